# rocprofv3 kernel stats of the RoboCup Partial workload (step kernel + observation kernel)
export TMPDIR=/tmp
rm -rf gpurun_out/kt_rcp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_rcp -- python3 bench.py --workload robocup_partial --steps 120 --warmup 30 --no-cpu-baseline > gpurun_out/kt_rcp.json 2> gpurun_out/kt_rcp.err
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kt_rcp/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("rc_"):
            print("%-28s calls %6s avg %10.1f us" % (r["Name"][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
