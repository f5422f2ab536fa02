# rocprofv3 kernel stats of the Driving Partial workload
export TMPDIR=/tmp
rm -rf gpurun_out/kt_dp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_dp -- python3 bench.py --workload driving_partial --steps 600 --warmup 100 --no-cpu-baseline > gpurun_out/kt_dp.json 2> gpurun_out/kt_dp.err
python3 - <<'PY'
import csv, glob, os
fs = glob.glob("gpurun_out/kt_dp/*/*kernel_stats.csv"); fs.sort(key=os.path.getmtime)
for r in csv.DictReader(open(fs[-1])):
    if r["Name"].startswith("drv_"):
        print("%-36s calls %6s avg %10.1f us" % (r["Name"][:36], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
