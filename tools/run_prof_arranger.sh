# rocprofv3 kernel stats of the arranger kernels at BASELINE sizes (4096 envs)
export TMPDIR=/tmp
rm -rf gpurun_out/kt_arr
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_arr -- python3 tools/bench_arranger.py > gpurun_out/arr_prof.jsonl 2> gpurun_out/arr_prof.err
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kt_arr/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("arr_") or "drv_step" in r["Name"] or "rc_step" in r["Name"]:
            print("%-28s calls %6s avg %10.1f us" % (r["Name"][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
