# instruction-cache behaviour of the two step kernels
mkdir -p gpurun_out
rm -rf gpurun_out/ic_rc gpurun_out/ic_drv
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "ICACHE\|SQC_INST" | head -20 > gpurun_out/icache_counters.txt
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/ic_rc -- python3 bench.py --workload robocup --steps 60 --warmup 0 --no-cpu-baseline > gpurun_out/ic_rc.json 2> gpurun_out/ic_rc.err
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/ic_drv -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/ic_drv.json 2> gpurun_out/ic_drv.err
python3 - <<'PY'
import csv, glob, collections, os
for d, k in (("gpurun_out/ic_rc", "rc_step_kernel"), ("gpurun_out/ic_drv", "drv_step_kernel")):
    fs = glob.glob(d + "/*/*counter_collection.csv"); fs.sort(key=os.path.getmtime)
    if not fs: print(d, "no output"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if k in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in agg.items()})
PY
cat gpurun_out/icache_counters.txt | head -12
tail -2 gpurun_out/ic_rc.err
