# per-wave instruction counts of a workload's step kernel for the library DYNENV_HIP_LIB selects (A/B of build variants on the judged
# counters: SALU and branches per wave-step).  Usage (GPU box): DYNENV_HIP_LIB=$PWD/dynenv_amd/libdynenv_hip_x.so bash tools/sq_quick.sh robocup tag
W=${1:-robocup}; T=${2:-x}
export TMPDIR=/tmp
D=gpurun_out/sq_quick_$T
rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $D -- python3 bench.py --workload $W --roofline-only > /dev/null 2> $D.err || exit 1
python3 - "$D" "$T" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "step" in k and "SQ_WAVES" in c:
        w = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])
        print(sys.argv[2], k, "per wave-step:", {n: round(sum(v) / len(v) / w) for n, v in c.items() if n.startswith("SQ_INSTS")}, "wave cycles %.0f" % (sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"]) / w))
PY
