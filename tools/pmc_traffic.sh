#!/bin/bash
# HBM-side traffic of one bench.py workload's step kernel: rocprofv3 FETCH_SIZE / WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md "HBM": they do not fit one pass; FETCH_SIZE x2 on gfx950) + the VMEM instruction counts.
# Usage (GPU box): bash tools/pmc_traffic.sh <workload> <kernel name> <tag> [steps]     (DYNENV_HIP_LIB selects a build)
W=${1:-robocup}; K=${2:-rc_step_kernel}; TAG=${3:-x}; STEPS=${4:-60}
mkdir -p gpurun_out
export TMPDIR=/tmp
for P in "f FETCH_SIZE" "w WRITE_SIZE" "i SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  set -- $P; N=$1; shift
  rm -rf gpurun_out/pt_${TAG}_$N
  rocprofv3 --kernel-trace --pmc $@ --output-format csv -d gpurun_out/pt_${TAG}_$N -- python3 bench.py --workload $W --steps $STEPS --warmup 0 --no-cpu-baseline > gpurun_out/pt_${TAG}_$N.json 2> gpurun_out/pt_${TAG}_$N.err
done
python3 - "$K" "$TAG" <<'PY'
import csv, glob, sys, collections
kern, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pt_%s_*/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(agg):
    v = agg[c]
    print("%-20s launches=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    f = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"]) * 1024 * 2   # KB -> B, x2: gfx950 tallies 128-B requests at 64 B
    w = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"]) * 1024
    print("traffic per launch: fetch %.1f MB (x2 corrected) + write %.1f MB = %.1f MB" % (f / 1e6, w / 1e6, (f + w) / 1e6))
PY
