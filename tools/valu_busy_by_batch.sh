#!/bin/bash
# SIMD VALU-busy of drv_step_kernel against the number of environments per launch / sub-batches on streams:
# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU over tools/split_batch_probe.py, launches grouped by grid size.
# VALU busy >= SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch duration x 2.4 GHz)  (fp64 holds the SIMD longer: a lower bound).
# Usage (GPU box): bash tools/valu_busy_by_batch.sh > gpurun_out/valu_busy_by_batch.txt
export TMPDIR=/tmp
D=gpurun_out/vbb
rm -rf $D; mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $D -- python3 tools/split_batch_probe.py --ks 1,2 --sizes 8192,16384,32768 --steps 200 > $D.log 2> $D.err || { tail -5 $D.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob("gpurun_out/vbb/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("drv_step_kernel") and r["Counter_Name"] == "SQ_INSTS_VALU":
            a = agg[int(r["Grid_Size"]) // 64]
            a[0] += float(r["Counter_Value"]); a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[2] += 1
print("envs per launch | launches | mean launch us | VALU instructions per launch | SIMD VALU busy (lower bound) | agent-steps/s of the launch alone")
for envs in sorted(agg):
    v, ns, n = agg[envs]
    print("%6d | %5d | %8.1f | %.3g | %.1f %% | %.0f M" % (envs, n, ns / n / 1e3, v / n, 100 * v * 4 / (1024 * ns * 2.4), envs * 10 * n / (ns * 1e-9) / 1e6))
print("(first 200 steps of an episode; 2048-environment launches = two sub-batches on two streams, overlapping in time: their per-launch duration is longer than their share of the wall clock)")
PY
