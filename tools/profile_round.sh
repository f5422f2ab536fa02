#!/bin/bash
# The judged profiling artifacts of a round, ONE workload per rocprofv3 run (no 1-env launches mixed in: --no-extra-legs):
#   1. rocprofv3 --kernel-trace --stats        -> gpurun_out/<TAG>_kernel_stats_<workload>.csv   (+ the bench line printed under it)
#   2. rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) for EVERY kernel of the step, the Partial paths' deferred /
#      finalize launches included -> gpurun_out/pmc_traffic.json, stamped with the hash of the kernel sources (bench.py reports
#      roofline.traffic from profiles/pmc_traffic.json only while that hash matches)
#   3. SQ counters (two passes) of every workload's step kernel -> gpurun_out/<TAG>_sq_breakdown_<workload>.txt and
#      gpurun_out/sq_counters.json (valu_busy, wave_time_shares per kernel, stamped like pmc_traffic.json: bench.py's roofline reads it)
# Every run is `bench.py --workload W --roofline-only`: the launches a profiler sees are the roofline leg's two passes over ONE
# whole episode at 4096 envs (+ 7 first-touch launches), i.e. the average of the stats csv is that leg's launch_ms.
# Usage (GPU box): bash tools/profile_round.sh r03_a ["driving robocup driving_partial robocup_partial"]; then copy into profiles/.
TAG=${1:-rXX}
WORKLOADS=${2:-"driving robocup driving_partial robocup_partial hbm"}
export TMPDIR=/tmp
mkdir -p gpurun_out
B="--roofline-only"
for W in $WORKLOADS; do
  case $W in robocup*) STEPS=240;; *) STEPS=600;; esac
  D=gpurun_out/${TAG}_prof_$W
  rm -rf $D ${D}_f ${D}_w
  # "hbm": the arranger's padded-tensor kernel and the transport's expansion kernel (tools/hbm_kernels_run.py), the two HBM-bound kernels
  case $W in hbm) PROG="tools/hbm_kernels_run.py";; *) PROG="bench.py --workload $W $B";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $PROG > gpurun_out/${TAG}_bench_${W}_under_rocprof.json 2> $D.err || exit 1
  cp "$(ls $D/*/*kernel_stats.csv | head -1)" gpurun_out/${TAG}_kernel_stats_$W.csv
  echo "[profile_round] $W kernel stats done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${D}_f -- python3 $PROG > /dev/null 2> ${D}_f.err || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${D}_w -- python3 $PROG > /dev/null 2> ${D}_w.err || exit 1
  echo "[profile_round] $W traffic passes done"
  case $W in driving|robocup|driving_partial|robocup_partial)
    rm -rf ${D}_s1 ${D}_s2
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d ${D}_s1 -- python3 bench.py --workload $W $B > /dev/null 2> ${D}_s1.err || exit 1
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d ${D}_s2 -- python3 bench.py --workload $W $B > /dev/null 2> ${D}_s2.err || exit 1
    echo "[profile_round] $W SQ passes done";;
  esac
done
python3 tools/profile_collect.py $TAG $WORKLOADS
# regression gates: the step kernels' average launch durations of this round's kernel-stats passes
python3 - "$TAG" <<'PY'
import csv, sys
tag = sys.argv[1]
GATES = {"driving": (["drv_step_kernel"], 165.0), "driving_partial": (["drv_step_partial_kernel", "drv_partial_obs_deferred_kernel"], 236.0),
         "robocup": (["rc_step_kernel"], 1280.0), "robocup_partial": (["rc_step_partial_kernel", "rc_partial_obs_deferred_kernel", "rc_partial_finalize_kernel"], 1380.0)}
for w, (ks, gate_us) in GATES.items():   # (the Partial steps as the SUM of their launches: the fused-vision deadline moves work between them)
    try:
        rows = list(csv.DictReader(open("gpurun_out/%s_kernel_stats_%s.csv" % (tag, w))))
    except OSError:
        continue
    steps = max(int(r["Calls"]) for r in rows if r["Name"].split("(")[0] == ks[0])
    us = [float(r["AverageNs"]) / 1e3 for k in ks for r in rows if r["Name"].split("(")[0] == k and int(r["Calls"]) == steps]
    if len(us) == len(ks):
        print("[profile_round] gate %-16s %-58s %8.1f us  (<= %.1f us: %s)" % (w, " + ".join(ks), sum(us), gate_us, "ok" if sum(us) <= gate_us else "REGRESSION"))
PY
# the RoboCup code's instruction-cache phase (RC_LAYOUT_PAD_WORDS): still within 0.5 % of the best candidate? (needs the candidate
# libraries of `python3 tools/rc_layout_sweep.py build`; skipped without them)
python3 tools/rc_layout_sweep.py check || echo "[profile_round] RoboCup code phase NOT within 0.5 % of the best candidate: see gpurun_out/rc_layout_sweep.txt"
