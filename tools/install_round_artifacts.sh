#!/bin/bash
# After a freeze run on the GPU box (GPU tests, tools/profile_round.sh PROF, bench.py twice as BENCH_bench_default / _driver_style): copy the
# judged artifacts from gpurun_out/ into profiles/, drop the previous set and point the documents at the new names.
# Usage: bash tools/install_round_artifacts.sh r06_v r06_m r06_u r06_l   (new profile tag, new bench tag, old profile tag, old bench tag)
set -e
P=$1; B=$2; OP=$3; OB=$4
cd "$(dirname "$0")/.."
for f in gpurun_out/${P}_kernel_stats_*.csv gpurun_out/${P}_sq_breakdown_*.txt gpurun_out/${P}_bench_*_under_rocprof.json gpurun_out/${P}_profile_round.log \
         gpurun_out/${B}_bench_default.json gpurun_out/${B}_bench_driver_style.json; do cp "$f" profiles/; done
cp gpurun_out/pmc_traffic.json gpurun_out/sq_counters.json profiles/
[ -f gpurun_out/${B}_gpu_tests.txt ] && cp gpurun_out/${B}_gpu_tests.txt profiles/${B}_gpu_tests_tail.txt
if [ -n "$OP" ] && [ "$OP" != "$P" ]; then rm -f profiles/${OP}_kernel_stats_*.csv profiles/${OP}_sq_breakdown_*.txt profiles/${OP}_bench_*_under_rocprof.json profiles/${OP}_profile_round.log; fi
if [ -n "$OB" ] && [ "$OB" != "$B" ]; then rm -f profiles/${OB}_bench_default.json profiles/${OB}_bench_driver_style.json profiles/${OB}_gpu_tests_tail.txt; fi
for d in DESIGN.md README.md INTEGRATION.md profiles/HISTORY.md; do
  [ -n "$OP" ] && sed -i "s/${OP}_/${P}_/g" $d
  [ -n "$OB" ] && sed -i "s/${OB}_bench/${B}_bench/g; s/${OB}_gpu_tests/${B}_gpu_tests/g" $d
done
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
t = json.load(open("profiles/pmc_traffic.json"))
print("kernel sources", bench.kernel_source_sha(), "| stamps", t["kernel_source_sha16"], json.load(open("profiles/sq_counters.json"))["kernel_source_sha16"])
assert t["kernel_source_sha16"] == bench.kernel_source_sha(), "the profile round was run on other kernel sources"
PY
