#!/bin/bash
# One GPU call that regenerates the judged artifacts of a round: HBM traffic of the four step kernels (-> profiles/pmc_traffic.json,
# read by bench.py), the driver-style bench line plain and under rocprofv3 --kernel-trace --stats, and the GPU test tail.
# Usage (GPU box): bash tools/refresh_artifacts.sh r02_b      (files land in gpurun_out/; copy them into profiles/)
TAG=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_traffic_all.sh > gpurun_out/${TAG}_pmc_traffic.log 2>&1 && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json &&
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_style.json 2> gpurun_out/${TAG}_bench.err &&
rm -rf gpurun_out/${TAG}_prof &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_style_under_rocprof.json 2> gpurun_out/${TAG}_bench_rocprof.err &&
cp $(ls gpurun_out/${TAG}_prof/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats_driver_style.csv &&
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${TAG}_pytest_gpu_tail.txt
cat gpurun_out/${TAG}_bench_driver_style.json; cat gpurun_out/${TAG}_pytest_gpu_tail.txt; head -8 gpurun_out/${TAG}_kernel_stats_driver_style.csv
