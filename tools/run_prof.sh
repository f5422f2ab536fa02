# Collect the round's rocprofv3 evidence for bench.py's default workload (Driving, 4096 envs) + RoboCup.
mkdir -p gpurun_out
rm -rf gpurun_out/kt gpurun_out/kt_rc gpurun_out/pmc1 gpurun_out/pmc_fetch gpurun_out/pmc_write
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --no-cpu-baseline > gpurun_out/kt.json 2> gpurun_out/kt.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/pmc1.json 2> gpurun_out/pmc1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_f.json 2> gpurun_out/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_w.json 2> gpurun_out/pmc_w.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_rc -- python3 bench.py --workload robocup --no-cpu-baseline > gpurun_out/kt_rc.json 2> gpurun_out/kt_rc.err
python3 tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary.txt
cat gpurun_out/pmc_summary.txt | head -30
