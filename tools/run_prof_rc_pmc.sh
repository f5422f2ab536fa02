# PMC passes for the RoboCup step kernel (separate passes; kernel-trace only)
mkdir -p gpurun_out
rm -rf gpurun_out/rcpmc1 gpurun_out/rcpmc2 gpurun_out/rcpmc_fetch gpurun_out/rcpmc_write
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/rcpmc1 -- python3 bench.py --workload robocup --steps 60 --warmup 0 --no-cpu-baseline > gpurun_out/rcpmc1.json 2> gpurun_out/rcpmc1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/rcpmc2 -- python3 bench.py --workload robocup --steps 60 --warmup 0 --no-cpu-baseline > gpurun_out/rcpmc2.json 2> gpurun_out/rcpmc2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/rcpmc_fetch -- python3 bench.py --workload robocup --steps 60 --warmup 0 --no-cpu-baseline > gpurun_out/rcpmc_f.json 2> gpurun_out/rcpmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/rcpmc_write -- python3 bench.py --workload robocup --steps 60 --warmup 0 --no-cpu-baseline > gpurun_out/rcpmc_w.json 2> gpurun_out/rcpmc_w.err
python3 tools/pmc_summary.py gpurun_out rc_step_kernel > gpurun_out/rc_pmc_summary.txt
grep -v "steps\|dispatch" gpurun_out/rc_pmc_summary.txt
tail -3 gpurun_out/rcpmc2.err
