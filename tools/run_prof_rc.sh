export TMPDIR=/tmp
rm -rf gpurun_out/rc_pmc*
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/rc_pmc1 -- python3 bench.py --workload robocup --steps 60 --warmup 60 --no-cpu-baseline > gpurun_out/rc_pmc1.json 2> gpurun_out/rc_pmc1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH --output-format csv -d gpurun_out/rc_pmc2 -- python3 bench.py --workload robocup --steps 60 --warmup 60 --no-cpu-baseline > gpurun_out/rc_pmc2.json 2> gpurun_out/rc_pmc2.err
python3 tools/pmc_summary.py gpurun_out rc_step_kernel | grep -v "^  steps\|dispatch"
