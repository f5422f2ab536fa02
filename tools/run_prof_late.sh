# PMC passes over late-episode steps of the Driving workload (steps 400-599: most substeps take the contact path)
export TMPDIR=/tmp
rm -rf gpurun_out/late_*
rocprofv3 --list-avail > gpurun_out/counters_avail.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/late_pmc1 -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/late_pmc1.json 2> gpurun_out/late_pmc1.err
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d gpurun_out/late_pmc2 -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/late_pmc2.json 2> gpurun_out/late_pmc2.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/late_pmc3 -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/late_pmc3.json 2> gpurun_out/late_pmc3.err
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/late_pmc4 -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/late_pmc4.json 2> gpurun_out/late_pmc4.err
python3 tools/pmc_summary.py gpurun_out drv_step_kernel | grep -v "^  steps\|dispatch"
tail -3 gpurun_out/late_pmc2.err gpurun_out/late_pmc4.err
