# extra SQ counters for the Driving step kernel: where does a wave's lifetime go?
mkdir -p gpurun_out
rm -rf gpurun_out/dpmc2 gpurun_out/dpmc3 gpurun_out/dpmc4
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d gpurun_out/dpmc2 -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/dpmc2.json 2> gpurun_out/dpmc2.err
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/dpmc3 -- python3 bench.py --steps 600 --warmup 0 --no-cpu-baseline > gpurun_out/dpmc3.json 2> gpurun_out/dpmc3.err
python3 tools/pmc_summary.py gpurun_out drv_step_kernel 2>/dev/null | grep -v "steps\|dispatch" | sort -u > gpurun_out/drv_pmc2_summary.txt
grep "SQ_ACTIVE\|SQ_WAIT\|IFETCH\|BRANCH\|SALU\|SMEM\|LEVEL\|WAVE_CYC" gpurun_out/drv_pmc2_summary.txt
grep -i "error\|invalid\|not found" gpurun_out/dpmc2.err gpurun_out/dpmc3.err | head -5
