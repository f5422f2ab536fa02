mkdir -p gpurun_out
for w in 1 2 3; do echo "waves/SIMD cap $w"; DYNENV_HIP_LIB=$PWD/dynenv_amd/libdynenv_hip_w$w.so python bench.py --no-cpu-baseline --steps 600 --warmup 600 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 100 --warmup 300 --no-cpu-baseline > gpurun_out/pmc1.json 2> gpurun_out/pmc1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 100 --warmup 300 --no-cpu-baseline > gpurun_out/pmc_f.json 2> gpurun_out/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 100 --warmup 300 --no-cpu-baseline > gpurun_out/pmc_w.json 2> gpurun_out/pmc_w.err
ls gpurun_out/pmc1/* | head
