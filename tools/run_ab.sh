mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in "" _w2 _w3; do echo "variant '$v'"; DYNENV_HIP_LIB=$PWD/dynenv_amd/libdynenv_hip$v.so python bench.py --no-cpu-baseline --steps 600 --warmup 600 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
