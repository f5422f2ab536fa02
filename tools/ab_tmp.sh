set -e
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -x -q -k "robocup or rc" > gpurun_out/ab_tests.txt 2>&1 || { tail -30 gpurun_out/ab_tests.txt; exit 1; }
tail -2 gpurun_out/ab_tests.txt
for p in 1 2 3; do
  DYNENV_HIP_LIB=dynenv_amd/libdynenv_hip_ab_prev.so python tools/episode_time.py robocup 2 2>/dev/null | grep ms/step
  python tools/episode_time.py robocup 2 2>/dev/null | grep ms/step
done
