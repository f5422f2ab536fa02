# RoboCup: per-environment cycles of the SAME environments (ids 0..1023, the same actions) when the launch holds 1024 (one wave per
# SIMD) and 4096 (four per SIMD) environments: what the neighbours on its SIMD / CU cost a slow environment = the upper bound of what
# SIMD isolation can recover.  Usage (GPU box): bash tools/rc_interference_probe.sh [step]
STEP=${1:-100}
for E in 256 1024 4096; do
  PROFILE_ENVS=$E PROFILE_SAVE=gpurun_out/rcw_$E.npy python3 tools/robocup_profile.py $STEP > gpurun_out/rc_profile_E$E.txt 2>&1 || exit 1
done
python3 - <<'PY'
import numpy as np
d = {E: np.load("gpurun_out/rcw_%d.npy" % E) for E in (256, 1024, 4096)}
for base, n in ((256, 256), (1024, 1024)):
    top = np.argsort(-d[base][:n])[:16]
    print("the 16 slowest of environments 0..%d, cycles at E=%d | E=1024 (ratio) | E=4096 (ratio)" % (n - 1, base))
    for k in top:
        print("%4d %8d | %8d (%.3f) | %8d (%.3f)" % (k, d[base][k], d[1024][k], d[1024][k] / d[base][k], d[4096][k], d[4096][k] / d[base][k]))
    print("mean ratio of the 16 slowest: E=1024 %.3f, E=4096 %.3f; max: %d / %d / %d" % (np.mean(d[1024][top] / d[base][top]), np.mean(d[4096][top] / d[base][top]), d[base][:n].max(), d[1024][:n].max(), d[4096][:n].max()))
PY
