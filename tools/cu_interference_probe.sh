# per-environment cycles of the SAME environments (ids 0..255, step 560) when the launch holds 256 (one wave per CU), 1024 (one per SIMD)
# and 4096 (four per SIMD, no isolation) environments: what neighbours on the SIMD / on the CU cost the slow ones
for E in 256 1024 4096; do
  DYNENV_NO_ISOLATION=1 PROFILE_ENVS=$E python tools/contact_profile.py 560 > /dev/null 2>&1
  cp gpurun_out/dbgw.txt gpurun_out/dbgw_$E.txt
done
python - <<'PY'
import numpy as np
d = {E: np.loadtxt("gpurun_out/dbgw_%d.txt" % E)[:256, 0] for E in (256, 1024, 4096)}
top = np.argsort(-d[256])[:16]
print("env  cycles at E=256 | E=1024 (ratio) | E=4096 (ratio)")
for k in top:
    print("%4d %8d | %8d (%.3f) | %8d (%.3f)" % (k, d[256][k], d[1024][k], d[1024][k] / d[256][k], d[4096][k], d[4096][k] / d[256][k]))
print("mean ratio of the 16 slowest: E=1024 %.3f, E=4096 %.3f" % (np.mean(d[1024][top] / d[256][top]), np.mean(d[4096][top] / d[256][top])))
PY
