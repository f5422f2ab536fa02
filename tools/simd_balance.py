"""Where do the environments of one Driving launch run?  Reads gpurun_out/dbgw.txt written by tools/contact_profile.py
(-DDRV_PROFILE build): column 0 = wall cycles of the env's wave, 1 = contact-path substeps, 11 = XCC_ID<<32 | HW_ID."""
import numpy as np, collections
d = np.loadtxt("gpurun_out/dbgw.txt", dtype=np.float64)
raw = [int(x) for x in np.loadtxt("gpurun_out/dbgw.txt", dtype=np.uint64)[:, 11]]
hw = np.array([r & 0xFFFFFFFF for r in raw]); xcc = np.array([(r >> 32) & 0xF for r in raw])
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; wave = hw & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
skey = key * 4 + simd
print("distinct XCC %d  SE %d  SH %d  CU-in-SE %d  -> CUs %d, SIMDs %d" % (len(set(xcc)), len(set(se)), len(set(sh)), len(set(cu)), len(set(key)), len(set(skey))))
cyc, nc = d[:, 0], d[:, 1]
heavy = nc >= 10
per = collections.defaultdict(list)
for i, k in enumerate(skey): per[k].append(i)
sizes = collections.Counter(len(v) for v in per.values())
print("envs per SIMD histogram:", sorted(sizes.items()))
nh = np.array([sum(heavy[i] for i in per[skey[i0]]) for i0 in range(len(cyc))])
for k in range(0, 9):
    m = heavy & (nh == k)
    if m.sum(): print("heavy envs sharing a SIMD with %d heavy envs in total: n=%4d  mean wall %8.0f  max %8.0f" % (k, m.sum(), cyc[m].mean(), cyc[m].max()))
for k in range(0, 9):
    m = (~heavy) & (nh == k)
    if m.sum(): print("light envs on a SIMD with %d heavy envs: n=%4d  mean wall %8.0f" % (k, m.sum(), cyc[m].mean()))
smax = np.array([max(cyc[i] for i in v) for v in per.values()])
ssum_h = np.array([sum(heavy[i] for i in v) for v in per.values()])
print("per-SIMD finish time: mean %.0f  p90 %.0f  max %.0f" % (smax.mean(), np.percentile(smax, 90), smax.max()))
for k in range(0, 9):
    m = ssum_h == k
    if m.sum(): print("  SIMDs with %d heavy envs: n=%4d mean finish %8.0f" % (k, m.sum(), smax[m].mean()))
b = np.arange(len(cyc))
print("block -> xcc (first 16):", list(xcc[:16]), " cu:", list(cu[:16]), " simd:", list(simd[:16]), "se:", list(se[:16]))
