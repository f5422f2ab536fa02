"""Where do the 4096 regular blocks of a Driving step land?  The SIMD isolation assumes that blocks g, g + 1024,
g + 2048, g + 3072 share a SIMD; the library checks that on the device after every launch.  This prints what the record says on
THIS box: distinct SIMDs, blocks per SIMD, how many of the 1024 groups hold, and for the groups that do not, what they look like.
Usage (GPU box): python tools/placement_probe.py [steps]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dynenv_amd import BatchedDynEnv, DynEnvType

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B2B = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # launches back to back (no host sync in between) before each look
env = BatchedDynEnv(DynEnvType.DRIVE, 4096, 10, seed=42)
env.reset_flat()
g = torch.Generator(device="cuda").manual_seed(1)
acts = [torch.randint(0, 3, (4096, 10, 2), dtype=torch.int32, device="cuda", generator=g) for _ in range(16)]
for s in range(steps):
    for j in range(B2B):
        env.step_flat(acts[(s + j) & 15], auto_reset=False)
    hw = env.debug_placement()
    if hw.size == 0:
        print("this handle does not record placements (isolation mode %d)" % env.debug_counters()["isolation_mode"]); break
    k = hw.reshape(4, 1024)
    same = (k[0] == k[1]) & (k[0] == k[2]) & (k[0] == k[3])
    per = collections.Counter(hw.tolist())
    sizes = collections.Counter(per.values())
    print("step %d: %d distinct SIMDs, blocks per SIMD %s, groups that hold: %d / 1024 (of 256..511, the ones isolation uses: %d / 256), counters %s" %
          (s, len(per), sorted(sizes.items()), int(same.sum()), int(same[256:512].sum()), {k_: v for k_, v in env.debug_counters().items() if "isol" in k_ or "place" in k_}))
    if not same.all():
        bad = np.nonzero(~same)[0][:8]
        for b in bad:
            print("   group %4d: %s" % (b, ["xcc %d se %d sh %d cu %2d simd %d" % (x >> 16, (x >> 13) & 7, (x >> 12) & 1, (x >> 8) & 15, (x >> 4) & 3) for x in k[:, b].tolist()]))
        # is it the same SIMD set shifted?  blocks per quarter per SIMD
        q = [collections.Counter(k[i].tolist()) for i in range(4)]
        print("   every quarter of the grid covers each SIMD exactly once:", [sorted(collections.Counter(c.values()).items()) for c in q])
        # which block offsets share a SIMD with block g of the first quarter
        where = collections.defaultdict(list)
        for b, x in enumerate(hw.tolist()): where[x].append(b)
        d = collections.Counter(tuple(np.diff(sorted(v)).tolist()) for v in where.values())
        print("   block-index differences within a SIMD (most common):", d.most_common(6))
        # agreement pattern of each group's four members (A = first member's SIMD, B = next new one, ...), by range of g
        def pat(col):
            names, out = {}, ""
            for x in col:
                names.setdefault(x, "ABCD"[len(names)])
                out += names[x]
            return out
        pats = [pat(k[:, g_].tolist()) for g_ in range(1024)]
        for lo in range(0, 1024, 256):
            print("   groups %4d..%4d:" % (lo, lo + 255), sorted(collections.Counter(pats[lo:lo + 256]).items()))
        # members of a SIMD as (quarter, group) pairs for a few SIMDs that are not a clean group
        shown = 0
        for x, v in where.items():
            gs = sorted((b // 1024, b % 1024) for b in v)
            if len({g_ for _, g_ in gs}) > 1 and shown < 10:
                print("   SIMD %08x holds (quarter, group):" % x, gs); shown += 1
env.close()
