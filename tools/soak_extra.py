#!/usr/bin/env python3
"""Extra soaks beyond the ones in tests/: other seeds, other player counts, every configuration, 4096 environments, whole episodes,
bit for bit against the oracle.  Usage (GPU box): python tools/soak_extra.py > gpurun_out/soak_extra.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_parity  # noqa: E402

for cfg, seed, players in (("driving", 777, None), ("driving", 20261004, None), ("driving", 5, 6), ("driving", 6, 2), ("robocup", 777, None),
                           ("robocup", 8, 3), ("driving_partial", 31, None), ("robocup_partial", 31, None), ("robocup_partial", 32, 2)):
    t0 = time.time()
    soak_parity.run(cfg, 4096 if cfg != "robocup_partial" else 2048, seed, players)
    print("soak %s seed %d players %s OK in %.0f s" % (cfg, seed, players, time.time() - t0), flush=True)
