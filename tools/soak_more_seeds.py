import sys
sys.path.insert(0, "tools")
import soak_parity
for seed in (7, 990017):
    for cfg, E in (("driving", 4096), ("robocup", 4096), ("driving_partial", 2048), ("robocup_partial", 1024)):
        soak_parity.run(cfg, E, seed)
