"""Whole episodes at full batch size against the oracle, bit for bit (tools/soak_parity.py): rewards and dones of every step
of every environment, observations every 25th step, episode statistics.  The exact shortcuts of the step kernels are only
exercised in their full variety at this scale."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,E", [("driving", 4096), ("robocup", 4096), ("driving_partial", 4096), ("robocup_partial", 4096)])
def test_full_episode_full_batch_parity(oracle_built, cfg, E):
    import soak_parity
    soak_parity.run(cfg, E, 20261003)
