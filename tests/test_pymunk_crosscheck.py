"""tools/pymunk_crosscheck.py: the tool a maintainer with pymunk runs to turn "parity: partial" into a number.  Here:
its kat_general engine reproduces a committed trajectory of tests/golden/driving_contacts.npz bit for bit THROUGH the public-pymunk-API code
path the real engine uses (substep log, post_solve order log), stays within 1e-9 of the oracle substep by substep; an engine with
another pair order is reported as such; and `--engine pymunk` says clearly what is missing.  CPU only; needs /root/reference."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "DynEnv")), reason="imports the reference's Python")


@needs_reference
def test_kat_general_engine_reproduces_the_committed_fixture_and_agrees_substep_by_substep(oracle_built, capsys):
    import pymunk_crosscheck as pc
    res = pc.main(["--engine", "kat_general", "--fixture", "j"])[0]
    out = capsys.readouterr().out
    assert "ARE bit for bit the committed fixture's" in out
    assert res["first"][1e-9] is None and res["max_dev"] < 1e-10 and res["reward_dev"] < 1e-12
    assert res["substeps"] == 2000 and res["touching"] > 100 and res["foreign"] == 0


@needs_reference
def test_an_engine_with_another_pair_order_is_reported(oracle_built, capsys):
    import pymunk_crosscheck as pc
    try:
        res = pc.main(["--engine", "kat_general", "--kat-order", "reversed", "--fixture", "a"])[0]
    finally:
        pc.install_engine("kat_general", "canonical")
    out = capsys.readouterr().out
    assert res["foreign"] > 0 and "NOT ascending shape ids" in out or res["first"][1e-9] is None
    assert res["foreign"] > 0


def test_the_pymunk_engine_says_what_is_missing():
    import importlib.util
    import pymunk_crosscheck as pc
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "pymunk" or k.startswith("pymunk.")}   # (a stand-in left by another test)
    try:
        if importlib.util.find_spec("pymunk") is not None:
            pytest.skip("pymunk is installed here: run tools/pymunk_crosscheck.py --engine pymunk")
        with pytest.raises(SystemExit) as e:
            pc.main(["--engine", "pymunk", "--fixture", "a"])
        assert "needs the `pymunk` module" in str(e.value) and "INTEGRATION.md" in str(e.value)
    finally:
        sys.modules.update(saved)
