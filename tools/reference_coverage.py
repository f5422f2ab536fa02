#!/usr/bin/env python3
"""Which lines of the reference's hot-path files do the fixture generators (tests/golden/gen_golden*.py) actually EXECUTE?  Runs every
generator (each in a scratch copy of tests/golden, nothing in the tree is touched) under sys.settrace restricted to /root/reference/DynEnv
and prints, per function of the hot-path files, the executable lines that no generator reached - the parts of the reference's behaviour
that no fixture pins.  Build container only (it imports /root/reference).

   python3 tools/reference_coverage.py > profiles/r05_reference_coverage.txt        (~10 min)
"""
import ast
import os
import runpy
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DynEnv"
FILES = ["DrivingEnvironment.py", "RoboCupEnvironment.py", "Car.py", "Robot.py", "Ball.py", "Pedestrian.py", "Obstacle.py", "Goalpost.py", "Road.py",
         "cutils.py", "environment_base.py", "models/models.py"]
ONLY = {"models/models.py": ("Indexer", "InOutArranger")}   # of these files only the named classes belong to the path
hit = {}


def tracer(frame, event, arg):
    fn = frame.f_code.co_filename
    if not fn.startswith(REF):
        return None
    if event == "line":
        hit.setdefault(fn, set()).add(frame.f_lineno)
    return tracer


def executable_lines(path):
    """line -> enclosing function name, for every line that starts a statement inside a function"""
    tree = ast.parse(open(path).read())
    out = {}

    def walk(node, fn):
        for ch in ast.iter_child_nodes(node):
            if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef)):
                walk(ch, (fn + "." if fn else "") + ch.name)
            elif isinstance(ch, ast.ClassDef):
                walk(ch, ch.name)
            else:
                if isinstance(ch, ast.stmt) and fn and not (isinstance(ch, ast.Expr) and isinstance(getattr(ch, "value", None), ast.Constant)):
                    out[ch.lineno] = fn
                walk(ch, fn)
    walk(tree, "")
    return out


def run_one(gold, g, out):
    """one generator, in this (fresh) process, traced; the lines it reached go to `out` as JSON"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, gold)
    sys.stdout = open(os.devnull, "w")
    os.chdir(gold)
    sys.settrace(tracer)
    try:
        runpy.run_path(os.path.join(gold, g), run_name="__main__")
    except SystemExit:
        pass
    finally:
        sys.settrace(None)
    json.dump({k: sorted(v) for k, v in hit.items()}, open(out, "w"))


def main():
    import json
    import subprocess
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        return run_one(sys.argv[2], sys.argv[3], sys.argv[4])
    scratch = tempfile.mkdtemp(prefix="refcov_")
    gold = os.path.join(scratch, "golden")
    shutil.copytree(os.path.join(ROOT, "tests", "golden"), gold)
    gens = sorted(f for f in os.listdir(gold) if f.startswith("gen_golden") and f.endswith(".py"))
    for g in gens:  # (each generator installs its own stand-ins for the reference's missing modules: one process each)
        out = os.path.join(scratch, g + ".json")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", gold, g, out], check=True)
        for k, v in json.load(open(out)).items():
            hit.setdefault(k, set()).update(v)
        print("ran %s" % g, file=sys.stderr)
    shutil.rmtree(scratch, ignore_errors=True)
    print("Lines of the reference's hot-path files that NO fixture generator executes (function: line numbers; %d generators)" % len(gens))
    print("What is left when this was last looked at (round 5): rendering and the Image observation (drawStaticObjects, _render_internal, the\n"
          "cv2 / projectPoints / getConicPoints / estimateConic / colorize blocks of getAgentVision: out of scope, SURVEY section 2), the `raise` lines of the\n"
          "sanity checks, continuous actions (Car.accelerate :65-69) and the 3-element action form, argparse / __str__ helpers, defaults of keyword\n"
          "arguments nobody omits, code the reference itself disabled (`if False:`, getLineInRadius), two noise multipliers for sighting types that\n"
          "Driving's isSeenInRadius never returns (cutils.py:517, 519), EnvironmentBase.reset / set_random_seed (the vec-env wrapper's job).\n"
          "Everything else of step / processAction / tick / move / isBallOutOfField / the collision callbacks / getFullState / getAgentVision runs.")
    for f in FILES:
        path = os.path.join(REF, f)
        ex = executable_lines(path)
        if f in ONLY:
            ex = {ln: fn for ln, fn in ex.items() if fn.split(".")[0] in ONLY[f]}
        got = hit.get(path, set())
        miss = {}
        for ln, fn in sorted(ex.items()):
            if ln not in got:
                miss.setdefault(fn, []).append(ln)
        total, cov = len(ex), sum(1 for ln in ex if ln in got)
        print("\n%s: %d of %d statement lines inside functions executed (%.0f %%)" % (f, cov, total, 100.0 * cov / max(total, 1)))
        for fn, lns in miss.items():
            n_fn = sum(1 for v in ex.values() if v == fn)
            print("   %-45s %3d of %3d missed: %s" % (fn, len(lns), n_fn, " ".join(str(x) for x in lns)))


if __name__ == "__main__":
    main()
