"""Randomised differential runs under `-m gpu`: a time-boxed slice of tests/manual/stress.py (the sorter over every
forced code path, int32 / int64, host and device entry points, every suffix array bit-compared with the oracle) and of
tests/manual/stress_bsdiff.py (Diff.Create's raw streams against the oracle's restated scan loop, Patch.Apply round
trips).  The seed changes with the tree -- `git rev-parse HEAD` where there is a repository, else a digest of the
product's sources (the GPU box gets a snapshot without .git) -- so every round tests inputs no earlier round has
passed; it is printed, a failing input is saved under gpurun_out/ by the scripts, and DQ_STRESS_SEED pins it.

Why they are here: the fixed-seed suite let a not-bit-exact match search (round 2) and a device memory fault (round 3)
through; both were found only by these scripts run by hand."""
import glob
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

# (90 s a slice: with the round-6 additions the GPU suite stays under 9 of its 20 minutes; longer slices are run by hand
# -- `python tests/manual/stress.py 600 <seed>` -- and their logs kept under profiles/)
BUDGET_S = int(os.environ.get("DQ_STRESS_SECONDS", "90"))


def tree_seed() -> int:
    if os.environ.get("DQ_STRESS_SEED"):
        return int(os.environ["DQ_STRESS_SEED"])
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        if head:
            return int(head[:8], 16) & 0x7fffffff
    except (OSError, subprocess.CalledProcessError):
        pass
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "deltaq_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
                    + glob.glob(os.path.join(ROOT, "deltaq_amd", "*.py"))):
        h.update(open(p, "rb").read())
    return int(h.hexdigest()[:8], 16) & 0x7fffffff


def run_script(name: str, seed: int):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = os.path.join(ROOT, "gpurun_out", f"{name}_seed{seed}.last")
    env = dict(os.environ, STRESS_LOG=log)
    for k in [k for k in env if k.startswith("DQ_") and k not in ("DQ_STRESS_SEED", "DQ_STRESS_SECONDS")]:
        env.pop(k)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "manual", name + ".py"), str(BUDGET_S), str(seed)],
                       capture_output=True, text=True, env=env, timeout=BUDGET_S + 240)
    tail = (p.stdout + p.stderr)[-3000:]
    print(f"{name} seed {seed}: {tail.strip().splitlines()[-1] if tail.strip() else ''}")
    assert p.returncode == 0, f"{name}.py seed {seed} (DQ_STRESS_SEED={seed} repeats it; input under gpurun_out/):\n{tail}"
    assert "OK:" in p.stdout, tail
    return p.stdout


def test_sorter_randomised_differential_slice():
    seed = tree_seed()
    out = run_script("stress", seed)
    # a slice that tested nothing proves nothing
    n_inputs = int(out.split("stress OK:")[1].split("inputs")[0])
    assert n_inputs >= 50, out


def test_bsdiff_randomised_differential_slice():
    seed = tree_seed() ^ 0x5bd1e995
    out = run_script("stress_bsdiff", seed & 0x7fffffff)
    n_pairs = int(out.split("bsdiff stress OK:")[1].split("file pairs")[0])
    assert n_pairs >= 20, out
