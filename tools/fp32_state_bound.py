"""What can ANY engine that holds its state in fp32 achieve against the fp64 oracle?  The fp64 build of the engine source (= the oracle to 1e-10 on every
fixture, checked here) is stepped with its state (qpos, qvel, warm start) rounded to fp32 after every substep -- exact arithmetic, fp32 state, i.e. the best
possible fp32-state engine -- in several variants (plain rounding, and rounding moved by up to one ulp at random: no engine can control its last bit).  A
snapshot on which one of these ideal engines leaves the oracle by more than 1e-4 is ILL-POSED for fp32 state: the reference's own step is discontinuous
(a contact or limit row switching on within the step) inside the fp32 resolution of the state.  Engine-independent: nothing of the fp32 arithmetic is involved.

    python tools/fp32_state_bound.py [family ...]          -> tests/golden/fp32_state_bound.json   {family: {component: [snapshot, ...]}}
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "tests", "golden", "fp32_state_bound.json")
VARIANTS = 6      # dither seeds 0 (plain rounding) .. 5


def run(fam, dither):
    code = f"""
import sys, ctypes, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); sys.path.insert(0, {os.path.join(ROOT, 'tools')!r})
import emu_tolerances as T
sys.argv = ['x']
L = ctypes.CDLL(T.build(True)); L.emu_create.restype = ctypes.c_void_p; L.emu_create.argtypes = [ctypes.c_void_p] * 3
idx, e, st, comps = T.run_family(L, {fam!r}, True, round_inputs={dither >= 0})
np.save('/tmp/fp32_bound_{fam}_{dither}.npy', e)
"""
    env = dict(os.environ, GRX_RND_MASK="96" if dither >= 0 else "0", GRX_RND_DITHER=str(max(dither, 0)))      # bits 5 | 6: the state after the solve / before the kinematics
    subprocess.check_call([sys.executable, "-c", code], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return np.load(f"/tmp/fp32_bound_{fam}_{dither}.npy")


def main(argv):
    from tolerance_cases import CASES
    import emu_tolerances as T
    fams = [a for a in argv if a in T.FAMILY_TO_TASK] or list(T.FAMILY_TO_TASK)
    table = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for fam in fams:
        exact = run(fam, -1)      # no rounding at all: must BE the oracle
        assert exact.max() < 1e-7, (fam, exact.max())
        worst = np.zeros_like(exact)
        for d in range(VARIANTS):
            worst = np.maximum(worst, run(fam, d))
        comps = CASES[fam][3]
        table[fam] = {}
        for comp, cols in comps.items():
            err = worst[:, cols].max(axis=1)
            bad = [int(i) for i in np.nonzero(err >= 1e-4)[0]]
            table[fam][comp] = bad
            print(f"{fam:18s} {comp:26s} ideal fp32-state engines: p50 {np.median(err):.1e} p99 {np.quantile(err, .99):.1e} max {err.max():.1e}  ill-posed snapshots ({len(bad)} of {len(err)}): {bad[:40]}", flush=True)
        json.dump(table, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
