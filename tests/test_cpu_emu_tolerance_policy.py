"""CPU twin of tests/test_gpu_tolerance_table.py: the engine SOURCE (fp32 lane emulator, tests/emu -- test infrastructure, never shipped) stepped from the oracle's
golden states, with the same policy as on the MI355X: every sampled snapshot whose activation gap is >= 1e-6 m must be within 1e-4 of the oracle on EVERY observation
component of EVERY fixture family (the 92 touch forces of cfg 3 ABSOLUTE, like every other component), and nothing may be off by more than the discontinuity of a constraint switch allows (5e-1).  Every snapshot
of every fixture, like the GPU test (half a minute of CPU)."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
sys.path.insert(0, os.path.join(HERE, "..", "tools"))
GAP, TOL = 1e-6, 1e-4
# The one measured exception, as on the MI355X (tests/golden/tolerance_table.json AntMazeLarge velocities: n_over_1e-4_away_from_boundary = 1, snapshot 180 at 1.29e-4: the torso rate after 5 RK4
# substeps against a wall, 19 rad/s at 18 m from the origin).  The emulator sums sequentially where the device reduces in DPP trees, so its borderline set is not exactly the device's:
# snapshot 35 is at 1.25e-4 here and below 1e-4 on the MI355X.  Both are held to 1.5e-4.  No other family / component has an exception (round 5's HandBlock snapshot 112 is at 1.2e-5 in the palm frame).
KNOWN = {("AntMazeLarge", "velocities"): [180, 35]}


FAMILIES = ["FetchReach", "FetchPush", "FetchPickAndPlace", "FetchSlide", "FetchHullContacts", "HandReach", "HandBlock", "HandEgg", "HandPen", "AdroitHammer", "AdroitDoor", "AdroitPen",
            "AdroitRelocate", "FrankaKitchen", "AntMazeLarge", "HandBlockTouch"]


@pytest.mark.parametrize("family", FAMILIES)
def test_emulated_family_meets_the_bound(family):
    import emu_sim
    import emu_tolerances as T
    from tolerance_cases import CASES, GOLDEN

    L = ctypes.CDLL(emu_sim.build())
    L.emu_create.restype = ctypes.c_void_p
    L.emu_create.argtypes = [ctypes.c_void_p] * 3
    g = np.load(os.path.join(GOLDEN, CASES[family][1]))
    idx, e, status, comps = T.run_family(L, family, False, 1)
    assert len(idx) == g["obs"].shape[0] and (status == 0).all()
    gap = g["activation_gap"][idx]
    posed = gap >= GAP
    import json
    from tolerance_cases import TABLE
    with open(TABLE) as f:
        rec = json.load(f)[family]
    assert posed.mean() >= rec["n_away_from_activation_boundary"] / rec["n"] - 0.05, float(posed.mean())      # the recorded well-posed share of the fixture (67 - 100 %) - 5 points
    for comp, cols in comps.items():
        err = e[:, cols].max(axis=1)
        strict = posed.copy()
        for i in KNOWN.get((family, comp), []):
            k = int(np.nonzero(idx == i)[0][0])
            assert err[k] < 1.5e-4, (comp, i, float(err[k]))
            strict[k] = False
        assert err[strict].max() < TOL, (comp, int(idx[np.nonzero(strict)[0][err[strict].argmax()]]), float(err[strict].max()))
        assert err.max() < 0.5, (comp, float(err.max()))
