"""CPU twin of tests/test_gpu_horizons.py: FREE-RUNNING rollouts of the engine SOURCE (fp32 lane emulator, tests/emu -- test infrastructure, never shipped) against the
oracle's recorded rollouts (/root/reference/tests/test_envs.py:62-117 is the reference's seeded-rollout test).  Every 4th fixture snapshot is a start; the emulated world
keeps its own state for 10 env.step() calls on the recorded actions and is compared after 1, 2, 5 and 10 steps.

  * horizons 1 and 2: every well-posed start (tests/tolerance_cases.py::posed_starts: activation gap >= 1e-6 m over the oracle's steps and not one of the starts the measured table lists
    as ill-conditioned for the reference algorithm itself) is within 1e-4 on EVERY component, the touch forces ABSOLUTE -- no allow-list;
  * every horizon: median < 1e-4 and >= 85 % of ALL starts within 1e-4 on every component; the one exception is MEASURED, not listed: a component whose GPU-recorded share at that
    horizon (tests/golden/tolerance_table.json "horizons") is itself below 90 % is held to that share - 5 points (FetchSlide's puck rotation at horizon 10)."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
sys.path.insert(0, os.path.join(HERE, "..", "tools"))
TOL = 1e-4
FAMILIES = ["FetchPush", "FetchPickAndPlace", "FetchSlide", "HandReach", "HandBlock", "HandEgg", "HandPen", "AdroitHammer", "AdroitDoor", "AdroitPen", "AdroitRelocate", "FrankaKitchen",
            "HandBlockTouch"]


@pytest.mark.parametrize("family", FAMILIES)
def test_emulated_free_running_rollout(family):
    import emu_sim
    import emu_tolerances as T

    L = ctypes.CDLL(emu_sim.build())
    L.emu_create.restype = ctypes.c_void_p
    L.emu_create.argtypes = [ctypes.c_void_p] * 3
    import json

    from tolerance_cases import TABLE, posed_starts

    with open(TABLE) as f:
        recorded = json.load(f)["horizons"][family]
    res, comps, g = T.run_family_horizons(L, family, every=4)
    for h, (idx, e, gap) in res.items():
        posed = posed_starts(family, h, idx, gap)
        for comp, cols in comps.items():
            err = (e[:, cols] / np.maximum(1.0, np.abs(g["obs"][idx + h - 1][:, cols]))).max(axis=1) if comp.endswith("_relative") else e[:, cols].max(axis=1)
            if h <= 2 and posed.any():
                assert err[posed].max() < TOL, (h, comp, int(idx[posed][err[posed].argmax()]), float(err[posed].max()))
            share = recorded[str(h)][comp]["frac_within_1e-4"]
            floor = 0.85 if share >= 0.90 else share - 0.05
            assert np.mean(err < TOL) >= floor, (h, comp, float(np.mean(err < TOL)), floor)
            if share >= 0.90:
                assert np.median(err) < TOL, (h, comp, float(np.median(err)))
