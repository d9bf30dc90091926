"""CPU twin of tests/test_gpu_horizons.py: FREE-RUNNING rollouts of the engine SOURCE (fp32 lane emulator, tests/emu -- test infrastructure, never shipped) against the
oracle's recorded rollouts (/root/reference/tests/test_envs.py:62-117 is the reference's seeded-rollout test).  Every 4th fixture snapshot is a start; the emulated world
keeps its own state for 10 env.step() calls on the recorded actions and is compared after 1, 2, 5 and 10 steps.

  * horizons 1 and 2: every start whose oracle steps keep an activation gap >= 1e-6 m is within 1e-4 on every component (HandBlock velocities: the documented 1.5e-4
    snapshot; touch channels relative, >= 75 % within 1e-4 and < 1e-3);
  * every horizon: median < 1e-4 and >= 85 % of ALL starts within 1e-4 on every non-touch component except FetchSlide's puck rotation."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
sys.path.insert(0, os.path.join(HERE, "..", "tools"))
GAP, TOL = 1e-6, 1e-4
KNOWN_1E4 = {("HandBlock", "velocities")}
CHAOTIC = {("FetchSlide", "puck_rotation"), ("FetchSlide", "puck_rot_velocity")}
FAMILIES = ["FetchPush", "FetchPickAndPlace", "FetchSlide", "HandReach", "HandBlock", "HandEgg", "HandPen", "AdroitHammer", "AdroitDoor", "AdroitPen", "AdroitRelocate", "FrankaKitchen",
            "HandBlockTouch"]


@pytest.mark.parametrize("family", FAMILIES)
def test_emulated_free_running_rollout(family):
    import emu_sim
    import emu_tolerances as T

    L = ctypes.CDLL(emu_sim.build())
    L.emu_create.restype = ctypes.c_void_p
    L.emu_create.argtypes = [ctypes.c_void_p] * 3
    res, comps, g = T.run_family_horizons(L, family, every=4)
    for h, (idx, e, gap) in res.items():
        posed = gap >= GAP
        for comp, cols in comps.items():
            err = (e[:, cols] / np.maximum(1.0, np.abs(g["obs"][idx + h - 1][:, cols]))).max(axis=1) if comp.endswith("_relative") else e[:, cols].max(axis=1)
            touch = comp.startswith("touch")
            if h <= 2 and posed.any():
                if touch:
                    assert np.mean(err[posed] < TOL) >= 0.75 and err[posed].max() < 1e-3, (h, comp, float(np.mean(err[posed] < TOL)), float(err[posed].max()))
                else:
                    assert err[posed].max() < (1.5e-4 if (family, comp) in KNOWN_1E4 else TOL), (h, comp, int(idx[posed][err[posed].argmax()]), float(err[posed].max()))
            if not touch and (family, comp) not in CHAOTIC:
                assert np.median(err) < TOL and np.mean(err < TOL) >= 0.85, (h, comp, float(np.median(err)), float(np.mean(err < TOL)))
