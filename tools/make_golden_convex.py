"""Golden fixtures for the families that go through the general convex (MPR) narrow phase: FetchSlide-v4 (cylinder puck on the
box table) and HandManipulateEgg*-v1 (ellipsoid against the hand's capsules and boxes).  Same teacher-forcing layout and the same
oracle as tools/make_golden.py / tools/make_golden_hand.py ("parity unpinned": the oracle restates MuJoCo).

    python tools/make_golden_convex.py
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import OUT, snapshots as fetch_snapshots  # noqa: E402
from make_golden_hand import block_snapshots  # noqa: E402

if __name__ == "__main__":
    d = fetch_snapshots("FetchSlide", 6, True)
    path = os.path.join(OUT, "fetch_FetchSlide_teacher.npz")
    np.savez_compressed(path, **d)
    print("FetchSlide", d["obs"].shape, "max nefc", d["nefc"].max(), "max ncon", d["ncon"].max(), f"{os.path.getsize(path)/1024:.0f} KiB")
    d = block_snapshots("HandManipulateEggRotate-v1", episodes=6, steps=40)
    path = os.path.join(OUT, "hand_EggRotate_teacher.npz")
    np.savez_compressed(path, **d)
    print("HandManipulateEggRotate", d["obs"].shape, "max nefc", d["nefc"].max(), "max ncon", d["ncon"].max(), "reset attempts", d["reset_attempts"],
          f"{os.path.getsize(path)/1024:.0f} KiB")
    d = block_snapshots("HandManipulateEgg_ContinuousTouchSensors-v1", episodes=4, steps=30)
    path = os.path.join(OUT, "hand_Egg_touch_teacher.npz")
    np.savez_compressed(path, **d)
    print("HandManipulateEgg_ContinuousTouchSensors", d["obs"].shape, "steps with active zones", int((d["obs"][:, 61:] > 0).any(axis=1).sum()),
          "max zones", int((d["obs"][:, 61:] > 0).sum(axis=1).max()), f"{os.path.getsize(path)/1024:.0f} KiB")
