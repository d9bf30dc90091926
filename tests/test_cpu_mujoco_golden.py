"""The ORACLE (oracle/grx_oracle.c + the Python task layers) against TRUE-MuJoCo fixtures recorded with the reference itself by tools/record_golden.py
(`tests/golden/mujoco_<id>.npz`).  This is the test that turns "parity unpinned" into "parity pinned": the build image has neither mujoco
nor gymnasium, so the fixtures do not exist yet and the comparison skips -- or FAILS with GRX_REQUIRE_MUJOCO_GOLDEN=1.  Runs on CPU; the HIP path is compared with
the same files in tests/test_gpu_mujoco_golden.py."""
import glob
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mujoco_*.npz")))
# self-check twins written by the in-repo oracle in the recorder's exact format (tools/record_selfcheck.py): they pin nothing, they keep the consumers below
# exercised end to end while the real fixtures are absent
SELF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "selfcheck_*.npz")))


def _env_id(path):
    b = os.path.basename(path)
    return b[b.index("_") + 1:-len(".npz")]
REQUIRE = os.environ.get("GRX_REQUIRE_MUJOCO_GOLDEN", "0") not in ("", "0")


def test_mujoco_fixtures_are_present_when_required():
    if REQUIRE:
        assert FILES, "GRX_REQUIRE_MUJOCO_GOLDEN is set but tests/golden/mujoco_*.npz do not exist: run `python tools/record_golden.py` where mujoco + gymnasium-robotics are installed"
    elif not FILES:
        pytest.skip("no MuJoCo-recorded fixtures committed: the oracle's physics stays 'parity unpinned' (set GRX_REQUIRE_MUJOCO_GOLDEN=1 to make this a failure)")


@pytest.mark.parametrize("path", FILES + SELF)
def test_oracle_teacher_forced_step_matches_mujoco(path):
    """One env.step() of the oracle from MuJoCo's own pre-step state, compared with what the reference returned (fp64 against fp64: 1e-6 on every observation
    component would be two implementations of the same equations; 1e-4 is north_star's bound)."""
    env_id = _env_id(path)
    g = np.load(path)
    assert bytes(g["mujoco_version"]).startswith(b"SELFCHECK") == os.path.basename(path).startswith("selfcheck_")      # a self-check file can never pass for a MuJoCo fixture
    from mujoco_golden_cases import assert_policy, oracle_replay

    out = oracle_replay(env_id, g)
    if out is None:
        pytest.skip(f"{env_id}: compared on the device only (tests/test_gpu_mujoco_golden.py); the oracle-side loader covers the Fetch, hand and maze families")
    errs, gaps = out
    # same policy as the device-side consumer and tests/test_gpu_tolerance_table.py: well-posed snapshots strictly within north_star's 1e-4, at most 1 % of all beyond it
    assert_policy(env_id, errs, gaps)
    assert errs.max() < 5e-3, (env_id, float(errs.max()))
