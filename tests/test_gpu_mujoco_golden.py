"""Parity against TRUE MuJoCo fixtures recorded with the reference (tools/record_golden.py).  The build image has no MuJoCo, so the
fixtures do not exist yet and these tests skip; once `tests/golden/mujoco_<id>.npz` are committed they pin the HIP path (and, through
the shared goldens, the oracle) against the reference: observations within 1e-4 (north_star tolerance), flags bit-exact away from
the thresholds."""
import glob
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mujoco_*.npz")))
# self-check twins written by the in-repo oracle in the recorder's exact format (tools/record_selfcheck.py): they pin nothing, they keep the consumers below
# exercised end to end while the real fixtures are absent
SELF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "selfcheck_*.npz")))


def _env_id(path):
    b = os.path.basename(path)
    return b[b.index("_") + 1:-len(".npz")]
REQUIRE = os.environ.get("GRX_REQUIRE_MUJOCO_GOLDEN", "0") not in ("", "0")


def test_mujoco_fixtures_are_present_when_required():
    """GRX_REQUIRE_MUJOCO_GOLDEN=1: the absence of the MuJoCo-recorded fixtures is a FAILURE, not a skip (until they exist every "1e-4" in this
    repo means "against the in-repo fp64 restatement", tools/record_golden.py is the one-command recorder)."""
    if REQUIRE:
        assert FILES, "GRX_REQUIRE_MUJOCO_GOLDEN is set but tests/golden/mujoco_*.npz do not exist: run `python tools/record_golden.py` where mujoco + gymnasium-robotics are installed"
    elif not FILES:
        pytest.skip("no MuJoCo-recorded fixtures committed; parity stays pinned to the in-repo oracle only (set GRX_REQUIRE_MUJOCO_GOLDEN=1 to make this a failure)")


@pytest.mark.parametrize("path", FILES + SELF)
def test_teacher_forced_step_matches_mujoco(path):
    import torch

    from gymnasium_robotics_amd.envs.fetch import FetchVecEnv
    from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv, HandReachVecEnv

    env_id = _env_id(path)
    g = np.load(path)
    assert bytes(g["mujoco_version"]).startswith(b"SELFCHECK") == os.path.basename(path).startswith("selfcheck_")
    n = g["obs"].shape[0]
    if not env_id.startswith(("Fetch", "HandReach", "HandManipulate")):
        return _plain_family(env_id, g, n)
    cls = FetchVecEnv if env_id.startswith("Fetch") else (HandReachVecEnv if env_id.startswith("HandReach") else HandBlockVecEnv)
    env = cls(env_id, num_envs=n, device="cuda:0", output="numpy", autoreset_mode="disabled", max_episode_steps=None)
    env.reset(seed=0)
    put = lambda name, arr: env.load_world_rows({name: arr})      # a MuJoCo-recorded state is in the MJCF's world frame; the device rows live in the model's (mjcf.CompiledModel.origin)
    nq, nv = env.nq, env.nv                       # the device model drops the visual-only target body of the hand-manipulation MJCFs
    put("qpos", g["qpos"][:, :nq]); put("qvel", g["qvel"][:, :nv]); put("qacc_ws", g["qacc_ws"][:, :nv]); put("goal", g["goal"])
    if env_id.startswith("Fetch"):
        put("mocap", g["mocap"]); put("aux", g["aux"])
    obs, r, _, _, info = env.step(g["action"])
    # the tolerance-table policy (tests/test_gpu_tolerance_table.py), with the activation gaps the ORACLE sees when it replays the same snapshots (a MuJoCo-recorded file cannot
    # carry them): every well-posed snapshot within 1e-4 on observation AND achieved goal, at most max(1, 1 %) of all snapshots beyond it
    from mujoco_golden_cases import assert_policy, oracle_replay

    rep = oracle_replay(env_id, g)
    gaps = None if rep is None else rep[1]
    assert_policy(env_id, np.abs(obs["observation"] - g["obs"]).max(axis=1), gaps)
    assert_policy(env_id, np.abs(obs["achieved_goal"] - g["achieved"]).reshape(n, -1).max(axis=1), gaps, "achieved_goal")


def _plain_family(env_id, g, n):
    """Adroit / FrankaKitchen / maze fixtures of tools/record_golden.py:record_plain"""
    import torch

    import gymnasium_robotics_amd as grx

    kw = dict(robot_noise_ratio=0.0, object_noise_ratio=0.0) if env_id.startswith("FrankaKitchen") else {}
    env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="numpy", autoreset_mode="disabled", max_episode_steps=None, **kw)
    env.reset(seed=0)
    put = lambda name, arr: env.load_world_rows({name: arr})
    put("qpos", g["qpos"][:, :env.nq]); put("qvel", g["qvel"][:, :env.nv]); put("qacc_ws", g["qacc_ws"][:, :env.nv])
    if env_id.startswith("AdroitHand"):
        from gymnasium_robotics_amd.envs.adroit_spec import group_shift

        shifts = group_shift(env.model, quat=g["edit"]) if env.task_name == "pen" else group_shift(env.model, pos=g["edit"][:, :3])
        put("shift", shifts)
        if env.target is not None:
            put("target", g["target"])
    elif env_id.startswith("FrankaKitchen"):
        put("last_qpos", g["last_qpos"])
    else:
        put("goal", g["goal"])
    out = env.step(g["action"])
    obs = out[0]["observation"] if isinstance(out[0], dict) else out[0]
    from mujoco_golden_cases import assert_policy, oracle_replay

    rep = oracle_replay(env_id, g)
    assert_policy(env_id, np.abs(obs - g["obs"]).max(axis=1), None if rep is None else rep[1])
