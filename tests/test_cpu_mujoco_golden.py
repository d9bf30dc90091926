"""The ORACLE (oracle/grx_oracle.c + the Python task layers) against TRUE-MuJoCo fixtures recorded with the reference itself by tools/record_golden.py
(`tests/golden/mujoco_<id>.npz`).  This is the test that turns "parity unpinned" into "parity pinned": the build image has neither mujoco
nor gymnasium, so the fixtures do not exist yet and the comparison skips -- or FAILS with GRX_REQUIRE_MUJOCO_GOLDEN=1.  Runs on CPU; the HIP path is compared with
the same files in tests/test_gpu_mujoco_golden.py."""
import glob
import os

import numpy as np
import pytest

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
    if env_id.startswith("Fetch"):
        from gymnasium_robotics_amd.envs.fetch import load_fetch_model
        from gymnasium_robotics_amd.envs.fetch_spec import parse_env_id
        from oracle.fetch_oracle import OracleFetchEnv

        task, reward_type = parse_env_id(env_id)
        env = OracleFetchEnv(load_fetch_model(task), task, reward_type)
        env.reset(seed=0)
    elif env_id.startswith(("AntMaze", "PointMaze")):
        from gymnasium_robotics_amd.envs import maze_spec
        from gymnasium_robotics_amd.envs.point_maze import load_point_maze_model
        from oracle.maze_oracle import OracleAntMazeEnv, OraclePointMazeEnv

        ant = env_id.startswith("AntMaze")
        layout, reward_type, _ = (maze_spec.parse_ant_maze_id if ant else maze_spec.parse_point_maze_id)(env_id)
        maze = maze_spec.Maze(maze_spec.MAPS[layout], *((maze_spec.ANT_MAZE_SIZE_SCALING, maze_spec.ANT_MAZE_HEIGHT) if ant else (maze_spec.POINT_MAZE_SIZE_SCALING, maze_spec.POINT_MAZE_HEIGHT)))
        env = (OracleAntMazeEnv if ant else OraclePointMazeEnv)(load_point_maze_model(maze, layout, None, "ant" if ant else "point"), maze, reward_type)
        env.reset(seed=0)
    else:
        pytest.skip(f"{env_id}: compared on the device only (tests/test_gpu_mujoco_golden.py); the oracle-side loader covers the Fetch and maze families")
    s = env.sim
    errs = []
    for i in range(g["obs"].shape[0]):
        s.qpos[:], s.qvel[:], s.qacc_warmstart[:] = g["qpos"][i, :s.nq], g["qvel"][i, :s.nv], g["qacc_ws"][i, :s.nv]
        env.goal = np.array(g["goal"][i], dtype=np.float64)
        if env_id.startswith("Fetch"):
            s.mocap_pos[:], s.mocap_quat[:] = g["mocap"][i, :3], g["mocap"][i, 3:7]
            s.forward()
        # Fetch: _set_action snaps the mocap onto the gripper body's pose of the LAST forward pass (stale by one integration step): the fixture's `aux`
        # (found by the self-check fixtures: fresh kinematics here cost 7e-4 on every FetchPickAndPlace snapshot)
        kw = dict(aux=np.asarray(g["aux"][i][:7], dtype=np.float64)) if env_id.startswith("Fetch") else {}
        obs, r, _, _, info = env.step(np.asarray(g["action"][i], dtype=np.float32), **kw)
        errs.append(np.abs(obs["observation"] - g["obs"][i]).max())
    errs = np.array(errs)
    assert np.quantile(errs, 0.98) < 1e-4 and errs.max() < 5e-3, (env_id, float(np.quantile(errs, 0.98)), float(errs.max()))
