"""CPU tests of the Maze family: the reference's own golden vectors for reset (RNG draw order + cell -> xy mapping),
reset properties, registry, and the PointMaze kernel source (lane emulator) against the fp64 oracle."""
import os

import numpy as np
import pytest

from gymnasium_robotics_amd.core import np_random
from gymnasium_robotics_amd.envs.maze_spec import MAPS, Maze, parse_point_maze_id, sample_maze_reset
from gymnasium_robotics_amd.mjcf import load_model

MODELS = os.path.join(os.path.dirname(__file__), "..", "gymnasium_robotics_amd", "models")


def test_reference_golden_reset_cell():
    """/root/reference/tests/envs/maze/test_point_maze.py:20-31: reset(seed=42, options={"reset_cell": [1, 2]}) -> obs [0.67929896, 0.59868401, 0, 0]"""
    maze = Maze([[1, 1, 1, 1], [1, "r", "r", 1], [1, "r", "g", 1], [1, 1, 1, 1]], 1.0, 0.4)
    goal, reset_pos = sample_maze_reset(maze, np_random(42)[0], 0.25, {"reset_cell": [1, 2]})
    np.testing.assert_almost_equal(np.array([0.67929896, 0.59868401]), reset_pos, decimal=7)


def test_reference_golden_goal_cell():
    """test_point_maze.py:34-45: reset(seed=42, options={"goal_cell": [2, 1]}) -> desired_goal [-0.36302198, -0.53056078]"""
    maze = Maze([[1, 1, 1, 1], [1, "r", "g", 1], [1, "g", "g", 1], [1, 1, 1, 1]], 1.0, 0.4)
    goal, reset_pos = sample_maze_reset(maze, np_random(42)[0], 0.25, {"goal_cell": [2, 1]})
    np.testing.assert_almost_equal(np.array([-0.36302198, -0.53056078]), goal, decimal=7)


def test_reset_never_starts_inside_goal_radius():
    """test_point_maze.py:9-17 / test_ant_maze.py:12-21: 1000 resets, distance to the goal always > 0.45"""
    maze = Maze(MAPS["UMaze"], 1.0, 0.4)
    rng = np_random(0)[0]
    for _ in range(1000):
        goal, pos = sample_maze_reset(maze, rng, 0.25)
        assert np.linalg.norm(pos - goal) > 0.45
    with pytest.raises(AssertionError):
        sample_maze_reset(maze, rng, 0.25, {"goal_cell": [0, 0]})


def test_registry_and_wall_layouts():
    assert parse_point_maze_id("PointMaze_UMaze-v3") == ("UMaze", "sparse", 300)
    assert parse_point_maze_id("PointMaze_Medium_Diverse_GRDense-v3") == ("Medium_Diverse_GR", "dense", 600)
    assert parse_point_maze_id("PointMaze_Large_Diverse_G-v3")[2] == 800
    with pytest.raises(KeyError):
        parse_point_maze_id("PointMaze_Huge-v3")
    for base in ("UMaze", "Open", "Medium", "Large"):
        m = load_model(os.path.join(MODELS, f"point_{base}.npz"))
        maze = Maze(MAPS[base], 1.0, 0.4)
        assert m.dim("ngeom") == len(maze.walls) + 2 and m.dim("npair") == len(maze.walls) + 1  # + ground plane, particle
        assert (m.dim("nq"), m.dim("nv"), m.dim("nu")) == (2, 2, 2) and m.info["unsupported_pairs"] == 0
        for v in ("_Diverse_G", "_Diverse_GR"):  # same walls, different goal/reset cells
            if base + v in MAPS:
                assert [(i, j) for i, j, _ in Maze(MAPS[base + v], 1.0, 0.4).walls] == [(i, j) for i, j, _ in maze.walls]
    d = Maze(MAPS["Large_Diverse_GR"], 1.0, 0.4)
    assert len(d.unique_goal_locations) == 8 and len(d.unique_reset_locations) == 8  # 8 combined cells


def test_emulated_point_kernel_matches_oracle():
    from emu_sim import EmuSim

    from gymnasium_robotics_amd import _native
    from oracle.maze_oracle import OraclePointMazeEnv

    model = load_model(os.path.join(MODELS, "point_UMaze.npz"))
    maze = Maze(MAPS["UMaze"], 1.0, 0.4)
    env = OraclePointMazeEnv(model, maze)
    emu = EmuSim(model, _native.PointTaskStruct(1, 1, 1, 0, 0.45, 5.0))
    rng = np.random.default_rng(3)
    worst, hits = 0.0, 0
    for ep in range(3):
        obs, _ = env.reset(seed=ep)
        emu.qpos[:], emu.qvel[:], emu.qacc_ws[:] = obs["observation"][:2], 0, 0
        drive = rng.uniform(-1, 1, 2)
        for t in range(150):
            a = np.clip(drive + 0.3 * rng.uniform(-1, 1, 2), -1, 1).astype(np.float32)  # persistent push: runs into walls
            obs, r, te, tr, info = env.step(a.astype(np.float64))
            emu.point_step(a)
            assert emu.status.value == 0
            worst = max(worst, np.abs(emu.obs - obs["observation"]).max())
            hits += env.sim.nefc > 1
    assert hits > 20, "the rollout never touched a wall"
    assert worst < 1e-4, worst


def test_emulated_ant_kernel_matches_oracle_teacher_forced():
    """AntMaze step (RK4, capsule/sphere contacts, joint limits) through the lane emulator vs the oracle, every step
    restarted from the oracle's state (legged contact dynamics are chaotic)."""
    from emu_sim import EmuSim

    from gymnasium_robotics_amd import _native
    from oracle.maze_oracle import OracleAntMazeEnv

    model = load_model(os.path.join(MODELS, "ant_UMaze.npz"))
    maze = Maze(MAPS["UMaze"], 4.0, 0.5)
    env = OracleAntMazeEnv(model, maze)
    emu = EmuSim(model, _native.PointTaskStruct(5, 1, 1, 1, 0.45, 5.0))
    rng = np.random.default_rng(0)
    errs = []
    obs, _ = env.reset(seed=1)
    for t in range(60):
        a = rng.uniform(-1, 1, 8).astype(np.float32)
        s = env.sim
        emu.qpos[:], emu.qvel[:], emu.qacc_ws[:] = s.qpos, s.qvel, s.qacc_warmstart
        obs, r, te, tr, info = env.step(a.astype(np.float64))
        emu.point_step(a)
        assert emu.status.value == 0 and s.bad_state == 0
        errs.append(max(np.abs(emu.obs - obs["observation"]).max(), np.abs(emu.achieved[:2] - obs["achieved_goal"]).max()))
    errs = np.array(errs)
    # measured on this rollout: p50 1.4e-6, p90 3.2e-6, max 2.3e-5 (RK4, contacts and limits included): every step inside the 1e-4 of north_star
    assert errs.max() < 1e-4 and np.median(errs) < 5e-6, (np.quantile(errs, [0.5, 0.9, 1.0]))


def test_emulated_ant_large_kernel_matches_golden_with_wall_contacts():
    """The AntMaze_Large model (BASELINE configs[3]: wall lattice, 76 geoms) through the lane emulator on the wall-contact fixture of
    tools/make_golden_antmaze.py; the same fixture is stepped on the MI355X in tests/test_gpu_maze.py."""
    from emu_sim import EmuSim

    from gymnasium_robotics_amd import _native

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ant_Large_teacher.npz"))
    emu = EmuSim(load_model(os.path.join(MODELS, "ant_Large.npz")), _native.PointTaskStruct(5, 1, 1, 1, 0.45, 5.0))
    pe, ve = [], []
    for i in range(0, len(g["obs"]), 2):
        emu.qpos[:], emu.qvel[:], emu.qacc_ws[:] = g["qpos"][i], g["qvel"][i], g["qacc_ws"][i]
        emu.point_step(g["action"][i])
        assert emu.status.value == 0
        e = np.abs(emu.obs - g["obs"][i])
        pe.append(max(e[:13].max(), np.abs(emu.achieved[:2] - g["achieved"][i]).max()))
        ve.append(e[13:].max() / max(1.0, np.abs(g["obs"][i, 13:]).max()))
    assert (g["wall_contact_substeps"][::2] > 0).sum() >= 40
    assert max(pe) < 1e-4 and max(ve) < 1e-4, (max(pe), max(ve))     # measured: 2.9e-6 / 2.4e-5


def test_redraw_goal_draw_order_and_contract():
    """MazeEnv.update_goal (maze_v4.py:400-418): goal cell index, x noise, y noise per attempt, until farther than 0.45."""
    from gymnasium_robotics_amd.core import np_random
    from gymnasium_robotics_amd.envs.maze_spec import GOAL_RADIUS, MAPS, POINT_MAZE_HEIGHT, POINT_MAZE_SIZE_SCALING, Maze, redraw_goal

    maze = Maze(MAPS["Large_Diverse_GR"], POINT_MAZE_SIZE_SCALING, POINT_MAZE_HEIGHT)
    assert len(maze.unique_goal_locations) > 1
    a, b = np_random(9)[0], np_random(9)[0]
    at = maze.unique_goal_locations[0].copy()
    new = redraw_goal(maze, a, at, at, 0.25)
    assert np.linalg.norm(new - at) > GOAL_RADIUS
    g = at.copy()
    while np.linalg.norm(at - g) <= GOAL_RADIUS:
        g = maze.unique_goal_locations[b.integers(low=0, high=len(maze.unique_goal_locations))].copy()
        g[0] += b.uniform(low=-0.25, high=0.25) * maze.maze_size_scaling
        g[1] += b.uniform(low=-0.25, high=0.25) * maze.maze_size_scaling
    assert np.array_equal(new, g) and a.uniform() == b.uniform()
    single = Maze(MAPS["UMaze"], POINT_MAZE_SIZE_SCALING, POINT_MAZE_HEIGHT)   # one goal cell only: nothing to redraw
    if len(single.unique_goal_locations) <= 1:
        assert np.array_equal(redraw_goal(single, a, at, at, 0.25), at)


def test_map_tables_equal_the_reference_module():
    """tests/golden/ref_maze_maps.json was written by importing the reference's own maps.py (tools/make_reference_vectors.py):
    every registered layout of this package is the same table, marker for marker."""
    import json

    from gymnasium_robotics_amd.envs import maze_spec

    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_maze_maps.json")) as f:
        ref = json.load(f)
    assert (maze_spec.R, maze_spec.G, maze_spec.C) == (ref["markers"]["R"], ref["markers"]["G"], ref["markers"]["C"])
    assert set(ref["maps"]) == set(maze_spec.MAPS)
    for name, table in ref["maps"].items():
        assert maze_spec.MAPS[name] == table, name


def test_degenerate_map_guard_refuses_only_a_coinciding_reset_and_goal_cell():
    """the reference's generate_reset_pos (/root/reference/gymnasium_robotics/envs/maze/maze_v4.py:284-297) redraws the reset cell centre until it is farther than
    half a cell from the noisy goal: with ONE cell that is both the only reset and the only goal cell it never returns (refused: ValueError); an 'r' cell NEXT to a 'g' cell is
    a valid maze (centre to noisy goal >= 0.75 cells) and must get past the map checks -- on this GPU-less machine that means reaching the 'no HIP device' error, not ValueError."""
    import torch

    from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv, PointMazeVecEnv

    if torch.cuda.is_available():
        pytest.skip("the map checks are exercised through the constructor's GPU-less error path")
    adjacent = [[1, 1, 1, 1], [1, "r", "g", 1], [1, 1, 1, 1]]
    single = [[1, 1, 1], [1, "c", 1], [1, 1, 1]]
    for cls, env_id in ((PointMazeVecEnv, "PointMaze_UMaze-v3"), (AntMazeVecEnv, "AntMaze_UMaze-v5")):
        with pytest.raises(RuntimeError, match="no HIP device"):
            cls(env_id, num_envs=2, maze_map=adjacent)
        with pytest.raises(ValueError, match="single reset cell"):
            cls(env_id, num_envs=2, maze_map=single)
    # the geometry the guard relies on: distinct lattice cells are at least one pitch apart
    m = Maze(adjacent, 1.0, 0.4)
    ur, ug = np.asarray(m.unique_reset_locations).reshape(-1, 2), np.asarray(m.unique_goal_locations).reshape(-1, 2)
    assert np.linalg.norm(ur[0] - ug[0]) >= 1.0 - 1e-12
