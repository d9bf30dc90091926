"""GPU parity tests of the PointMaze path (C ABI grx_point_step) against the fp64 oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(env_id, n, **kw):
    import torch

    from gymnasium_robotics_amd.envs.point_maze import PointMazeVecEnv

    assert torch.cuda.is_available()
    return PointMazeVecEnv(env_id, num_envs=n, device="cuda:0", **kw)


@pytest.mark.parametrize("env_id", ["PointMaze_UMaze-v3", "PointMaze_Large_Diverse_GRDense-v3"])
def test_free_running_rollout_matches_oracle(env_id):
    """Every world is compared with its own single-world oracle env (same seed, same actions) for 120 steps, walls included."""
    from oracle.maze_oracle import OraclePointMazeEnv

    n = 6
    env = _mk(env_id, n)
    obs, info = env.reset(seed=11)
    orcs = [OraclePointMazeEnv(env.model, env.maze, reward_type=env.reward_type) for _ in range(n)]
    for i, o in enumerate(orcs):
        oo, oi = o.reset(seed=11 + i)
        assert np.abs(obs["observation"][i] - oo["observation"]).max() < 1e-6
        assert np.abs(obs["desired_goal"][i] - oo["desired_goal"]).max() < 1e-6 and info["success"][i] == oi["success"]
    rng = np.random.default_rng(0)
    drive = rng.uniform(-1, 1, (n, 2))
    worst, wall_steps = 0.0, 0
    for t in range(120):
        a = np.clip(drive + 0.3 * rng.uniform(-1, 1, (n, 2)), -1, 1).astype(np.float32)
        obs, r, term, trunc, info = env.step(a)
        assert int(np.abs(info["status"]).max()) == 0
        for i, o in enumerate(orcs):
            oo, ro, to, _, io = o.step(a[i].astype(np.float64))
            worst = max(worst, np.abs(obs["observation"][i] - oo["observation"]).max())
            wall_steps += o.sim.nefc > 1
            d = np.linalg.norm(oo["achieved_goal"] - oo["desired_goal"])
            if abs(d - 0.45) > 1e-5:
                assert info["success"][i] == io["success"] and term[i] == to
                assert abs(r[i] - ro) < 1e-5
    assert wall_steps > 50
    assert worst < 1e-4, worst


def test_reward_invariant_termination_and_time_limit():
    env = _mk("PointMaze_Open-v3", 32, continuing_task=False, max_episode_steps=40)
    obs, _ = env.reset(seed=5)
    rng = np.random.default_rng(1)
    saw_term = False
    for t in range(40):
        # steer every world straight at its goal so that some reach it
        dirn = obs["desired_goal"] - obs["achieved_goal"]
        a = np.clip(dirn / (np.linalg.norm(dirn, axis=1, keepdims=True) + 1e-9), -1, 1).astype(np.float32)
        obs, r, term, trunc, info = env.step(a)
        rc = env.compute_reward(obs["achieved_goal"], obs["desired_goal"], {})
        fresh = ~(env._elapsed == 0)  # worlds that were just auto-reset report reward 0
        assert np.array_equal(rc[fresh], r[fresh])  # core.py:59-62 invariant, bit-exact
        d = np.linalg.norm(obs["achieved_goal"].astype(np.float64) - obs["desired_goal"].astype(np.float64), axis=1)      # maze_v4.py:381-388 on the returned goals, fp64
        assert np.array_equal(term[fresh], (d <= 0.45)[fresh])
        assert np.array_equal(env.compute_terminated(obs["achieved_goal"], obs["desired_goal"])[fresh], term[fresh])
        saw_term |= term.any()
    assert saw_term
    with pytest.raises(ValueError, match="Action dimension mismatch"):
        env.step(np.zeros((32, 3), np.float32))


def test_velocity_is_clipped_before_the_step():
    """point.py:57,73-77: qvel is clipped to +-5 BEFORE the physics step"""
    import torch

    env = _mk("PointMaze_Open-v3", 2)
    env.reset(seed=0)
    env.qpos.zero_()
    env.qvel.copy_(torch.tensor([[50.0, -50.0], [1.0, 2.0]], device=env.device))
    obs, *_ = env.step(np.zeros((2, 2), np.float32))
    # damping 1, mass 4.19, dt 0.01: v' = v * (1 - h*d/(m + h*d)) from the clipped value
    k = 1 - 0.01 * 1.0 / (4.18879 + 0.01 * 1.0)
    assert np.allclose(obs["observation"][0, 2:], [5 * k, -5 * k], atol=1e-4)
    assert np.allclose(obs["observation"][1, 2:], [1 * k, 2 * k], atol=1e-4)


def test_ant_maze_teacher_forced_matches_oracle():
    """AntMaze-v5 on the GPU (RK4, capsule/sphere contacts vs floor and walls, joint limits): a batch of snapshots taken
    along an oracle rollout is stepped once from the oracle's own pre-step states."""
    import torch

    from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv
    from oracle.maze_oracle import OracleAntMazeEnv

    n = 96
    env = AntMazeVecEnv("AntMaze_UMaze-v5", num_envs=n, device="cuda:0", autoreset_mode="disabled", max_episode_steps=None)
    obs0, info0 = env.reset(seed=3)
    assert obs0["observation"].shape == (n, 27) and obs0["achieved_goal"].shape == (n, 2)
    orc = OracleAntMazeEnv(env.model, env.maze)
    o, _ = orc.reset(seed=3)
    assert np.abs(obs0["observation"][0] - o["observation"]).max() < 1e-6 and np.abs(obs0["desired_goal"][0] - o["desired_goal"]).max() < 1e-6
    rng = np.random.default_rng(0)
    pre_q, pre_v, pre_w, acts, exp_obs, exp_ag = [], [], [], [], [], []
    drive = rng.uniform(-1, 1, 8)
    for t in range(n):
        if t % 24 == 0:
            drive = rng.uniform(-1, 1, 8)
        a = np.clip(drive + 0.5 * rng.uniform(-1, 1, 8), -1, 1).astype(np.float32)
        s = orc.sim
        pre_q.append(s.qpos.copy()); pre_v.append(s.qvel.copy()); pre_w.append(s.qacc_warmstart.copy()); acts.append(a)
        o, r, te, tr, info = orc.step(a.astype(np.float64))
        exp_obs.append(o["observation"]); exp_ag.append(o["achieved_goal"])
        assert s.bad_state == 0
    f = lambda x: torch.from_numpy(np.asarray(x, dtype=np.float32)).cuda()
    env.qpos.copy_(f(pre_q)); env.qvel.copy_(f(pre_v)); env.qacc_ws.copy_(f(pre_w))
    obs, r, term, trunc, info = env.step(np.asarray(acts))
    assert int(np.abs(info["status"]).max()) == 0
    err = np.maximum(np.abs(obs["observation"] - np.asarray(exp_obs)).max(axis=1), np.abs(obs["achieved_goal"] - np.asarray(exp_ag)).max(axis=1))
    print("ant teacher-forced err p50 %.2e p90 %.2e max %.2e" % tuple(np.quantile(err, [0.5, 0.9, 1.0])))
    assert err.max() < 1e-4   # every snapshot (measured max 1.2e-5, tests/golden/tolerance_table.json)


def test_ant_maze_large_teacher_forced_matches_golden():
    """BASELINE.json configs[3] on ITS OWN kernel instantiation (76 geoms, 819 wall pairs through the wall lattice): the 240 snapshots of
    tests/golden/ant_Large_teacher.npz (tools/make_golden_antmaze.py: the ant pushed against one wall or into a corner of its cell, 99 snapshots with
    wall contacts in some substep) stepped once from the oracle's own pre-step states.  Reference: __init__.py:936-958, maze_v4.py:179-212,
    ant_maze_v5.py:295-320."""
    import torch

    from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ant_Large_teacher.npz"))
    n = g["obs"].shape[0]
    env = AntMazeVecEnv("AntMaze_Large_Diverse_GR-v5", num_envs=n, device="cuda:0", autoreset_mode="disabled", max_episode_steps=None)
    env.reset(seed=0)
    env.load_world_rows({k: g[k] for k in ("qpos", "qvel", "qacc_ws", "goal")})
    obs, r, term, trunc, info = env.step(g["action"])
    assert int(np.abs(info["status"]).max()) == 0
    e = np.abs(obs["observation"] - g["obs"])
    pe = np.maximum(e[:, :13].max(axis=1), np.abs(obs["achieved_goal"] - g["achieved"]).max(axis=1))
    ve = e[:, 13:].max(axis=1) / np.maximum(1.0, np.abs(g["obs"][:, 13:]).max(axis=1))   # the kicked ants move at up to 16 m/s
    wall = g["wall_contact_substeps"] > 0
    print("AntMaze_Large teacher-forced: positions p50 %.1e max %.1e (wall snapshots max %.1e), velocities (relative) p50 %.1e max %.1e, %d wall snapshots"
          % (np.median(pe), pe.max(), pe[wall].max(), np.median(ve), ve.max(), wall.sum()))
    assert wall.sum() >= 90
    assert pe.max() < 1e-4 and ve.max() < 1e-4              # north_star's bound on every snapshot
    assert np.array_equal(info["success"], g["success"]) and np.array_equal(r, g["reward"].astype(r.dtype))
    assert np.abs(obs["desired_goal"] - g["goal"]).max() < 1e-6
    env.close()


def test_ant_maze_large_runs_and_flags():
    """AntMaze_Large_Diverse_GR-v5 (BASELINE.json configs[3]): 819 candidate pairs, 8 combined goal/reset cells."""
    from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv

    env = AntMazeVecEnv("AntMaze_Large_Diverse_GR-v5", num_envs=256, device="cuda:0")
    assert env.max_episode_steps == 1000
    obs, info = env.reset(seed=0)
    assert not info["success"].any()
    rng = np.random.default_rng(1)
    for t in range(10):
        obs, r, term, trunc, info = env.step(rng.uniform(-1, 1, (256, 8)).astype(np.float32))
    assert int(np.abs(info["status"]).max()) == 0 and np.isfinite(obs["observation"]).all()
    assert not term.any() and not trunc.any()
    d = np.linalg.norm(obs["achieved_goal"] - obs["desired_goal"], axis=1)
    assert np.array_equal(info["success"], d <= 0.45)
    assert 0.2 < obs["observation"][:, 0].mean() < 0.9  # torso height stays physical


def test_reset_target_redraws_goal_like_the_oracle():
    """reset_target=True (maze_v4.py:400-418): the step that reaches the goal still reports the old goal, the next one a new goal;
    world-by-world identical to the oracle env driven with the same seed and actions."""
    from oracle.maze_oracle import OraclePointMazeEnv

    n = 4
    env = _mk("PointMaze_Large_Diverse_GR-v3", n, reset_target=True, max_episode_steps=None)
    obs, _ = env.reset(seed=3)
    orcs = [OraclePointMazeEnv(env.model, env.maze, reward_type=env.reward_type, reset_target=True) for _ in range(n)]
    for i, o in enumerate(orcs):
        o.reset(seed=3 + i)
    changed = 0
    for t in range(400):
        d = obs["desired_goal"] - np.stack([o.sim.qpos[:2] for o in orcs])
        act = np.clip(d * 3.0, -1, 1).astype(np.float32)            # drive straight at the goal (walls permitting)
        prev_goal = obs["desired_goal"].copy()
        import torch
        for name in ("qpos", "qvel", "qacc_ws"):   # teacher forcing: every step starts from the oracle's state
            src = np.stack([getattr(o.sim, "qacc_warmstart" if name == "qacc_ws" else name) for o in orcs]).astype(np.float32)
            getattr(env, name).copy_(torch.from_numpy(src).to(env.device))
        obs, r, term, trunc, info = env.step(act)
        assert np.array_equal(obs["desired_goal"], prev_goal) or t > 0   # the reaching step reports the goal it reached
        for i, o in enumerate(orcs):
            oo, orr, _, _, oi = o.step(act[i])
            assert np.abs(obs["observation"][i] - oo["observation"]).max() < 2e-3
            assert np.abs(obs["desired_goal"][i] - oo["desired_goal"]).max() < 1e-6
            assert bool(info["success"][i]) == oi["success"] or abs(np.linalg.norm(oo["achieved_goal"] - oo["desired_goal"]) - 0.45) < 1e-4
            if oi["success"]:
                changed += 1
                assert np.abs(env.goal[i].double().cpu().numpy() - o.goal).max() < 1e-6 and np.linalg.norm(o.goal - oo["achieved_goal"]) > 0.45
        assert not term.any()
    assert changed >= 2
    env.close()


@pytest.mark.parametrize("env_id,output", [("PointMaze_UMaze-v3", "torch"), ("AntMaze_UMaze-v5", "torch"), ("AntMaze_Medium_Diverse_GR-v5", "numpy")])
def test_same_step_autoreset_matches_next_step(env_id, output):
    """the reset kernel of the same-step path (grx_maze_reset_rows, pinned staging) against the next-step path: bit-equal (see tests/autoreset_cases.py)"""
    import gymnasium_robotics_amd as grx
    from autoreset_cases import check_same_step_against_next_step

    n = 48
    make = lambda **kw: grx.make_vec(env_id, num_envs=n, device="cuda:0", **kw)
    A, B = check_same_step_against_next_step(make, horizon=6, steps=14, act_dim=2 if "Point" in env_id else 8, output=output)
    # the packed rows (cross-rank gather, HER) of the rows reset in the last step carry the finished episode's reward / success and the NEW goal
    pk, goal = A.packed.cpu().numpy(), A.goal.cpu().numpy()
    od = A.obs_dim
    assert np.array_equal(pk[:, od + 2: od + 4], goal) and np.array_equal(pk[:, :od], A.obs.cpu().numpy())


@pytest.mark.parametrize("env_id", ["PointMaze_UMaze-v3", "PointMaze_Medium_Diverse_GR-v3", "AntMaze_Large_Diverse_GR-v5", "AntMaze_Open_Diverse_G-v5"])
def test_device_reset_draws_equal_numpy_bit_for_bit(env_id):
    """MazeEnv.reset's draws are made ON THE DEVICE (grx_maze_sample_resets_device): goal cell and reset cell through Generator.integers (32-bit Lemire on the buffered halves of
    the 64-bit outputs), the rejection loop of generate_reset_pos, the four uniform noise draws (maze_v4.py:299-358).  Over several ragged reset lists and with the
    options of reset(), start / goal rows equal the host routine fed by numpy generators (maze_spec.sample_maze_reset = the reference's draw order, pinned by the reference's
    own goldens in tests/test_cpu_maze.py) rounded to float32, and every world's stream -- position AND the buffered 32-bit half -- ends where numpy's does."""
    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.core import np_random
    from gymnasium_robotics_amd.envs.maze_spec import sample_maze_reset

    n = 48
    env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="numpy")
    assert env._device_draws
    seeds = [500 + 3 * i for i in range(n)]
    rngs = [np_random(s)[0] for s in seeds]
    env.reset(seed=seeds)
    free = [(i, j) for i, row in enumerate(env.maze.maze_map) for j, c in enumerate(row) if c != 1]
    # (a goal cell that is the maze's ONLY reset cell would loop forever in the reference's generate_reset_pos -- and in sample_maze_reset: not a case to test)
    far_goal = [c for c in free if len(env.maze.unique_reset_locations) > 1 or np.abs(env.maze.cell_rowcol_to_xy(c) - env.maze.unique_reset_locations[0]).max() > 1e-9]
    calls = [(np.arange(n), None), (np.arange(0, n, 5), None), (np.array([n - 1, 2]), None), (np.arange(n), {"goal_cell": np.array(far_goal[0])}),
             (np.arange(1, n, 2), {"reset_cell": np.array(free[-1])}), (np.arange(n)[::-1].copy(), {"goal_cell": np.array(far_goal[-1]), "reset_cell": np.array(free[0])}), (np.arange(n), None)]
    skip = env.OBS_SKIP
    for c, (idx, options) in enumerate(calls):
        if c:
            torch.cuda.set_sync_debug_mode("error")      # only enqueued: index list through pinned memory, draws by a kernel
            try:
                env._reset_worlds(idx, options)
            finally:
                torch.cuda.set_sync_debug_mode("default")
        goal, qpos = env.goal.cpu().numpy(), env.qpos.cpu().numpy()
        for w in idx:
            g_ref, s_ref = sample_maze_reset(env.maze, rngs[w], env.position_noise_range, options)
            assert np.array_equal(goal[w], g_ref.astype(np.float32)), (c, w)
            assert np.array_equal(qpos[w, :2], s_ref.astype(np.float32)), (c, w)
    for w in range(n):
        a, b = env.world_rng(w).bit_generator.state, rngs[w].bit_generator.state
        assert a["state"] == b["state"] and a["has_uint32"] == b["has_uint32"] and (not a["has_uint32"] or a["uinteger"] == b["uinteger"]), w
    env.close()


@pytest.mark.parametrize("env_id,parts", [("AntMaze_Large_Diverse_GR-v5", 5), ("AntMaze_UMaze-v5", 2), ("AntMaze_Medium-v5", 3), ("PointMaze_Large_Diverse_GR-v3", 2)])
def test_split_step_is_the_plain_step(monkeypatch, env_id, parts):
    """Round 6: the maze step launch with P workgroups per world, each running its share of the frame_skip substeps and handing the world on through its own state row (include/grx_capi.h
    grx_point_buffers.split_parts), against the plain launch: state rows, observations, achieved goals, rewards, flags, status words and packed rows are BIT-IDENTICAL after every
    step -- RK4 ants (the four stages of a substep stay in one part; 5 substeps in 2, 3 and 5 uneven shares) and the Euler point (velocity clip once per step), same-step autoresets
    at a short time limit included.  The reference's step is one env.step() whatever the launch geometry (/root/reference/gymnasium_robotics/envs/maze/ant_maze_v5.py:295-310)."""
    import torch

    import gymnasium_robotics_amd as grx

    n, envs = 2048, []
    for p_ in (1, parts):
        monkeypatch.setenv("GRX_MAZE_SPLIT", str(p_))
        e = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=7)
        e.reset(seed=5)
        envs.append(e)
    plain, split = envs
    assert plain._split == 1 and split._split == min(parts, split.N_SUBSTEPS)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(9)
    act_dim = plain.single_action_space.shape[0]
    for t in range(18):
        a = torch.rand(n, act_dim, device="cuda:0", generator=gen) * 2 - 1
        outs = [e.step(a) for e in envs]
        for name in ("qpos", "qvel", "qacc_ws", "obs", "achieved", "reward", "success", "terminated", "status", "packed", "goal"):
            assert torch.equal(getattr(split, name), getattr(plain, name)), (t, name, int((getattr(split, name) != getattr(plain, name)).sum()))
        assert torch.equal(torch.as_tensor(outs[0][3]), torch.as_tensor(outs[1][3]))
        assert split._split == 1 or int(split._split_state.abs().max()) == 0, t      # every world's words are clean again (the point's single substep cannot be split: one part)
    assert int((split.status & 1).max()) == 0 and torch.isfinite(split.qpos).all()
