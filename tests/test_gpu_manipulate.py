"""GPU parity tests of the HandManipulateBlock family (grx_hand_step_kernel, kind 1) against the oracle's golden fixture."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "hand_BlockRotateXYZ_teacher.npz")


@pytest.fixture(scope="module")
def env_and_golden():
    import torch

    from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv

    g = np.load(GOLDEN)
    env = HandBlockVecEnv("HandManipulateBlockRotateXYZ-v1", num_envs=g["obs"].shape[0], device="cuda:0", output="numpy", autoreset_mode="disabled",
                          max_episode_steps=None)
    return env, g, torch


def test_teacher_forced_step_matches_golden(env_and_golden):
    env, g, torch = env_and_golden
    env.reset(seed=0)
    env.load_world_rows({k: g[k] for k in ("qpos", "qvel", "qacc_ws", "goal")})
    obs, r, term, trunc, info = env.step(g["action"])
    assert int(info["status"].max()) == 0
    e = np.abs(obs["observation"] - g["obs"])
    pe, ve = np.maximum(e[:, :24].max(axis=1), e[:, 54:].max(axis=1)), e[:, 24:54].max(axis=1)
    far = g["activation_gap"] >= 1e-6
    # every snapshot's positions inside north_star's 1e-4 (measured max 2e-6); velocities (up to 20 rad/s): inside on every snapshot away from an activation boundary
    # but one at 1.05e-4 (snapshot 112), 99.6 % of all
    assert pe.max() < 1e-4 and ve.max() < 2e-4 and np.mean(ve < 1e-4) >= 0.99 and np.mean(ve[far] < 1e-4) >= 0.99, (float(pe.max()), float(ve.max()))
    assert np.median(pe) < 1e-6 and np.median(ve) < 5e-5
    from gymnasium_robotics_amd.envs.manipulate_spec import block_goal_distance
    _, d_rot = block_goal_distance(g["achieved"], g["goal"], "ignore", "xyz")
    clear = np.abs(d_rot - 0.1) > 1e-3
    assert np.array_equal(r[clear], g["reward"][clear].astype(np.float32))
    assert np.array_equal(info["is_success"][clear], g["success"][clear].astype(np.float32))
    assert not term.any() and not trunc.any()


def test_reset_settles_block_on_palm_like_the_oracle(env_and_golden):
    env, g, _ = env_and_golden
    obs, _ = env.reset(seed=0)
    k = len(g["reset_seed"])
    assert (env.reset_attempts[:k] == g["reset_attempts"]).all()
    assert (obs["observation"][:, 56] > 0.04).all()                       # every world ends with the block on the palm
    # 200 fp32 substeps of contact-rich settling vs the fp64 oracle: same resting pose to a fraction of a millimetre / degree
    assert np.abs(obs["observation"][:k, 54:57] - g["reset_obs"][:, 54:57]).max() < 1e-3
    qa, qb = obs["observation"][:k, 57:61], g["reset_obs"][:, 57:61]
    assert (2 * np.arccos(np.clip(np.abs((qa * qb).sum(axis=1)), 0, 1))).max() < 2e-2
    assert np.abs(obs["desired_goal"][:k, 3:] - g["reset_goal"][:, 3:]).max() < 1e-6   # goal rotation is pure RNG (same draws)
    assert np.array_equal(obs["desired_goal"][:, :3].astype(np.float32), obs["achieved_goal"][:, :3].astype(np.float32))  # 'ignore': goal position = settled position


def test_step_reward_equals_compute_reward_bitwise(env_and_golden):
    env, g, torch = env_and_golden
    env.reset(seed=2)
    rng = np.random.default_rng(0)
    for _ in range(3):
        obs, r, _, _, info = env.step(rng.uniform(-1, 1, (env.num_envs, 20)).astype(np.float32))
    r2 = env.compute_reward(obs["achieved_goal"].astype(np.float32), obs["desired_goal"].astype(np.float32), info)
    assert np.array_equal(r, r2) and r.dtype == np.float32 and set(np.unique(r)) <= {-1.0, 0.0}
    q = torch.nn.functional.normalize(torch.randn(5, 9, 4, device=env.device), dim=-1)
    pose = torch.cat([torch.zeros(5, 9, 3, device=env.device), q], dim=-1)
    assert float(env.compute_reward(pose, pose, None).abs().max()) == 0.0 and tuple(env.compute_reward(pose, pose, None).shape) == (5, 9)
    with pytest.raises(ValueError):
        env.compute_reward(np.zeros((4, 15), np.float32), np.zeros((4, 15), np.float32), None)


def test_variants_and_truncation():
    from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv

    for env_id in ("HandManipulateBlockRotateZ-v1", "HandManipulateBlockRotateParallel-v1", "HandManipulateBlockFullDense-v1"):
        env = HandBlockVecEnv(env_id, num_envs=4, device="cuda:0", output="numpy", max_episode_steps=3)
        obs, _ = env.reset(seed=7)
        assert obs["observation"].shape == (4, 61) and obs["desired_goal"].shape == (4, 7)
        if "Full" in env_id:
            off = obs["desired_goal"][:, :3] - obs["achieved_goal"][:, :3]
            assert (off[:, 2] >= -1e-6).all() and (off[:, 2] <= 0.06 + 1e-6).all() and np.abs(off).max() > 1e-3
        for t in range(3):
            obs, r, term, trunc, info = env.step(np.zeros((4, 20), np.float32))
        assert trunc.all() and not term.any() and int(info["status"].max()) == 0
        assert (r.dtype == np.float64) == ("Dense" in env_id)
        env.close()


def test_touch_sensor_variants_match_golden():
    import torch

    from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hand_BlockRotateXYZ_touch_teacher.npz"))
    n = g["obs"].shape[0]
    outs = {}
    for env_id in ("HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", "HandManipulateBlockRotateXYZ_BooleanTouchSensors-v1"):
        env = HandBlockVecEnv(env_id, num_envs=n, device="cuda:0", output="numpy", autoreset_mode="disabled", max_episode_steps=None)
        obs, _ = env.reset(seed=0)
        assert obs["observation"].shape == (n, 153)                                    # 61 + 92 (SURVEY.md 8(d) cfg 3)
        assert (obs["observation"][:, 61:] > 0).any(axis=1).all()                       # a settled block presses on some zone
        env.load_world_rows({k: g[k] for k in ("qpos", "qvel", "qacc_ws", "goal")})
        obs, r, _, _, info = env.step(g["action"])
        assert int(info["status"].max()) == 0
        outs[env_id] = obs["observation"]
        env.close()
    cont, boolean = outs["HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1"], outs["HandManipulateBlockRotateXYZ_BooleanTouchSensors-v1"]
    ref = g["obs"][:, 61:]
    far = g["activation_gap"] >= 1e-6
    assert np.array_equal(cont[far, 61:] > 0, ref[far] > 0)
    scale = np.maximum(1.0, ref.max(axis=1, keepdims=True))
    assert (np.abs(cont[:, 61:] - ref) / scale)[far].max() < 2e-3
    assert np.array_equal(boolean[:, 61:], (cont[:, 61:] > 0).astype(np.float64))
    assert np.abs(cont[:, :61] - boolean[:, :61]).max() == 0.0
    assert (ref[far] > 0).sum() > 40
    assert np.median((np.abs(cont[:, 61:] - ref) / scale).max(axis=1)) < 1e-4   # all snapshots, including the ones next to an activation threshold


def test_pen_variant_matches_golden_and_reference_distance():
    import torch

    from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv
    from gymnasium_robotics_amd.envs.manipulate_spec import block_goal_distance

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hand_PenRotate_teacher.npz"))
    n = g["obs"].shape[0]
    env = HandBlockVecEnv("HandManipulatePenRotate-v1", num_envs=n, device="cuda:0", output="numpy", autoreset_mode="disabled", max_episode_steps=None)
    obs, _ = env.reset(seed=0)
    k = len(g["reset_seed"])
    assert (env.reset_attempts[:k] == g["reset_attempts"]).all() and (obs["observation"][:, 56] > 0.04).all()
    assert np.abs(obs["observation"][:k, 54:57] - g["reset_obs"][:, 54:57]).max() < 2e-3
    env.load_world_rows({name: g[name] for name in ("qpos", "qvel", "qacc_ws", "goal")})
    obs, r, _, _, info = env.step(g["action"])
    assert int(info["status"].max()) == 0
    e = np.abs(obs["observation"] - g["obs"])
    pe = np.maximum(e[:, :24].max(axis=1), e[:, 54:].max(axis=1))
    far = g["activation_gap"] >= 1e-6
    assert pe[far].max() < 1e-4 and pe.max() < 5e-3 and np.median(pe) < 1e-5
    # the device's ignore-z goal distance against vectors produced by the reference's own rotations.py (dense reward = -d_rot here)
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_rotations.npz"))
    pose = lambda q: np.concatenate([np.zeros((len(q), 3)), q], axis=1).astype(np.float32)
    dense = HandBlockVecEnv("HandManipulatePenRotateDense-v1", num_envs=1, device="cuda:0")
    d_dev = -dense.compute_reward(pose(ref["qa"]), pose(ref["qb"]), None)
    d_ref = ref["angle_diff_ignore_z"]
    wrap = np.minimum(np.abs(d_dev - d_ref), np.abs(np.abs(d_dev - d_ref) - 2 * np.pi))   # identical orientations: 0 or 2 pi by rounding (reference quirk)
    assert wrap.max() < 2e-3 and np.abs(d_dev - d_ref)[16:].max() < 2e-3
    pose64 = lambda q: np.concatenate([np.zeros((len(q), 3)), q], axis=1)
    _, d_host = block_goal_distance(pose64(ref["qa"]), pose64(ref["qb"]), "ignore", "xyz", ignore_z=True)
    assert np.abs(d_host - d_ref).max() < 1e-6
    env.close(); dense.close()


@pytest.mark.parametrize("env_id", ["HandManipulateEgg-v1", "HandManipulatePen-v1", "HandManipulateBlock-v1", "HandManipulateEgg_ContinuousTouchSensors-v1",
                                    "HandManipulatePen_BooleanTouchSensors-v1", "HandManipulateBlock_BooleanTouchSensors-v1"])
def test_serialize_deserialize_with_constructor_override(env_id):
    """/root/reference/tests/envs/hand/test_manipulate.py:19-28 and test_manipulate_touch_sensors.py:17-26: gym.make(id, target_position="fixed"),
    reset, pickle round trip."""
    import pickle

    import gymnasium_robotics_amd as grx

    env1 = grx.make_vec(env_id, num_envs=2, device="cuda:0", target_position="fixed")
    obs, _ = env1.reset(seed=0)
    env2 = pickle.loads(pickle.dumps(env1))
    assert env1.target_position == env2.target_position == "fixed"
    # 'fixed': the goal position is the object's settled position (manipulate.py:239-242); positions count in the distance (not 'ignore')
    assert np.array_equal(obs["desired_goal"][:, :3].astype(np.float32), obs["achieved_goal"][:, :3].astype(np.float32))
    assert env1.task.ignore_position == 0
    with pytest.raises(ValueError, match="Unknown target_rotation"):
        grx.make_vec(env_id, num_envs=2, device="cuda:0", target_rotation="fixed")


@pytest.mark.parametrize("env_id", ["HandManipulateBlockRotateXYZ-v1", "HandManipulateEggRotate_ContinuousTouchSensors-v1"])
def test_same_step_autoreset_matches_next_step(env_id):
    """The overlapped same-step reset (settle chains on side streams started two steps ahead, committed by grx_hand_commit_rows) against the sequential
    next-step reset.  Same draws and the same ten settle steps; the chain's first solve starts from the warm start the world had when the chain was
    started, so the settled state agrees to solver tolerance, not bit for bit (tests/autoreset_cases.py)."""
    import gymnasium_robotics_amd as grx
    from autoreset_cases import check_same_step_against_next_step

    make = lambda **kw: grx.make_vec(env_id, num_envs=40, device="cuda:0", **kw)
    # the egg: a rolling object whose contact set flips under the different warm start -- its velocity components are what diverges
    check_same_step_against_next_step(make, horizon=5, steps=12, act_dim=20, tol=2e-4, tol_max=0.3 if "Egg" in env_id else 1e-2, outlier_rows=0.3 if "Egg" in env_id else 0.15, output="torch",
                                      touch_from=61 if "Touch" in env_id else None)


def test_abandoned_settle_chains_give_their_draws_back():
    """same-step autoreset starts the settle chains of the worlds that are about to hit their time limit one or two steps AHEAD, drawing the reset poses from the
    worlds' own generators.  An explicit reset() in that window abandons the chains; the draws must be handed back, so that the reset observations equal those of an
    environment that never started a chain (the reference's sequential _reset_sim, manipulate.py:154-224, draws only when the episode has really ended)."""
    import gymnasium_robotics_amd as grx

    mk = lambda mode: grx.make_vec("HandManipulateBlockRotateXYZ-v1", num_envs=12, device="cuda:0", output="numpy", autoreset_mode=mode, max_episode_steps=6)
    A, B = mk("same_step"), mk("next_step")
    A.reset(seed=4); B.reset(seed=4)
    rng = np.random.default_rng(0)
    for t in range(5):      # the fifth step starts with two steps left: A starts (and draws for) its chains there, B draws nothing before the time limit
        a = rng.uniform(-1, 1, (12, 20)).astype(np.float32)
        A.step(a); B.step(a)
    assert len(A._chains) > 0
    oa, _ = A.reset()
    ob, _ = B.reset()
    for k in ("observation", "desired_goal"):
        assert np.array_equal(oa[k], ob[k]), k
    # reset(seed=...) with chains in flight: the seed must win -- the abandoned chains' draws go back to the OLD generators, the new ones start untouched
    for t in range(5):
        a = rng.uniform(-1, 1, (12, 20)).astype(np.float32)
        A.step(a); B.step(a)
    assert len(A._chains) > 0
    oa, _ = A.reset(seed=11)
    ob, _ = B.reset(seed=11)      # same history (the settle steps start from the finished episode's warm start, as in the reference), no chain ever started
    for k in ("observation", "desired_goal"):
        assert np.array_equal(oa[k], ob[k]), k
    A.close(); B.close()


def test_overflow_lane_polling_equals_the_serialised_rerun(monkeypatch):
    """hand + touch sensors: the polling workgroups of the standing lane launch against the serialised re-run (see tests/test_gpu_adroit.py): bit-identical rollouts"""
    import torch

    from gymnasium_robotics_amd import make_vec

    n, envs = 4096, []
    for poll in ("16", "0"):
        monkeypatch.setenv("GRX_LANE_POLL", poll)
        e = make_vec("HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled")
        assert e.lane is not None and e.lane.poll_grid == int(poll)
        e.reset(seed=4)
        envs.append(e)
    g = torch.Generator(device="cuda:0"); g.manual_seed(6)
    entered = 0
    for t in range(30):
        a = torch.rand(n, 20, device="cuda:0", generator=g) * 2 - 1
        outs = [e.step(a) for e in envs]
        entered += len(envs[1].lane.entered_last_step())
        o0, o1 = outs[0][0]["observation"], outs[1][0]["observation"]
        if not torch.equal(o0, o1):
            d = (o0 != o1).nonzero()
            raise AssertionError(f"step {t}: {len(d)} observation entries differ; worlds {sorted(set(d[:, 0].tolist()))[:8]} columns {sorted(set(d[:, 1].tolist()))[:12]} "
                                 f"values {[(float(o0[w, c]), float(o1[w, c])) for w, c in d[:4].tolist()]}; entrants {envs[0].lane.entered_last_step().tolist()} / {envs[1].lane.entered_last_step().tolist()}")
        assert torch.equal(envs[0].qpos, envs[1].qpos) and torch.equal(envs[0].qvel, envs[1].qvel), t
        for e, o in zip(envs, (o0, o1)):      # the observation row belongs to the state row it is returned with (a discarded fast-kernel run must not leave its row behind)
            assert torch.equal(o[:, :24], e.qpos[:, :24]) and torch.equal(o[:, 24:48], e.qvel[:, :24]), t
    assert entered >= 1, entered


@pytest.mark.parametrize("env_id", ["HandManipulateBlockRotateXYZ-v1", "HandManipulateEggRotate_ContinuousTouchSensors-v1", "HandManipulatePenRotate-v1"])
def test_repeat_launch_is_the_sequence_of_launches(env_id):
    """Round 6: grx_hand_step_repeat -- the ten settle steps of MujocoManipulateEnv._reset_sim (/root/reference/gymnasium_robotics/envs/shadow_dexterous_hand/manipulate.py:205-224:
    `for _ in range(10): self._set_action(np.zeros(20)); mj_step`) as ONE launch -- against ten launches of grx_hand_step on the same rows: state rows, observations (touch
    words included), achieved goals, rewards, success flags, status words and packed rows are BIT-IDENTICAL, from freshly reset worlds (object dropping onto the palm: contacts
    switching on) and from worlds mid-rollout; a repeat of 1 is the plain step; launches with an overflow lane are refused."""
    import ctypes

    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd import _native

    n, L = 96, _native.lib()
    envs = [grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None) for _ in range(2)]
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(4)
    names = ("qpos", "qvel", "qacc_ws", "obs", "achieved", "palm", "reward", "success", "status", "packed")
    for e in envs:
        e.reset(seed=21)
    seq, rep = envs
    for phase, repeat in (("after reset", 10), ("mid rollout", 10), ("single", 1)):
        if phase == "mid rollout":
            for t in range(4):
                a = torch.rand(n, 20, device="cuda:0", generator=gen) * 2 - 1
                for e in envs:
                    e.step(a)
            for name in names:
                assert torch.equal(getattr(seq, name), getattr(rep, name)), (phase, "rollout", name)
        act = torch.zeros(n, 20, device="cuda:0") if phase != "single" else torch.rand(n, 20, device="cuda:0", generator=gen) * 2 - 1
        for e in envs:
            e.action.copy_(act)
        sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(repeat):
            _native.check(L.grx_hand_step(seq._h, ctypes.byref(seq.task), ctypes.byref(seq._bufs), n, 0, sp))
        _native.check(L.grx_hand_step_repeat(rep._h, ctypes.byref(rep.task), ctypes.byref(rep._bufs), n, repeat, sp))
        torch.cuda.synchronize()
        for name in names:
            assert torch.equal(getattr(seq, name), getattr(rep, name)), (phase, name, int((getattr(seq, name) != getattr(rep, name)).sum()))
        assert torch.isfinite(rep.qpos).all() and int((rep.status & 1).max()) == 0
    b = rep._make_bufs(None)
    b.lane.skip = rep.mask.data_ptr()
    with pytest.raises(RuntimeError, match="overflow lane"):
        _native.check(L.grx_hand_step_repeat(rep._h, ctypes.byref(rep.task), ctypes.byref(b), n, 10, None))
    with pytest.raises(RuntimeError, match="repeat must be"):
        _native.check(L.grx_hand_step_repeat(rep._h, ctypes.byref(rep.task), ctypes.byref(rep._bufs), n, 0, None))
