"""GPU tests that mirror the reference's own generic env tests (/root/reference/tests/test_envs.py, tests/envs/hand/*.py):
determinism of seeded rollouts, reset state == documented initial state, pickling by constructor value."""
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(env_id, **kw):
    import gymnasium_robotics_amd as grx

    return grx.make_vec(env_id, num_envs=3, device="cuda:0", output="numpy", **kw)


ALL_IDS = ["FetchReach-v4", "FetchPush-v4", "FetchSlide-v4", "FetchPickAndPlaceDense-v4", "HandManipulateEggFull-v1", "HandReach-v3", "HandManipulateBlockRotateXYZ-v1",
           "HandManipulateBlock_BooleanTouchSensors-v1", "HandManipulatePen_ContinuousTouchSensors-v1", "PointMaze_UMaze-v3", "AntMaze_UMaze-v5"]


def _assert_equal(a, b):
    if isinstance(a, dict):
        assert a.keys() == b.keys()
        for k in a:
            _assert_equal(a[k], b[k])
    else:
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("env_id", ALL_IDS)
def test_env_determinism_rollout(env_id):
    """tests/test_envs.py:64-117: two envs, same seed, same actions -> identical observations, rewards, flags and infos."""
    e1, e2 = _make(env_id), _make(env_id)
    o1, i1 = e1.reset(seed=0)
    o2, i2 = e2.reset(seed=0)
    _assert_equal(o1, o2)
    e1.action_space.seed(0)
    for _ in range(12):
        a = e1.action_space.sample()
        s1, s2 = e1.step(a), e2.step(a)
        for x, y in zip(s1, s2):
            _assert_equal(x, y)
        assert e1.observation_space.contains(s1[0])
    e1.close(); e2.close()


@pytest.mark.parametrize("env_id", ["FetchReach-v4", "FetchPush-v4", "FetchPickAndPlace-v4", "HandReach-v3"])
def test_robot_env_reset(env_id):
    """tests/test_envs.py:181-232: after reset the state is the documented initial state (object xy excluded for the Fetch object tasks)."""
    env = _make(env_id)
    for seed in (24, 10):
        env.reset(seed=seed)
        qpos, qvel = env.qpos.cpu().numpy(), env.qvel.cpu().numpy()
        init = np.broadcast_to(env.initial_qpos.cpu().numpy() if hasattr(env, "initial_qpos") else env._initial_qpos.cpu().numpy(), qpos.shape)
        if env_id.startswith(("FetchPush", "FetchPickAndPlace")):
            qpos, init = np.delete(qpos, np.s_[-7:-5], axis=1), np.delete(init, np.s_[-7:-5], axis=1)
        assert np.array_equal(qpos, init)
        assert not qvel.any() or env_id.startswith("Fetch")   # Fetch keeps the settled initial_qvel (robot_env.py:298)
    env.close()


@pytest.mark.parametrize("env_id", ALL_IDS)
def test_pickle_env(env_id):
    """tests/test_envs.py:164-178: a pickled copy behaves like the original."""
    env = _make(env_id)
    twin = pickle.loads(pickle.dumps(env))
    _assert_equal(env.reset(seed=5)[0], twin.reset(seed=5)[0])
    a = np.zeros((3, env.single_action_space.shape[0]), np.float32)
    for x, y in zip(env.step(a), twin.step(a)):
        _assert_equal(x, y)
    env.close(); twin.close()


def test_serialize_deserialize_keeps_constructor_arguments():
    """tests/envs/hand/test_reach.py, tests/envs/hand/test_manipulate.py."""
    env1 = _make("HandReach-v3", distance_threshold=1e-6)
    env1.reset()
    env2 = pickle.loads(pickle.dumps(env1))
    assert env1.distance_threshold == env2.distance_threshold == 1e-6
    assert env2.task.distance_threshold == 1e-6
    env1.close(); env2.close()


def test_order_by_cost_slots_places_three_cheap_worlds_on_the_slots_that_run_three():
    """grx_order_by_cost_slots in the two-worlds-per-slot regime (512 worlds per XCD slice on 256 slots): with K stragglers (cost > 2 x the cheapest) M = K workgroups
    are a slot's third world; the M cheapest worlds end the first round, the next M start the second round, the next M end the launch, everything else keeps the
    decreasing order; a permutation inside every slice.  slots = 0 and other regimes: the plain order."""
    import ctypes

    import torch

    from gymnasium_robotics_amd import _native

    L = _native.lib()
    n, per, slots = 4096, 512, 256
    rng = np.random.default_rng(3)
    cost = rng.integers(1000, 1400, n).astype(np.int32)
    strag = {s: rng.choice(per, size=3 + (s % 3), replace=False) + s * per for s in range(8)}
    for s in range(8):
        cost[strag[s]] = rng.integers(2900, 3400, len(strag[s]))
    ct = torch.from_numpy(cost).to("cuda:0")
    plain = torch.full((n,), -1, device="cuda:0", dtype=torch.int32)
    tail = torch.full((n,), -1, device="cuda:0", dtype=torch.int32)
    _native.check(L.grx_order_by_cost_slots(ct.data_ptr(), None, 0.0, n, 0, plain.data_ptr(), None))
    _native.check(L.grx_order_by_cost_slots(ct.data_ptr(), None, 0.0, n, slots, tail.data_ptr(), None))
    torch.cuda.synchronize()
    P, T_ = plain.cpu().numpy().reshape(per, 8), tail.cpu().numpy().reshape(per, 8)
    for s in range(8):
        d, t = P[:, s], T_[:, s]                      # d: the decreasing list of the slice
        assert sorted(t.tolist()) == sorted(d.tolist()) == list(range(s * per, (s + 1) * per))
        M = len(strag[s])
        assert set(d[:M].tolist()) == set(strag[s].tolist())
        assert (t[:slots - M] == d[:slots - M]).all()                                   # stragglers and the expensive half first
        assert (t[slots - M:slots] == d[per - M:]).all()                                # the M cheapest end the first round
        assert (t[slots:slots + M] == d[per - 2 * M:per - M]).all()                     # the next M start the second round
        assert (t[slots + M:per - M] == d[slots - M:per - 3 * M]).all()                 # the middle, still decreasing
        assert (t[per - M:] == d[per - 3 * M:per - 2 * M]).all()                        # the next M are dispatched last
    # three rounds and more (1024 worlds per slice on 256 slots): the plain order
    n2 = 8192
    c2 = torch.from_numpy(rng.integers(1000, 4000, n2).astype(np.int32)).to("cuda:0")
    o1 = torch.empty(n2, device="cuda:0", dtype=torch.int32); o2 = torch.empty(n2, device="cuda:0", dtype=torch.int32)
    _native.check(L.grx_order_by_cost_slots(c2.data_ptr(), None, 0.0, n2, 0, o1.data_ptr(), None))
    _native.check(L.grx_order_by_cost_slots(c2.data_ptr(), None, 0.0, n2, slots, o2.data_ptr(), None))
    torch.cuda.synchronize()
    assert (o1 == o2).all()


def test_order_by_cost_and_balance_invariance():
    """grx_order_by_cost: per XCD slice (n / 8 contiguous worlds) decreasing cost, ties by world index, workgroup b -> slice b & 7.
    The dispatch order is a scheduling choice only: a balanced and an unbalanced env produce bit-identical outputs."""
    import ctypes

    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd import _native

    L = _native.lib()
    n = 2048
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    cost = torch.randint(0, 5000, (n,), device="cuda:0", generator=g, dtype=torch.int32)
    order = torch.full((n,), -1, device="cuda:0", dtype=torch.int32)
    _native.check(L.grx_order_by_cost(cost.data_ptr(), None, 0.0, n, order.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    o, c = order.cpu().numpy().reshape(n // 8, 8), cost.cpu().numpy()
    assert sorted(o.ravel().tolist()) == list(range(n))                       # a permutation
    per = n // 8
    for s in range(8):
        w = o[:, s]
        assert (w // per == s).all()                                          # slice s stays on XCD s
        key = np.stack([-c[w].astype(np.int64), w.astype(np.int64)], axis=1)
        assert (np.lexsort((key[:, 1], key[:, 0])) == np.arange(per)).all()   # decreasing cost, ties by index
    with pytest.raises(RuntimeError, match="multiple of 8"):
        _native.check(L.grx_order_by_cost(cost.data_ptr(), None, 0.0, 1001, order.data_ptr(), None))
    ema = torch.full((n,), 100.0, device="cuda:0")
    _native.check(L.grx_order_by_cost(cost.data_ptr(), ema.data_ptr(), 0.25, n, order.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.allclose(ema.cpu().numpy(), 75.0 + 0.25 * c, rtol=1e-6)
    a = grx.make_vec("FetchPickAndPlace-v4", num_envs=1024, device="cuda:0", balance=True)
    b = grx.make_vec("FetchPickAndPlace-v4", num_envs=1024, device="cuda:0", balance=False)
    assert a.balance and not b.balance
    a.reset(seed=5); b.reset(seed=5)
    rng = np.random.default_rng(0)
    for _ in range(6):
        act = rng.uniform(-1, 1, (1024, 4)).astype(np.float32)
        sa, sb = a.step(act), b.step(act)
        assert np.array_equal(sa[0]["observation"], sb[0]["observation"]) and np.array_equal(sa[1], sb[1])
    assert int(a.cost.min()) >= 12 * 20 and not np.array_equal(a.order.cpu().numpy(), np.arange(1024).reshape(8, 128).T.ravel())


@pytest.mark.parametrize("env_id,var", [("FrankaKitchen-v1", "GRX_KITCHEN_BALANCE"), ("AdroitHandHammer-v2", "GRX_ADROIT_BALANCE"), ("AdroitHandRelocate-v2", "GRX_ADROIT_BALANCE")])
def test_cost_order_of_kitchen_and_adroit_is_scheduling_only(monkeypatch, env_id, var):
    """Round 6: the kitchen and the Adroit step launches are dispatched longest-world-first from the durations the previous launches measured (include/grx_capi.h
    grx_kitchen_buffers.order / .cost): the order is a permutation of the worlds that changes after the first steps, every world's duration is recorded, and the rollout -- observations,
    rewards, flags, state rows, same-step autoresets included -- is bit-identical to the launch in index order."""
    import torch

    import gymnasium_robotics_amd as grx

    n, envs = 1024, []
    for on in ("1", "0"):
        monkeypatch.setenv(var, on)
        e = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=6)
        e.reset(seed=11)
        envs.append(e)
    a, b = envs
    assert a.balance and a.order is not None and not b.balance and b.order is None
    identity = a.order.clone()
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(2)
    act_dim = a.single_action_space.shape[0]
    for t in range(9):
        act = torch.rand(n, act_dim, device="cuda:0", generator=gen) * 2 - 1
        oa, ob = a.step(act), b.step(act)
        obs_a, obs_b = (oa[0]["observation"], ob[0]["observation"]) if isinstance(oa[0], dict) else (oa[0], ob[0])
        assert torch.equal(obs_a, obs_b) and torch.equal(torch.as_tensor(oa[1]), torch.as_tensor(ob[1])), t
        assert torch.equal(torch.as_tensor(oa[2]), torch.as_tensor(ob[2])) and torch.equal(torch.as_tensor(oa[3]), torch.as_tensor(ob[3])), t
        for name in ("qpos", "qvel", "qacc_ws", "status"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (t, name)
    torch.cuda.synchronize()
    assert sorted(a.order.cpu().tolist()) == list(range(n)) and not torch.equal(a.order, identity)
    assert int(a.cost.min()) > 0 and int(a.cost.max()) < 10_000_000      # 80 ns units: every world was timed, no start stamp was left behind


def test_device_rewards_equal_the_reference_run_vectors():
    """The batched reward kernels (the HER relabelling entry points) against rewards computed by the reference's own methods
    (tests/golden/ref_host_logic.npz, tools/make_reference_host_vectors.py); fp32 device arithmetic: pairs whose distance sits on a
    threshold are excluded from the exact comparison."""
    import os

    import gymnasium_robotics_amd as grx

    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_host_logic.npz"))
    for env_id, key in (("FetchPush-v4", "sparse"), ("FetchPushDense-v4", "dense")):
        env = grx.make_vec(env_id, num_envs=2, device="cuda:0")
        r = env.compute_reward(ref["fetch_ag"], ref["fetch_dg"], {})
        # the device decides d > 0.05 in fp64 on the fp32-rounded goals it is handed: exactly the reference's answer for those goals
        d = np.linalg.norm(ref["fetch_ag"].astype(np.float32).astype(np.float64) - ref["fetch_dg"].astype(np.float32).astype(np.float64), axis=-1)
        assert r.shape == ref[f"fetch_reward_{key}"].shape
        assert np.array_equal(r, -(d > 0.05).astype(np.float32)) if key == "sparse" else np.allclose(r, -d, rtol=0, atol=2e-8)
        clear = np.abs(np.linalg.norm(ref["fetch_ag"] - ref["fetch_dg"], axis=-1) - 0.05) > 1e-6      # the reference's own fp64 goals: equal where fp32 input rounding cannot matter
        assert np.allclose(r[clear], ref[f"fetch_reward_{key}"][clear], rtol=0, atol=1e-6)
    for env_id, key in (("HandReach-v3", "sparse"), ("HandReachDense-v3", "dense")):
        env = grx.make_vec(env_id, num_envs=2, device="cuda:0")
        r = env.compute_reward(ref["hand_reach_ag"], ref["hand_reach_dg"], {})
        d = np.linalg.norm(ref["hand_reach_ag"].astype(np.float32).astype(np.float64) - ref["hand_reach_dg"].astype(np.float32).astype(np.float64), axis=-1)
        assert np.array_equal(r, -(d > 0.01).astype(np.float32)) if key == "sparse" else np.allclose(r, -d, rtol=0, atol=2e-8)
        clear = np.abs(np.linalg.norm(ref["hand_reach_ag"] - ref["hand_reach_dg"], axis=-1) - 0.01) > 1e-6
        assert np.allclose(r[clear], ref[f"hand_reach_reward_{key}"][clear], rtol=0, atol=1e-6)
    for env_id, tag, key in (("HandManipulateBlockRotateXYZ-v1", "ignore_xyz", "sparse"), ("HandManipulateBlockRotateXYZDense-v1", "ignore_xyz", "dense"),
                             ("HandManipulateBlockFull-v1", "random_xyz", "sparse"), ("HandManipulateBlockFullDense-v1", "random_xyz", "dense")):
        env = grx.make_vec(env_id, num_envs=2, device="cuda:0")
        r = env.compute_reward(ref["manip_ga"], ref["manip_gb"], {})
        clear = (np.abs(ref[f"manip_{tag}_dpos"] - 0.01) > 1e-5) & (np.abs(ref[f"manip_{tag}_drot"] - 0.1) > 1e-4)
        # dense: -(10 d_pos + d_rot); the fp32 angle of two nearly identical orientations carries ~3e-4 rad of rounding (2 atan2 of a 1e-4 sine)
        assert np.allclose(r[clear], ref[f"manip_{tag}_reward_{key}"][clear], rtol=0, atol=1e-3 if key == "dense" else 0), (env_id, np.abs(r - ref[f"manip_{tag}_reward_{key}"])[clear].max())


def test_step_async_wait_equals_step():
    import gymnasium_robotics_amd as grx

    a = grx.make_vec("FetchPush-v4", num_envs=64, device="cuda:0", output="torch")
    b = grx.make_vec("FetchPush-v4", num_envs=64, device="cuda:0", output="torch")
    a.reset(seed=1); b.reset(seed=1)
    act = np.random.default_rng(0).uniform(-1, 1, (64, 4)).astype(np.float32)
    with pytest.raises(RuntimeError, match="without a pending"):
        a.step_wait()
    a.step_async(act)
    oa = a.step_wait()
    ob = b.step(act)
    assert bool((oa[0]["observation"] == ob[0]["observation"]).all()) and bool((oa[1] == ob[1]).all())


@pytest.mark.parametrize("mode", ["next_step", "same_step"])
@pytest.mark.parametrize("env_id", ["FetchPush-v4", "PointMaze_UMazeDense-v3", "AntMaze_UMaze-v5", "HandReach-v3", "HandManipulateBlockRotateZ-v1"])
def test_packed_row_reward_equals_reward_on_autoreset_steps(env_id, mode):
    """The packed rows [obs | achieved | desired | reward | success] the step kernels write (cross-rank gather, HER) must carry the reward / success the
    step REPORTS, also in the step that resets a world: 0 in next-step mode (gymnasium >= 1.0: the reset step returns reward 0), the finished episode's
    values in same-step mode."""
    import gymnasium_robotics_amd as grx

    env = grx.make_vec(env_id, num_envs=6, device="cuda:0", output="torch", autoreset_mode=mode, max_episode_steps=3)
    env.reset(seed=1)
    rng = np.random.default_rng(0)
    nu = env.action_space.shape[-1]
    resets = 0
    for t in range(9):
        obs, r, term, trunc, info = env.step(rng.uniform(-1, 1, (6, nu)).astype(np.float32))
        pk = env.packed.cpu().numpy()
        succ = info["is_success"] if "is_success" in info else info["success"]
        assert np.array_equal(pk[:, -2], r.float().cpu().numpy()), (t, pk[:, -2], r)
        assert np.array_equal(pk[:, -1] != 0, succ.cpu().numpy() != 0), t
        if mode == "next_step" and t in (3, 7):       # the reset steps of a 3-step time limit
            assert float(r.abs().max()) == 0.0
            resets += 1
        assert int(info["status"].max()) == 0 and "status_sticky" in info
    assert mode == "same_step" or resets == 2
    env.close()


# (HandManipulate* is the one family that is not in this list: its reset is a data-dependent RETRY loop -- "the object fell off the palm, draw again",
# manipulate.py:205-224 / robot_env.py:163-171 -- whose verdict is read back from the settle chain that ran one or two steps ahead on a side stream.)
@pytest.mark.parametrize("env_id,steps", [("FetchPickAndPlace-v4", 3), ("FetchReach-v4", 3), ("FetchSlide-v4", 3), ("HandReach-v3", 3), ("AdroitHandHammer-v2", 3), ("AdroitHandDoor-v2", 3),
                                          ("AdroitHandPen-v2", 3), ("AdroitHandRelocate-v2", 3), ("PointMaze_UMaze-v3", 3), ("AntMaze_UMaze-v5", 3), ("FrankaKitchen-v1", 3)])
def test_torch_step_only_enqueues(env_id, steps):
    """step(output="torch") of every family only ENQUEUES work, also in the steps that reset worlds (same-step autoreset at a short time limit): with torch's
    synchronisation debug mode set to "error" any blocking read-back or host-side wait for the device raises.  (The reset draws of the non-kitchen families are made
    by the worlds' numpy generators on the host WHILE the step kernel runs -- the time limit is host-side bookkeeping -- and reach the device through pinned staging.)"""
    import torch

    import gymnasium_robotics_amd as grx

    n = 64
    env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=steps)
    env.reset(seed=0)
    act = torch.zeros(n, env.single_action_space.shape[0], device="cuda:0")
    for _ in range(steps + 1):      # one full episode incl. its resets outside the checked region (first-use allocations)
        env.step(act)
    torch.cuda.synchronize()
    try:
        torch.cuda.set_sync_debug_mode("error")
    except Exception:
        pytest.skip("this torch build has no synchronisation debug mode")
    try:
        for _ in range(2 * steps + 1):
            out = env.step(act)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert out[1].is_cuda      # rewards stay on the device; terminated / truncated are host-side bookkeeping (the time limit) in the goal families, device tensors in the kitchen
    env.close()


def test_many_environments_of_one_id_share_two_model_slots():
    """The native side has 32 model descriptor slots per process; environments of the same compiled tables share one reference-counted handle (fast + overflow-lane
    tables = two slots per id), so a process can hold far more than 16 environments; closing one leaves the others stepping, the last close frees the slots."""
    import numpy as np

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd import _native

    before = len(_native._MODELS)
    envs = [grx.make_vec("FetchPickAndPlace-v4", num_envs=8, device="cuda:0") for _ in range(20)]
    assert len(_native._MODELS) <= before + 2      # (== before when an environment of this id is still alive elsewhere in the process: its two handles are simply shared)
    assert all(e._h.value == envs[0]._h.value and e._h_big.value == envs[0]._h_big.value for e in envs)
    a = np.zeros((8, 4), np.float32)
    ref = None
    for e in envs[:3]:
        e.reset(seed=5)
        obs = e.step(a)[0]["observation"]
        ref = obs if ref is None else ref
        assert np.array_equal(obs, ref)
    envs[0].close()
    envs[1].step(a)
    for e in envs[1:]:
        e.close()
    assert len(_native._MODELS) == before
    e = grx.make_vec("FetchPickAndPlace-v4", num_envs=8, device="cuda:0")     # the slots were really freed and can be taken again
    e.reset(seed=5)
    assert np.array_equal(e.step(a)[0]["observation"], ref)
    e.close()


def test_world_rng_round_trip_on_device_streams():
    """the families whose reset draws live on the device expose world_rng(i) / set_world_rng(i, g): positioning a world's stream from the host changes exactly that
    world's next reset, and _rng_state assignment (FetchVecEnv) uploads instead of being dropped."""
    import gymnasium_robotics_amd as grx

    for env_id in ("FetchPush-v4", "PointMaze_UMaze-v3", "AdroitHandHammer-v2"):
        a = grx.make_vec(env_id, num_envs=6, device="cuda:0", output="numpy")
        b = grx.make_vec(env_id, num_envs=6, device="cuda:0", output="numpy")
        a.reset(seed=3); b.reset(seed=40)
        for i in range(6):
            b.set_world_rng(i, a.world_rng(i))
            assert b.world_rng(i).bit_generator.state == a.world_rng(i).bit_generator.state
        oa, _ = a.reset(); ob, _ = b.reset()
        _assert_equal(oa, ob)
        a.close(); b.close()
    f = grx.make_vec("FetchPush-v4", num_envs=4, device="cuda:0")
    f.reset(seed=1)
    st = f._rng_state.copy()
    st[2] = st[0]
    f._rng_state = st
    assert np.array_equal(f._rng_state, st)
    f.close()


def test_degenerate_maze_is_refused_and_an_exhausted_rejection_loop_is_flagged():
    """a device rejection loop must never hand back a sample that violates the reference's condition.  (i) A maze whose single reset cell is its single goal cell is
    refused by the constructor (the reference's generate_reset_pos would spin, maze_v4.py:400-418).  (ii) Fetch with an object range that can never leave the 0.1 m ring around the
    gripper: the loop gives up after 65 536 draws, the sample is NaN, the next step flags the world (sticky GRX_STATUS_BADNUM) instead of stepping a silently wrong state."""
    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.envs.point_maze import PointMazeVecEnv

    with pytest.raises(ValueError, match="single reset cell"):
        PointMazeVecEnv("PointMaze_UMaze-v3", num_envs=2, device="cuda:0", maze_map=[[1, 1, 1], [1, "c", 1], [1, 1, 1]])
    # one 'r' cell NEXT TO one 'g' cell is a valid maze (the reset cell's centre is >= 0.75 cells from the noisy goal: the reference's loop ends on its first draw)
    # (a custom map needs the MJCF assets to compile its walls; the GPU box has none, so the packaged U-maze tables stand in -- the reset logic under test reads the MAP, not the walls)
    from gymnasium_robotics_amd.envs.maze_spec import MAPS, POINT_MAZE_HEIGHT, POINT_MAZE_SIZE_SCALING, Maze
    from gymnasium_robotics_amd.envs.point_maze import load_point_maze_model

    walls = load_point_maze_model(Maze(MAPS["UMaze"], POINT_MAZE_SIZE_SCALING, POINT_MAZE_HEIGHT), "UMaze", None, "point")
    adj = PointMazeVecEnv("PointMaze_UMaze-v3", num_envs=3, device="cuda:0", maze_map=[[1, 1, 1, 1], [1, "r", "g", 1], [1, 1, 1, 1]], model=walls)
    obs, _ = adj.reset(seed=5)
    d = np.linalg.norm(obs["achieved_goal"] - obs["desired_goal"], axis=1)
    assert (d > 0.5 * adj.maze.maze_size_scaling).all() and (d < 1.5 * adj.maze.maze_size_scaling).all() and int(np.abs(adj.status.cpu().numpy()).max()) == 0
    adj.close()
    env = grx.make_vec("FetchPush-v4", num_envs=4, device="cuda:0", output="numpy")
    env.cfg = dict(env.cfg, obj_range=0.01)
    env.reset(seed=0)
    out = env.step(np.zeros((4, 4), np.float32))
    assert (out[4]["status_sticky"] & 1).all()
    env.close()


@pytest.mark.parametrize("env_id,n,var,act_dim", [("FetchPickAndPlace-v4", 2050, "GRX_FETCH_SPLIT", 4), ("FetchPickAndPlace-v4", 1001, "GRX_FETCH_SPLIT", 4),
                                                  ("AntMaze_Large_Diverse_GR-v5", 3083, "GRX_MAZE_SPLIT", 8), ("AntMaze_Large_Diverse_GR-v5", 100, "GRX_MAZE_SPLIT", 8)])
def test_split_steps_at_batch_sizes_that_are_no_multiple_of_eight(monkeypatch, env_id, n, var, act_dim):
    """The split launches (include/grx_capi.h grx_fetch_buffers.split_parts / grx_point_buffers.split_parts) map workgroup -> (part, slot) -> world over a grid rounded up to 8 slots per
    part: the padding workgroups of every part return before they wait for anything, and the rollout is the plain launch's bit for bit (same-step autoresets included)."""
    import torch

    import gymnasium_robotics_amd as grx

    envs = []
    for p in ("1", "2"):
        monkeypatch.setenv(var, p)
        e = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=5)
        e.reset(seed=3)
        envs.append(e)
    assert envs[0]._split == 1 and envs[1]._split == 2
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    for t in range(12):
        a = torch.rand(n, act_dim, device="cuda:0", generator=g) * 2 - 1
        for e in envs:
            e.step(a)
        for name in ("qpos", "qvel", "qacc_ws", "obs", "reward", "status", "packed"):
            assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)), (t, name)
