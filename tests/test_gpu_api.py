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
    assert env2.task.distance_threshold == np.float32(1e-6)
    env1.close(); env2.close()


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
    _native.check(L.grx_order_by_cost(cost.data_ptr(), n, order.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
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
        _native.check(L.grx_order_by_cost(cost.data_ptr(), 1001, order.data_ptr(), None))
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
