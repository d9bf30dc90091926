"""Every id the reference registers (121), through the checks the reference's generic suite applies to every spec -- /root/reference/tests/test_envs.py:39-58 hands each env to
gymnasium.utils.env_checker.check_env [3P, absent here: no gymnasium in the container] and :64-117 runs the seeded two-env rollout.  What check_env verifies according to its
documentation is restated for the vector façade: spaces exist and agree with what reset / step return (shape, dtype, membership, finiteness), reset(seed=) is deterministic and
a second seed changes the episode, step returns (obs, reward, terminated, truncated, info) with numeric / boolean arrays of the batch shape, seeded rollouts are reproducible,
out-of-range actions are accepted (robot_env.py:132 clips them), goal environments satisfy reward == compute_reward(achieved, desired, info) (core.py:59-62), close() can be
called twice.  MI355X, through the C ABI, two worlds per environment."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ids():
    import gymnasium_robotics_amd as grx

    return grx.registered_env_ids()


def _flat(x, prefix=""):
    if isinstance(x, dict):
        for k, v in x.items():
            yield from _flat(v, f"{prefix}{k}.")
    else:
        yield prefix.rstrip("."), np.asarray(x)


def _equal(a, b):
    fa, fb = dict(_flat(a)), dict(_flat(b))
    return fa.keys() == fb.keys() and all(np.array_equal(fa[k], fb[k]) for k in fa)


def _close(a, b, atol):
    fa, fb = dict(_flat(a)), dict(_flat(b))
    return fa.keys() == fb.keys() and all(np.allclose(fa[k], fb[k], rtol=1e-5, atol=atol) for k in fa)


def _check_obs(env, obs, n):
    assert env.observation_space.contains(obs), "observation outside observation_space"
    for name, arr in _flat(obs):
        assert arr.shape[0] == n and np.isfinite(arr).all() and arr.dtype == np.float64, (name, arr.shape, arr.dtype)


@pytest.mark.parametrize("env_id", _ids())
def test_registered_id_conforms(env_id):
    import gymnasium_robotics_amd as grx

    n = 2
    e1 = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="numpy")
    e2 = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="numpy")
    for e in (e1, e2):
        assert e.num_envs == n and e.single_action_space.shape == e.action_space.shape[1:] and e.action_space.shape[0] == n
        assert e.single_action_space.dtype in (np.float32, np.float64)
    goal_env = isinstance(e1.single_observation_space, dict)
    if goal_env:
        assert {"observation", "achieved_goal", "desired_goal"} <= set(e1.single_observation_space.keys())      # core.py:33-43
    # reset: (obs, info), obs in the space, the same seed reproduces it, another seed gives another episode.  Both environments go through the SAME sequence of resets:
    # the manipulation resets run their ten settle steps from the solver warm start the previous episode left, in the reference as here (manipulate.py:205-224), so two
    # environments agree bit for bit only after the same history, and a repeated reset(seed=) of one environment agrees to solver tolerance (fp32 here) -- which is also what
    # check_env asks for: it compares the two resets with data_equivalence's tolerance, not bit for bit [3P]
    o3, _ = e1.reset(seed=456)
    e2.reset(seed=456)
    o1, i1 = e1.reset(seed=123)
    o2, i2 = e2.reset(seed=123)
    assert isinstance(i1, dict) and _equal(o1, o2)
    _check_obs(e1, o1, n)
    assert not _equal(o1, o3), "a different seed gave the same first observation"
    o1b, _ = e1.reset(seed=123)
    e2.reset(seed=123)
    assert _close(o1, o1b, 5e-5), "reset(seed=) is not deterministic"
    # step: types, shapes, membership; the seeded two-env rollout of test_envs.py:64-117
    e1.action_space.seed(0)
    for t in range(4):
        a = e1.action_space.sample()
        assert a.shape == e1.action_space.shape
        if t == 2:
            a = a * 3.0      # outside [-1, 1]: clipped like the reference (robot_env.py:132), not an error
        s1, s2 = e1.step(a), e2.step(a)
        assert len(s1) == 5
        obs, rew, term, trunc, info = s1
        _check_obs(e1, obs, n)
        rew, term, trunc = np.asarray(rew), np.asarray(term), np.asarray(trunc)
        assert rew.shape == (n,) and rew.dtype.kind == "f" and np.isfinite(rew).all()
        assert term.shape == (n,) and term.dtype == np.bool_ and trunc.shape == (n,) and trunc.dtype == np.bool_
        assert isinstance(info, dict)
        for x, y in zip(s1[:4], s2[:4]):
            assert _equal(x, y), f"step {t}: two environments with the same seed and actions disagree"
        assert _equal({k: v for k, v in info.items() if not k.startswith("_")}, {k: v for k, v in s2[4].items() if not k.startswith("_")})
        if goal_env and hasattr(e1, "compute_reward") and not isinstance(obs["achieved_goal"], dict):
            r = np.asarray(e1.compute_reward(obs["achieved_goal"], obs["desired_goal"], info))
            assert np.array_equal(r, rew), "reward != compute_reward(achieved_goal, desired_goal, info)"      # core.py:59-62
    e1.close(); e1.close(); e2.close()
