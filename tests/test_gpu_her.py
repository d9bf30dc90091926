"""On-device HER relabel + replay write (gymnasium_robotics_amd/her.py, C ABI grx_her_relabel) against a plain numpy restatement of what a HER replay
does with the reference's compute_reward (README.md:72-76): gather a transition, substitute a goal achieved later in the episode, recompute the
reward.  Copies must be bit-exact; the recomputed reward must equal env.compute_reward (the batched kernel) bit for bit, and the host restatement of
the reference's reward (fetch_spec / hand_spec / manipulate_spec / maze_spec, float64) to fp32 rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rollout(env_id, n, steps, seed=0):
    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.her import HerReplay

    env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None)
    buf = HerReplay(env, horizon=steps, capacity=4096, seed=seed)
    env.reset(seed=seed)
    buf.begin_episode(env.packed)
    g = torch.Generator(device="cuda:0"); g.manual_seed(seed)
    A = env.single_action_space.shape[0]
    for t in range(steps):
        a = torch.rand(n, A, device="cuda:0", generator=g) * 2 - 1
        env.step(a)
        buf.append(a, env.packed)
    return env, buf


@pytest.mark.parametrize("env_id", ["FetchPush-v4", "FetchPickAndPlaceDense-v4", "HandReach-v3", "HandManipulateBlockRotateXYZ-v1", "HandManipulatePenRotateDense-v1",
                                    "PointMaze_UMaze-v3", "AntMaze_UMazeDense-v5"])
def test_relabelled_rows_match_numpy_her(env_id):
    import torch

    steps = 12
    env, buf = _rollout(env_id, 24, steps)
    rows = buf.relabel(batch=1024, k_future=4).clone()
    # the same draws again (the generator is seeded; re-create it to replay the index stream)
    buf.reseed(0)
    t, w, tg = (x.cpu().numpy().astype(np.int64) for x in buf.sample_indices(1024, 4))
    assert (tg < 0).any() and (tg >= 0).any() and np.all((tg < 0) | ((tg > t) & (tg <= steps)))
    ep, acts = buf.episode.cpu().numpy(), buf.actions.cpu().numpy()
    o, gd = buf.obs_dim, buf.goal_dim
    assert t.min() >= 0 and t.max() == steps - 1
    r0, r1 = ep[t, w], ep[t + 1, w]
    goal = np.where((tg >= 0)[:, None], ep[np.maximum(tg, 0), w][:, o: o + gd], r0[:, o + gd: o + 2 * gd])
    parts = buf.split(rows)
    got = {k: v.cpu().numpy() for k, v in parts.items()}
    assert np.array_equal(got["observation"], r0[:, :o]) and np.array_equal(got["achieved_goal"], r0[:, o: o + gd]) and np.array_equal(got["desired_goal"], goal)
    assert np.array_equal(got["action"], acts[t + 1, w]) and np.array_equal(got["next_observation"], r1[:, :o]) and np.array_equal(got["next_achieved_goal"], r1[:, o: o + gd])
    # reward: bit-equal to the batched compute_reward kernel ...
    ag1 = torch.from_numpy(r1[:, o: o + gd].copy()).cuda()
    rk = env.compute_reward(ag1, torch.from_numpy(goal.copy()).cuda(), None)
    rk = rk.cpu().numpy() if hasattr(rk, "cpu") else np.asarray(rk)
    assert np.array_equal(got["reward"][:, 0], rk.astype(np.float32))
    # ... and, where the goal was NOT substituted, to what the step kernel itself wrote for that transition (reward == compute_reward(ag, dg): core.py:59-62)
    own = tg < 0
    assert np.array_equal(got["reward"][own, 0], r1[own, -2]) and np.array_equal(got["success"][own, 0], r1[own, -1])
    # relabelled with the goal achieved at the very next row: the transition reaches its goal
    nxt = tg == t + 1
    if nxt.any():
        assert (got["success"][nxt, 0] == 1.0).all()
        if "Dense" not in env_id:
            assert (got["reward"][nxt, 0] == (1.0 if "Maze" in env_id else 0.0)).all()


def test_ring_buffer_and_argument_checks():
    import ctypes

    import torch

    from gymnasium_robotics_amd import _native

    env, buf = _rollout("FetchReach-v4", 8, 5)
    assert buf.OW == 2 * 10 + 3 * 3 + 4 + 2 and buf.rows.shape == (4096, buf.OW)
    for _ in range(5):
        v = buf.relabel(1000)
        assert v.shape == (1000, buf.OW) and torch.isfinite(v).all()
    assert buf.head == 1000 and buf.size == 4000          # the fifth batch wrapped to the start of the ring
    with pytest.raises(ValueError):
        buf.relabel(5000)
    with pytest.raises(RuntimeError, match="episode buffer is full"):
        buf.append(torch.zeros(8, 4, device="cuda:0"), env.packed)
    a = _native.HerArgsStruct()
    assert _native.lib().grx_her_relabel(ctypes.byref(a), 4, None) != 0 and b"null buffer" in _native.lib().grx_last_error()


def test_continuous_ring_respects_episode_boundaries():
    """same-step autoreset with staggered episodes: rows wrap around the ring, every sampled transition and its substituted goal lie inside the
    world's CURRENT episode (never across a reset), and the rows equal a numpy gather from a full host-side log of the rollout."""
    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.her import HerReplay

    n, H, steps = 32, 10, 37
    env = grx.make_vec("FetchPush-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=H)
    buf = HerReplay(env, horizon=H, capacity=2048, seed=3, continuous=True)
    env.reset(seed=0)
    env._elapsed[:] = np.arange(n) % H
    buf.begin_episode(env.packed)
    log_rows, log_acts, starts = [env.packed.cpu().numpy().copy()], [np.zeros((n, 4), np.float32)], np.zeros(n, np.int64)
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    for t in range(steps):
        a = torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1
        _, _, term, trunc, _ = env.step(a)
        done = (term | trunc)
        buf.append(a, env.packed, done)
        log_rows.append(env.packed.cpu().numpy().copy()); log_acts.append(a.cpu().numpy().copy())
        starts[done.numpy()] = t + 1
    assert np.array_equal(buf.episode_start.cpu().numpy(), starts)
    rows = buf.relabel(1024).clone()
    buf.reseed(3)
    t, w, tg = (x.cpu().numpy().astype(np.int64) for x in buf.sample_indices(1024, 4))
    lo = np.maximum(starts[w], steps - H)
    assert np.all(t >= lo) and np.all(t < steps) and np.all((tg < 0) | ((tg > t) & (tg <= steps)))
    L, A = np.stack(log_rows), np.stack(log_acts)
    o, gd = buf.obs_dim, buf.goal_dim
    got = {k: v.cpu().numpy() for k, v in buf.split(rows).items()}
    goal = np.where((tg >= 0)[:, None], L[np.maximum(tg, 0), w][:, o: o + gd], L[t, w][:, o + gd: o + 2 * gd])
    assert np.array_equal(got["observation"], L[t, w][:, :o]) and np.array_equal(got["next_observation"], L[t + 1, w][:, :o])
    assert np.array_equal(got["action"], A[t + 1, w]) and np.array_equal(got["desired_goal"], goal)
    d = np.linalg.norm(L[t + 1, w][:, o: o + gd].astype(np.float64) - goal, axis=1)
    assert np.array_equal(got["reward"][:, 0], -(d > 0.05).astype(np.float32))        # fetch_env.py:74-80, sparse: the fp64 compare, every sample


def test_nothing_to_sample_right_after_a_lockstep_reset():
    """all worlds reset in the same step (episodes in lock-step): the ring holds no transition of any current episode -> relabel() writes nothing"""
    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.her import HerReplay

    env = grx.make_vec("FetchReach-v4", num_envs=8, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=3)
    buf = HerReplay(env, horizon=3, capacity=256, continuous=True)
    env.reset(seed=0)
    buf.begin_episode(env.packed)
    sizes = []
    for t in range(7):
        a = torch.zeros(8, 4, device="cuda:0")
        _, _, term, trunc, _ = env.step(a)
        buf.append(a, env.packed, term | trunc)
        sizes.append(len(buf.relabel(32)))
    assert sizes == [32, 32, 0, 32, 32, 0, 32]
    torch.cuda.synchronize()


def test_index_sampler_distribution():
    """grx_her_sample: worlds uniform among those with a transition, transitions uniform inside the current episode, k / (k + 1) of the goals substituted"""
    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.her import HerReplay

    n, H = 64, 20
    env = grx.make_vec("FetchReach-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None)
    buf = HerReplay(env, horizon=H, capacity=256, seed=9, continuous=True)
    env.reset(seed=0)
    buf.begin_episode(env.packed)
    for _ in range(12):
        buf.append(torch.zeros(n, 4, device="cuda:0"), env.packed)
    starts = np.zeros(n, np.int64); starts[:8] = 12; starts[8:16] = 7      # eight worlds just reset (nothing to sample), eight five rows into a new episode
    buf.set_episode_start(starts)
    B = 1 << 17
    t, w, tg = (x.cpu().numpy().astype(np.int64) for x in buf.sample_indices(B, 4))
    cnt = np.bincount(w, minlength=n)
    assert (cnt[:8] == 0).all() and np.abs(cnt[8:] - B / 56).max() < 6 * np.sqrt(B / 56)
    assert np.all(t >= starts[w]) and np.all(t < 12) and np.all((tg < 0) | ((tg > t) & (tg <= 12)))
    old = w >= 16                                                           # episodes that began at row 0: t uniform on 0 .. 11
    ct = np.bincount(t[old], minlength=12)
    assert np.abs(ct - old.sum() / 12).max() < 6 * np.sqrt(old.sum() / 12)
    assert abs((tg >= 0).mean() - 0.8) < 0.01
    last = t == 11                                                          # the newest transition has one later row only
    assert (tg[last & (tg >= 0)] == 12).all()
    t2, w2, tg2 = (x.cpu().numpy() for x in buf.sample_indices(B, 4))       # the stream advances with every call and replays after reseed()
    assert not np.array_equal(t2, t)
    buf.reseed(9)
    t3, w3, tg3 = (x.cpu().numpy() for x in buf.sample_indices(B, 4))
    assert np.array_equal(t3, t) and np.array_equal(w3, w) and np.array_equal(tg3, tg)


def test_last_transition_of_every_episode_is_stored_under_same_step_autoreset():
    """README.md:66-76 / core.py:45-67: a HER replay sees WHOLE episodes.  Under same-step autoreset the row a world writes in the step that resets it is the
    first row of its new episode; the reset kernel parks the terminal row (FetchVecEnv.final_packed) and HerReplay.append(final_rows=...) keeps the finished
    episode sampleable for that one step, last transition included.  Checked against a numpy HER over a complete host-side log (terminal rows from
    info["final_obs"])."""
    import torch

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.her import HerReplay

    n, H, steps = 32, 10, 40              # staggered: in every step n / H worlds reach their time limit
    env = grx.make_vec("FetchPush-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=H)
    buf = HerReplay(env, horizon=H, capacity=8192, seed=5, continuous=True)
    env.reset(seed=0)
    env._elapsed[:] = np.arange(n) % H
    buf.begin_episode(env.packed)
    L, A = [env.packed.cpu().numpy().copy()], [np.zeros((n, 4), np.float32)]
    starts, prev, term_t, term_rows = np.zeros(n, np.int64), np.zeros(n, np.int64), -np.ones(n, np.int64), np.zeros((n, env.packed.shape[1]), np.float32)
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    o, gd = buf.obs_dim, buf.goal_dim
    for t in range(steps):
        a = torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1
        _, r, term, trunc, info = env.step(a)
        done = (term | trunc).numpy()
        buf.append(a, env.packed, term | trunc, final_rows=env.final_packed)
        L.append(env.packed.cpu().numpy().copy()); A.append(a.cpu().numpy().copy())
        if done.any():
            fo = info["final_obs"]
            rows = np.nonzero(done)[0]
            fp = env.final_packed.cpu().numpy()
            assert np.array_equal(fp[rows, :o], fo["observation"].cpu().numpy()) and np.array_equal(fp[rows, o: o + gd], fo["achieved_goal"].cpu().numpy())
            assert np.array_equal(fp[rows, -2], r.cpu().numpy()[rows])         # the reported reward of the step IS the terminal row's
            prev[rows], term_t[rows], starts[rows], term_rows[rows] = starts[rows], t + 1, t + 1, fp[rows]
    assert np.array_equal(buf.episode_start.cpu().numpy(), starts) and np.array_equal(buf.term_t.cpu().numpy(), term_t) and np.array_equal(buf.prev_start.cpu().numpy(), prev)
    just = term_t == steps
    assert just.sum() == n // H + (1 if n % H > (steps - 1) % H else 0) or just.sum() >= 3
    B = 4096
    rows = buf.relabel(B).clone()
    buf.reseed(5)
    t, w, tg = (x.cpu().numpy().astype(np.int64) for x in buf.sample_indices(B, 4))
    lo = np.where(just[w], np.maximum(prev[w], steps - H), np.maximum(starts[w], steps - H))
    assert np.all(t >= lo) and np.all(t < steps) and np.all((tg < 0) | ((tg > t) & (tg <= steps)))
    assert just[w].any() and (~just[w]).any()
    Ls, As = np.stack(L), np.stack(A)
    row_at = lambda tt, ww: np.where(((tt == term_t[ww]) & just[ww])[:, None], term_rows[ww], Ls[tt, ww])     # the finished episode's row `steps` is its terminal row
    r0, r1 = Ls[t, w], row_at(t + 1, w)
    goal = np.where((tg >= 0)[:, None], row_at(np.maximum(tg, 0), w)[:, o: o + gd], r0[:, o + gd: o + 2 * gd])
    got = {k: v.cpu().numpy() for k, v in buf.split(rows).items()}
    assert np.array_equal(got["observation"], r0[:, :o]) and np.array_equal(got["next_observation"], r1[:, :o]) and np.array_equal(got["next_achieved_goal"], r1[:, o: o + gd])
    assert np.array_equal(got["action"], As[t + 1, w]) and np.array_equal(got["desired_goal"], goal)
    last = just[w] & (t == steps - 1)
    assert last.sum() > 0, "the last transition of a just-finished episode must be drawn"
    assert not np.array_equal(r1[last][:, :o], Ls[steps, w[last]][:, :o])                # ... and its next observation is the terminal one, not the reset one
    own = last & (tg < 0)
    if own.any():                                                                          # un-relabelled: the reward the env reported for that step
        assert np.array_equal(got["reward"][own, 0], term_rows[w[own], -2]) and np.array_equal(got["success"][own, 0], term_rows[w[own], -1])
    d = np.linalg.norm(r1[:, o: o + gd].astype(np.float64) - goal, axis=1)
    assert np.array_equal(got["reward"][:, 0], -(d > 0.05).astype(np.float32))        # fetch_env.py:74-80, sparse: the fp64 compare, every sample
    # a world that resets in a step is no longer skipped: the buffer always has something to sample
    env2 = grx.make_vec("FetchReach-v4", num_envs=8, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=3)
    b2 = HerReplay(env2, horizon=3, capacity=256, continuous=True)
    env2.reset(seed=0)
    b2.begin_episode(env2.packed)
    sizes = []
    for k in range(7):
        z = torch.zeros(8, 4, device="cuda:0")
        _, _, term, trunc, _ = env2.step(z)
        b2.append(z, env2.packed, term | trunc, final_rows=env2.final_packed)
        sizes.append(len(b2.relabel(32)))
    assert sizes == [32] * 7
