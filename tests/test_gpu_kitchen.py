"""GPU parity tests for FrankaKitchen-v1 (pytest -m gpu): everything goes through the C ABI (grx_kitchen_step, grx_sample_uniform_rows) via KitchenVecEnv;
the oracle / golden fixtures are only the checker."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _env(n, **kw):
    import torch

    from gymnasium_robotics_amd import make_vec

    assert torch.cuda.is_available(), "these tests need the GPU"
    return make_vec("FrankaKitchen-v1", num_envs=n, device="cuda:0", **kw)


def test_noise_sampler_equals_numpy():
    """grx_sample_uniform_rows advances numpy's PCG64 streams bit-exactly: the 59 draws of an observation = four Generator.uniform(-1, 1) calls (9, 9, 21, 20)"""
    from gymnasium_robotics_amd import _native
    from gymnasium_robotics_amd.core import np_random

    L = _native.lib()
    st, mask = np.zeros((3, 4), np.uint64), (1 << 64) - 1
    gens = []
    for i, sd in enumerate((0, 7, 123456)):
        g = np_random(sd)[0]
        s = g.bit_generator.state["state"]
        st[i] = [s["state"] >> 64, s["state"] & mask, s["inc"] >> 64, s["inc"] & mask]
        gens.append(g)
    out = np.zeros((3, 59), np.float32)
    for rep in range(2):     # two consecutive observations: the stream position carries over
        _native.check(L.grx_sample_uniform_rows(st.ctypes.data, None, 3, 59, out.ctypes.data))
        want = np.stack([np.concatenate([g.uniform(low=-1.0, high=1.0, size=k) for k in (9, 9, 21, 20)]) for g in gens])
        assert np.array_equal(out, want.astype(np.float32))


def test_teacher_forced_step_matches_golden():
    """All 248 fixtures in one launch, fed the recorded noise draws; the specialised kernel must be the one that ran (shape 30)."""
    import torch

    g = np.load(os.path.join(GOLDEN, "kitchen_teacher.npz"))
    n = g["obs"].shape[0]
    env = _env(n, autoreset_mode="disabled", max_episode_steps=None)
    assert env._L.grx_model_dim(env._h, b"shape") == 30
    env.reset(seed=0)
    f32 = lambda k: torch.from_numpy(g[k].astype(np.float32)).to(env.device)
    env.qpos.copy_(f32("qpos")); env.qvel.copy_(f32("qvel")); env.qacc_ws.copy_(f32("qacc_ws")); env.last_qpos.copy_(f32("last_qpos"))
    env._draw_noise = lambda idx=None: env.noise.copy_(f32("noise"))      # the fixture's draws instead of the env's own streams
    obs, r, term, trunc, info = env.step(g["action"])
    assert int(info["status"].max()) & 0xFFFF == 0
    e = np.abs(obs["observation"] - g["obs"])
    pos, vel = np.concatenate([e[:, :9], e[:, 18:39]], axis=1).max(axis=1), np.concatenate([e[:, 9:18], e[:, 39:]], axis=1).max(axis=1)
    print(f"positions p50 {np.median(pos):.2e} p95 {np.quantile(pos, 0.95):.2e} max {pos.max():.2e}; velocities p50 {np.median(vel):.2e} p90 {np.quantile(vel, 0.9):.2e} max {vel.max():.2e}")
    # north_star's bound on every snapshot away from an activation boundary (tests/test_gpu_tolerance_table.py) and on >= 99 % of all: round 3 had 14 of the 248
    # fixtures off by 1e-4 ... 7e-2 (hull / cylinder contacts through the fp32 portal routine); two remain, both with a finger-pad contact listed or not within
    # 2e-7 m of its margin in one of the 40 substeps
    posed = g["activation_gap"] >= 1e-6
    assert posed.mean() > 0.7 and pos[posed].max() < 1e-4 and vel[posed].max() < 1e-4, (float(pos[posed].max()), float(vel[posed].max()))
    assert np.mean(pos < 1e-4) >= 0.99 and np.mean(vel < 1e-4) >= 0.99 and pos.max() < 1e-2 and vel.max() < 0.5
    assert np.array_equal(env.completed.cpu().numpy(), g["completed"])
    # every fixture world is a fresh episode with all seven tasks open: reward = number of tasks inside their threshold (kitchen_env.py:340-354)
    assert np.array_equal(r, np.array([bin(int(c)).count("1") for c in g["completed"]], dtype=np.float64))


def test_reset_matches_golden_and_reference_draws():
    g = np.load(os.path.join(GOLDEN, "kitchen_teacher.npz"))
    n = len(g["reset_seed"])
    env = _env(n)
    obs, info = env.reset(seed=int(g["reset_seed"][0]))
    assert obs["observation"].shape == (n, 59) and obs["observation"].dtype == np.float64 and set(obs["achieved_goal"]) == set(obs["desired_goal"]) and len(obs["achieved_goal"]) == 7
    assert np.abs(obs["observation"] - g["reset_obs"]).max() < 1e-5            # world i = the reference env seeded seed + i, noise included
    assert (info["tasks_to_complete"] == 127).all()


def test_api_contract_tasks_and_autoreset():
    env = _env(6, tasks_to_complete=["microwave", "kettle"], max_episode_steps=4, autoreset_mode="same_step", robot_noise_ratio=0.0, object_noise_ratio=0.0)
    obs, info = env.reset(seed=1)
    assert set(obs["achieved_goal"]) == {"microwave", "kettle"} and obs["achieved_goal"]["kettle"].shape == (6, 7)
    with pytest.raises(ValueError, match="Action dimension mismatch"):
        env.step(np.zeros((6, 8), np.float32))
    with pytest.raises(ValueError, match="cannot be found"):
        _env(2, tasks_to_complete=["toaster"])
    mw = 1 << 5
    env.qpos[2, 22] = -0.75                         # world 2: microwave door at its goal -> one completion, the task leaves the open list
    obs, r, term, trunc, info = env.step(np.zeros((6, 9), np.float32))
    assert r[2] == 1.0 and r.sum() == 1.0 and info["step_task_completions"][2] == mw and info["tasks_to_complete"][2] == (1 << 6) and not term.any()
    assert env.task_names(info["episode_task_completions"][2]) == ["microwave"]
    obs, r, term, trunc, info = env.step(np.zeros((6, 9), np.float32))
    assert r.sum() == 0.0                           # remove_task_when_completed: no second reward for the same task
    for _ in range(2):
        obs, r, term, trunc, info = env.step(np.zeros((6, 9), np.float32))
    assert trunc.all() and "final_obs" in info and (info["tasks_to_complete"] == (mw | 1 << 6)).all()      # autoreset restored the task lists
    assert np.abs(obs["observation"][:, :9] - env._init_qpos[:9].cpu().numpy()).max() < 1e-6


def test_worlds_are_independent_and_deterministic():
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, (6, 16, 9)).astype(np.float32)
    outs = []
    for _ in range(2):
        env = _env(16)
        env.reset(seed=0)
        outs.append(np.stack([env.step(a)[0]["observation"] for a in acts]))
    assert np.array_equal(outs[0], outs[1])
    env1 = _env(1)
    env1.reset(seed=5)
    solo = np.stack([env1.step(a[5:6])[0]["observation"][0] for a in acts])
    assert np.array_equal(solo, outs[0][:, 5])


@pytest.mark.parametrize("radius", [0.1, 0.004])
def test_skin_list_is_exact(radius):
    """The broad-phase skin list (grx_collision: candidates within `radius` of the bounding test, rebuilt when a geom has moved radius / 2) must not change a
    single bit: same seeds and actions with the list and with the full sweep of all 3 736 candidates in every substep (skin_radius=0).  radius 4 mm
    forces a rebuild almost every substep, 0.1 m (the default) about once per env.step; the rollout crosses a time-limit autoreset."""
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (7, 48, 9)).astype(np.float32)
    outs = []
    for r in (0.0, radius):
        env = _env(48, skin_radius=r, max_episode_steps=4, autoreset_mode="same_step")
        assert (env._skin is None) == (r == 0.0)
        env.reset(seed=11)
        rows = []
        for a in acts:
            obs, rew, term, trunc, info = env.step(a)
            rows.append(np.concatenate([obs["observation"], rew[:, None], env.qacc_ws.cpu().numpy()], axis=1))
        outs.append(np.stack(rows))
        if r:
            hdr = env._skin[:, :2].cpu().numpy()
            assert (hdr[:, 1] == 1).all() and (hdr[:, 0] > 0).all() and (hdr[:, 0] < 3736).all()      # every world owns a valid list that is shorter than the flat one
            print(f"skin {r}: list length p50 {np.median(hdr[:, 0]):.0f} max {hdr[:, 0].max()}")
    assert np.array_equal(outs[0], outs[1])


def test_device_noise_streams_equal_numpy():
    """grx_uniform_rows_device: the worlds' numpy PCG64 streams advanced on the MI355X (128-bit LCG in 64-bit halves, fp64 uniform(-1, 1), float32 rows) are bit-equal to
    Generator.uniform(-1, 1) -- the four draws of an observation (franka_env.py:118-127: 9 + 9, kitchen_env.py:361-369: 21 + 20); masked-out worlds keep state and rows"""
    import torch

    from gymnasium_robotics_amd import _native
    from gymnasium_robotics_amd.core import np_random

    L = _native.lib()
    n = 300
    st, m64 = np.zeros((n, 4), np.uint64), (1 << 64) - 1
    gens = []
    for i in range(n):
        g = np_random(1000 + 7 * i)[0]
        s = g.bit_generator.state["state"]
        st[i] = [s["state"] >> 64, s["state"] & m64, s["inc"] >> 64, s["inc"] & m64]
        gens.append(g)
    d_st = torch.from_numpy(st.view(np.int64)).to("cuda:0")
    out = torch.full((n, 59), 9.0, device="cuda:0")
    mask = torch.ones(n, dtype=torch.uint8, device="cuda:0")
    mask[5] = 0; mask[299] = 0
    for rep in range(3):
        _native.check(L.grx_uniform_rows_device(d_st.data_ptr(), mask.data_ptr(), n, 59, out.data_ptr(), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i in range(n):
            if i in (5, 299):
                assert (got[i] == 9.0).all()
                continue
            want = np.concatenate([gens[i].uniform(low=-1.0, high=1.0, size=k) for k in (9, 9, 21, 20)]).astype(np.float32)
            assert np.array_equal(got[i], want), (rep, i)
    host = d_st.cpu().numpy().view(np.uint64)
    assert np.array_equal(host[5], st[5]) and np.array_equal(host[299], st[299])
    s0 = gens[0].bit_generator.state["state"]
    assert int(host[0, 0]) == s0["state"] >> 64 and int(host[0, 1]) == s0["state"] & m64


@pytest.mark.parametrize("mode", ["same_step", "next_step"])
def test_device_path_equals_host_path(mode):
    """output="torch" (noise streams, bookkeeping, TimeLimit and autoreset on the device: no host synchronisation) against output="numpy" (the host path the
    fixtures pin): the same seed gives bit-identical observations -- the noise is the same bits -- rewards, flags and task masks, step by step, resets included."""
    import torch

    kw = dict(tasks_to_complete=["microwave", "kettle"], max_episode_steps=4, autoreset_mode=mode)
    a, b = _env(6, output="numpy", **kw), _env(6, output="torch", **kw)
    assert b._device_path and not a._device_path
    oa, ia = a.reset(seed=11)
    ob, ib = b.reset(seed=11)
    assert np.array_equal(oa["observation"].astype(np.float32), ob["observation"].cpu().numpy())
    rng = np.random.default_rng(0)
    for t in range(11):
        if t in (1, 6):            # world 2: microwave door at its goal -> a completion (and, at t = 6, kettle too for world 4: both tasks done -> terminated)
            for e in (a, b):
                e.qpos[2, 22] = -0.75
        if t == 6:
            for e in (a, b):
                e.qpos[4, 22] = -0.75
                e.qpos[4, 23:30] = torch.tensor([-0.23, 0.75, 1.62, 0.99, 0.0, 0.0, -0.06], device="cuda:0")
        act = rng.uniform(-1, 1, (6, 9)).astype(np.float32)
        oa, ra, ta, ua, ia = a.step(act)
        ob, rb, tb, ub, ib = b.step(torch.from_numpy(act).to("cuda:0"))
        assert np.array_equal(oa["observation"].astype(np.float32), ob["observation"].cpu().numpy()), t
        for k in oa["achieved_goal"]:
            assert np.array_equal(oa["achieved_goal"][k].astype(np.float32), ob["achieved_goal"][k].cpu().numpy())
        assert np.array_equal(ra, rb.cpu().numpy().astype(np.float64)) and np.array_equal(ta, tb.cpu().numpy()) and np.array_equal(ua, ub.cpu().numpy()), t
        for k in ("tasks_to_complete", "step_task_completions", "episode_task_completions"):
            assert np.array_equal(ia[k], ib[k].cpu().numpy().astype(np.int64)), (t, k)
        if mode == "same_step":
            done = ta | ua
            assert np.array_equal(done, ib["_final_obs"].cpu().numpy())
            if done.any():
                idx = np.nonzero(done)[0]
                assert np.array_equal(ia["final_obs"]["observation"].astype(np.float32), ib["final_obs"]["observation"].cpu().numpy()[idx])
                for k in ("tasks_to_complete", "step_task_completions", "episode_task_completions"):
                    assert np.array_equal(ia["final_info"][k], ib["final_info"][k].cpu().numpy()[idx].astype(np.int64)), (t, k)
    assert (ta | ua).any() or t > 0


def test_torch_step_does_not_synchronise():
    """KitchenVecEnv.step(output="torch") only enqueues: with torch's synchronisation debug mode set to "error" any blocking read-back / stream wait raises"""
    import torch

    env = _env(64, output="torch", autoreset_mode="same_step", max_episode_steps=3)
    env.reset(seed=0)
    act = torch.zeros(64, 9, device="cuda:0")
    env.step(act)
    torch.cuda.synchronize()
    try:
        torch.cuda.set_sync_debug_mode("error")
    except Exception:
        pytest.skip("this torch build has no synchronisation debug mode")
    try:
        for _ in range(5):
            obs, r, term, trunc, info = env.step(act)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert r.is_cuda and term.is_cuda and info["tasks_to_complete"].is_cuda and bool((info["_final_obs"] == (term | trunc)).all())


def test_overflow_lane_polling_equals_the_serialised_rerun(monkeypatch):
    """kitchen: the polling workgroups of the standing lane launch against the serialised re-run (see tests/test_gpu_adroit.py): bit-identical rollouts"""
    import torch

    n, envs = 8192, []
    for poll in ("16", "0"):
        monkeypatch.setenv("GRX_LANE_POLL", poll)
        e = _env(n, output="torch", autoreset_mode="disabled", max_episode_steps=None)
        assert e.lane is not None and e.lane.poll_grid == int(poll)
        e.reset(seed=8)
        envs.append(e)
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    entered = 0
    for t in range(70):
        a = torch.rand(n, 9, device="cuda:0", generator=g) * 2 - 1
        outs = [e.step(a) for e in envs]
        entered += len(envs[1].lane.entered_last_step())
        assert torch.equal(envs[0].qpos, envs[1].qpos) and torch.equal(envs[0].qvel, envs[1].qvel) and torch.equal(outs[0][0]["observation"], outs[1][0]["observation"]), t
    assert entered >= 1, entered


@pytest.mark.parametrize("parts", [2, 5])
def test_split_step_is_the_plain_step(monkeypatch, parts):
    """Round 6: the kitchen step launch with P workgroups per world, each running its share of the 40 substeps and handing the world on through a carrier row (include/grx_capi.h
    grx_kitchen_buffers.split_parts; a part rebuilds the broad-phase skin list instead of reading what another CU wrote), against the plain launch: state rows, observations,
    completion bits, last_qpos and status words are BIT-IDENTICAL after every step -- same-step autoresets at a short time limit, the noise streams on, the standing overflow lane with its
    polling workgroups, the cost-ordered dispatch.  The reference's step is one env.step() whatever the launch geometry (/root/reference/gymnasium_robotics/envs/franka_kitchen/kitchen_env.py:399-423)."""
    import torch

    import gymnasium_robotics_amd as grx

    n, envs = 2048, []
    for p_ in (1, parts):
        monkeypatch.setenv("GRX_KITCHEN_SPLIT", str(p_))
        e = grx.make_vec("FrankaKitchen-v1", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=7)
        e.reset(seed=5)
        envs.append(e)
    plain, split = envs
    assert plain._split == 1 and split._split == parts
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(9)
    for t in range(16):
        a = torch.rand(n, 9, device="cuda:0", generator=gen) * 2 - 1
        outs = [e.step(a) for e in envs]
        for name in ("qpos", "qvel", "qacc_ws", "obs", "completed", "last_qpos", "status"):
            assert torch.equal(getattr(split, name), getattr(plain, name)), (t, name, int((getattr(split, name) != getattr(plain, name)).sum()))
        assert int(split._split_state.abs().max()) == 0, t      # every world's words are clean again
    assert int((split.status & 1).max()) == 0 and torch.isfinite(split.qpos).all()
