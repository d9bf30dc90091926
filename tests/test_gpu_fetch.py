"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything goes through the C ABI of
libgrx_hip.so via FetchVecEnv; the oracle / golden fixtures are only the checker.

Tolerances (stated by BASELINE.json north_star): obs/reward within 1e-4 of the reference path,
done/success flags bit-exact.  "Teacher-forced" = every compared step starts from the oracle's
pre-step state, which is how per-step parity is defined for chaotic contact dynamics (SURVEY.md §7 hard part 2).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4
IDS = {"FetchReach": "FetchReach-v4", "FetchPush": "FetchPush-v4", "FetchPickAndPlace": "FetchPickAndPlace-v4", "FetchSlide": "FetchSlide-v4"}


def _env(task, n, **kw):
    import torch

    from gymnasium_robotics_amd.envs.fetch import FetchVecEnv

    assert torch.cuda.is_available(), "these tests need the GPU"
    return FetchVecEnv(IDS[task], num_envs=n, device="cuda:0", **kw)


def _load_state(env, g, sel):
    import torch

    dev = env.device
    env.load_world_rows({k: g[k][sel] for k in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal")})


@pytest.mark.parametrize("task", ["FetchReach", "FetchPush", "FetchPickAndPlace"])
def test_teacher_forced_step_matches_golden(task):
    g = np.load(os.path.join(GOLDEN, f"fetch_{task}_teacher.npz"))
    n = g["obs"].shape[0]
    env = _env(task, n, autoreset_mode="disabled", max_episode_steps=None)
    env.reset(seed=0)
    _load_state(env, g, slice(None))
    obs, r, term, trunc, info = env.step(g["action"])
    assert int(np.abs(info["status"]).max()) == 0
    err = np.abs(obs["observation"] - g["obs"]).max(axis=1)
    # fp32 vs the fp64 oracle.  MuJoCo's soft constraints are discontinuous where a row switches on or off (a contact is listed when dist < margin, and the reference
    # acceleration -b*v - k*d*r jumps by b*v): the oracle records per snapshot how close any such switch came to its threshold during the step ("activation_gap",
    # tests/test_gpu_tolerance_table.py).  Asserted: (1) EVERY snapshot whose gap is at least 1e-6 m -- ten times the resolution an fp32 state has at 1 m -- is
    # within 1e-4, (2) at least 99 % of ALL snapshots are, (3) nothing is off by more than 5e-3.
    posed = g["activation_gap"] >= 1e-6
    assert posed.mean() > 0.7
    assert err[posed].max() < TOL, f"worst snapshot {np.nonzero(posed)[0][err[posed].argmax()]} err {err[posed].max():.3e}"
    assert np.mean(err < TOL) >= 0.99, f"only {100 * np.mean(err < TOL):.1f}% of snapshots within 1e-4"
    assert err.max() < 5e-3, f"snapshot {err.argmax()} err {err.max():.3e} gap {g['activation_gap'][err.argmax()]:.2e}"
    assert np.abs(obs["achieved_goal"] - g["achieved"])[posed].max() < TOL
    print(f"{task}: {posed.sum()}/{n} snapshots away from activation boundaries, max err there {err[posed].max():.2e}; all: p50 {np.median(err):.2e} "
          f"p99 {np.quantile(err, 0.99):.2e} max {err.max():.2e}")
    # flags / sparse reward: EXACTLY the reference's functions (fetch_env.py:74-80, 168-170: fp64 norm, d < 0.05 / d > 0.05) of the returned goals -- every snapshot,
    # no band around the threshold (the device takes the distance and the compare in fp64, csrc/grx_fetch_task.h grx_goal_distance3)
    d_hip = np.linalg.norm(obs["achieved_goal"].astype(np.float64) - obs["desired_goal"].astype(np.float64), axis=-1)
    assert np.array_equal(info["is_success"], (d_hip < 0.05).astype(np.float32))
    assert np.array_equal(r, -(d_hip > 0.05).astype(np.float32))
    # ... and therefore the oracle's flags wherever the difference of the two achieved goals cannot flip the compare (it can for no snapshot of these fixtures)
    d = np.linalg.norm(g["achieved"] - g["goal"], axis=-1)
    decided = np.abs(d - 0.05) > np.abs(d_hip - d)
    assert decided.all(), int((~decided).sum())
    assert np.array_equal(r[decided], g["reward"][decided].astype(np.float32))
    assert np.array_equal(info["is_success"][decided], g["success"][decided].astype(np.float32))
    assert not term.any() and not trunc.any()
    # post-step state
    assert np.abs(env.qpos.cpu().numpy() - g["qpos_next"])[posed].max() < TOL


def test_teacher_forced_hull_contacts_match_golden():
    """Hull-vs-convex narrow phase (mesh-mesh / mesh-box pairs of the Fetch links, assets/fetch/robot.xml:16-93): 168 snapshots of scripted
    rollouts that fold the arm into the head / torso and press the wrist / gripper housing onto the table (tools/make_golden_hull.py), 147 of
    them with hull contacts.  The portal refinement and its support functions run in fp64 on the device (GRX_MPR_REAL; the fp32 vertex scan's winner is refined
    over the hull graph): every snapshot away from an activation boundary is within north_star's 1e-4 (measured max 4e-6; round 3: 98.2 %, max 1.7e-3)."""
    g = np.load(os.path.join(GOLDEN, "fetch_hull_teacher.npz"))
    n = g["obs"].shape[0]
    env = _env("FetchPickAndPlace", n, autoreset_mode="disabled", max_episode_steps=None)
    env.reset(seed=0)
    _load_state(env, g, slice(None))
    obs, r, term, trunc, info = env.step(g["action"])
    assert int((info["status"] & ~6).max()) == 0   # no bad number / solver failure (capacity flags may fire in the folded poses)
    err = np.abs(obs["observation"] - g["obs"]).max(axis=1)
    hull = g["hull_contacts"] > 0
    print(f"hull snapshots: p50 {np.median(err[hull]):.2e} p90 {np.quantile(err[hull], 0.9):.2e} max {err[hull].max():.2e}; others max {err[~hull].max():.2e}")
    assert hull.sum() > 120
    posed = g["activation_gap"] >= 1e-6
    assert posed.mean() > 0.9 and err[posed].max() < TOL and np.mean(err < TOL) >= 0.99, (float(err[posed].max()), float(np.mean(err < TOL)))
    assert np.median(err[hull]) < 2e-6 and err.max() < 5e-3


def test_hull_caches_do_not_change_the_rollout(monkeypatch):
    """The per-world HBM row of the hull pairs (cached separating directions, guessed support vertices of persistent contacts: csrc/grx_engine.h, grx_mesh_pairs /
    grx_mesh_support) only replaces work whose outcome it proves: free-running from the 168 folded-arm poses of the hull fixture -- arm on head, wrist on the table, hull
    contacts in most steps -- the rollout with the row is bit-identical to the rollout without it (GRX_NO_HULLCACHE: every portal search scans its hulls)."""
    import torch

    g = np.load(os.path.join(GOLDEN, "fetch_hull_teacher.npz"))
    n = g["obs"].shape[0]
    envs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("GRX_NO_HULLCACHE", "1")
        e = _env("FetchPickAndPlace", n, autoreset_mode="disabled", max_episode_steps=None, output="torch")
        assert (e.hullcache is None) == off
        e.reset(seed=0)
        _load_state(e, g, slice(None))
        envs.append(e)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(3)
    a0 = torch.from_numpy(g["action"].astype(np.float32)).cuda()
    for t in range(30):
        a = a0 if t < 10 else (a0 * 0.2 + 0.3 * (torch.rand(n, 4, device="cuda:0", generator=gen) * 2 - 1))     # keep pressing, with a little noise
        for e in envs:
            e.step(a)
        assert torch.equal(envs[0].qpos, envs[1].qpos) and torch.equal(envs[0].qvel, envs[1].qvel) and torch.equal(envs[0].obs, envs[1].obs), t
    used = envs[0].hullcache[:, 21:].abs().sum(dim=1) > 0
    assert int(used.sum()) > 20      # the guesses were really in play: worlds with a persistent hull contact wrote their rows


def test_hull_candidate_lists_do_not_change_the_rollout(monkeypatch):
    """The support-candidate lists of the hulls (GrxModel::mesh_cellhdr: a support evaluation reads the <= 64 records of its direction's cube-map cell instead of scanning the
    hull) hold, for every direction, every vertex the scan could pick or tie with (tests/test_cpu_hull_cells.py): free-running from the 168 folded-arm poses of the hull
    fixture, the rollout of a model created WITH the lists is bit-identical to the rollout of one created without them (GRX_NO_HULLCELLS: every evaluation whose guess fails
    scans the hull), with and without the per-world guesses."""
    import torch

    g = np.load(os.path.join(GOLDEN, "fetch_hull_teacher.npz"))
    n = g["obs"].shape[0]
    envs = []
    for cells_off, cache_off in ((False, False), (True, False), (False, True), (True, True)):
        for var, off in (("GRX_NO_HULLCELLS", cells_off), ("GRX_NO_HULLCACHE", cache_off)):
            if off:
                monkeypatch.setenv(var, "1")
            else:
                monkeypatch.delenv(var, raising=False)
        e = _env("FetchPickAndPlace", n, autoreset_mode="disabled", max_episode_steps=None, output="torch")
        e.reset(seed=0)
        _load_state(e, g, slice(None))
        envs.append(e)
    assert len({e._h.value for e in envs[:2]}) == 2      # two native models: one with the lists, one without
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(3)
    a0 = torch.from_numpy(g["action"].astype(np.float32)).cuda()
    for t in range(30):
        a = a0 if t < 10 else (a0 * 0.2 + 0.3 * (torch.rand(n, 4, device="cuda:0", generator=gen) * 2 - 1))
        for e in envs:
            e.step(a)
        for e in envs[1:]:
            assert torch.equal(envs[0].qpos, e.qpos) and torch.equal(envs[0].qvel, e.qvel) and torch.equal(envs[0].obs, e.obs), t


@pytest.mark.parametrize("task,parts,start", [("FetchPickAndPlace", 2, "rollout"), ("FetchPickAndPlace", 4, "rollout"), ("FetchPickAndPlace", 3, "hull_poses"), ("FetchPush", 2, "rollout"), ("FetchReach", 5, "rollout"),
                                              ("FetchSlide", 2, "rollout")])
def test_split_step_is_the_plain_step(monkeypatch, task, parts, start):
    """Round 6: the step launch with P workgroups per world, each running 1 / P of the substeps (include/grx_capi.h, grx_fetch_buffers.split_parts: the state travels through the world's
    hand-off row, a part waits for the one before it) against the plain launch (one workgroup per world): state rows, observations, rewards, flags, status words and the hull
    caches are BIT-IDENTICAL after every step -- staggered episodes with same-step autoresets, the hull fixture (portal searches in every substep), block_gripper tasks (21 passes:
    uneven shares), the cost-ordered dispatch on.  The reference's step is one env.step() whatever the launch geometry (/root/reference/gymnasium_robotics/envs/robot_env.py:114-152)."""
    import torch

    envs = []
    for p_ in (1, parts):
        monkeypatch.setenv("GRX_FETCH_SPLIT", str(p_))
        if start == "hull_poses":
            g = np.load(os.path.join(GOLDEN, "fetch_hull_teacher.npz"))
            n = g["obs"].shape[0]
            e = _env(task, n, autoreset_mode="disabled", max_episode_steps=None, output="torch")
            e.reset(seed=0)
            _load_state(e, g, slice(None))
        else:
            n = 2048
            e = _env(task, n, autoreset_mode="same_step", max_episode_steps=50, output="torch")
            e.reset(seed=7)
            e._elapsed[:] = np.arange(n) % 50
        envs.append(e)
    plain, split = envs
    assert plain._split == 1 and split._split == parts
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(3)
    for t in range(30 if start == "hull_poses" else 110):
        a = torch.rand(n, 4, device="cuda:0", generator=gen) * 2 - 1
        outs = [e.step(a) for e in envs]
        for name in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "obs", "achieved", "reward", "success", "packed", "goal", "status"):
            assert torch.equal(getattr(split, name), getattr(plain, name)), (t, name, int((getattr(split, name) != getattr(plain, name)).sum()))
        assert torch.equal(outs[0][3], outs[1][3])
        assert int(split._split_state[:, 0].abs().max()) == 0, t      # every world's word is clean again
    assert int(split.status.abs().max()) == 0
    assert split.cost is None or int(split.cost.min()) > 0      # the measured durations of the parts add up to a cost for the next launch's order


@pytest.mark.parametrize("start", ["hull_poses", "rollout"])
def test_handoff_is_the_full_kernels_rollout(monkeypatch, start):
    """Round 6: FetchPickAndPlace steps on a FAST kernel without the hull-pair routine (168 VGPRs, three waves per SIMD, ten worlds per CU); a world in which a hull pair passes
    the bounding-box filter, or that exceeds the fast tables, is handed off MID-STEP -- at the substep in question, state untouched -- to the standing lane, which runs the
    model's full kernel and resumes AT that substep (include/grx_capi.h, grx_fetch_buffers.handoff).  Opt-in (GRX_FETCH_HANDOFF=1: measured no faster).  Against
    the same rollout with every world on the full kernel (the default, what rounds 1 - 5 ran): state rows, observations, rewards, flags and status words are BIT-IDENTICAL after every step, from the 168 folded-arm poses of
    the hull fixture (every world has hull activity from the first substep) and over staggered episodes with same-step autoresets (worlds enter and leave the lane)."""
    import torch

    envs = []
    monkeypatch.setenv("GRX_LANE_FIRST", "1"); monkeypatch.setenv("GRX_LANE_SPACER", "50000")      # the launch order the A/B found best (the lane's workgroups resident before the fast launch)
    for off in (False, True):
        monkeypatch.setenv("GRX_FETCH_HANDOFF", "0" if off else "1")
        if start == "hull_poses":
            g = np.load(os.path.join(GOLDEN, "fetch_hull_teacher.npz"))
            n = g["obs"].shape[0]
            e = _env("FetchPickAndPlace", n, autoreset_mode="disabled", max_episode_steps=None, output="torch")
            e.reset(seed=0)
            _load_state(e, g, slice(None))
        else:
            n = 1024
            e = _env("FetchPickAndPlace", n, autoreset_mode="same_step", max_episode_steps=50, output="torch")
            e.reset(seed=11)
            e._elapsed[:] = np.arange(n) % 50
        envs.append(e)
    fast, full = envs
    assert fast._h_fast is not None and full._h_fast is None and fast.lane.mode == "lane" and full.lane.mode == "entry"
    assert fast.lds_bytes < full.lds_bytes and (160 * 1024) // (-(-fast.lds_bytes // 1280) * 1280) >= 10      # ten worlds per CU by LDS
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    lane_sizes, entrants = [], 0
    for t in range(40 if start == "hull_poses" else 130):
        a = torch.rand(n, 4, device="cuda:0", generator=gen) * 2 - 1
        outs = [e.step(a) for e in envs]
        for name in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "obs", "achieved", "reward", "success", "packed", "goal"):
            assert torch.equal(getattr(fast, name), getattr(full, name)), (t, name, int((getattr(fast, name) != getattr(full, name)).sum()))
        assert torch.equal(fast.status & 0xFFFF, full.status & 0xFFFF) and torch.equal(fast.status >> 16, full.status >> 16), t
        assert torch.equal(outs[0][3], outs[1][3])
        assert int(fast.handoff[:, 0].view(torch.int32).abs().max()) == 0, t      # every hand-off row was consumed inside the step
        lane_sizes.append(fast.lane.count()); entrants += len(fast.lane.entered_last_step())
    assert int(fast.status.abs().max()) == 0
    assert max(lane_sizes) > (100 if start == "hull_poses" else 20) and entrants > 20, (max(lane_sizes), entrants)      # the mechanism was really in play


@pytest.mark.parametrize("task,output,order", [("FetchPickAndPlace", "torch", "before"), ("FetchPickAndPlace", "numpy", "before"), ("FetchPickAndPlace", "torch", "after"),
                                               ("FetchSlide", "torch", "before"), ("FetchReach", "torch", "before")])
def test_overlapped_reset_is_the_inline_reset(monkeypatch, task, output, order):
    """Same-step autoreset, two ways: the reset kernel of the worlds a step will truncate run AHEAD of it on a side stream into staged rows and committed behind the step
    kernel (grx_fetch_commit_rows, the default), against the in-line reset behind the step (GRX_FETCH_AHEAD_RESET=0).  Staggered episodes (some worlds reset in every
    step, a few steps reset none), 130 steps = every world through two or three resets: state rows, outputs, packed and parked terminal rows, goals, status words, the
    device-resident PCG64 streams, flags and info["final_obs"] are bit-identical after every step."""
    import torch

    n, horizon = 192, 50
    envs = []
    monkeypatch.setenv("GRX_FETCH_AHEAD_ORDER", order)
    for on in ("1", "0"):
        monkeypatch.setenv("GRX_FETCH_AHEAD_RESET", on)
        e = _env(task, n, autoreset_mode="same_step", max_episode_steps=horizon, output=output)
        e.reset(seed=11)
        e._elapsed[:] = (np.arange(n) * 7) % 41      # phases 41 .. 49 are empty: 9 steps of every 50 reset no world
        envs.append(e)
    assert envs[0]._ahead is not None and envs[1]._ahead is None
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    resets = 0
    for t in range(130):
        a = torch.rand(n, 4, device="cuda:0", generator=gen) * 2 - 1
        outs = [e.step(a if output == "torch" else a.cpu().numpy()) for e in envs]
        (o0, r0, te0, tr0, i0), (o1, r1, te1, tr1, i1) = outs
        eq = torch.equal if output == "torch" else np.array_equal
        for k in o0:
            assert eq(o0[k], o1[k]), (t, k)
        assert eq(r0, r1) and eq(te0, te1) and eq(tr0, tr1) and eq(i0["is_success"], i1["is_success"]), t
        assert ("final_obs" in i0) == ("final_obs" in i1), t
        if "final_obs" in i0:
            resets += 1
            for k in i0["final_obs"]:
                assert eq(i0["final_obs"][k], i1["final_obs"][k]), (t, k)
        a_, b_ = envs
        for name in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal", "obs", "achieved", "reward", "success", "status", "packed", "final_packed", "_rng_dev"):
            assert torch.equal(getattr(a_, name), getattr(b_, name)), (t, name)
        assert np.array_equal(a_._elapsed, b_._elapsed)
    assert 100 <= resets < 130


def test_compacted_reset_kernel_matches_masked_forward():
    """grx_fetch_reset (compacted list, initial rows + host draws applied on the device) gives the rows the old path produced:
    initial state written by the host + masked grx_fetch_forward."""
    import ctypes

    import torch

    from gymnasium_robotics_amd import _native

    env = _env("FetchPickAndPlace", 64)
    env.reset(seed=11)
    ref = {k: getattr(env, k).clone() for k in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal", "obs", "achieved", "reward", "success")}
    # same worlds through the masked forward path, from the rows the reset kernel was told to write
    env.qacc_ws.fill_(7.0)   # must be zeroed by both paths
    q = env.initial_qpos.unsqueeze(0).repeat(64, 1)
    q[:, env._obj_qadr: env._obj_qadr + 2] = ref["qpos"][:, env._obj_qadr: env._obj_qadr + 2]
    env.qpos.copy_(q); env.qvel.copy_(env.initial_qvel.unsqueeze(0).repeat(64, 1)); env.qacc_ws.zero_(); env.mocap.copy_(env._mocap0.unsqueeze(0).repeat(64, 1))
    env.mask.fill_(1)
    _native.check(env._L.grx_fetch_forward(env._h, ctypes.byref(env.task), ctypes.byref(env._bufs_masked), 64, 0, env._stream()))
    torch.cuda.synchronize()
    for k, v in ref.items():
        assert torch.equal(getattr(env, k), v), k


@pytest.mark.parametrize("task", ["FetchReach-v4", "FetchPush-v4", "FetchSlide-v4", "FetchPickAndPlace-v4"])
def test_device_reset_draws_equal_numpy_bit_for_bit(task):
    """The reset draws are made ON THE DEVICE from device-resident PCG64 streams (grx_fetch_sample_resets_device): over several episodes with ragged reset lists the
    sample rows equal the reference's np_random.uniform sequence (rejection loop + goal, fetch_env.py:153-166, 388-391) rounded to float32, and the streams end at
    numpy's position.  The host draws nothing: reset() itself runs under the sync-debug guard."""
    import torch

    from gymnasium_robotics_amd.core import np_random
    from gymnasium_robotics_amd.envs.fetch import FetchVecEnv, sample_fetch_reset

    n = 96
    env = FetchVecEnv(task, num_envs=n, device="cuda:0")
    seeds = [1000 + 7 * i for i in range(n)]
    env.reset(seed=seeds)
    rngs = [np_random(s)[0] for s in seeds]
    lists = [np.arange(n), np.arange(0, n, 3), np.array([5, 95, 17]), np.arange(n)[::-1].copy(), np.array([0])]
    for ep, idx in enumerate(lists):
        if ep:
            torch.cuda.set_sync_debug_mode("error")
            try:
                staged = env._stage_reset(idx)
            finally:
                torch.cuda.set_sync_debug_mode("default")
            env._launch_reset(staged, idx)
        else:
            staged = (n, None, env._reset_stage.view(-1)[n: 6 * n])
        rows = staged[2].reshape(len(idx), 5).cpu().numpy()
        for k, w in enumerate(idx):
            o_ref, g_ref = sample_fetch_reset(env.cfg, rngs[w], np.asarray(env.initial_gripper_xpos, np.float64), float(env.height_offset))
            assert np.array_equal(rows[k, 2:5], g_ref.astype(np.float32)), (ep, w)
            if o_ref is not None:
                assert np.array_equal(rows[k, 0:2], o_ref.astype(np.float32)), (ep, w)
        goal = env.goal.cpu().numpy()
        assert np.array_equal(goal[idx], rows[:, 2:5])
    for w in (0, 5, 17, 95):
        assert env.world_rng(w).bit_generator.state == rngs[w].bit_generator.state


@pytest.mark.parametrize("task", ["FetchReach", "FetchPush", "FetchPickAndPlace"])
def test_reset_matches_golden(task):
    """reset(seed=s) gives world i the start state + goal of the reference's reset(seed=s+i) (PCG64 draw order)."""
    g = np.load(os.path.join(GOLDEN, f"fetch_{task}_teacher.npz"))
    n = len(g["reset_seed"])
    env = _env(task, n)
    obs, info = env.reset(seed=int(g["reset_seed"][0]))
    assert np.abs(env.initial_gripper_xpos - g["initial_gripper_xpos"]).max() < 2e-5
    assert np.abs(obs["desired_goal"] - g["reset_goal"]).max() < 2e-5
    assert np.abs(obs["observation"] - g["reset_obs"]).max() < TOL
    assert np.abs(env.qpos.cpu().numpy() - g["reset_qpos"]).max() < TOL


def test_live_oracle_free_running_reach():
    """Contact-free task: a whole free-running episode stays within tolerance of the fp64 oracle."""
    from oracle.fetch_oracle import OracleFetchEnv

    env = _env("FetchReach", 2)
    obs, _ = env.reset(seed=3)
    orc = OracleFetchEnv(env.model, "FetchReach")
    o, _ = orc.reset(seed=3)
    rng = np.random.default_rng(5)
    worst = np.abs(obs["observation"][0] - o["observation"]).max()
    for _ in range(50):
        a = rng.uniform(-1, 1, (2, 4)).astype(np.float32)
        obs, r, _, _, info = env.step(a)
        o, ro, _, _, io = orc.step(a[0].astype(np.float64))
        worst = max(worst, np.abs(obs["observation"][0] - o["observation"]).max())
    assert worst < TOL, worst


def test_determinism_rollout_bit_identical():
    """tests/test_envs.py:62-117 of the reference: same seed + same actions => identical outputs."""
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (20, 64, 4)).astype(np.float32)
    outs = []
    for _ in range(2):
        env = _env("FetchPickAndPlace", 64)
        obs, _ = env.reset(seed=0)
        traj = [obs["observation"].copy()]
        for a in acts:
            obs, r, te, tr, info = env.step(a)
            traj += [obs["observation"].copy(), r.copy(), info["is_success"].copy()]
        outs.append(traj)
        env.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def test_reward_invariant_and_her_recompute():
    """core.py:59-62: reward == compute_reward(achieved_goal, desired_goal, info), bit-exactly, and
    compute_reward accepts arbitrary leading batch dims (HER relabelling)."""
    for env_id in ("FetchPickAndPlace-v4", "FetchPickAndPlaceDense-v4"):
        from gymnasium_robotics_amd.envs.fetch import FetchVecEnv

        env = FetchVecEnv(env_id, num_envs=128, device="cuda:0")
        obs, _ = env.reset(seed=1)
        rng = np.random.default_rng(2)
        ags, dgs, rs = [], [], []
        for _ in range(5):
            obs, r, _, _, info = env.step(rng.uniform(-1, 1, (128, 4)).astype(np.float32))
            ags.append(obs["achieved_goal"]); dgs.append(obs["desired_goal"]); rs.append(r)
        ag, dg, rr = np.stack(ags), np.stack(dgs), np.stack(rs)
        rc = env.compute_reward(ag, dg, {})
        assert rc.shape == (5, 128) and rc.dtype == rr.dtype
        assert np.array_equal(rc, rr)
        # relabel with shuffled goals against a numpy restatement of fetch_env.py:74-80
        perm = rng.permutation(128)
        rl = env.compute_reward(ag, dg[:, perm], {})
        d = np.linalg.norm(ag.astype(np.float64) - dg[:, perm].astype(np.float64), axis=-1)       # the reference's fp64 norm of the (fp32) goals handed in
        ref = -(d > 0.05).astype(np.float32) if env.reward_type == "sparse" else (-d).astype(np.float32)
        assert np.array_equal(rl, ref) if env.reward_type == "sparse" else np.allclose(rl, ref, rtol=0, atol=2e-8)


def test_time_limit_and_autoreset_next_step():
    env = _env("FetchReach", 4, max_episode_steps=5)
    env.reset(seed=0)
    a = np.zeros((4, 4), np.float32)
    for t in range(5):
        obs, r, term, trunc, info = env.step(a)
        assert not term.any()
        assert trunc.all() == (t == 4)
    goal_before = obs["desired_goal"].copy()
    obs, r, term, trunc, info = env.step(a)  # this call resets instead of stepping
    assert not trunc.any() and (r == 0).all()
    assert not np.allclose(obs["desired_goal"], goal_before)
    assert (env._elapsed == 0).all()


def test_action_shape_error_and_step_before_reset():
    env = _env("FetchReach", 2)
    with pytest.raises(RuntimeError):
        env.step(np.zeros((2, 4), np.float32))
    env.reset(seed=0)
    with pytest.raises(ValueError, match="Action dimension mismatch"):
        env.step(np.zeros((2, 3), np.float32))


def test_large_batch_worlds_independent():
    """4096 worlds: world i of the big batch == the same world stepped in a batch of 8 (tile-sharding invariant)."""
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (3, 4096, 4)).astype(np.float32)
    big = _env("FetchPickAndPlace", 4096)
    big.reset(seed=100)
    small = _env("FetchPickAndPlace", 8, seed_offset=1000)
    small.reset(seed=100)
    for a in acts:
        ob, rb, _, _, ib = big.step(a)
        os_, rs, _, _, is_ = small.step(a[1000:1008])
    assert np.array_equal(ob["observation"][1000:1008], os_["observation"])
    assert int(np.abs(ib["status"]).max()) == 0


def test_capacity_overflow_is_rerun_not_truncated():
    """No dropped contacts (the reference never truncates, robot_env.py:341): worlds that exceed a table capacity in some substep write nothing and are stepped again
    on the same model with larger tables (grx_fetch_buffers.redo).  Here the step kernel gets deliberately small tables (64 rows) and the contact-rich hull fixture:
    most worlds overflow, every one of them must come out bit-identical to an environment whose tables are large to begin with, with no overflow flag left."""
    import torch

    from gymnasium_robotics_amd.envs.fetch import FetchVecEnv, load_fetch_model

    g = np.load(os.path.join(GOLDEN, "fetch_hull_teacher.npz"))
    n = g["obs"].shape[0]
    base = load_fetch_model("FetchPickAndPlace")
    small = FetchVecEnv("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="numpy", autoreset_mode="disabled", max_episode_steps=None,
                        model=base.with_capacity(maxefc=64, jpool=640))
    trunc = FetchVecEnv("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="numpy", autoreset_mode="disabled", max_episode_steps=None,
                        model=base.with_capacity(maxefc=64, jpool=640), overflow_rerun=False)
    big = FetchVecEnv("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="numpy", autoreset_mode="disabled", max_episode_steps=None,
                      model=base.with_capacity(maxefc=256, jpool=4080), overflow_rerun=False)
    outs = []
    for env in (small, trunc, big):
        env.reset(seed=0)
        env.load_world_rows({k: g[k] for k in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal")})
        obs, r, _, _, info = env.step(g["action"])
        outs.append((obs["observation"], r, info["status"], info["status_sticky"], env.qpos.cpu().numpy(), env.qvel.cpu().numpy()))
    redone = np.zeros(n, bool)
    redone[small.lane.entered_last_step()] = True
    assert redone.sum() >= 5, int(redone.sum())                                    # the small tables do overflow on this fixture
    assert (outs[1][3][redone] & 6).all() and not (outs[1][3][~redone] & 6).any()     # ... and without the re-run exactly those worlds are flagged (contacts dropped)
    assert not (outs[0][3] & 6).any() and not (outs[2][3] & 6).any()                   # re-run: no flag left
    for k, (a, b) in enumerate(zip(outs[0], outs[2])):
        d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
        assert np.array_equal(a[~redone], b[~redone]), k                           # worlds that stayed on the fast kernel: bit for bit
        assert d.max() < 5e-5, (k, float(d.max()))                                 # re-run worlds: the large-table launch is a separately compiled instantiation of the same source (fused-multiply-add
                                                                                   # contraction may differ): rounding level, against 1e-3 and more when contacts are dropped
    assert np.abs(outs[1][0][redone] - outs[2][0][redone]).max() > 1e-6            # and truncation is not harmless
    e = np.abs(outs[0][0] - g["obs"]).max(axis=1)
    assert np.mean(e < 1e-4) > 0.95                                                # against the oracle: the fixture's own bar (tests/golden/tolerance_table.json FetchHullContacts)
    for env in (small, trunc, big):
        env.close()
