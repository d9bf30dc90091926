"""GPU parity tests for AdroitHandHammer / Door / Pen / Relocate-v2 (pytest -m gpu): everything goes through the C ABI (grx_adroit_step) via AdroitVecEnv;
the oracle / golden fixtures are only the checker."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _env(n, env_id="AdroitHandHammer-v2", **kw):
    import torch

    from gymnasium_robotics_amd import make_vec

    assert torch.cuda.is_available(), "these tests need the GPU"
    return make_vec(env_id, num_envs=n, device="cuda:0", **kw)


ENV_ID = {"hammer": "AdroitHandHammer-v2", "door": "AdroitHandDoor-v2", "pen": "AdroitHandPen-v2", "relocate": "AdroitHandRelocate-v2"}


@pytest.mark.parametrize("task", ["hammer", "door", "pen", "relocate"])
def test_teacher_forced_step_matches_golden(task):
    """All 420 fixtures of the task in one launch; same per-component bounds as the emulator test (tests/adroit_cases.py) -- plus the specialised
    kernel must be the one that ran (grx_model_dim "shape" = 20..23)."""
    import torch

    from adroit_cases import check

    g = np.load(os.path.join(GOLDEN, f"adroit_{task}_teacher.npz"))
    n = g["obs"].shape[0]
    env = _env(n, env_id=ENV_ID[task], autoreset_mode="disabled", max_episode_steps=None)
    assert env._L.grx_model_dim(env._h, b"shape") == {"hammer": 20, "door": 21, "pen": 22, "relocate": 23}[task]
    env.reset(seed=0)
    dev = env.device
    f32 = lambda k: torch.from_numpy(g[k].astype(np.float32)).to(dev)
    env.qpos.copy_(f32("qpos")); env.qvel.copy_(f32("qvel")); env.qacc_ws.copy_(f32("qacc_ws")); env.shift.copy_(f32("shift"))
    if env.target is not None:
        env.target.copy_(f32("target"))
    obs, r, term, trunc, info = env.step(g["action"])
    assert int(info["status"].max()) == 0 and not term.any() and not trunc.any()
    print(check(task, obs, g["obs"], r, g["reward"]))
    assert np.array_equal(info["success"], g["success"].astype(bool))


@pytest.mark.parametrize("task", ["hammer", "door", "pen", "relocate"])
def test_reset_matches_golden_and_reference_draws(task):
    """reset(seed=s): world i gets the model edits the reference's reset_model draws for seed s + i (board height / door frame position / target
    quaternion / ball offset + target site) and the observation of mj_forward at init_qpos with them."""
    g = np.load(os.path.join(GOLDEN, f"adroit_{task}_teacher.npz"))
    n = len(g["reset_seed"])
    env = _env(n, env_id=ENV_ID[task])
    obs, info = env.reset(seed=int(g["reset_seed"][0]))
    assert obs.shape == (n, g["reset_obs"].shape[1]) and obs.dtype == np.float64
    assert np.abs(env.model_edit - g["reset_edit"]).max() == 0.0
    if task == "relocate":
        assert np.abs(env.target_pos - g["reset_target"]).max() == 0.0
    assert np.abs(obs - g["reset_obs"]).max() < 1e-5
    # hammer / door / relocate draw ON THE DEVICE (grx_adroit_sample_resets_device): over further ragged resets the fp64 edit rows, the fp32 shift poses and the stream
    # positions stay bit-equal to the host routine fed by numpy generators (adroit_spec.sample_reset_batch = the reference's draw order)
    if task != "pen":
        import torch

        from gymnasium_robotics_amd.core import np_random
        from gymnasium_robotics_amd.envs.adroit_spec import sample_reset_batch

        assert env._device_draws
        seeds = [int(g["reset_seed"][0]) + i for i in range(n)]
        rngs = [np_random(sd)[0] for sd in seeds]
        edit = np.tile(np.asarray(env.model.info["shift_pos0"], dtype=np.float64), (n, 1))
        shift, target = np.zeros((n, 7), np.float32), np.zeros((n, 3), np.float32)
        for idx in (np.arange(n), np.arange(0, n, 3), np.array([n - 1, 0]), np.arange(n)[::-1].copy()):
            if len(idx) < n or not np.array_equal(idx, np.arange(n)) or shift.any():
                torch.cuda.set_sync_debug_mode("error")      # the draws are enqueued, nothing is read back
                try:
                    env._reset_worlds(idx)
                finally:
                    torch.cuda.set_sync_debug_mode("default")
            d = sample_reset_batch(task, [rngs[w] for w in idx], env.model, current=edit[idx] if task in ("hammer", "relocate") else None)
            edit[idx] = d["edit"]; shift[idx] = d["shift"].astype(np.float32)
            if task == "relocate":
                target[idx] = d["target"].astype(np.float32)
            assert np.array_equal(env.model_edit, edit) and np.array_equal(env.shift.cpu().numpy(), shift)
            if task == "relocate":
                assert np.array_equal(env.target.cpu().numpy(), target) and np.array_equal(env.target_pos[idx], d["target"])
        st = env._rng_dev.cpu().numpy().view(np.uint64)
        for w in (0, n - 1):
            s = rngs[w].bit_generator.state["state"]["state"]
            assert (int(st[w, 0]) << 64 | int(st[w, 1])) == s


@pytest.mark.parametrize("task", ["door", "pen", "relocate"])
def test_state_round_trip_and_rollout(task):
    """get_env_state / set_env_state as the reference's tests use them (tests/envs/adroit_hand/test_adroit_*.py: set the state, read it back), and a
    200-step random rollout with autoreset that stays finite and flag-free."""
    env = _env(16, env_id=ENV_ID[task], max_episode_steps=50, autoreset_mode="same_step")
    obs, _ = env.reset(seed=11)
    rng = np.random.default_rng(2)
    nu = env.single_action_space.shape[0]
    for t in range(120):
        obs, r, term, trunc, info = env.step(rng.uniform(-1, 1, (16, nu)).astype(np.float32))
        assert np.isfinite(obs).all() and np.isfinite(r).all()
    assert int(info["status"].max()) & 0xFFFF == 0
    st = env.get_env_state()
    keys = {"door": ("qpos", "qvel", "door_body_pos"), "pen": ("qpos", "qvel", "desired_orien"), "relocate": ("qpos", "qvel", "obj_pos", "target_pos")}[task]
    edit = env.model_edit.copy()
    env.set_env_state({k: st[k] for k in keys})
    st2 = env.get_env_state()
    assert np.allclose(st2["qpos"], st["qpos"]) and np.allclose(st2["qvel"], st["qvel"]) and np.allclose(env.model_edit, edit, atol=1e-5)


def test_api_contract_and_autoreset():
    env = _env(8, max_episode_steps=5, autoreset_mode="same_step")
    obs, _ = env.reset(seed=3)
    rng = np.random.default_rng(0)
    with pytest.raises(ValueError, match="Action dimension mismatch"):
        env.step(np.zeros((8, 25), np.float32))
    z0 = env.board_z.copy()
    for t in range(5):
        obs, r, term, trunc, info = env.step(rng.uniform(-1, 1, (8, 26)).astype(np.float32))
        assert r.shape == (8,) and r.dtype == np.float64 and info["success"].dtype == bool and not term.any()
    assert trunc.all() and "final_obs" in info and info["final_obs"].shape == (8, 46)
    assert not np.array_equal(env.board_z, z0)                      # the autoreset drew new board heights
    assert np.abs(obs[:, :27]).max() < 1e-6                          # and the returned observation is the reset one (qpos = init_qpos = 0)
    st = env.get_env_state()
    assert st["qpos"].shape == (8, 33) and st["board_pos"].shape == (8, 3) and np.allclose(st["board_pos"][:, 2], env.board_z)
    st["qpos"][:, 0] = 0.1
    env.set_env_state(st)
    assert np.allclose(env.qpos[:, 0].cpu().numpy(), 0.1)


def test_worlds_are_independent_and_deterministic():
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, (12, 32, 26)).astype(np.float32)
    outs = []
    for _ in range(2):
        env = _env(32)
        env.reset(seed=0)
        traj = [env.step(a)[0] for a in acts]
        outs.append(np.stack(traj))
    assert np.array_equal(outs[0], outs[1])
    env1 = _env(1)
    env1.reset(seed=5)       # world 5 of the batch = a single env seeded 5
    solo = np.stack([env1.step(a[5:6])[0][0] for a in acts])
    assert np.array_equal(solo, outs[0][:, 5])


def test_overflow_lane_polling_equals_the_serialised_rerun(monkeypatch):
    """The polling workgroups of the standing lane launch (grx_overflow_lane.ready / progress / poll_*) re-run an entrant while the fast launch is still running; the
    serialised launch behind the fast kernel takes whatever they did not claim.  Who re-runs a world must not matter: a rollout with 16 polling workgroups, one with a single
    polling workgroup (so that the entry launch gets work too) and one without polling are bit-identical, world by world, and entrants did occur."""
    import torch

    from gymnasium_robotics_amd import make_vec

    n, envs = 4096, []
    for poll in ("16", "1", "0"):
        monkeypatch.setenv("GRX_LANE_POLL", poll)
        e = make_vec("AdroitHandDoor-v2", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
        assert e.lane is not None and e.lane.mode == "lane" and e.lane.poll_grid == int(poll)
        e.reset(seed=3)
        envs.append(e)
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    entered = 0
    for t in range(14):
        a = torch.rand(n, envs[0].single_action_space.shape[0], device="cuda:0", generator=g) * 2 - 1
        outs = [e.step(a) for e in envs]
        entered += len(envs[2].lane.entered_last_step())
        for e, o in zip(envs[:2], outs[:2]):
            assert torch.equal(e.qpos, envs[2].qpos) and torch.equal(e.qvel, envs[2].qvel) and torch.equal(o[0], outs[2][0]) and torch.equal(o[1], outs[2][1]), t
            assert torch.equal(e.status & 0xFFFF, envs[2].status & 0xFFFF)
    assert entered >= 3, entered      # the rollout does push worlds over the fast kernel's tables


@pytest.mark.parametrize("task", ["hammer", "door", "pen", "relocate"])
def test_compacted_reset_launch_is_the_masked_one(monkeypatch, task):
    """The reset-time forward pass as one workgroup per reset world (grx_adroit_buffers.compact, the default) against the masked launch over all N worlds
    (GRX_ADROIT_COMPACT_RESET=0): staggered same-step autoresets, state rows, observations, rewards and status words bit-identical after every step."""
    import torch

    n, horizon = 256, 20
    envs = []
    for on in ("1", "0"):
        monkeypatch.setenv("GRX_ADROIT_COMPACT_RESET", on)
        e = _env(n, ENV_ID[task], output="torch", autoreset_mode="same_step", max_episode_steps=horizon)
        e.reset(seed=7)
        e._elapsed[:] = (np.arange(n) * 3) % 17      # phases 17 .. 19 are empty: some steps reset no world
        envs.append(e)
    assert envs[0]._compact_resets and not envs[1]._compact_resets
    assert torch.equal(envs[0].obs, envs[1].obs) and float(envs[0].obs.abs().max()) > 0.1      # reset(): the compacted launch covers every world
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    for t in range(50):
        a = torch.rand(n, envs[0].single_action_space.shape[0], device="cuda:0", generator=g) * 2 - 1
        (o0, r0, _, tr0, _), (o1, r1, _, tr1, _) = [e.step(a) for e in envs]
        assert torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(tr0, tr1), t
        for name in ("qpos", "qvel", "qacc_ws", "shift", "status"):
            assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)), (t, name)


@pytest.mark.parametrize("task,output", [("hammer", "torch"), ("hammer", "numpy"), ("door", "torch"), ("relocate", "torch"), ("pen", "torch")])
def test_overlapped_reset_is_the_inline_reset(monkeypatch, task, output):
    """Same-step autoreset of the device-draw tasks, two ways: draws + forward pass of the worlds a step will truncate on a side stream beside the step kernel, committed
    behind it (grx_adroit_commit_rows, the default), against the in-line reset (GRX_ADROIT_AHEAD_RESET=0).  Staggered episodes, 70 steps = three or four resets per world:
    state rows, per-world model edits (shift / target rows, the fp64 edit rows), outputs, status words, PCG64 streams and info["final_obs"] bit-identical after every step."""
    import torch

    n, horizon = 256, 20
    envs = []
    for on in ("1", "0"):
        monkeypatch.setenv("GRX_ADROIT_AHEAD_RESET", on)
        e = _env(n, ENV_ID[task], output=output, autoreset_mode="same_step", max_episode_steps=horizon)
        e.reset(seed=7)
        e._elapsed[:] = (np.arange(n) * 3) % 17
        envs.append(e)
    assert envs[0]._ahead is not None and envs[1]._ahead is None
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    eq = torch.equal if output == "torch" else np.array_equal
    resets = 0
    for t in range(70):
        a = torch.rand(n, envs[0].single_action_space.shape[0], device="cuda:0", generator=g) * 2 - 1
        (o0, r0, _, tr0, i0), (o1, r1, _, tr1, i1) = [e.step(a if output == "torch" else a.cpu().numpy()) for e in envs]
        assert eq(o0, o1) and eq(r0, r1) and eq(tr0, tr1) and eq(i0["success"], i1["success"]), t
        assert ("final_obs" in i0) == ("final_obs" in i1)
        if "final_obs" in i0:
            resets += 1
            assert eq(i0["final_obs"], i1["final_obs"]), t
        for name in ("qpos", "qvel", "qacc_ws", "shift", "obs", "reward", "success", "status") + (("_rng_dev", "_edit_dev") if task != "pen" else ()) + (("target", "_target_dev") if task == "relocate" else ()):
            assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)), (t, name)
        assert np.array_equal(envs[0].model_edit, envs[1].model_edit), t
        if task == "pen":
            assert all(a_.bit_generator.state == b_.bit_generator.state for a_, b_ in zip(envs[0].np_randoms[::37], envs[1].np_randoms[::37])), t
    assert 50 <= resets < 70


@pytest.mark.parametrize("env_id,parts", [("AdroitHandHammer-v2", 5), ("AdroitHandPen-v2", 2), ("AdroitHandRelocate-v2", 5), ("AdroitHandDoor-v2", 3)])
def test_split_step_is_the_plain_step(monkeypatch, env_id, parts):
    """Round 6: the Adroit step launch with P workgroups per world, each running its share of the 5 substeps and handing the world on through a carrier row (include/grx_capi.h
    grx_adroit_buffers.split_parts), against the plain launch: state rows, observations, rewards, success flags and status words are BIT-IDENTICAL after every step -- same-step autoresets at a
    short time limit (compact reset launches, the overlapped reset), the cost-ordered dispatch on, the standing overflow lane of door / relocate with its polling workgroups counting every part.
    The reference's step is one env.step() whatever the launch geometry (/root/reference/gymnasium_robotics/envs/adroit_hand/adroit_hammer.py:291-357)."""
    import torch

    import gymnasium_robotics_amd as grx

    n, envs = 2048, []
    for p_ in (1, parts):
        monkeypatch.setenv("GRX_ADROIT_SPLIT", str(p_))
        e = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", max_episode_steps=9)
        e.reset(seed=5)
        envs.append(e)
    plain, split = envs
    assert plain._split == 1 and split._split == parts
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(9)
    act_dim = plain.single_action_space.shape[0]
    for t in range(24):
        a = torch.rand(n, act_dim, device="cuda:0", generator=gen) * 2 - 1
        outs = [e.step(a) for e in envs]
        for name in ("qpos", "qvel", "qacc_ws", "obs", "reward", "success", "status"):
            assert torch.equal(getattr(split, name), getattr(plain, name)), (t, name, int((getattr(split, name) != getattr(plain, name)).sum()))
        assert torch.equal(torch.as_tensor(outs[0][3]), torch.as_tensor(outs[1][3]))
        assert int(split._split_state.abs().max()) == 0, t      # every world's words are clean again
    assert int((split.status & 1).max()) == 0 and torch.isfinite(split.qpos).all()
    assert split.cost is None or (int(split.cost.min()) > 0 and int(split.cost.max()) < 10_000_000)      # (the pen runs without the cost order)
