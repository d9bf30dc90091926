"""GPU parity tests for AdroitHandHammer-v2 (pytest -m gpu): everything goes through the C ABI (grx_adroit_step) via AdroitHammerVecEnv;
the oracle / golden fixtures are only the checker."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _env(n, **kw):
    import torch

    from gymnasium_robotics_amd import make_vec

    assert torch.cuda.is_available(), "these tests need the GPU"
    return make_vec("AdroitHandHammer-v2", num_envs=n, device="cuda:0", **kw)


def test_teacher_forced_step_matches_golden():
    """Same fixtures, same per-component bounds as the emulator test (tests/test_cpu_adroit.py) -- plus the specialised kernel must be the one
    that ran (grx_model_dim "shape" = 20)."""
    import torch

    g = np.load(os.path.join(GOLDEN, "adroit_hammer_teacher.npz"))
    n = g["obs"].shape[0]
    env = _env(n, autoreset_mode="disabled", max_episode_steps=None)
    assert env._L.grx_model_dim(env._h, b"shape") == 20
    env.reset(seed=0)
    dev = env.device
    env.qpos.copy_(torch.from_numpy(g["qpos"].astype(np.float32)).to(dev)); env.qvel.copy_(torch.from_numpy(g["qvel"].astype(np.float32)).to(dev))
    env.qacc_ws.copy_(torch.from_numpy(g["qacc_ws"].astype(np.float32)).to(dev))
    env.shift[:, 2] = torch.from_numpy((g["board_z"] - env._board_z0).astype(np.float32)).to(dev)
    obs, r, term, trunc, info = env.step(g["action"])
    assert int(info["status"].max()) == 0 and not term.any() and not trunc.any()
    e = np.abs(obs - g["obs"])
    e_q, e_vel, e_rot = e[:, :27].max(axis=1), e[:, 27:33].max(axis=1), e[:, 39:42].max(axis=1)
    e_pos = np.maximum(e[:, 33:39].max(axis=1), e[:, 42:45].max(axis=1))
    print(f"qpos max {e_q.max():.2e}; positions max {e_pos.max():.2e}; hammer velocity p50 {np.median(e_vel):.2e} p90 {np.quantile(e_vel, 0.9):.2e} max {e_vel.max():.2e}; "
          f"hammer euler p50 {np.median(e_rot):.2e} max {e_rot.max():.2e}; reward p50 {np.median(np.abs(r - g['reward'])):.2e}")
    assert e_q.max() < 1e-4 and e_pos.max() < 2e-4
    assert np.median(e_vel) < 5e-3 and np.quantile(e_vel, 0.9) < 6e-2 and e_vel.max() < 0.5
    assert np.median(e_rot) < 1e-4 and e_rot.max() < 5e-3
    assert e[:, 45].max() < 1e-3
    assert np.median(np.abs(r - g["reward"])) < 1e-4 and np.abs(r - g["reward"]).max() < 5e-3
    assert np.array_equal(info["success"], g["success"].astype(bool))


def test_reset_matches_golden_and_reference_draws():
    """reset(seed=s): world i gets the board height the reference draws for seed s + i (np_random.uniform(0.1, 0.25), adroit_hammer.py:374) and the
    observation of mj_forward at init_qpos with that board."""
    g = np.load(os.path.join(GOLDEN, "adroit_hammer_teacher.npz"))
    n = len(g["reset_seed"])
    env = _env(n)
    obs, info = env.reset(seed=int(g["reset_seed"][0]))
    assert obs.shape == (n, 46) and obs.dtype == np.float64
    assert np.abs(env.board_z - g["reset_board_z"]).max() == 0.0
    assert np.abs(obs - g["reset_obs"]).max() < 1e-5


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
