"""CPU tests of the host logic: C ABI surface, env-id registry, reset RNG draw order, golden fixtures sanity."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_c_abi_exports_every_declared_symbol():
    from gymnasium_robotics_amd import _native

    L = _native.lib()  # dlopen works without a GPU
    header = open(os.path.join(ROOT, "include", "grx_capi.h")).read()
    declared = set(re.findall(r"\b(grx_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), f"{name} not exported by libgrx_hip.so"


def test_model_create_fails_loudly_without_gpu(fetch_models):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gymnasium_robotics_amd import _native
    from gymnasium_robotics_amd.envs.fetch import FetchVecEnv

    H, I, F = fetch_models["FetchReach"].pack()
    h = ctypes.c_void_p()
    rc = _native.lib().grx_model_create(H.ctypes.data, H.size, I.ctypes.data, I.size, F.ctypes.data, F.size, 0, ctypes.byref(h))
    assert rc != 0 and _native.lib().grx_last_error()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FetchVecEnv("FetchReach-v4", num_envs=2)


def test_env_id_registry():
    from gymnasium_robotics_amd.envs.fetch_spec import parse_env_id

    assert parse_env_id("FetchPickAndPlace-v4") == ("FetchPickAndPlace", "sparse")
    assert parse_env_id("FetchReachDense-v4") == ("FetchReach", "dense")
    with pytest.raises(KeyError):
        parse_env_id("FetchFly-v4")


def test_task_struct_mirrors_c_layout(fetch_models):
    from gymnasium_robotics_amd.envs.fetch_spec import FetchTaskStruct, make_fetch_task

    assert ctypes.sizeof(FetchTaskStruct) == 4 * (4 + 1 + 3 + 4 + 2 + 4 + 2 + 1 + 1) + 8      # ... dt, padding, the fp64 distance threshold
    t = make_fetch_task(fetch_models["FetchPickAndPlace"], "FetchPickAndPlace", "sparse")
    assert (t.obs_dim, t.has_object, t.block_gripper, t.n_substeps) == (25, 1, 0, 20)
    assert abs(t.dt - 0.04) < 1e-9  # 25 Hz control (robot_env.py:83-85)
    t = make_fetch_task(fetch_models["FetchReach"], "FetchReach", "dense")
    assert (t.obs_dim, t.has_object, t.block_gripper, t.sparse_reward) == (10, 0, 1, 0)


@pytest.mark.parametrize("task", ["FetchReach", "FetchPush", "FetchPickAndPlace"])
def test_reset_draw_order_matches_oracle_env(fetch_models, task):
    """The host-side sampler consumes the PCG64 stream exactly like the reference's _reset_sim + _sample_goal."""
    from gymnasium_robotics_amd.core import np_random
    from gymnasium_robotics_amd.envs.fetch import sample_fetch_reset
    from gymnasium_robotics_amd.envs.fetch_spec import FETCH_TASKS
    from oracle.fetch_oracle import OracleFetchEnv

    env = OracleFetchEnv(fetch_models[task], task)
    for seed in (0, 1, 42):
        obs, _ = env.reset(seed=seed)
        rng, _ = np_random(seed)
        oxy, goal = sample_fetch_reset(FETCH_TASKS[task], rng, env.initial_gripper_xpos, getattr(env, "height_offset", 0.0))
        assert np.array_equal(goal, obs["desired_goal"])
        if oxy is not None:
            assert np.array_equal(oxy, env.sim.qpos[env.jq["object0:joint"]: env.jq["object0:joint"] + 2])
        # second episode continues the same stream (no reseed)
        obs2, _ = env.reset()
        _, goal2 = sample_fetch_reset(FETCH_TASKS[task], rng, env.initial_gripper_xpos, getattr(env, "height_offset", 0.0))
        assert np.array_equal(goal2, obs2["desired_goal"])


def test_golden_fixtures_are_reproducible(fetch_models):
    """tests/golden/*.npz were produced by tools/make_golden.py from the oracle: re-derive a few rows."""
    from oracle.fetch_oracle import OracleFetchEnv

    g = np.load(os.path.join(ROOT, "tests", "golden", "fetch_FetchPickAndPlace_teacher.npz"))
    env = OracleFetchEnv(fetch_models["FetchPickAndPlace"], "FetchPickAndPlace")
    s = env.sim
    for i in (0, 57, 399):
        s.qpos[:], s.qvel[:], s.qacc_warmstart[:] = g["qpos"][i], g["qvel"][i], g["qacc_ws"][i]
        a = np.clip(g["action"][i].astype(np.float64), -1, 1)
        s.ctrl[:] = s.qpos[[env.jq["robot0:l_gripper_finger_joint"], env.jq["robot0:r_gripper_finger_joint"]]] + a[3]
        s.mocap_pos[:] = g["aux"][i][:3] + 0.05 * a[:3]
        s.mocap_quat[:] = g["aux"][i][3:7] + np.array([1.0, 0, 1, 0])
        s.step(20)
        assert np.allclose(s.qpos, g["qpos_next"][i], atol=1e-12)


def test_spaces_and_goal_contract():
    from gymnasium_robotics_amd.spaces import Box, Dict, batch_space

    single = Dict(dict(observation=Box(-np.inf, np.inf, (25,), np.float64), achieved_goal=Box(-np.inf, np.inf, (3,), np.float64),
                       desired_goal=Box(-np.inf, np.inf, (3,), np.float64)))
    b = batch_space(single, 7)
    assert b["observation"].shape == (7, 25) and b["achieved_goal"].dtype == np.float64
    a = Box(-1.0, 1.0, (4,), np.float32)
    a.seed(0)
    x = a.sample()
    assert a.contains(x) and x.dtype == np.float32


def test_native_pcg64_reset_sampler_is_bit_exact_with_numpy():
    """grx_fetch_sample_resets advances numpy PCG64 states in C; draws must equal Generator.uniform bit for bit."""
    from gymnasium_robotics_amd import _native
    from gymnasium_robotics_amd.core import np_random
    from gymnasium_robotics_amd.envs.fetch import sample_fetch_reset
    from gymnasium_robotics_amd.envs.fetch_spec import FETCH_TASKS

    L = _native.lib()
    g0 = np.array([1.3419, 0.7491, 0.5347])
    mask = (1 << 64) - 1
    for task in ("FetchReach", "FetchPush", "FetchSlide", "FetchPickAndPlace"):
        cfg = FETCH_TASKS[task]
        seeds = [0, 1, 42, 123456789, 2**31]
        rngs = [np_random(s)[0] for s in seeds]
        st = np.zeros((len(seeds), 4), np.uint64)
        for i, s in enumerate(seeds):
            b = np_random(s)[0].bit_generator.state["state"]
            st[i] = [b["state"] >> 64, b["state"] & mask, b["inc"] >> 64, b["inc"] & mask]
        toff = np.ascontiguousarray(np.broadcast_to(np.asarray(cfg["target_offset"], dtype=np.float64), (3,)))
        for episode in range(4):  # several consecutive resets continue the same streams
            idx = np.arange(len(seeds), dtype=np.int64)
            oxy = np.zeros((len(seeds), 2)); goal = np.zeros((len(seeds), 3))
            rc = L.grx_fetch_sample_resets(st.ctypes.data, idx.ctypes.data, len(seeds), int(cfg["has_object"]), int(cfg["target_in_the_air"]),
                                           float(cfg["obj_range"]), float(cfg["target_range"]), toff.ctypes.data, g0.ctypes.data, 0.4249,
                                           oxy.ctypes.data, goal.ctypes.data)
            assert rc == 0
            for i, rng in enumerate(rngs):
                o_ref, g_ref = sample_fetch_reset(cfg, rng, g0, 0.4249)
                assert np.array_equal(goal[i], g_ref)
                if o_ref is not None:
                    assert np.array_equal(oxy[i], o_ref)
