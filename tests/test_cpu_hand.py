"""CPU tests of the HandReach family: host logic, the oracle's documented anchors, and the DEVICE ENGINE SOURCE run through the
sequential lane emulator against the oracle's golden fixtures (tendon-limit rows, friction-loss rows, capsule-capsule pairs)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "hand_HandReach_teacher.npz")


@pytest.fixture(scope="module")
def hand_model():
    from gymnasium_robotics_amd.envs.hand import load_hand_reach_model

    return load_hand_reach_model()


def test_hand_model_structure(hand_model):
    """SURVEY.md 8(a) cfg 3 robot: 24 hinges, 20 position actuators, 24 friction-loss dofs, limited fixed tendons, explicit condim-1 pairs."""
    m, T = hand_model, hand_model.tables
    assert (m.dim("nq"), m.dim("nv"), m.dim("nu"), m.dim("njnt")) == (24, 24, 20, 24)
    assert int((T["dof_frictionloss"] > 0).sum()) == 24 and np.allclose(T["dof_frictionloss"], 0.001)     # hand/shared.xml:13
    assert len(T["tendon_adr"]) == 44 and T["tendon_limited"].all()                                         # hand/shared.xml:53-198
    assert sorted(set(T["tendon_num"].tolist())) == [1, 2]
    cond1 = T["pair_condim"] == 1
    assert int(cond1.sum()) == 18   # 19 <pair> elements in hand/shared.xml:31-51, one of them listed twice
    t1, t2 = T["geom_type"][T["pair_geom1"][cond1]], T["geom_type"][T["pair_geom2"][cond1]]
    assert set(zip(t1.tolist(), t2.tolist())) == {(3, 3), (3, 6)}   # capsule-capsule, capsule-box(palm)
    assert np.allclose(T["jnt_margin"], 0.01) and np.allclose(T["act_biasprm"][:, 1], -T["act_gainprm"][:, 0])


def test_oracle_initial_fingertips_match_reference_documentation():
    """The reference documents the fingertip positions of the initial pose (shadow_dexterous_hand/reach.py:352-370, three
    significant digits): a golden vector for kinematics + site placement of the compiled model."""
    g = np.load(GOLDEN)
    doc = np.array([[0.99, 0.8, 0.15], [1.02, 0.8, 0.15], [1.04, 0.81, 0.155], [1.07, 0.82, 0.16], [0.95, 0.84, 0.16]])
    assert np.abs(g["initial_goal"].reshape(5, 3) - doc).max() < 6e-3


def test_goal_sampler_contract():
    from gymnasium_robotics_amd.envs.hand_spec import hand_reach_reward, parse_hand_reach_id, sample_hand_reach_goal

    g = np.load(GOLDEN)
    init, palm = g["initial_goal"], g["palm_xpos"]
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(0)))
    stay = moved = 0
    for _ in range(400):
        goal = sample_hand_reach_goal(rng, init, palm).reshape(5, 3)
        diff = np.abs(goal - init.reshape(5, 3)).max(axis=1) > 0
        if not diff.any():
            stay += 1
            continue
        moved += 1
        assert diff[4] and diff[:4].sum() == 1                     # the thumb and exactly one finger meet (reach.py:99-120)
        k = int(np.nonzero(diff[:4])[0][0])
        assert 0.0 < np.linalg.norm(goal[4] - goal[k]) <= 0.0100001  # each sits 5 mm off the common meeting point
        assert np.linalg.norm(goal[4] - (palm + [0.0, -0.09, 0.05])) < 0.05
    assert 15 <= stay <= 70   # 10 % branch (reach.py:122-125)
    assert parse_hand_reach_id("HandReach-v3") == "sparse" and parse_hand_reach_id("HandReachDense-v3") == "dense"
    with pytest.raises(ValueError):
        parse_hand_reach_id("HandReach-v9")
    a, b = np.zeros(15), np.full(15, 0.004)
    assert hand_reach_reward(a, b, "sparse") == -1.0 and hand_reach_reward(a, a + 1e-3, "sparse") == 0.0
    assert np.isclose(hand_reach_reward(a, b, "dense"), -np.linalg.norm(b))


def test_choice_equals_integers_stream():
    """The sampler replaces np_random.choice(finger_names) (reach.py:103) by an index draw: same value, same generator state after."""
    from gymnasium_robotics_amd.core import np_random

    names = ["robot0:S_fftip", "robot0:S_mftip", "robot0:S_rftip", "robot0:S_lftip"]
    for seed in range(300):
        a, b = np_random(seed)[0], np_random(seed)[0]
        assert a.choice(names) == names[b.integers(0, len(names))]
        assert a.normal(scale=0.005, size=3).tolist() == b.normal(scale=0.005, size=3).tolist() and a.uniform() == b.uniform()


def test_reset_goals_match_golden():
    """Seeded goal sampling reproduces the goals the oracle env drew (same host sampler, same numpy bit stream)."""
    from gymnasium_robotics_amd.core import np_random
    from gymnasium_robotics_amd.envs.hand_spec import sample_hand_reach_goal

    g = np.load(GOLDEN)
    for seed, goal in zip(g["reset_seed"], g["reset_goal"]):
        assert np.array_equal(sample_hand_reach_goal(np_random(int(seed))[0], g["initial_goal"], g["palm_xpos"]), goal)


def test_emulated_hand_step_matches_golden(hand_model):
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.hand_spec import make_hand_task

    g = np.load(GOLDEN)
    emu = EmuSim(hand_model, make_hand_task(hand_model))
    errs, tendon_steps, contact_steps = [], 0, 0
    for i in range(0, g["obs"].shape[0], 2):
        emu.load_world(g, i, ("qpos", "qvel", "qacc_ws"))
        emu.hand_step(g["action"][i])
        assert emu.status.value == 0
        err = np.abs(emu.hand_obs[:63] - g["obs"][i]).max()
        errs.append(err)
        tendon_steps += int(g["ntendon_rows"][i] > 0); contact_steps += int(g["ncon"][i] > 0)
        assert err < (1e-4 if g["activation_gap"][i] >= 1e-6 else 5e-3), (i, err)
        assert np.array_equal(emu.hand_achieved, emu.hand_obs[48:63])
    assert tendon_steps > 50 and contact_steps > 30   # the fixture exercises tendon-limit rows and the explicit contact pairs
    assert np.median(errs) < 2e-5


def test_emulated_hand_reset_forward_matches_golden(hand_model):
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.hand_spec import initial_qpos_vector, make_hand_task

    g = np.load(GOLDEN)
    emu = EmuSim(hand_model, make_hand_task(hand_model))
    emu.qpos[:] = initial_qpos_vector(hand_model)
    emu.qvel[:] = 0
    emu.qacc_ws[:] = 0
    emu.hand_step(np.zeros(20, np.float32), forward_only=True)
    assert np.abs(emu.hand_obs[:63] - g["reset_obs"][0]).max() < 2e-6
    assert np.abs(emu.palm - g["palm_xpos"]).max() < 1e-6


def test_batched_goal_sampler_equals_the_per_world_one():
    from gymnasium_robotics_amd.core import np_random
    from gymnasium_robotics_amd.envs.hand_spec import sample_hand_reach_goal, sample_hand_reach_goal_batch

    rng = np.random.default_rng(0)
    init, palm = rng.uniform(-0.05, 0.05, 15) + 1.0, np.array([1.0, 0.9, 0.15])
    a = [np_random(k)[0] for k in range(200)]
    b = [np_random(k)[0] for k in range(200)]
    one = np.stack([sample_hand_reach_goal(r, init, palm) for r in a])
    many = sample_hand_reach_goal_batch(b, init, palm)
    assert np.allclose(one, many, rtol=0, atol=1e-15)      # the squared norm is summed in a different order (dot vs explicit): last bit
    assert (np.abs(one - init).max(axis=1) == 0).sum() > 5            # the 10 % stay branch occurred
    assert all(x.uniform() == y.uniform() for x, y in zip(a, b))
