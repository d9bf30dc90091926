"""The reference's own BaseRobotEnv.step (robot_env.py:114-152) with MujocoFetchEnv's _set_action / _step_callback / _get_obs and the
mujoco_utils helpers (ctrl_set_action, mocap_set_action + reset_mocap2body_xpos, robot_get_obs, get_site_xpos / xvelp / xvelr / xmat,
set_joint_qpos) EXECUTED on top of the oracle's physics (tests/ref_harness.py), against the oracle environment's restated task layer
on an identical second simulation: observations, achieved goal, reward, success flag and the post-step state must be bit-identical.
This pins SURVEY.md section 8(a) rows P1, P3-P9 of the checker against the reference code itself; what stays unpinned is the
physics (P2).  Skips where /root/reference is not mounted."""
import numpy as np
import pytest

import ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="the reference tree is not mounted here")


@pytest.mark.parametrize("task,reward_type", [("FetchReach", "sparse"), ("FetchPush", "dense"), ("FetchSlide", "sparse"), ("FetchPickAndPlace", "sparse")])
def test_reference_step_on_oracle_physics_equals_the_oracle_env(fetch_models, task, reward_type):
    from oracle.fetch_oracle import OracleFetchEnv

    a_env = OracleFetchEnv(fetch_models[task], task, reward_type=reward_type)      # restated task layer
    b_env = OracleFetchEnv(fetch_models[task], task, reward_type=reward_type)      # physics host for the reference's task layer
    oa, _ = a_env.reset(seed=11)
    ob, _ = b_env.reset(seed=11)
    assert np.array_equal(oa["observation"], ob["observation"])
    ref = ref_harness.fetch_on_oracle(b_env, reward_type)
    # the reference's _get_obs on the freshly reset state
    o0 = ref._get_obs()
    assert np.array_equal(o0["observation"], oa["observation"]) and np.array_equal(o0["achieved_goal"], oa["achieved_goal"]) and np.array_equal(o0["desired_goal"], oa["desired_goal"])
    rng = np.random.default_rng(3)
    for t in range(12):
        act = rng.uniform(-1.3, 1.3, 4).astype(np.float32)                         # beyond [-1, 1]: step() clips
        if t == 5:
            act[:3] = 0
        oa, ra, ta, tra, ia = a_env.step(act)
        ob, rb, tb, trb, ib = ref.step(act)
        assert np.array_equal(ob["observation"], oa["observation"]), (t, np.abs(ob["observation"] - oa["observation"]).max())
        assert np.array_equal(ob["achieved_goal"], oa["achieved_goal"]) and np.array_equal(ob["desired_goal"], oa["desired_goal"])
        assert rb == ra and type(rb) is type(ra) and ib["is_success"] == ia["is_success"]
        assert tb is False and trb is False
        assert np.array_equal(a_env.sim.qpos, b_env.sim.qpos) and np.array_equal(a_env.sim.qvel, b_env.sim.qvel)
        assert np.array_equal(a_env.sim.ctrl, b_env.sim.ctrl) and np.array_equal(a_env.sim.mocap_pos, b_env.sim.mocap_pos) and np.array_equal(a_env.sim.mocap_quat, b_env.sim.mocap_quat)
    with pytest.raises(ValueError, match="Action dimension mismatch"):
        ref.step(np.zeros(3, np.float32))


def _same_step(a_out, b_out, a_sim, b_sim, t):
    (oa, ra, _, _, ia), (ob, rb, tb, trb, ib) = a_out, b_out
    for k in ("observation", "achieved_goal", "desired_goal"):
        assert np.array_equal(ob[k], oa[k]), (t, k, np.abs(ob[k] - oa[k]).max())
    # dense manipulation reward: -(10 d_pos + d_rot) with d_rot = 2 acos(w) of a quaternion product -- the restated product associates
    # differently and may differ in the last bit
    assert abs(float(rb) - float(ra)) <= 1e-12 and float(ib["is_success"]) == float(ia["is_success"]) and tb is False and trb is False
    assert np.array_equal(a_sim.qpos, b_sim.qpos) and np.array_equal(a_sim.qvel, b_sim.qvel) and np.array_equal(a_sim.ctrl, b_sim.ctrl)


@pytest.mark.parametrize("reward_type", ["sparse", "dense"])
def test_reference_hand_reach_step_on_oracle_physics(reward_type):
    """hand_env.py:42-61 (_set_action, absolute control), reach.py:92-97,398-428 (_get_obs, reward, success) executed as is."""
    from gymnasium_robotics_amd.envs.hand import load_hand_reach_model
    from oracle.hand_oracle import OracleHandReachEnv

    model = load_hand_reach_model()
    a_env, b_env = OracleHandReachEnv(model, reward_type), OracleHandReachEnv(model, reward_type)
    a_env.reset(seed=4); b_env.reset(seed=4)
    ref = ref_harness.hand_on_oracle(b_env, "reach", reward_type=reward_type, distance_threshold=0.01)
    o0 = ref._get_obs()
    assert np.array_equal(o0["observation"], a_env._obs()["observation"])
    rng = np.random.default_rng(1)
    for t in range(8):
        act = rng.uniform(-1.2, 1.2, 20).astype(np.float32)
        _same_step(a_env.step(act), ref.step(act), a_env.sim, b_env.sim, t)


@pytest.mark.parametrize("env_id", ["HandManipulateBlockRotateXYZ-v1", "HandManipulateBlockFullDense-v1", "HandManipulateEggRotate-v1", "HandManipulatePenRotateDense-v1"])
def test_reference_manipulate_step_on_oracle_physics(env_id):
    """manipulate.py:87-142,298-316 (_goal_distance incl. the pen's ignore-z path, compute_reward, _is_success, _get_obs) executed as is."""
    from gymnasium_robotics_amd.envs.hand import load_hand_block_model
    from gymnasium_robotics_amd.envs.manipulate_spec import OBJECTS, object_of, parse_block_id
    from oracle.manipulate_oracle import OracleHandBlockEnv

    tp, tr, rt, _ = parse_block_id(env_id)
    obj = object_of(env_id)
    model = load_hand_block_model(obj=obj)
    a_env, b_env = OracleHandBlockEnv(model, tp, tr, rt, "off", obj), OracleHandBlockEnv(model, tp, tr, rt, "off", obj)
    a_env.reset(seed=2); b_env.reset(seed=2)
    ref = ref_harness.hand_on_oracle(b_env, "manipulate", reward_type=rt, target_position=tp, target_rotation=tr, rotation_threshold=0.1,
                                     distance_threshold=OBJECTS[obj]["distance_threshold"], ignore_z_target_rotation=OBJECTS[obj]["ignore_z_target_rotation"])
    assert np.array_equal(ref._get_obs()["observation"], a_env._obs()["observation"])
    rng = np.random.default_rng(5)
    for t in range(6):
        act = (0.4 * rng.uniform(-1, 1, 20)).astype(np.float32)
        _same_step(a_env.step(act), ref.step(act), a_env.sim, b_env.sim, t)
