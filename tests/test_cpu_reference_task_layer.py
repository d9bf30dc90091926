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
