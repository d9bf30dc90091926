"""Host-side task logic against vectors produced by EXECUTING the reference's own code (tools/make_reference_host_vectors.py imports
fetch_env.py / reach.py / manipulate.py from /root/reference under stand-ins for the absent gymnasium / mujoco modules and calls the
reference methods on plain namespaces): same np_random seeds -> the same draws in the same order, the same arithmetic."""
import os

import numpy as np
import pytest

REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_host_logic.npz"))


def _rng(seed):
    from gymnasium_robotics_amd.core import np_random

    return np_random(seed)[0]


@pytest.mark.parametrize("key,task,height", [("reach", "FetchReach", 0.0), ("push", "FetchPush", 0.42473), ("slide", "FetchSlide", 0.41401),
                                              ("pick", "FetchPickAndPlace", 0.42473)])
def test_fetch_reset_and_goal_sampling_is_the_reference_draw_for_draw(key, task, height):
    """fetch_env.py:375-402 (_reset_sim: object xy rejection loop) then :153-166 (_sample_goal) on one np_random stream."""
    from gymnasium_robotics_amd.envs.fetch import sample_fetch_reset
    from gymnasium_robotics_amd.envs.fetch_spec import FETCH_TASKS

    g0 = REF["fetch_gripper_xpos"]
    for seed in range(64):
        oxy, goal = sample_fetch_reset(FETCH_TASKS[task], _rng(seed), g0, height)
        assert np.array_equal(goal, REF[f"fetch_{key}_goal"][seed]), (seed, goal, REF[f"fetch_{key}_goal"][seed])
        if FETCH_TASKS[task]["has_object"]:
            assert np.array_equal(oxy, REF[f"fetch_{key}_object_xy"][seed])
        else:
            assert oxy is None


def test_fetch_reward_and_success_of_the_checker_equal_the_reference():
    """fetch_env.py:16-18,74-80,168-170 on (5, 33, 3) batches incl. distances at the threshold (strict comparisons)."""
    from oracle.fetch_oracle import goal_distance

    ag, dg = REF["fetch_ag"], REF["fetch_dg"]
    d = goal_distance(ag, dg)
    assert np.array_equal(-(d > 0.05).astype(np.float32), REF["fetch_reward_sparse"]) and REF["fetch_reward_sparse"].dtype == np.float32
    assert np.array_equal(-d, REF["fetch_reward_dense"])
    assert np.array_equal((d < 0.05).astype(np.float32), REF["fetch_success"])


def test_hand_reach_goal_sampling_and_reward():
    """reach.py:98-134 (_sample_goal: finger choice, meeting point, noise, 10 % keep-initial branch) and :92-97."""
    from gymnasium_robotics_amd.envs.hand_spec import hand_reach_reward, sample_hand_reach_goal

    init, palm = REF["hand_reach_initial_goal"], REF["hand_reach_palm"]
    for seed in range(64):
        g = sample_hand_reach_goal(_rng(seed), init, palm)
        assert np.array_equal(g, REF["hand_reach_goal"][seed]), seed
    ag, dg = REF["hand_reach_ag"], REF["hand_reach_dg"]
    assert np.array_equal(hand_reach_reward(ag, dg, "sparse"), REF["hand_reach_reward_sparse"])
    assert np.array_equal(hand_reach_reward(ag, dg, "dense"), REF["hand_reach_reward_dense"])
    assert np.array_equal((np.linalg.norm(ag - dg, axis=-1) < 0.01).astype(np.float32), REF["hand_reach_success"])


@pytest.mark.parametrize("tp,tr", [("ignore", "z"), ("ignore", "parallel"), ("ignore", "xyz"), ("random", "xyz")])
def test_manipulate_reset_pose_and_goal_sampling(tp, tr):
    """manipulate.py:154-224 (_reset_sim pose randomisation) then :226-279 (_sample_goal), one stream, registered variants."""
    from gymnasium_robotics_amd.envs import manipulate_spec as ms

    pq = ms.canonical_parallel_quats()
    obj0 = np.array([1.0, 0.87, 0.2, 1.0, 0.0, 0.0, 0.0])
    for seed in range(48):
        rng = _rng(seed)
        pose = ms.sample_reset_object_pose(rng, obj0[:3], obj0[3:], tp, tr, pq)
        assert np.allclose(pose, REF[f"manip_{tp}_{tr}_reset_pose"][seed], rtol=0, atol=1e-15), (seed, pose - REF[f"manip_{tp}_{tr}_reset_pose"][seed])
        goal = ms.sample_block_goal(rng, obj0, tp, tr, pq)          # the recording double answered the goal query with the un-randomised pose
        assert np.allclose(goal, REF[f"manip_{tp}_{tr}_goal"][seed], rtol=0, atol=1e-15), seed


@pytest.mark.parametrize("tp,tr", [("ignore", "xyz"), ("random", "xyz"), ("random", "ignore")])
def test_manipulate_distance_reward_success(tp, tr):
    """manipulate.py:87-142 on a batch with identical, antipodal and near-threshold pairs."""
    from gymnasium_robotics_amd.envs import manipulate_spec as ms

    ga, gb = REF["manip_ga"], REF["manip_gb"]
    dp, dr = ms.block_goal_distance(ga, gb, tp, tr)
    assert np.allclose(dp, REF[f"manip_{tp}_{tr}_dpos"], rtol=0, atol=1e-15)
    # 2 acos(w) near w = 1 amplifies rounding: identical orientations give 0 +- 3e-8 in either implementation
    assert np.allclose(dr, REF[f"manip_{tp}_{tr}_drot"], rtol=0, atol=1e-7)
    clear = (np.abs(REF[f"manip_{tp}_{tr}_dpos"] - 0.01) > 1e-9) & (np.abs(REF[f"manip_{tp}_{tr}_drot"] - 0.1) > 1e-6)
    suc = ms.block_is_success(ga, gb, tp, tr)
    assert np.array_equal(suc[clear], REF[f"manip_{tp}_{tr}_success"][clear])
    assert np.array_equal(ms.block_reward(ga, gb, tp, tr, "sparse")[clear], REF[f"manip_{tp}_{tr}_reward_sparse"][clear])
    assert np.allclose(ms.block_reward(ga, gb, tp, tr, "dense"), REF[f"manip_{tp}_{tr}_reward_dense"], rtol=0, atol=1e-7)
