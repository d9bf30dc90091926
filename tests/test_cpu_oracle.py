"""Known-answer anchors for the fp64 oracle (oracle/) -- the checker itself must be pinned before it is trusted.
No MuJoCo is available ("parity unpinned"), so the anchors are the documented / analytic ones:
the Fetch start pose from the reference docstrings, the closed-form rest penetration of MuJoCo's soft-contact
model, and conservation laws on models compiled from MJCF snippets."""
import os
import tempfile

import numpy as np
import pytest

from gymnasium_robotics_amd.mjcf import compile_mjcf
from oracle.fetch_oracle import OracleFetchEnv
from oracle.oracle_sim import OracleSim


def _compile(xml: str):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.xml")
        open(p, "w").write(xml)
        return compile_mjcf(p)


def test_fetch_documented_start_pose(fetch_models):
    """pick_and_place.py:104: "the gripper is placed in ... (x,y,z) = [1.3419 0.7491 0.555]" -- that is the mocap
    target (kinematics at the initial qpos + the fixed offset of fetch_env.py:411-413); base at [0.405, 0.48, 0]."""
    env = OracleFetchEnv(fetch_models["FetchPickAndPlace"], "FetchPickAndPlace")
    assert np.allclose(env.sim.mocap_pos, [1.3419, 0.7491, 0.555], atol=5e-5)
    assert np.allclose(env.sim.qpos[:3], [0.405, 0.48, 0.0], atol=1e-6)
    # after the 10x20 settle steps the weld has pulled the gripper onto the target in x,y; z still lags (0.5347:
    # the value every Fetch user sees as the third entry of the first observation)
    assert np.allclose(env.initial_gripper_xpos[:2], [1.3419, 0.7491], atol=1e-4)
    assert abs(env.initial_gripper_xpos[2] - 0.5347) < 1e-4
    # FetchSlide documents [1, 0.75, 0.41] for a different offset; Reach shares the P&P pose
    env_r = OracleFetchEnv(fetch_models["FetchReach"], "FetchReach")
    assert np.allclose(env_r.initial_gripper_xpos, env.initial_gripper_xpos, atol=2e-4)


def test_object_rest_height_matches_soft_contact_closed_form(fetch_models):
    """Block (2 kg, 4 corner contacts x 4 pyramid rows) resting on the table: force balance of MuJoCo's soft
    constraint model gives 16*D*k*d(r)*|r| = m*g with R = 2*mu^2*(1-d)/d*(1+mu^2)/m."""
    env = OracleFetchEnv(fetch_models["FetchPickAndPlace"], "FetchPickAndPlace")
    z = env.height_offset
    r = 0.425 - z
    assert 0 < r < 5e-4 and abs(z - 0.42) < 0.01  # docs: "fixed height of (z) = [0.42] m"
    dmin, dmax, width, mid, power = 0.9, 0.95, 0.001, 0.5, 2
    x = r / width
    d = dmin + (x * x / mid) * (dmax - dmin)
    R = 2 * 1.0 * ((1 - d) / d * (0.5 + 0.5))
    k = 1 / (dmax ** 2 * 0.02 ** 2)
    assert abs(16 * (1 / R) * k * d * r - 2 * 9.81) < 1e-3 * 2 * 9.81


FREE_BODY = """<mujoco><option timestep="0.002" gravity="0 0 0"/><worldbody>
<body pos="0 0 1"><freejoint/><geom type="box" size="0.1 0.2 0.3" mass="3" contype="0" conaffinity="0"/></body>
</worldbody></mujoco>"""

PENDULUM = """<mujoco><compiler angle="radian"/><option timestep="0.0005"/><worldbody>
<body pos="0 0 2"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.05" pos="0.5 0 0" mass="1" contype="0" conaffinity="0"/>
<body pos="0.5 0 0"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.05" pos="0.4 0 0" mass="0.5" contype="0" conaffinity="0"/></body>
</body></worldbody></mujoco>"""


def test_free_body_conserves_momentum_and_energy():
    m = _compile(FREE_BODY)
    s = OracleSim(m)
    s.qvel[:] = [0.3, -0.2, 0.1, 1.0, 2.0, -1.5]
    I = np.array([m.tables["body_inertia"][1][k] for k in range(3)])

    def energy():
        return 0.5 * 3 * np.dot(s.qvel[:3], s.qvel[:3]) + 0.5 * np.dot(I * s.qvel[3:], s.qvel[3:])

    def ang_mom_world():
        s.forward()
        R = s.xmat[9:18].reshape(3, 3)
        return R @ (I * s.qvel[3:])

    e0, l0, p0 = energy(), ang_mom_world(), s.qvel[:3].copy()
    s.step(2000)
    assert np.allclose(s.qvel[:3], p0, atol=1e-12)
    assert abs(energy() - e0) / e0 < 5e-3          # torque-free precession, semi-implicit Euler
    assert np.allclose(ang_mom_world(), l0, rtol=5e-3, atol=1e-4)
    assert abs(np.linalg.norm(s.qpos[3:7]) - 1) < 1e-12
    assert np.allclose(s.qpos[:3], [0.3 * 4, -0.2 * 4, 1 + 0.1 * 4], atol=1e-9)


def test_double_pendulum_energy_and_mass_matrix():
    m = _compile(PENDULUM)
    s = OracleSim(m)
    s.qpos[:] = [0.3, -0.5]
    s.forward()
    M = s.M.reshape(2, 2)
    # closed-form mass matrix of a planar double pendulum with point masses (+ sphere inertias 0.4 m r^2)
    l1, l2, m1, m2, q2 = 0.5, 0.4, 1.0, 0.5, -0.5
    i1, i2 = 0.4 * m1 * 0.05 ** 2, 0.4 * m2 * 0.05 ** 2
    m22 = m2 * l2 ** 2 + i2
    m12 = m22 + m2 * l1 * l2 * np.cos(q2)
    m11 = m1 * l1 ** 2 + i1 + m2 * (l1 ** 2 + l2 ** 2 + 2 * l1 * l2 * np.cos(q2)) + i2
    assert np.allclose(M, [[m11, m12], [m12, m22]], atol=1e-12)

    def energy():
        s.forward()
        ke = 0.5 * s.qvel @ s.M.reshape(2, 2) @ s.qvel
        pe = 9.81 * (m1 * (s.xpos[5] + s.xmat[9:18].reshape(3, 3)[2] @ [0.5, 0, 0]) + m2 * (s.xpos[8] + s.xmat[18:27].reshape(3, 3)[2] @ [0.4, 0, 0]))
        return ke + pe

    e0 = energy()
    s.step(4000)  # 2 s
    assert abs(energy() - e0) < 2e-2 * abs(e0)


def test_model_dimensions_match_survey_table(fetch_models):
    """FetchReach 15/15/0 nmocap 1; FetchPickAndPlace 22/21/2"""
    r, p = fetch_models["FetchReach"], fetch_models["FetchPickAndPlace"]
    assert (r.dim("nq"), r.dim("nv"), r.dim("nu"), r.dim("nmocap"), r.dim("neq")) == (15, 15, 0, 1, 1)
    assert (p.dim("nq"), p.dim("nv"), p.dim("nu"), p.dim("nmocap"), p.dim("neq")) == (22, 21, 2, 1, 1)
    assert abs(p.tables["body_mass"][p.names["body"]["object0"]] - 2.0) < 1e-12
    lim = r.tables["jnt_limited"].ravel()
    assert lim.sum() == 9  # torso, head x2, shoulder x2, elbow_flex, wrist_flex, 2 fingers (robot.xml:16-93)


def test_oracle_step_is_deterministic_and_healthy(fetch_models):
    outs = []
    for _ in range(2):
        env = OracleFetchEnv(fetch_models["FetchPush"], "FetchPush")
        env.reset(seed=7)
        rng = np.random.default_rng(0)
        tr = []
        for _ in range(20):
            o, r, *_ = env.step(rng.uniform(-1, 1, 4))
            tr.append(o["observation"])
        assert env.sim.bad_state == 0
        outs.append(np.array(tr))
    assert np.array_equal(outs[0], outs[1])


def test_fixtures_are_contiguous_rollouts():
    """The free-running parity tests (tests/test_gpu_horizons.py, tests/test_cpu_emu_horizons.py) replay the fixtures as ROLLOUTS: within an episode the recorded pre-step
    state of snapshot i + 1 must be the oracle's post-step state of snapshot i bit for bit (fixtures that record qpos_next), goals / model edits must not change inside an
    episode, and the hand fixtures' next joint angles are the previous observation's."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tolerance_cases import CASES, GOLDEN, ROLLOUT_FAMILIES, episode_runs

    for fam in ROLLOUT_FAMILIES:
        g = np.load(os.path.join(GOLDEN, CASES[fam][1]))
        run = episode_runs(g)
        cont = run[:-1] > 1
        assert cont.sum() >= 0.9 * len(run), fam
        if "qpos_next" in g.files:
            assert np.array_equal(g["qpos"][1:][cont], g["qpos_next"][:-1][cont]) and np.array_equal(g["qvel"][1:][cont], g["qvel_next"][:-1][cont]), fam
        else:
            assert np.array_equal(g["qpos"][1:, :24][cont], g["obs"][:-1, :24][cont]), fam
        for k in ("goal", "shift", "target"):
            if k in g.files:
                assert np.array_equal(g[k][1:][cont], g[k][:-1][cont]), (fam, k)
