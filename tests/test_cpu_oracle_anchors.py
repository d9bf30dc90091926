"""Analytic anchors for the physics of the fp64 oracle (oracle/grx_oracle.c), round 2.

No MuJoCo can be run in the build container, so the oracle's `mj_step` restatement is pinned against everything that
follows in closed form from MuJoCo's *documented* constraint model (impedance d(r), reference
acceleration aref = -b v - k d r with b = 2 / (dmax tc), k = 1 / (dmax^2 tc^2 dampratio^2), regulariser
R = (1 - d) / d * diagApprox, pyramid rows n +- mu t with R_py = 2 mu^2 R_first, friction-loss boxes, implicit Euler,
RK4).  Every test below is a case where that model has an exact answer that does NOT depend on this repo's code, and each
one fails for a wrong R / aref / pyramid scaling / row count / invweight (the factor that breaks it is named in the test).
All models are compiled from MJCF snippets by the product compiler, so the compiler's defaults and inertia code are on
the tested path as well.
"""
import os
import tempfile

import numpy as np
import pytest
from scipy.optimize import brentq

from gymnasium_robotics_amd.mjcf import compile_mjcf
from oracle.oracle_sim import OracleSim

G = 9.81


def _compile(xml: str):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.xml")
        with open(p, "w") as f:
            f.write(xml)
        return compile_mjcf(p)


def impedance(r, dmin=0.9, dmax=0.95, width=0.001, mid=0.5, power=2.0):
    """MuJoCo's documented impedance sigmoid d(|r|) (XML reference, solimp)."""
    x = min(abs(r) / width, 1.0)
    if x >= 1:
        return dmax
    y = x ** power / mid ** (power - 1) if x <= mid else 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
    return dmin + y * (dmax - dmin)


def stiffness(dmax=0.95, tc=0.02, dr=1.0):
    return 1.0 / (dmax ** 2 * tc ** 2 * dr ** 2)


def rest_depth(weight_factor, solimp=(0.9, 0.95, 0.001, 0.5, 2.0), tc=0.02, dr=1.0):
    """|r| solving  weight_factor * d(r)^2 / (1 - d(r)) * k * r = g   (force balance of ONE soft row family at rest:
    f = D * k * d * |r| per row with D = d / ((1 - d) diagApprox); weight_factor collects row count, mass and diagApprox)."""
    k = stiffness(solimp[1], tc, dr)
    f = lambda r: weight_factor * impedance(r, *solimp) ** 2 / (1 - impedance(r, *solimp)) * k * r - G
    return brentq(f, 1e-12, 5e-2, xtol=1e-16)


def _settle(s, n=4000):
    s.step(n)
    assert np.abs(s.qvel).max() < 1e-9, "did not come to rest"


# --------------------------------------------------------------------------------------------------------- contacts
SPHERE = """<mujoco><option timestep="0.001"/><worldbody>
<geom name="floor" type="plane" size="1 1 0.1" condim="{cd}" friction="{mu} {spin} 0.0001"/>
<body pos="0 0 0.1"><freejoint/><geom type="sphere" size="0.1" mass="{mass}" condim="{cd}" friction="{mu} {spin} 0.0001"/></body>
</worldbody></mujoco>"""


def test_anchor_01_frictionless_contact_rest_depth():
    """condim 1: one row, diagApprox = 1/m  =>  d^2/(1-d) k r = g, independent of the mass.  Breaks for a wrong R, a wrong k
    (dmax^2 factor) or a body_invweight0 that is not 1/m for a free body."""
    for mass in (0.3, 2.5):
        s = OracleSim(_compile(SPHERE.format(cd=1, mu=1, spin=0.005, mass=mass)))
        _settle(s)
        assert s.nefc == 1
        assert abs((0.1 - s.qpos[2]) / rest_depth(1.0) - 1) < 1e-7


@pytest.mark.parametrize("mu", [0.3, 0.7, 1.0])
def test_anchor_02_pyramid_condim3_rest_depth_depends_on_mu(mu):
    """condim 3: four rows n +- mu t, each with R = 2 mu^2 (1-d)/d (1 + mu^2)/m and the same aref at rest, so
    4 D k d r = m g  <=>  [2 / (mu^2 (1 + mu^2))] d^2/(1-d) k r = g.  mu = 1 (the Fetch value) hides every power of mu: the
    other two values catch a wrong 2 mu^2 scaling and a wrong diagApprox = tran (1 + mu^2)."""
    s = OracleSim(_compile(SPHERE.format(cd=3, mu=mu, spin=0.005, mass=1.3)))
    _settle(s)
    assert s.nefc == 4
    assert abs((0.1 - s.qpos[2]) / rest_depth(2.0 / (mu * mu * (1 + mu * mu))) - 1) < 1e-7


def test_anchor_03_pyramid_condim4_has_six_rows_sharing_R():
    """condim 4 (the Fetch finger pads, the hand's object contacts): 2 (4 - 1) = 6 rows; every row carries the R of the FIRST
    pair (2 mu^2 (1-d)/d tran (1 + mu^2)) even though the torsional pair's own diagApprox (rot invweight) differs:
    6 D k d r = m g."""
    mu = 0.6
    s = OracleSim(_compile(SPHERE.format(cd=4, mu=mu, spin=0.02, mass=0.8)))
    _settle(s)
    assert s.nefc == 6
    R = s.efc("R")
    assert np.allclose(R, R[0], rtol=0, atol=0)
    assert abs((0.1 - s.qpos[2]) / rest_depth(3.0 / (mu * mu * (1 + mu * mu))) - 1) < 1e-7


def test_anchor_03b_pyramid_condim6_has_ten_rows_sharing_R():
    """condim 6 (FrankaKitchen's finger pads, franka_assets/assets.xml:51-55): 2 (6 - 1) = 10 rows -- two tangents, torsion, two rolling axes -- all with
    the R of the first pair: 10 D k d r = m g, i.e. the rest depth of the condim-3 formula with the factor 2 replaced by 5.  The MJCF friction
    attribute only carries (slide, spin, roll); the compiler must spread them to the five pyramid coefficients (slide, slide, spin, roll, roll)."""
    mu = 0.7
    xml = SPHERE.format(cd=6, mu=mu, spin=0.03, mass=0.9).replace("0.0001", "0.002")
    m = _compile(xml)
    s = OracleSim(m)
    _settle(s)
    assert s.nefc == 10
    R = s.efc("R")
    assert np.allclose(R, R[0], rtol=0, atol=0)
    assert abs((0.1 - s.qpos[2]) / rest_depth(5.0 / (mu * mu * (1 + mu * mu))) - 1) < 1e-7
    fr = np.asarray(m.tables["pair_friction"]).reshape(-1, 5)[0]
    assert np.allclose(fr, [mu, mu, 0.03, 0.002, 0.002])
    # rolling resistance is real: the same ball given a roll about y decelerates with the roll rows in place and keeps rolling without them (condim 3)
    def roll_speed_after(cd):
        b = OracleSim(_compile(SPHERE.format(cd=cd, mu=mu, spin=0.03, mass=0.9).replace("0.0001", "0.002")))
        b.step(500)
        b.qvel[:] = [0.1, 0, 0, 0, 1.0, 0]          # v = w x r: pure rolling along x
        b.step(400)
        return abs(b.qvel[4])
    assert roll_speed_after(6) < 0.9 * roll_speed_after(3)


def test_anchor_04_impedance_sigmoid_spot_values():
    """R / diagApprox = (1 - d)/d read back at prescribed penetrations of a frictionless sphere: d(0) = dmin, d(width) = dmax,
    d(mid * width) = dmin + mid (dmax - dmin) for any power, and the power-law value below the midpoint."""
    xml = """<mujoco><option timestep="0.001"/><worldbody>
    <geom type="plane" size="1 1 0.1" condim="1" solimp="0.8 0.98 0.004 0.25 3"/>
    <body pos="0 0 0.1"><freejoint/><geom type="sphere" size="0.1" mass="1" condim="1" solimp="0.8 0.98 0.004 0.25 3"/></body>
    </worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    for depth, y in ((1e-9, 0.0), (0.004, 1.0), (0.001, 0.25), (0.0005, 0.125 ** 3 / 0.25 ** 2), (0.002, 1 - 0.5 ** 3 / 0.75 ** 2)):
        s.qpos[2] = 0.1 - depth
        s.forward()
        d = 1.0 / (1.0 + s.efc("R")[0] / s.efc("diagApprox")[0])
        assert abs(d - (0.8 + y * 0.18)) < 1e-7, (depth, d)


def test_anchor_05_two_body_contact_conserves_momentum():
    """Head-on soft collision of two free spheres in zero gravity: the contact Jacobian acts with opposite signs on the two
    bodies, so total linear momentum is conserved to rounding through the whole contact episode and the spheres separate."""
    xml = """<mujoco><option timestep="0.0005" gravity="0 0 0"/><worldbody>
    <body pos="-0.15 0 0"><freejoint/><geom type="sphere" size="0.1" mass="1.0" condim="3"/></body>
    <body pos="0.15 0.02 0"><freejoint/><geom type="sphere" size="0.1" mass="3.0" condim="3"/></body>
    </worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    s.qvel[0], s.qvel[6] = 1.0, -0.5
    p0 = 1.0 * s.qvel[0:3] + 3.0 * s.qvel[6:9]
    touched = False
    for _ in range(1200):
        s.step(1)
        touched |= s.nefc > 0
        assert np.allclose(1.0 * s.qvel[0:3] + 3.0 * s.qvel[6:9], p0, atol=1e-11)
    assert touched and s.nefc == 0 and s.qvel[0] < 0 < s.qvel[6] + 0.5   # light sphere bounced back, contact released


# --------------------------------------------------------------------------------------------------------- limits / tendons / welds
def test_anchor_06_joint_limit_rest_depth_with_custom_solref_solimp():
    """A mass on a vertical slide resting on its lower limit: one row, diagApprox = dof_invweight0 = 1/m, so
    d^2/(1-d) k r = g with the JOINT's solreflimit (tc 0.01 -> refsafe keeps it, dampratio 0.7) and solimplimit (power 3)."""
    xml = """<mujoco><option timestep="0.0005"/><worldbody>
    <body pos="0 0 1"><joint type="slide" axis="0 0 1" limited="true" range="0 1" solreflimit="0.01 0.7" solimplimit="0.8 0.99 0.002 0.3 3"/>
    <geom type="sphere" size="0.05" mass="4" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    s.step(8000)
    assert np.abs(s.qvel).max() < 1e-9 and s.nefc == 1
    assert abs(-s.qpos[0] / rest_depth(1.0, (0.8, 0.99, 0.002, 0.3, 3.0), tc=0.01, dr=0.7) - 1) < 1e-7


def test_anchor_07_refsafe_clamps_time_constant_to_two_timesteps():
    """solref timeconst 0.001 with h = 0.002 is raised to 2 h = 0.004 (refsafe, on by default): the rest depth follows tc = 0.004."""
    xml = """<mujoco><option timestep="0.002"/><worldbody>
    <body pos="0 0 1"><joint type="slide" axis="0 0 1" limited="true" range="0 1" solreflimit="0.001 1"/>
    <geom type="sphere" size="0.05" mass="1" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    _settle(s)
    assert abs(-s.qpos[0] / rest_depth(1.0, tc=0.004) - 1) < 1e-7


def test_anchor_08_limit_response_is_critically_damped_with_the_solref_time_constant():
    """With dmin = dmax = d the solver gives a = d * aref for a single row on a 1-dof mass (D / (m + D) = d), and
    aref = -(2 / (d tc)) v - (1 / (d^2 tc^2)) d r, so the penetration obeys r'' = -(2 / tc) r' - r / tc^2:
    r(t) = r0 (1 + t / tc) exp(-t / tc) -- MuJoCo's documented meaning of solref = (timeconst, dampratio = 1) -- for as long as the reference
    acceleration pushes outwards (t < tc; after that the unilateral row switches off and the mass coasts, also checked).
    A wrong b or k (e.g. a missing dmax factor) changes the decay."""
    tc, r0, h = 0.05, -0.01, 1e-5
    xml = f"""<mujoco><option timestep="{h}" gravity="0 0 0"/><worldbody>
    <body pos="0 0 1"><joint type="slide" axis="0 0 1" limited="true" range="0 1" solreflimit="{tc} 1" solimplimit="0.6 0.6 0.001 0.5 2"/>
    <geom type="sphere" size="0.05" mass="2" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    s.qpos[0] = r0
    for frac in (0.25, 0.5, 0.95):
        n = int(round(frac * tc / h)) - int(round(s.time[0] / h))
        s.step(n)
        t = s.time[0]
        assert abs(s.qpos[0] / (r0 * (1 + t / tc) * np.exp(-t / tc)) - 1) < 2e-4, frac   # O(h) integrator error only
    s.step(int(round(0.5 * tc / h)))   # t > tc: aref < 0, a unilateral row cannot pull -> inactive, constant velocity
    s.forward()
    assert s.qacc[0] == 0.0 and abs(s.qvel[0] - (-r0 / tc) * np.exp(-1.0)) < 2e-4 * abs(r0 / tc)


def test_anchor_09_tendon_limit_rest_depth():
    """Two equal masses on vertical slides coupled by the fixed tendon L = q1 + q2 >= 0: the single tendon-limit row has
    J = (1, 1), tendon_invweight0 = J M^-1 J' = 2 / m, and carries m g on each dof: d^2/(1-d) k r / 2 = g."""
    xml = """<mujoco><option timestep="0.0005"/><worldbody>
    <body pos="0 0 1"><joint name="a" type="slide" axis="0 0 1"/><geom type="sphere" size="0.05" mass="1.5" contype="0" conaffinity="0"/></body>
    <body pos="1 0 1"><joint name="b" type="slide" axis="0 0 1"/><geom type="sphere" size="0.05" mass="1.5" contype="0" conaffinity="0"/></body>
    </worldbody><tendon><fixed name="t" limited="true" range="0 1"><joint joint="a" coef="1"/><joint joint="b" coef="1"/></fixed></tendon></mujoco>"""
    s = OracleSim(_compile(xml))
    s.step(8000)
    assert np.abs(s.qvel).max() < 1e-9 and s.nefc == 1
    assert abs(s.qpos[0] - s.qpos[1]) < 1e-12
    assert abs(-(s.qpos[0] + s.qpos[1]) / rest_depth(0.5) - 1) < 1e-7


def test_anchor_10_weld_to_mocap_sags_by_the_soft_constraint_offset():
    """A free body welded to a mocap body (the Fetch gripper construction, assets/fetch/shared.xml:38-40) hangs below the
    target by the offset at which the weld's translational z row carries m g: d^2/(1-d) k r = g with eq_invweight = 1/m
    (the mocap side is the world); the five other weld rows stay at zero residual."""
    xml = """<mujoco><option timestep="0.001"/><worldbody>
    <body name="mocap" mocap="true" pos="0.3 0.2 1"/>
    <body name="b" pos="0.3 0.2 1"><freejoint/><geom type="box" size="0.05 0.04 0.03" mass="1.7" contype="0" conaffinity="0"/></body>
    </worldbody><equality><weld body1="mocap" body2="b" solref="0.02 1" solimp="0.9 0.95 0.001"/></equality></mujoco>"""
    s = OracleSim(_compile(xml))
    _settle(s, 6000)
    assert s.nefc == 6
    assert abs((1.0 - s.qpos[2]) / rest_depth(1.0) - 1) < 1e-6
    assert np.allclose(s.qpos[[0, 1]], [0.3, 0.2], atol=1e-10) and np.allclose(s.qpos[3:7], [1, 0, 0, 0], atol=1e-10)


# --------------------------------------------------------------------------------------------------------- friction loss, actuators
SLIDER = """<mujoco><option timestep="0.001"/><worldbody>
<body pos="0 0 1"><joint name="j" type="slide" axis="0 0 1" frictionloss="{fl}"/><geom type="sphere" size="0.05" mass="2" contype="0" conaffinity="0"/></body>
</worldbody>{extra}</mujoco>"""


def test_anchor_11_frictionloss_saturates_at_its_bound():
    """Friction loss below the weight: the row saturates, the mass falls with a = -(g - f / m) exactly."""
    s = OracleSim(_compile(SLIDER.format(fl=5.0, extra="")))
    s.step(50)
    s.forward()
    assert abs(s.qacc[0] + (G - 5.0 / 2)) < 1e-12


def test_anchor_12_frictionloss_creep_velocity_of_the_soft_row():
    """Friction loss above the weight: the row stays in its quadratic zone and the mass creeps at the speed where the row's
    damping term carries the weight: f = D b |v| = m g with D = d m / (1 - d) (d = dmin: a friction row has zero residual)
    and b = 2 / (dmax tc)  =>  |v| = g (1 - dmin) dmax tc / (2 dmin).  Breaks if friction rows are given a stiffness term or
    the wrong impedance."""
    s = OracleSim(_compile(SLIDER.format(fl=50.0, extra="")))
    s.step(3000)
    assert abs(-s.qvel[0] / (G * (1 - 0.9) * 0.95 * 0.02 / (2 * 0.9)) - 1) < 1e-9


def test_anchor_13_position_actuator_steady_state_and_forcerange():
    """<position kp>: force = kp (ctrl - q); at rest kp (c - q) = m g.  With a forcerange below the weight the force clamps and
    the mass accelerates at -(g - F / m)."""
    act = '<actuator><position joint="j" kp="400" {fr}/></actuator>'
    m2 = _compile(SLIDER.replace('frictionloss="{fl}"', 'damping="30"').format(extra=act.format(fr='ctrllimited="true" ctrlrange="-1 1"')))
    s = OracleSim(m2)
    s.ctrl[0] = 0.3
    s.step(6000)
    assert abs(s.qvel[0]) < 1e-9 and abs(s.qpos[0] - (0.3 - 2 * G / 400)) < 1e-9
    s.ctrl[0] = 5.0   # clamped to ctrlrange 1
    s.step(6000)
    assert abs(s.qpos[0] - (1.0 - 2 * G / 400)) < 1e-9
    m3 = _compile(SLIDER.format(fl=0, extra=act.format(fr='forcelimited="true" forcerange="-10 10"')))
    s = OracleSim(m3)
    s.ctrl[0] = 100.0
    s.forward()
    assert abs(s.qacc[0] - (10.0 / 2 - G)) < 1e-12


def test_anchor_14_general_affine_actuator():
    """<general biastype="affine">: force = gain ctrl + b0 + b1 q + b2 qdot (the Adroit hand's actuators, adroit_hammer.py:234-262)."""
    act = '<actuator><general joint="j" gainprm="10 0 0" biastype="affine" biasprm="3 -10 -2"/></actuator>'
    s = OracleSim(_compile(SLIDER.format(fl=0, extra=act)))
    s.qpos[0], s.qvel[0], s.ctrl[0] = 0.2, -0.5, 0.7
    s.forward()
    assert abs(s.qfrc_actuator[0] - (10 * 0.7 + 3 - 10 * 0.2 - 2 * -0.5)) < 1e-12


def test_anchor_15_implicit_joint_damping_matches_the_backward_euler_update():
    """Euler with joint damping integrates the damping implicitly: v' = (m v + h f) / (m + h c), which is stable for
    h c / m >> 1 (the Fetch base slides have damping 1e11)."""
    xml = """<mujoco><option timestep="0.002" gravity="0 0 0"/><worldbody>
    <body><joint type="slide" axis="1 0 0" damping="1e6"/><geom type="sphere" size="0.05" mass="2" contype="0" conaffinity="0"/></body>
    </worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    s.qvel[0] = 1.0
    s.step(1)
    assert abs(s.qvel[0] - 2.0 / (2.0 + 0.002 * 1e6)) < 1e-15


# --------------------------------------------------------------------------------------------------------- integrators
PEND = """<mujoco><compiler angle="radian"/><option timestep="{h}" integrator="{integ}"/><worldbody>
<body pos="0 0 1"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.01" pos="0.5 0 0" mass="1" contype="0" conaffinity="0"/></body>
</worldbody></mujoco>"""


def _pendulum_angle(integ, h, T=0.5):
    s = OracleSim(_compile(PEND.format(h=h, integ=integ)))
    s.qpos[0] = 0.4
    s.step(int(round(T / h)))
    return s.qpos[0]


def test_anchor_16_rk4_is_fourth_order_and_euler_first_order():
    """Order of convergence on a pendulum (AntMaze uses RK4, ant.xml:3): halving h divides the RK4 error by ~16, the
    semi-implicit Euler error by ~2."""
    ref = _pendulum_angle("RK4", 1e-4)
    e = [abs(_pendulum_angle("RK4", h) - ref) for h in (0.02, 0.01, 0.005)]
    assert 12 < e[0] / e[1] < 20 and 12 < e[1] / e[2] < 20
    e = [abs(_pendulum_angle("Euler", h) - ref) for h in (0.004, 0.002, 0.001)]
    assert 1.8 < e[0] / e[1] < 2.2 and 1.8 < e[1] / e[2] < 2.2


# --------------------------------------------------------------------------------------------------------- compiler constants
def test_anchor_17_geom_inertias_from_density():
    """Masses / principal inertias of the primitive geoms the models use, from the textbook formulae (density 1000 default)."""
    xml = """<mujoco><worldbody>
    <body name="box"><freejoint/><geom type="box" size="0.1 0.2 0.3"/></body>
    <body name="sph"><freejoint/><geom type="sphere" size="0.1"/></body>
    <body name="cyl"><freejoint/><geom type="cylinder" size="0.1 0.2"/></body>
    <body name="ell"><freejoint/><geom type="ellipsoid" size="0.1 0.2 0.3"/></body>
    <body name="cap"><freejoint/><geom type="capsule" size="0.1 0.2"/></body>
    </worldbody></mujoco>"""
    m = _compile(xml)
    T, B = m.tables, m.names["body"]
    rho, pi = 1000.0, np.pi
    mb = rho * 8 * 0.1 * 0.2 * 0.3
    ms = rho * 4 / 3 * pi * 1e-3
    mc = rho * pi * 0.01 * 0.4
    me = rho * 4 / 3 * pi * 0.1 * 0.2 * 0.3
    mcyl, msph = rho * pi * 0.01 * 0.4, rho * 4 / 3 * pi * 1e-3     # capsule = cylinder + two half spheres
    want = {
        "box": (mb, [mb / 3 * (0.04 + 0.09), mb / 3 * (0.01 + 0.09), mb / 3 * (0.01 + 0.04)]),
        "sph": (ms, [0.4 * ms * 0.01] * 3),
        "cyl": (mc, [mc * (3 * 0.01 + 0.16) / 12, mc * (3 * 0.01 + 0.16) / 12, mc * 0.01 / 2]),
        "ell": (me, [me / 5 * (0.04 + 0.09), me / 5 * (0.01 + 0.09), me / 5 * (0.01 + 0.04)]),
        "cap": (mcyl + msph, [mcyl * (3 * 0.01 + 0.16) / 12 + msph * (0.4 * 0.01 + 0.04 + 0.375 * 0.2 * 0.1 * 2)] * 2 + [mcyl * 0.01 / 2 + 0.4 * msph * 0.01]),
    }
    for name, (mass, inertia) in want.items():
        b = B[name]
        assert abs(T["body_mass"][b] / mass - 1) < 1e-12, name
        assert np.allclose(np.sort(T["body_inertia"][b][:3]), np.sort(inertia), rtol=1e-10), name


def test_anchor_18_invweight0_of_free_and_hinged_bodies():
    """body_invweight0 = (1/m, mean 1/I) for a free body at its COM; dof_invweight0 of a hinge = 1 / (I + m l^2); these feed
    every diagApprox above, here checked directly."""
    xml = """<mujoco><compiler angle="radian"/><worldbody>
    <body name="f" pos="0 0 1"><freejoint/><geom type="box" size="0.1 0.2 0.3" mass="3"/></body>
    <body name="h" pos="1 0 1"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.05" pos="0.5 0 0" mass="2"/></body>
    </worldbody></mujoco>"""
    m = _compile(xml)
    T, B = m.tables, m.names["body"]
    I = 3.0 / 3 * np.array([0.04 + 0.09, 0.01 + 0.09, 0.01 + 0.04])
    gf = [k for k, b in enumerate(T["geom_bodyid"].ravel()) if b == B["f"]][0]
    assert np.allclose(T["geom_invweight0"][gf], [1 / 3.0, np.mean(1 / I)], rtol=1e-12)   # the body's invweight0, carried by its geoms
    assert np.allclose(T["dof_invweight0"][:6].ravel(), [1 / 3.0] * 3 + [np.mean(1 / I)] * 3, rtol=1e-12)
    Ih = 0.4 * 2 * 0.05 ** 2 + 2 * 0.25
    assert abs(T["dof_invweight0"].ravel()[6] * Ih - 1) < 1e-12


@pytest.mark.parametrize("mu", [0.5, 1.0])
def test_anchor_19_friction_creep_closed_form_and_noslip_removes_it(mu):
    """A point mass (three slide joints, sphere geom) on a plane, gravity tilted by theta with tan(theta) < mu.  The regularised pyramid holds it only up to a
    creep: at steady state the rows n +- mu t carry f = D (k d |r| -+ b mu v), so the tangential balance is 2 mu^2 D b v = m g sin(theta) with
    1/D = R = 2 mu^2 (1 - d)/d (1 + mu^2)/m  =>  |v| = g sin(theta) (1 - d)(1 + mu^2) dmax tc / (2 d), d = d(r) at the rest depth r that carries
    g cos(theta) (anchor 02).  With the noslip post-solver (Adroit: adroit_assets.xml:3) the friction dimensions are re-solved WITHOUT regularisation: the
    creep is gone, the rest depth (normal direction: untouched) stays.  Breaks for a friction row with a stiffness term, a wrong R_py, or a noslip pass that
    also hardens the normal."""
    th = 0.15
    xml = """<mujoco><option timestep="0.001" gravity="{gx} 0 {gz}" noslip_iterations="{ns}" noslip_tolerance="1e-14"/><worldbody>
    <geom type="plane" size="2 2 0.1" condim="3" friction="{mu} 0.005 0.0001"/>
    <body pos="0 0 0.1"><joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 1 0"/><joint type="slide" axis="0 0 1"/>
    <geom type="sphere" size="0.1" mass="0.8" condim="3" friction="{mu} 0.005 0.0001"/></body></worldbody></mujoco>"""
    r0 = rest_depth(2.0 / (mu * mu * (1 + mu * mu)) / np.cos(th))                 # carries g cos(theta)
    d = impedance(r0)
    v_creep = G * np.sin(th) * (1 - d) * (1 + mu * mu) * 0.95 * 0.02 / (2 * d)
    out = {}
    for ns in (0, 30):
        s = OracleSim(_compile(xml.format(gx=G * np.sin(th), gz=-G * np.cos(th), ns=ns, mu=mu)))
        s.step(6000)
        out[ns] = (s.qvel[0], -s.qpos[2])
        assert abs(s.qvel[1]) < 1e-12 and abs(s.qvel[2]) < 1e-9
    assert abs(out[0][0] / v_creep - 1) < 1e-6 and abs(out[0][1] / r0 - 1) < 1e-6        # soft rows: the closed-form creep
    assert abs(out[30][0]) < 1e-9 * v_creep + 1e-12 and abs(out[30][1] / r0 - 1) < 1e-6     # noslip: no creep, same depth


@pytest.mark.parametrize("other", ['type="box" size="0.2 0.15 0.05"', 'type="capsule" size="0.04 0.2" euler="0 90 0"', 'type="sphere" size="0.12"'])
def test_anchor_20_portal_routine_agrees_with_the_analytic_sphere(other):
    """The general convex narrow phase (portal refinement; egg, puck, hammer head, kettle) against the analytic sphere routines: an ELLIPSOID with three equal
    radii is a sphere, so its contact with a box / capsule / sphere -- which goes through the portal search -- must be the contact the closed-form
    sphere-box / sphere-capsule / sphere-sphere routines give for a sphere primitive in the same pose: same distance, normal and position to the routine's
    tolerance (mpr_tolerance 1e-6).  Two unrelated algorithms, one answer."""
    xml = """<mujoco><option timestep="0.001"/><worldbody>
    <geom name="o" {other} pos="0 0 0"/>
    <body pos="0 0 0.5"><freejoint/><geom name="s" {shape} mass="0.3"/></body></worldbody></mujoco>"""
    sims = [OracleSim(_compile(xml.format(other=other, shape=sh))) for sh in ('type="sphere" size="0.07"', 'type="ellipsoid" size="0.07 0.07 0.07"')]
    rng = np.random.default_rng(4)

    def contact(s, pos, q):
        s.qpos[:] = np.r_[pos, q]
        s.qvel[:] = 0
        s.forward()
        return s.contacts()

    hits = 0
    for _ in range(40):
        dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        lo, hi = 0.0, 0.6                                   # along dirn: in contact at lo (centres coincide), free at hi
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            lo, hi = (mid, hi) if len(contact(sims[0], dirn * mid, q)) else (lo, mid)
        t = lo - rng.uniform(1e-4, 5e-4)                    # a physically relevant depth: soft contacts rest at fractions of a millimetre
        a, b = contact(sims[0], dirn * t, q), contact(sims[1], dirn * t, q)
        assert len(a) == 1 and len(b) == 1, (len(a), len(b))
        a, b = a[0], b[0]
        sgn = 1.0 if np.dot(a[4:7], b[4:7]) > 0 else -1.0   # the normal points from geom1 to geom2 and the pair order follows the geom types
        # a portal within eps = 1e-6 of the surface pins the distance to eps, the normal of a radius-r surface only to sqrt(2 eps / r) = 5e-3, the point to r times that
        # The DEPTH of a portal search is measured along the final portal's normal from the centre ray, not along the true minimal translation: against a flat face
        # (box) or along the centre line (sphere) it is exact to eps, against a curved surface met obliquely (capsule) it over-estimates by up to ~7 % of the
        # depth -- 23 micrometres at 0.35 mm in the worst of these 40 poses.  A property of the published algorithm, bounded by this test.
        assert -6e-4 < a[0] < 0 and abs(a[0] - b[0]) < 5e-6 + 0.09 * abs(a[0]) and np.abs(a[4:7] - sgn * b[4:7]).max() < 1.2e-2 and np.abs(a[1:4] - b[1:4]).max() < 1e-3, (a[:7], b[:7])
        if 'capsule' not in other:
            assert abs(a[0] - b[0]) < 5e-6, (a[0], b[0])
        hits += 1
    assert hits == 40


def test_anchor_21_joint_equality_couples_two_hinges():
    """<equality><joint polycoef="0 a 0 0 0"> (FrankaKitchen's knob <-> burner couplings, kitchen_franka/.../oven_asset.xml:40-46): a soft row with residual
    q1 - a q2, J = (1, -a), diagApprox = invweight0(dof1) + invweight0(dof2).  Two hinges about the gravity axis (no load), the first one driven to a
    fixed angle by a stiff position actuator: the coupled joint settles at q1 = a q2 exactly (no force left in the row), and the row's regulariser is
    (1 - d)/d (1/I1 + 1/I2)."""
    a = 3.0
    xml = f"""<mujoco><option timestep="0.002"/><worldbody>
    <body pos="0 0 0.2"><joint name="j1" type="hinge" axis="0 0 1" damping="0.05"/><geom type="box" size="0.1 0.02 0.02" pos="0.1 0 0" mass="0.5" contype="0" conaffinity="0"/></body>
    <body pos="0.5 0 0.2"><joint name="j2" type="hinge" axis="0 0 1" damping="0.05"/><geom type="box" size="0.05 0.02 0.02" pos="0.05 0 0" mass="0.2" contype="0" conaffinity="0"/></body>
    </worldbody>
    <equality><joint joint1="j1" joint2="j2" polycoef="0 {a} 0 0 0"/></equality>
    <actuator><position joint="j1" kp="50"/></actuator></mujoco>"""
    m = _compile(xml)
    s = OracleSim(m)
    s.ctrl[0] = 0.6
    s.forward()
    assert s.nefc == 1
    iw = np.asarray(m.tables["dof_invweight0"]).ravel()
    d0 = impedance(0.0)
    assert abs(s.efc("R")[0] / ((1 - d0) / d0 * (iw[0] + iw[1])) - 1) < 1e-12
    assert np.allclose(s.efc("J")[0], [1.0, -a])
    s.step(6000)
    assert np.abs(s.qvel).max() < 1e-7
    assert abs(s.qpos[0] - 0.6) < 1e-6 and abs(s.qpos[0] - a * s.qpos[1]) < 1e-7


def _write_cube_stl(path, half):
    """binary STL of an axis-aligned cube with the given half sizes (12 triangles)"""
    import struct

    hx, hy, hz = half
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = [t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))]
    with open(path, "wb") as f:
        f.write(b"\0" * 80 + struct.pack("<I", len(tris)))
        for t in tris:
            f.write(struct.pack("<3f", 0, 0, 0) + b"".join(struct.pack("<3f", *v[k]) for k in t) + b"\0\0")


@pytest.mark.parametrize("support", ["box", "mesh"])
def test_anchor_22_hull_contact_geometry(support):
    """The hull-vs-convex narrow phase (portal refinement over the mesh's convex hull: Fetch links, kitchen fixtures) against elementary geometry.  A 10 cm
    cube given as a MESH is placed into a static slab (a box primitive, or a second mesh) with a prescribed overlap: the routine must return ONE contact whose
    distance is minus the overlap and whose normal is the slab's face normal -- face on face (any yaw, any lateral offset) and edge on face (cube rolled by
    45 degrees: the overlap is that of its lowest edge).  Breaks for a depth measured along the centre ray instead of the normal, a hull that is not the mesh's
    convex hull, or a geom frame that is not the hull's centre of mass.  (A settle test is not possible here: on ONE contact point a cube rocks for ever, in
    MuJoCo as well -- which is why the force balance is pinned by the primitive anchors above and only the geometry here.)"""
    with tempfile.TemporaryDirectory() as d:
        _write_cube_stl(os.path.join(d, "cube.stl"), (0.05, 0.05, 0.05))
        _write_cube_stl(os.path.join(d, "slab.stl"), (0.3, 0.3, 0.05))
        slab = '<geom name="slab" type="box" size="0.3 0.3 0.05" pos="0 0 0.05" condim="1"/>' if support == "box" else \
               '<geom name="slab" type="mesh" mesh="slab" pos="0 0 0.05" condim="1"/>'
        xml = f"""<mujoco><option timestep="0.001"/><asset><mesh name="cube" file="cube.stl"/><mesh name="slab" file="slab.stl"/></asset><worldbody>
        {slab}
        <body pos="0 0 0.15"><freejoint/><geom name="cube" type="mesh" mesh="cube" mass="0.7" condim="1"/></body>
        </worldbody></mujoco>"""
        p = os.path.join(d, "m.xml")
        with open(p, "w") as f:
            f.write(xml)
        m = compile_mjcf(p)
    s = OracleSim(m)
    for overlap in (1e-4, 1e-3, 5e-3):
        for (x, y, yaw) in ((0.0, 0.0, 0.0), (0.013, -0.021, 0.0), (-0.05, 0.08, 0.6)):
            s.qpos[:] = [x, y, 0.15 - overlap, np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
            s.qvel[:] = 0
            s.forward()
            con = s.contacts()
            assert s.ncon == 1, (overlap, x, y, yaw, s.ncon)
            assert abs(-con[0, 0] / overlap - 1) < 2e-3 and abs(con[0, 0] + overlap) < 2e-6, (overlap, con[0, 0])      # mpr_tolerance 1e-6 on the portal distance
            assert abs(abs(con[0, 6]) - 1) < 1e-6 and np.abs(con[0, 4:6]).max() < 1e-3                                 # normal = the face normal
            assert abs(con[0, 1] - x) < 0.0501 + 1e-6 and abs(con[0, 2] - y) < 0.0501 + 1e-6 and abs(con[0, 3] - (0.1 - 0.5 * overlap)) < 1e-5   # a point of the overlap polygon, mid-way between the faces
        # edge on face: rolled by 45 degrees about x, the lowest edge is sqrt(2) * 0.05 below the centre
        h = np.sqrt(2.0) * 0.05
        s.qpos[:] = [0.01, 0.0, 0.1 + h - overlap, np.cos(np.pi / 8), np.sin(np.pi / 8), 0, 0]
        s.forward()
        con = s.contacts()
        assert s.ncon == 1 and abs(con[0, 0] + overlap) < 2e-6 and abs(abs(con[0, 6]) - 1) < 1e-5 and abs(con[0, 2]) < 1e-5, (overlap, con[0, :8])


@pytest.mark.parametrize("ground", ["plane", "box"])
@pytest.mark.parametrize("shape,ncon,z0", [('type="box" size="0.1 0.07 0.05"', 4, 0.05), ('type="capsule" size="0.04 0.12" euler="0 90 0"', 2, 0.04)])
def test_anchor_23_weight_splits_evenly_over_the_contacts_of_a_resting_primitive(shape, ncon, z0, ground):
    """A box lying on a plane has four corner contacts, a capsule lying on its side two (its cap centres): the analytic plane routines of the Fetch / hand /
    Adroit scenes.  By symmetry each contact carries m g / ncon on its four pyramid rows, so the rest depth solves
    [2 ncon / (mu^2 (1 + mu^2))] d^2/(1-d) k r = g (anchor 02 with ncon times the rows).  Breaks for a missing / duplicated contact, a contact at the wrong
    place (the body would tilt: checked), or a regulariser that depends on where on the body the contact sits."""
    mu = 0.8
    g = f'<geom type="plane" size="1 1 0.1" condim="3" friction="{mu} 0.005 0.0001"/>' if ground == "plane" else \
        f'<geom type="box" size="0.5 0.4 0.1" pos="0 0 -0.1" condim="3" friction="{mu} 0.005 0.0001"/>'      # a static slab: the box-box / capsule-box routines instead of the plane ones
    xml = f"""<mujoco><option timestep="0.001"/><worldbody>
    {g}
    <body pos="0 0 {z0}"><freejoint/><geom {shape} mass="1.1" condim="3" friction="{mu} 0.005 0.0001"/></body></worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    _settle(s, 6000)
    assert s.ncon == ncon and s.nefc == 4 * ncon, (s.ncon, s.nefc)
    assert abs((z0 - s.qpos[2]) / rest_depth(2.0 * ncon / (mu * mu * (1 + mu * mu))) - 1) < 1e-7
    assert np.abs(s.qpos[:2]).max() < 1e-5 and abs(abs(s.qpos[3]) - 1) < 1e-10            # no drift beyond the settling transient, no tilt
    d = s.contacts()[:, 0]
    assert np.abs(d - d.mean()).max() < 1e-12                                            # the same depth at every contact


@pytest.mark.parametrize("ground", ['type="plane" size="1 1 0.1" pos="0 0 0.1"', 'type="box" size="0.4 0.4 0.05" pos="0 0 0.05"'])
def test_anchor_24_off_centre_contact_force_and_turn(ground):
    """One contact that does not pass through the centre of mass: a cube tilted onto one corner, pressed into the ground by a prescribed overlap, at rest.
    The row sees A = J M^-1 J' = 1/m + |r x n|^2 / I (cube: I = 2/3 m a^2) while its regulariser uses the TRANSLATIONAL body weight only,
    R = (1 - d)/d / m (MuJoCo's diagApprox for a contact's normal), so f = (k d |r| + g) / (A + R), qacc = (f/m - g) n and, in the body frame of the free
    joint, I^-1 R' (r x n) f.  Breaks for a diagApprox that includes the rotational weight, a Jacobian with the wrong lever arm, or a world-frame omega."""
    a, m_ = 0.05, 0.9
    xml = f"""<mujoco><option timestep="0.001"/><worldbody><geom {ground} condim="1"/>
    <body pos="0 0 0.4"><freejoint/><geom type="box" size="{a} {a} {a}" mass="{m_}" condim="1"/></body></worldbody></mujoco>"""
    s = OracleSim(_compile(xml))
    qx = np.array([np.cos(0.26), np.sin(0.26), 0, 0]); qy = np.array([np.cos(0.17), 0, np.sin(0.17), 0])
    quat = np.array([qy[0] * qx[0] - qy[1:] @ qx[1:], *(qy[0] * qx[1:] + qx[0] * qy[1:] + np.cross(qy[1:], qx[1:]))])
    w, x, y, z = quat
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    corners = np.array([[sx, sy, sz] for sx in (-a, a) for sy in (-a, a) for sz in (-a, a)]) @ Rm.T
    low = corners[np.argmin(corners[:, 2])]
    inertia, rxn = 2.0 / 3.0 * m_ * a * a, np.cross(low, [0.0, 0.0, 1.0])
    for overlap in (2e-4, 1e-3, 3e-3):
        s.qpos[:] = np.r_[0.03, -0.02, 0.1 - low[2] - overlap, quat]
        s.qvel[:] = 0
        s.forward()
        d = impedance(overlap)
        A, R = 1.0 / m_ + rxn @ rxn / inertia, (1 - d) / d / m_
        f = (stiffness() * d * overlap + G) / (A + R)
        assert s.ncon == 1 and s.nefc == 1 and abs(s.efc("R")[0] / R - 1) < 1e-12
        assert abs(s.qacc[2] - (f / m_ - G)) < 1e-9 and np.abs(s.qacc[:2]).max() < 1e-12
        assert np.abs(s.qacc[3:] - Rm.T @ (rxn * f / inertia)).max() < 1e-8


DOUBLE_PENDULUM = """<mujoco><option timestep="0.001"/><worldbody>
<body pos="0 0 2"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.002" pos="0 0 -{l1}" mass="{m1}" contype="0" conaffinity="0"/>
<body pos="0 0 -{l1}"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.002" pos="0 0 -{l2}" mass="{m2}" contype="0" conaffinity="0"/></body></body>
</worldbody></mujoco>"""


def double_pendulum_acc(t1, t2, w1, w2, m1, m2, l1, l2, g=G):
    """textbook equations of motion of the planar double pendulum with point masses (absolute angles from the hanging position)"""
    den = 2 * m1 + m2 - m2 * np.cos(2 * t1 - 2 * t2)
    a1 = (-g * (2 * m1 + m2) * np.sin(t1) - m2 * g * np.sin(t1 - 2 * t2) - 2 * np.sin(t1 - t2) * m2 * (w2 * w2 * l2 + w1 * w1 * l1 * np.cos(t1 - t2))) / (l1 * den)
    a2 = (2 * np.sin(t1 - t2) * (w1 * w1 * l1 * (m1 + m2) + g * (m1 + m2) * np.cos(t1) + w2 * w2 * l2 * m2 * np.cos(t1 - t2))) / (l2 * den)
    return a1, a2


def test_anchor_25_double_pendulum_accelerations():
    """Two hinges, two point masses: the joint accelerations at arbitrary states against the textbook equations of motion -- composite-inertia mass matrix
    (off-diagonal coupling), centrifugal / Coriolis bias and gravity of the RNE pass in one number each.  MuJoCo's second joint angle is relative:
    theta2 = q1 + q2.  (The 2 mm spheres add 0.4 m r^2 of rotational inertia: 6e-6 of m l^2.)"""
    m1, m2, l1, l2 = 0.7, 0.4, 0.5, 0.35
    s = OracleSim(_compile(DOUBLE_PENDULUM.format(m1=m1, m2=m2, l1=l1, l2=l2)))
    rng = np.random.default_rng(2)
    for _ in range(20):
        q, v = rng.uniform(-2.5, 2.5, 2), rng.uniform(-6, 6, 2)
        s.qpos[:], s.qvel[:] = q, v
        s.forward()
        a1, a2 = double_pendulum_acc(q[0], q[0] + q[1], v[0], v[0] + v[1], m1, m2, l1, l2)
        scale = max(abs(a1), abs(a2), 1.0)
        assert abs(s.qacc[0] - a1) < 5e-5 * scale and abs(s.qacc[0] + s.qacc[1] - a2) < 5e-5 * scale, (q, v, s.qacc, a1, a2)


FREE_BOX = """<mujoco><option timestep="0.001" gravity="0 0 0"/><worldbody>
<body pos="0 0 1"><freejoint/><geom type="box" size="0.1 0.07 0.05" mass="1.3" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""


def euler_equations_acc(w, half=(0.1, 0.07, 0.05), mass=1.3):
    """torque-free rigid body, body frame: I w' = -w x (I w), I of a box = m/3 (b^2 + c^2, a^2 + c^2, a^2 + b^2)"""
    a, b, c = half
    inertia = mass / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    return -np.cross(w, inertia * w) / inertia


def test_anchor_26_torque_free_rotation_obeys_eulers_equations():
    """A free box spinning about a non-principal axis in zero gravity: the angular part of qacc (body frame, as MuJoCo keeps a free joint's angular velocity)
    is Euler's equations, the linear part stays zero.  The gyroscopic term of the bias force and the box inertia, in one vector."""
    s = OracleSim(_compile(FREE_BOX))
    rng = np.random.default_rng(5)
    for _ in range(10):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w = rng.uniform(-8, 8, 3)
        s.qpos[:] = np.r_[0.1, -0.2, 1.0, q]
        s.qvel[:] = np.r_[rng.uniform(-1, 1, 3), w]
        s.forward()
        assert np.abs(s.qacc[:3]).max() < 1e-12 and np.abs(s.qacc[3:] - euler_equations_acc(w)).max() < 1e-10


ARMATURE = """<mujoco><option timestep="0.001"/><worldbody>
<body pos="0 0 1"><joint name="j" type="hinge" axis="0 1 0" armature="{arm}" damping="{damp}"/><geom type="sphere" size="0.002" pos="0 0 -{l}" mass="{m}" contype="0" conaffinity="0"/></body>
</worldbody><actuator><motor joint="j" gear="{gear}"/></actuator></mujoco>"""


def test_anchor_27_armature_gear_and_damping_of_a_driven_hinge():
    """One hinge with a point mass, rotor inertia (armature), a geared motor and joint damping: qacc = (gear u - c w - m g l sin q) / (m l^2 + armature) in
    mj_forward (the damping enters explicitly there; the Euler step then integrates it implicitly, anchor 15).  The Fetch and Shadow-hand joints all carry
    armature and damping."""
    m, l, arm, gear, damp = 0.6, 0.4, 0.02, 3.0, 0.15
    s = OracleSim(_compile(ARMATURE.format(m=m, l=l, arm=arm, gear=gear, damp=damp)))
    rng = np.random.default_rng(3)
    for _ in range(10):
        q, w, u = rng.uniform(-3, 3), rng.uniform(-5, 5), rng.uniform(-1, 1)
        s.qpos[0], s.qvel[0], s.ctrl[0] = q, w, u
        s.forward()
        inertia = m * l * l + 0.4 * m * 0.002 ** 2 + arm
        assert abs(s.qacc[0] - (gear * u - damp * w - m * G * l * np.sin(q)) / inertia) < 1e-9


POINT_ON_PLANE = """<mujoco><option timestep="0.001"/><worldbody>
<geom type="plane" size="2 2 0.1" condim="3" friction="{mu} 0.005 0.0001"/>
<body pos="0 0 0.1"><joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 1 0"/><joint type="slide" axis="0 0 1"/>
<geom type="sphere" size="0.1" mass="0.8" condim="3" friction="{mu} 0.005 0.0001"/></body></worldbody></mujoco>"""


def sliding_deceleration(v, mu, d, dmax=0.95, tc=0.02):
    """point mass at its rest depth sliding slowly: the rows n +- mu t carry D (K -+ b mu v - (a_z +- mu a_x)) with 2 mu^2 D = m d / ((1 - d)(1 + mu^2)), the
    vertical balance is untouched and a_x = -b v d / ((1 - d)(1 + mu^2) + d), b = 2 / (dmax tc): the viscous regime of the soft friction cone"""
    return -(2.0 / (dmax * tc)) * v * d / ((1 - d) * (1 + mu * mu) + d)


@pytest.mark.parametrize("mu", [0.5, 1.0])
def test_anchor_28_slow_sliding_is_viscous_with_the_solref_damping(mu):
    r0 = rest_depth(2.0 / (mu * mu * (1 + mu * mu)))
    s = OracleSim(_compile(POINT_ON_PLANE.format(mu=mu)))
    for v in (1e-3, 5e-3):
        s.qpos[:] = [0, 0, -r0]
        s.qvel[:] = [v, 0, 0]
        s.forward()
        assert s.nefc == 4 and abs(s.qacc[2]) < 1e-6 and abs(s.qacc[1]) < 1e-12
        assert abs(s.qacc[0] / sliding_deceleration(v, mu, impedance(r0)) - 1) < 1e-6


def fast_sliding_acc(v, mu, r0, m=0.8, dmax=0.95, tc=0.02):
    """the same point mass sliding FAST along +x: the velocity term b mu v of the row n - mu t dwarfs the stiffness term, that row alone carries force --
    f = D (K + b mu v + g) / (1 + D (1 + mu^2) / m) from f = D (K + b mu v - a_z + mu a_x), m a_x = -mu f, m a_z = -m g + f -- and pushes the body up so hard
    (a_z > K) that the other three rows would need negative forces: they are off"""
    d = impedance(r0)
    D = m * d / ((1 - d) * 2 * mu * mu * (1 + mu * mu))
    K, b = stiffness(dmax, tc) * d * r0, 2.0 / (dmax * tc)
    f = D * (K + b * mu * v + G) / (1 + D * (1 + mu * mu) / m)
    ax, az = -mu * f / m, -G + f / m
    assert D * (K - b * mu * v - az - mu * ax) < 0 and D * (K - az) < 0 < f, "not the one-row regime"
    return ax, az


@pytest.mark.parametrize("mu", [0.5, 1.0])
def test_anchor_29_fast_sliding_switches_one_pyramid_row_off(mu):
    """The inequality side of the solver: at 0.3 m/s ONE pyramid row is active and three are off (they would need negative forces); the accelerations follow
    from that single row in closed form.  Breaks for a solver that keeps rows on (bilateral friction), a wrong active set, or a wrong velocity term in aref."""
    r0 = rest_depth(2.0 / (mu * mu * (1 + mu * mu)))
    s = OracleSim(_compile(POINT_ON_PLANE.format(mu=mu)))
    s.qpos[:] = [0, 0, -r0]
    s.qvel[:] = [0.3, 0, 0]
    s.forward()
    ax, az = fast_sliding_acc(0.3, mu, r0)
    assert abs(s.qacc[0] / ax - 1) < 1e-6 and abs(s.qacc[2] / az - 1) < 1e-6 and abs(s.qacc[1]) < 1e-12
    assert (s.efc("force") > 0).sum() == 1


def spin_deceleration(w, mu, mu_t, r0, m, radius, dmax=0.95, tc=0.02):
    """a ball at its condim-4 rest depth spinning slowly about the contact normal: the torsional rows n +- mu_t e_n carry D (K -+ b mu_t w - (a_z +- mu_t alpha))
    with the R of the FIRST pair (1/D = 2 mu^2 (1-d)/d (1+mu^2)/m), so I alpha = -2 mu_t^2 D (b w + alpha), I = 2/5 m r^2"""
    d = impedance(r0)
    D = m * d / ((1 - d) * 2 * mu * mu * (1 + mu * mu))
    inertia, b = 0.4 * m * radius * radius, 2.0 / (dmax * tc)
    return -2 * mu_t * mu_t * D * b * w / (inertia + 2 * mu_t * mu_t * D)


def test_anchor_30_torsional_friction_of_condim4():
    """condim 4 (the Fetch finger pads, the hand's object contacts): a sphere spinning slowly about the vertical is braked by the torsional pair, whose rows share
    the first pair's R (anchor 03) but enter with the torsional coefficient: alpha in closed form, no linear acceleration.  Breaks for a torsional pair with its
    own R, a missing pair, or a spin coefficient applied as a length-free number to the wrong axis."""
    mu, mu_t, m_, rad = 0.6, 0.02, 0.8, 0.1
    r0 = rest_depth(3.0 / (mu * mu * (1 + mu * mu)))
    s = OracleSim(_compile(SPHERE.format(cd=4, mu=mu, spin=mu_t, mass=m_)))
    for w in (0.01, 0.05):          # b mu_t w below the stiffness term: both torsional rows stay on
        s.qpos[:] = [0, 0, 0.1 - r0, 1, 0, 0, 0]
        s.qvel[:] = [0, 0, 0, 0, 0, w]
        s.forward()
        assert s.nefc == 6 and np.abs(s.qacc[:5]).max() < 1e-6
        assert abs(s.qacc[5] / spin_deceleration(w, mu, mu_t, r0, m_, rad) - 1) < 1e-6
