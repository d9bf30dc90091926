"""The closed-form anchors of tests/test_cpu_oracle_anchors.py on the GPU build, through the C ABI (pytest -m gpu): grx_point_step is a generic free-running
stepper (any compiled model, ctrl = action, observation = qpos | qvel), so the HIP kernels themselves -- generic shape, fp32 -- have to settle on the analytic
rest depths, contact counts and creep speeds.  No oracle, no fixture: the expected numbers follow from MuJoCo's documented constraint model alone."""
import ctypes
import os
import tempfile

import numpy as np
import pytest

from test_cpu_oracle_anchors import G, SLIDER, SPHERE, _write_cube_stl, impedance, rest_depth, stiffness

pytestmark = pytest.mark.gpu


def _settle_on_gpu(xml, steps, n=32, ctrl=None, state=None, files=()):
    import torch

    from gymnasium_robotics_amd import _native
    from gymnasium_robotics_amd.mjcf import compile_mjcf

    assert torch.cuda.is_available(), "these tests need the GPU"
    with tempfile.TemporaryDirectory() as d:
        for name, half in files:
            _write_cube_stl(os.path.join(d, name), half)
        p = os.path.join(d, "m.xml")
        with open(p, "w") as f:
            f.write(xml)
        m = compile_mjcf(p)
    L = _native.lib()
    H, I, F = m.pack()
    h = ctypes.c_void_p()
    _native.check(L.grx_model_create(H.ctypes.data, H.size, I.ctypes.data, I.size, F.ctypes.data, F.size, 0, ctypes.byref(h)))
    try:
        nq, nv = m.dim("nq"), m.dim("nv")
        dev = torch.device("cuda:0")
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
        bufs = dict(qpos=torch.from_numpy(np.tile(m.tables["qpos0"].astype(np.float32), (n, 1))).to(dev), qvel=z(n, nv), qacc_ws=z(n, nv), goal=z(n, 2), action=z(n, max(m.dim("nu"), 1)),
                    obs=z(n, nq + nv), achieved=z(n, 2), reward=z(n), success=z(n, dtype=torch.uint8), terminated=z(n, dtype=torch.uint8), status=z(n, dtype=torch.int32))
        b = _native.PointBuffersStruct()
        for k, t in bufs.items():
            setattr(b, k, t.data_ptr())
        b.mask = b.packed = None
        if ctrl is not None:
            bufs["action"][:] = torch.tensor(ctrl, dtype=torch.float32, device=dev)
        if state is not None:
            bufs["qpos"][:] = torch.tensor(state[0], dtype=torch.float32, device=dev)
            bufs["qvel"][:] = torch.tensor(state[1], dtype=torch.float32, device=dev)
        per_call = 250 if steps >= 250 else steps
        task = _native.PointTaskStruct(per_call, 1, 1, 1, 0.45, 5.0)     # agent = 1: ctrl = action, no velocity clip; 250 raw physics steps per launch
        for _ in range(steps // per_call):
            _native.check(L.grx_point_step(h, ctypes.byref(task), ctypes.byref(b), n, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        torch.cuda.synchronize()
        qpos, qvel, status = bufs["qpos"].cpu().numpy().astype(np.float64), bufs["qvel"].cpu().numpy().astype(np.float64), bufs["status"].cpu().numpy()
    finally:
        L.grx_model_destroy(h)
    assert (status & 0xFFFF == 0).all() and np.isfinite(qpos).all()
    assert np.array_equal(qpos, np.tile(qpos[:1], (n, 1))), "identical worlds must stay identical"
    return qpos[0], qvel[0]


@pytest.mark.parametrize("cd,mu,factor", [(1, 1.0, 1.0), (3, 0.7, 2.0 / (0.49 * 1.49)), (3, 1.0, 1.0)])
def test_gpu_sphere_rest_depth(cd, mu, factor):
    qpos, qvel = _settle_on_gpu(SPHERE.format(cd=cd, mu=mu, spin=0.005, mass=1.3), 4000)
    assert np.abs(qvel).max() < 3e-5 and abs((0.1 - qpos[2]) / rest_depth(factor) - 1) < 2e-3


@pytest.mark.parametrize("ground", ["plane", "box"])
@pytest.mark.parametrize("shape,ncon,z0", [('type="box" size="0.1 0.07 0.05"', 4, 0.05), ('type="capsule" size="0.04 0.12" euler="0 90 0"', 2, 0.04)])
def test_gpu_multi_contact_rest_depth(shape, ncon, z0, ground):
    mu = 0.8
    g = f'<geom type="plane" size="1 1 0.1" condim="3" friction="{mu} 0.005 0.0001"/>' if ground == "plane" else \
        f'<geom type="box" size="0.5 0.4 0.1" pos="0 0 -0.1" condim="3" friction="{mu} 0.005 0.0001"/>'
    qpos, qvel = _settle_on_gpu(f"""<mujoco><option timestep="0.001"/><worldbody>{g}
    <body pos="0 0 {z0}"><freejoint/><geom {shape} mass="1.1" condim="3" friction="{mu} 0.005 0.0001"/></body></worldbody></mujoco>""", 6000)
    assert np.abs(qvel).max() < 3e-5
    assert abs((z0 - qpos[2]) / rest_depth(2.0 * ncon / (mu * mu * (1 + mu * mu))) - 1) < 3e-3          # the weight is split over ncon contacts of four rows each
    assert np.abs(qpos[:2]).max() < 1e-4 and abs(abs(qpos[3]) - 1) < 1e-6


@pytest.mark.parametrize("mu", [0.5, 1.0])
def test_gpu_friction_creep_and_noslip(mu):
    th = 0.15
    xml = """<mujoco><option timestep="0.001" gravity="{gx} 0 {gz}" noslip_iterations="{ns}" noslip_tolerance="1e-9"/><worldbody>
    <geom type="plane" size="2 2 0.1" condim="3" friction="{mu} 0.005 0.0001"/>
    <body pos="0 0 0.1"><joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 1 0"/><joint type="slide" axis="0 0 1"/>
    <geom type="sphere" size="0.1" mass="0.8" condim="3" friction="{mu} 0.005 0.0001"/></body></worldbody></mujoco>"""
    r0 = rest_depth(2.0 / (mu * mu * (1 + mu * mu)) / np.cos(th))
    d = impedance(r0)
    v_creep = G * np.sin(th) * (1 - d) * (1 + mu * mu) * 0.95 * 0.02 / (2 * d)
    for ns in (0, 30):
        qpos, qvel = _settle_on_gpu(xml.format(gx=G * np.sin(th), gz=-G * np.cos(th), ns=ns, mu=mu), 6000)
        assert abs(-qpos[2] / r0 - 1) < 3e-3
        if ns == 0:
            assert abs(qvel[0] / v_creep - 1) < 2e-3            # the regularised pyramid creeps at the closed-form speed
        else:
            assert abs(qvel[0]) < 2e-3 * v_creep               # the noslip pass holds the body


def test_gpu_joint_limit_rest_depth_with_custom_solref_solimp():
    """anchor 06 on the device: one limit row with the JOINT's solreflimit (dampratio 0.7) and solimplimit (power-3 impedance)"""
    xml = """<mujoco><option timestep="0.0005"/><worldbody>
    <body pos="0 0 1"><joint type="slide" axis="0 0 1" limited="true" range="0 1" solreflimit="0.01 0.7" solimplimit="0.8 0.99 0.002 0.3 3"/>
    <geom type="sphere" size="0.05" mass="4" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""
    qpos, qvel = _settle_on_gpu(xml, 8000)
    assert abs(qvel[0]) < 1e-5 and abs(-qpos[0] / rest_depth(1.0, (0.8, 0.99, 0.002, 0.3, 3.0), tc=0.01, dr=0.7) - 1) < 2e-3


def test_gpu_frictionloss_saturation_and_creep():
    """anchors 11 / 12 on the device: below the weight the friction-loss row saturates (free fall minus f / m), above it the mass creeps at
    |v| = g (1 - dmin) dmax tc / (2 dmin)"""
    qpos, qvel = _settle_on_gpu(SLIDER.format(fl=5.0, extra=""), 200)
    assert abs(qvel[0] / (-(G - 5.0 / 2) * 0.2) - 1) < 1e-4
    qpos, qvel = _settle_on_gpu(SLIDER.format(fl=50.0, extra=""), 3000)
    assert abs(-qvel[0] / (G * (1 - 0.9) * 0.95 * 0.02 / (2 * 0.9)) - 1) < 1e-3


def test_gpu_actuators_and_implicit_damping():
    """anchors 13 / 14 on the device: a position servo holds kp (c - q) = m g, its ctrlrange clamps the command; a general-affine actuator
    (gain ctrl + b0 + b1 q + b2 qdot, the Adroit hand's kind) drives a damped mass to (gain c + b0 - m g) / -b1"""
    act = '<actuator><position joint="j" kp="400" ctrllimited="true" ctrlrange="-1 1"/></actuator>'
    xml = SLIDER.replace('frictionloss="{fl}"', 'damping="30"').format(extra=act)
    qpos, qvel = _settle_on_gpu(xml, 6000, ctrl=[0.3])
    assert abs(qvel[0]) < 1e-5 and abs(qpos[0] - (0.3 - 2 * G / 400)) < 1e-5
    qpos, qvel = _settle_on_gpu(xml, 6000, ctrl=[5.0])
    assert abs(qpos[0] - (1.0 - 2 * G / 400)) < 1e-5
    act = '<actuator><general joint="j" gainprm="10 0 0" biastype="affine" biasprm="3 -100 -20"/></actuator>'
    qpos, qvel = _settle_on_gpu(SLIDER.format(fl=0, extra=act), 6000, ctrl=[0.7])
    assert abs(qvel[0]) < 1e-5 and abs(qpos[0] - (10 * 0.7 + 3 - 2 * G) / 100) < 1e-5


def test_gpu_weld_to_mocap_sags_by_the_soft_constraint_offset():
    """anchor 10 on the device: a free body welded to a mocap body hangs below it by the rest depth of one soft row (the Fetch gripper's weld)"""
    xml = """<mujoco><option timestep="0.001"/><worldbody>
    <body name="mocap" mocap="true" pos="0.3 0.2 1"/>
    <body name="b" pos="0.3 0.2 1"><freejoint/><geom type="box" size="0.05 0.04 0.03" mass="1.7" contype="0" conaffinity="0"/></body>
    </worldbody><equality><weld body1="mocap" body2="b" solref="0.02 1" solimp="0.9 0.95 0.001"/></equality></mujoco>"""
    qpos, qvel = _settle_on_gpu(xml, 12000)
    # the generic stepper carries no mocap state (the maze models have none): the mocap body sits at the origin, the weld drags the body there and it
    # comes to rest one soft-row depth below it
    assert np.abs(qvel).max() < 3e-5 and abs(-qpos[2] / rest_depth(1.0) - 1) < 3e-3
    assert np.allclose(qpos[[0, 1]], [0.0, 0.0], atol=1e-6) and np.allclose(np.abs(qpos[3]), 1, atol=1e-6)


@pytest.mark.parametrize("cd,mu,rows_over_2", [(4, 0.6, 3.0), (6, 0.7, 5.0)])
def test_gpu_condim4_and_condim6_rest_depth(cd, mu, rows_over_2):
    """anchors 03 / 03b on the device: 2 (condim - 1) pyramid rows that all carry the R of the first pair (Fetch finger pads, the kitchen's condim-6 pads)"""
    xml = SPHERE.format(cd=cd, mu=mu, spin=0.03, mass=0.9).replace("0.0001", "0.002")
    qpos, qvel = _settle_on_gpu(xml, 4000)
    assert np.abs(qvel).max() < 3e-5 and abs((0.1 - qpos[2]) / rest_depth(rows_over_2 / (mu * mu * (1 + mu * mu))) - 1) < 2e-3


def test_gpu_tendon_limit_rest_depth():
    """anchor 09 on the device: two masses coupled by the fixed tendon q1 + q2 >= 0 (the Shadow hand's coupled joints): J = (1, 1), invweight 2 / m"""
    xml = """<mujoco><option timestep="0.0005"/><worldbody>
    <body pos="0 0 1"><joint name="a" type="slide" axis="0 0 1"/><geom type="sphere" size="0.05" mass="1.5" contype="0" conaffinity="0"/></body>
    <body pos="1 0 1"><joint name="b" type="slide" axis="0 0 1"/><geom type="sphere" size="0.05" mass="1.5" contype="0" conaffinity="0"/></body>
    </worldbody><tendon><fixed name="t" limited="true" range="0 1"><joint joint="a" coef="1"/><joint joint="b" coef="1"/></fixed></tendon></mujoco>"""
    qpos, qvel = _settle_on_gpu(xml, 8000)
    assert np.abs(qvel).max() < 1e-5 and abs(qpos[0] - qpos[1]) < 1e-6
    assert abs(-(qpos[0] + qpos[1]) / rest_depth(0.5) - 1) < 2e-3


def test_gpu_joint_equality_couples_two_hinges():
    """anchor 21 on the device: <equality><joint polycoef="0 a"> (the kitchen's knob <-> burner couplings): the coupled hinge settles at q1 = a q2"""
    a = 3.0
    xml = f"""<mujoco><option timestep="0.002"/><worldbody>
    <body pos="0 0 0.2"><joint name="j1" type="hinge" axis="0 0 1" damping="0.05"/><geom type="box" size="0.1 0.02 0.02" pos="0.1 0 0" mass="0.5" contype="0" conaffinity="0"/></body>
    <body pos="0.5 0 0.2"><joint name="j2" type="hinge" axis="0 0 1" damping="0.05"/><geom type="box" size="0.05 0.02 0.02" pos="0.05 0 0" mass="0.2" contype="0" conaffinity="0"/></body>
    </worldbody>
    <equality><joint joint1="j1" joint2="j2" polycoef="0 {a} 0 0 0"/></equality>
    <actuator><position joint="j1" kp="50"/></actuator></mujoco>"""
    qpos, qvel = _settle_on_gpu(xml, 6000, ctrl=[0.6])
    assert np.abs(qvel).max() < 1e-4 and abs(qpos[0] - 0.6) < 1e-4 and abs(qpos[0] - a * qpos[1]) < 1e-4       # fp32 jitter floor of the lightly damped pair


def test_gpu_two_body_contact_conserves_momentum():
    """anchor 05 on the device: two free spheres collide in zero gravity; contact forces are internal, so m1 v1 + m2 v2 is conserved through the contact
    (fp32: to 1e-5 of the momentum) -- equal and opposite Jacobian rows on the two bodies"""
    xml = """<mujoco><option timestep="0.001" gravity="0 0 0"/><worldbody>
    <body pos="0 0 0"><freejoint/><geom type="sphere" size="0.1" mass="1" condim="1"/></body>
    <body pos="0.5 0.02 0.01"><freejoint/><geom type="sphere" size="0.15" mass="3" condim="1"/></body></worldbody></mujoco>"""
    q0 = np.array([0, 0, 0, 1, 0, 0, 0, 0.5, 0.02, 0.01, 1, 0, 0, 0.0])
    v0 = np.array([1.0, 0, 0, 0, 0, 0, -0.5, 0, 0, 0, 0, 0.0])
    p0 = 1.0 * v0[0:3] + 3.0 * v0[6:9]
    qpos, qvel = _settle_on_gpu(xml, 1000, state=(q0, v0))
    assert np.abs(1.0 * qvel[0:3] + 3.0 * qvel[6:9] - p0).max() < 2e-5
    assert qvel[0] < 0 < qvel[6] + 0.5                     # the light sphere bounced back


def test_gpu_rk4_and_euler_orders_of_convergence():
    """anchor 16 on the device (AntMaze integrates with RK4, ant.xml:3): pendulum angle after 0.5 s against the fp64 oracle at h = 1e-4.  Halving h halves the
    semi-implicit Euler error; the RK4 error at h = 0.02 is four orders below Euler's at h = 0.004 and falls by ~16 per halving until it meets the fp32 floor."""
    from oracle.oracle_sim import OracleSim
    from test_cpu_oracle_anchors import PEND, _compile

    ref_sim = OracleSim(_compile(PEND.format(h=1e-4, integ="RK4")))
    ref_sim.qpos[0] = 0.4
    ref_sim.step(5000)
    ref = float(ref_sim.qpos[0])

    def angle(integ, h):
        qpos, _ = _settle_on_gpu(PEND.format(h=h, integ=integ), int(round(0.5 / h)), state=([0.4], [0.0]))
        return qpos[0]

    e = [abs(angle("Euler", h) - ref) for h in (0.004, 0.002, 0.001)]
    assert 1.8 < e[0] / e[1] < 2.2 and 1.8 < e[1] / e[2] < 2.2, e
    r = [abs(angle("RK4", h) - ref) for h in (0.02, 0.01)]
    assert r[0] < 1e-3 * e[0] and r[0] / max(r[1], 1e-7) > 8 or r[0] < 2e-6, (r, e)


@pytest.mark.parametrize("support", ["box", "mesh"])
def test_gpu_hull_contact_distance(support):
    """anchor 22 on the device: a cube given as a MESH stands on one VERTEX (body diagonal vertical: the contact point is unique and under the centre of mass)
    in a slab (box primitive / second mesh) with a prescribed overlap, at rest.  The stepper does not hand out contacts, but ONE step does: a single frictionless
    row through the centre of mass gives a = d k d |r| - (1 - d) g, so the velocity after one step measures the distance the hull routine reported
    (dv / d|r| = h k d^2 = 2.5 per metre: a micrometre of distance is 2.5e-6 m/s, far above fp32).  (Face-on the contact point is any point of the overlap polygon
    and the cube also turns: nothing closed-form to compare with.)"""
    slab = '<geom name="slab" type="box" size="0.3 0.3 0.05" pos="0 0 0.05" condim="1"/>' if support == "box" else \
           '<geom name="slab" type="mesh" mesh="slab" pos="0 0 0.05" condim="1"/>'
    xml = f"""<mujoco><option timestep="0.001"/><asset><mesh name="cube" file="cube.stl"/><mesh name="slab" file="slab.stl"/></asset><worldbody>
    {slab}<body pos="0 0 0.3"><freejoint/><geom name="cube" type="mesh" mesh="cube" mass="0.7" condim="1"/></body></worldbody></mujoco>"""
    h, k = 0.001, stiffness()
    u = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)                    # the body diagonal ...
    axis = np.cross(u, [0.0, 0.0, -1.0]); ang = np.arccos(-u[2])     # ... turned onto -z
    axis /= np.linalg.norm(axis)
    quat = np.r_[np.cos(ang / 2), np.sin(ang / 2) * axis]
    for overlap in (2e-4, 1e-3, 3e-3):
        for (x, y) in ((0.0, 0.0), (0.11, -0.07)):
            q0 = np.r_[x, y, 0.1 + np.sqrt(3.0) * 0.05 - overlap, quat]
            qpos, qvel = _settle_on_gpu(xml, 1, state=(q0, [0.0] * 6), files=(("cube.stl", (0.05, 0.05, 0.05)), ("slab.stl", (0.3, 0.3, 0.05))))
            d = impedance(overlap)
            v = h * (d * k * d * overlap - (1 - d) * G)
            assert abs(qvel[2] - v) < 2.5 * 3e-6 + 1e-6, (overlap, x, y, qvel[2], v)                 # the reported distance is the overlap to within 3 micrometres
            assert np.abs(qvel[[0, 1]]).max() < 1e-5 and np.abs(qvel[3:]).max() < 2e-3               # through the centre of mass: (almost) no turn


@pytest.mark.parametrize("shape", ['type="sphere" size="0.07"', 'type="ellipsoid" size="0.07 0.07 0.07"'])
@pytest.mark.parametrize("other", ['type="box" size="0.3 0.3 0.05" pos="0 0 0.05"', 'type="cylinder" size="0.3 0.05" pos="0 0 0.05"'])
def test_gpu_convex_pair_distance_through_the_one_step_response(shape, other):
    """anchor 20 on the device: a ball -- as a sphere primitive (closed-form routines; against the cylinder the lane-resident portal search) and as an ellipsoid with
    three equal radii (portal search + the smooth-surface extent) -- pressed into a slab's flat top by a prescribed overlap, at rest, at several places on the
    slab.  The contact normal passes through the centre of mass, so one step gives v = h (d k d |r| - (1 - d) g): the velocity measures the reported distance."""
    xml = f"""<mujoco><option timestep="0.001"/><worldbody><geom {other} condim="1"/>
    <body pos="0 0 0.3"><freejoint/><geom {shape} mass="0.4" condim="1"/></body></worldbody></mujoco>"""
    h, k = 0.001, stiffness()
    q = np.array([0.8, 0.1, -0.5, 0.3]); q /= np.linalg.norm(q)
    for overlap in (2e-4, 1e-3, 3e-3):
        for (x, y) in ((0.0, 0.0), (0.11, -0.07), (-0.18, 0.05)):
            q0 = np.r_[x, y, 0.1 + 0.07 - overlap, q]
            qpos, qvel = _settle_on_gpu(xml, 1, state=(q0, [0.0] * 6))
            d = impedance(overlap)
            v = h * (d * k * d * overlap - (1 - d) * G)
            assert abs(qvel[2] - v) < 2.5 * 3e-6 + 1e-6, (overlap, x, y, qvel[2], v)                 # the reported distance is the overlap to within 3 micrometres
            assert np.abs(qvel[[0, 1]]).max() < 1e-4        # normal within 1e-2 rad of the cap's (a portal within 1e-6 of a radius-r surface pins its normal to sqrt(2 eps / r) = 5e-3)


_U = np.ones(3) / np.sqrt(3.0)
_AX = np.cross(_U, [0.0, 0.0, -1.0]) / np.linalg.norm(np.cross(_U, [0.0, 0.0, -1.0]))
_VERTEX_DOWN = np.r_[np.cos(np.arccos(-_U[2]) / 2), np.sin(np.arccos(-_U[2]) / 2) * _AX]      # turns a cube's body diagonal onto -z


@pytest.mark.parametrize("ground", ['type="plane" size="1 1 0.1" pos="0 0 0.1"', 'type="box" size="0.3 0.3 0.05" pos="0 0 0.05"', 'type="cylinder" size="0.3 0.05" pos="0 0 0.05"'])
@pytest.mark.parametrize("shape,quat,reach", [('type="box" size="0.05 0.05 0.05"', _VERTEX_DOWN, np.sqrt(3.0) * 0.05),
                                               ('type="capsule" size="0.04 0.1"', np.array([1.0, 0, 0, 0]), 0.14),
                                               ('type="mesh" mesh="cube"', _VERTEX_DOWN, np.sqrt(3.0) * 0.05)])
def test_gpu_single_contact_distance_of_the_analytic_and_hull_routines(ground, shape, quat, reach):
    """One contact through the centre of mass -- a box or a mesh cube standing on a vertex, a capsule standing on its cap -- on a plane and on a box slab, at
    several places and depths: plane-box, plane-capsule, plane-mesh, box-box, capsule-box and hull-box routines, and on a cylinder's cap the portal search on a lane
    (box-cylinder, capsule-cylinder) and over the wave (hull-cylinder).  One step from rest gives
    v = h (d k d |r| - (1 - d) g) for the overlap the routine reported (see test_gpu_hull_contact_distance)."""
    # the whole scene sits 1.9 m from the world origin (the Fetch table is at x = 1.3): the routines must work on differences of geom positions
    ox, oy = 1.7, -0.85
    xml = f"""<mujoco><option timestep="0.001"/><asset><mesh name="cube" file="cube.stl"/></asset><worldbody><body pos="{ox} {oy} 0"><geom {ground} condim="1"/></body>
    <body pos="0 0 0.4"><freejoint/><geom {shape} mass="0.6" condim="1"/></body></worldbody></mujoco>"""
    h, k = 0.001, stiffness()
    for overlap in (2e-4, 1e-3, 3e-3):
        for (x, y) in ((0.0, 0.0), (0.11, -0.07), (-0.18, 0.05)):
            q0 = np.r_[ox + x, oy + y, 0.1 + reach - overlap, quat]
            qpos, qvel = _settle_on_gpu(xml, 1, state=(q0, [0.0] * 6), files=(("cube.stl", (0.05, 0.05, 0.05)),))
            d = impedance(overlap)
            v = h * (d * k * d * overlap - (1 - d) * G)
            assert abs(qvel[2] - v) < 2.5 * 3e-6 + 1e-6, (overlap, x, y, qvel[2], v)
            assert np.abs(qvel[[0, 1]]).max() < 1e-4


@pytest.mark.parametrize("ground", ['type="plane" size="1 1 0.1" pos="0 0 0.1"', 'type="box" size="0.4 0.4 0.05" pos="0 0 0.05"'])
def test_gpu_off_centre_contact_turns_the_body_by_the_closed_form(ground):
    """One contact that does NOT pass through the centre of mass: a cube tilted so that one corner is lowest, pressed into the ground by a prescribed overlap, at
    rest.  The row sees A = 1/m + |r x n|^2 / I (cube: I = 2/3 m a^2, isotropic) and R = (1 - d)/d / m (diagApprox has the translational weight only), so
    f = (k d |r| + g) / (A + R); one step gives v = h (f/m - g) n and omega = h I^-1 (r x n) f, the latter in the BODY frame for a free joint.  Pins the
    rotational half of the contact Jacobian and the inertia on the device."""
    a, m_ = 0.05, 0.9
    xml = f"""<mujoco><option timestep="0.001"/><worldbody><geom {ground} condim="1"/>
    <body pos="0 0 0.4"><freejoint/><geom type="box" size="{a} {a} {a}" mass="{m_}" condim="1"/></body></worldbody></mujoco>"""
    h, k = 0.001, stiffness()

    def quat_mul(p, q):
        return np.array([p[0] * q[0] - p[1:] @ q[1:], *(p[0] * q[1:] + q[0] * p[1:] + np.cross(p[1:], q[1:]))])

    qx = np.array([np.cos(0.26), np.sin(0.26), 0, 0]); qy = np.array([np.cos(0.17), 0, np.sin(0.17), 0])
    quat = quat_mul(qy, qx)
    w, x, y, z = quat
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    corners = np.array([[sx, sy, sz] for sx in (-a, a) for sy in (-a, a) for sz in (-a, a)]) @ Rm.T
    low = corners[np.argmin(corners[:, 2])]
    assert np.sort(corners[:, 2])[1] - low[2] > 5e-3                      # one corner only within reach
    inertia = 2.0 / 3.0 * m_ * a * a
    n = np.array([0.0, 0.0, 1.0])
    rxn = np.cross(low, n)
    for overlap in (2e-4, 1e-3, 3e-3):
        q0 = np.r_[0.03, -0.02, 0.1 - low[2] - overlap, quat]
        qpos, qvel = _settle_on_gpu(xml, 1, state=(q0, [0.0] * 6))
        d = impedance(overlap)
        A, R = 1.0 / m_ + rxn @ rxn / inertia, (1 - d) / d / m_
        f = (k * d * overlap + G) / (A + R)
        v_z, om_local = h * (f / m_ - G), Rm.T @ (h * rxn * f / inertia)
        assert abs(qvel[2] - v_z) < 1e-5 and np.abs(qvel[[0, 1]]).max() < 1e-6, (overlap, qvel[:3], v_z)
        assert np.abs(qvel[3:] - om_local).max() < 2e-3 * np.abs(om_local).max() + 1e-5, (overlap, qvel[3:], om_local)


def test_gpu_double_pendulum_accelerations():
    """anchor 25 on the device: one semi-implicit Euler step of a two-hinge, two-point-mass chain from arbitrary states; (v' - v) / h is the joint acceleration
    and must equal the textbook equations of motion -- mass matrix with its off-diagonal coupling, centrifugal / Coriolis bias and gravity of the HIP kernels."""
    from test_cpu_oracle_anchors import DOUBLE_PENDULUM, double_pendulum_acc

    m1, m2, l1, l2, h = 0.7, 0.4, 0.5, 0.35, 0.001
    xml = DOUBLE_PENDULUM.format(m1=m1, m2=m2, l1=l1, l2=l2)
    rng = np.random.default_rng(2)
    for _ in range(8):
        q, v = rng.uniform(-2.5, 2.5, 2), rng.uniform(-6, 6, 2)
        v = np.float32(v).astype(np.float64)                         # the state the device really starts from
        q = np.float32(q).astype(np.float64)
        qpos, qvel = _settle_on_gpu(xml, 1, state=(q, v))
        acc = (qvel - v) / h
        a1, a2 = double_pendulum_acc(q[0], q[0] + q[1], v[0], v[0] + v[1], m1, m2, l1, l2)
        scale = max(abs(a1), abs(a2), 1.0)
        assert abs(acc[0] - a1) < 2e-3 * scale and abs(acc[0] + acc[1] - a2) < 2e-3 * scale, (q, v, acc, a1, a2)      # fp32: v ~ 6 resolves acc to 6e-7 / h = 6e-4


def test_gpu_torque_free_rotation_obeys_eulers_equations():
    """anchor 26 on the device: (v' - v) / h of a free box spinning about a non-principal axis in zero gravity = Euler's equations in the body frame"""
    from test_cpu_oracle_anchors import FREE_BOX, euler_equations_acc

    rng = np.random.default_rng(5)
    for _ in range(6):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w = np.float32(rng.uniform(-8, 8, 3)).astype(np.float64)
        v = np.r_[np.float32(rng.uniform(-1, 1, 3)).astype(np.float64), w]
        qpos, qvel = _settle_on_gpu(FREE_BOX, 1, state=(np.r_[0.1, -0.2, 1.0, q], v))
        acc = (qvel - v) / 0.001
        want = euler_equations_acc(w)
        assert np.abs(acc[:3]).max() < 1e-3 and np.abs(acc[3:] - want).max() < 2e-3 * max(np.abs(want).max(), 1.0), (w, acc, want)


def test_gpu_armature_gear_and_damping_of_a_driven_hinge():
    """anchor 27 on the device: one Euler step of a hinge with rotor inertia, a geared motor and damping.  The step integrates the damping implicitly:
    v' = v + h (gear u - c v - m g l sin q) / (I + h c) with I = m l^2 + armature (anchor 15's update), which is what (v' - v) / h is compared with."""
    from test_cpu_oracle_anchors import ARMATURE

    m, l, arm, gear, damp, h = 0.6, 0.4, 0.02, 3.0, 0.15, 0.001
    xml = ARMATURE.format(m=m, l=l, arm=arm, gear=gear, damp=damp)
    rng = np.random.default_rng(3)
    inertia = m * l * l + 0.4 * m * 0.002 ** 2 + arm
    for _ in range(6):
        q, w, u = (float(np.float32(x)) for x in (rng.uniform(-3, 3), rng.uniform(-5, 5), rng.uniform(-1, 1)))
        qpos, qvel = _settle_on_gpu(xml, 1, state=([q], [w]), ctrl=[u])
        acc = (qvel[0] - w) / h
        want = (gear * u - damp * w - m * G * l * np.sin(q)) / (inertia + h * damp)
        assert abs(acc - want) < 2e-3 * max(abs(want), 1.0), (q, w, u, acc, want)


@pytest.mark.parametrize("ground", ['type="plane" size="1 1 0.1" pos="0 0 0.1"', 'type="box" size="0.4 0.4 0.05" pos="0 0 0.05"'])
@pytest.mark.parametrize("shape,ncon,reach", [('type="box" size="0.1 0.07 0.05"', 4, 0.05), ('type="capsule" size="0.04 0.12" euler="0 90 0"', 2, 0.04)])
def test_gpu_symmetric_multi_contact_response(ground, shape, ncon, reach):
    """A box lying face-on (four corner contacts) and a capsule lying on its side (two), centred, pressed in by a prescribed overlap, at rest: by symmetry the
    body only moves vertically and every contact sees the same acceleration, so with D = m d / (1 - d) per frictionless row
    a = (ncon d/(1-d) k d |r| - g) / (1 + ncon d/(1-d)).  One step measures it: all ncon contacts found, at the same depth, none doubled."""
    xml = f"""<mujoco><option timestep="0.001"/><worldbody><geom {ground} condim="1"/>
    <body pos="0 0 0.4"><freejoint/><geom {shape} mass="1.1" condim="1"/></body></worldbody></mujoco>"""
    h, k = 0.001, stiffness()
    for overlap in (2e-4, 1e-3, 3e-3):
        qpos, qvel = _settle_on_gpu(xml, 1, state=([0.0, 0.0, 0.1 + reach - overlap, 1, 0, 0, 0], [0.0] * 6))
        d = impedance(overlap)
        w = ncon * d / (1 - d)
        v = h * (w * k * d * overlap - G) / (1 + w)
        assert abs(qvel[2] - v) < 1e-5 and np.abs(qvel[[0, 1, 3, 4, 5]]).max() < 2e-4, (overlap, qvel, v)


@pytest.mark.parametrize("ground", ['type="box" size="0.3 0.3 0.05" pos="0 0 0.05"', 'type="cylinder" size="0.3 0.05" pos="0 0 0.05"'])
def test_gpu_standing_cylinder_on_a_flat_top(ground):
    """FetchSlide's puck configuration: a cylinder standing flat on a box (and on another cylinder's cap).  The convex route gives ONE contact; centred on the
    slab the centre ray is the common axis, so the contact passes through the centre of mass and one step gives v = h (d k d |r| - (1 - d) g) for the overlap."""
    xml = f"""<mujoco><option timestep="0.001"/><worldbody><geom {ground} condim="1"/>
    <body pos="0 0 0.4"><freejoint/><geom type="cylinder" size="0.04 0.03" mass="0.5" condim="1"/></body></worldbody></mujoco>"""
    h, k = 0.001, stiffness()
    for overlap in (2e-4, 1e-3, 3e-3):
        qpos, qvel = _settle_on_gpu(xml, 1, state=([0.0, 0.0, 0.1 + 0.03 - overlap, 1, 0, 0, 0], [0.0] * 6))
        d = impedance(overlap)
        v = h * (d * k * d * overlap - (1 - d) * G)
        assert abs(qvel[2] - v) < 2.5 * 3e-6 + 1e-6 and np.abs(qvel[[0, 1]]).max() < 1e-4 and np.abs(qvel[3:]).max() < 5e-2, (overlap, qvel, v)


@pytest.mark.parametrize("mu", [0.5, 1.0])
def test_gpu_slow_sliding_is_viscous_with_the_solref_damping(mu):
    """anchor 28 on the device: a point mass at its rest depth sliding slowly decelerates with a_x = -b v d / ((1 - d)(1 + mu^2) + d): the velocity term of the
    friction rows, the pyramid's R scaling and the impedance in one number"""
    from test_cpu_oracle_anchors import POINT_ON_PLANE, sliding_deceleration

    r0 = rest_depth(2.0 / (mu * mu * (1 + mu * mu)))
    for v in (1e-3, 5e-3):
        v32 = float(np.float32(v))
        qpos, qvel = _settle_on_gpu(POINT_ON_PLANE.format(mu=mu), 1, state=([0.0, 0.0, -r0], [v32, 0.0, 0.0]))
        acc = (qvel[0] - v32) / 0.001
        assert abs(acc / sliding_deceleration(v32, mu, impedance(r0)) - 1) < 5e-3 and abs(qvel[1]) < 1e-8, (v, acc)


@pytest.mark.parametrize("mu", [0.5, 1.0])
def test_gpu_fast_sliding_switches_one_pyramid_row_off(mu):
    """anchor 29 on the device: the active-set side of the Newton solver -- at 0.3 m/s ONE pyramid row is active, the other three are off; the accelerations follow from that row in closed form"""
    from test_cpu_oracle_anchors import POINT_ON_PLANE, fast_sliding_acc

    r0 = rest_depth(2.0 / (mu * mu * (1 + mu * mu)))
    v32 = float(np.float32(0.3))
    qpos, qvel = _settle_on_gpu(POINT_ON_PLANE.format(mu=mu), 1, state=([0.0, 0.0, -r0], [v32, 0.0, 0.0]))
    ax, az = fast_sliding_acc(v32, mu, r0)
    assert abs((qvel[0] - v32) / 0.001 / ax - 1) < 5e-3 and abs(qvel[2] / 0.001 / az - 1) < 5e-3 and abs(qvel[1]) < 1e-7, (qvel, ax, az)


def test_gpu_torsional_friction_of_condim4():
    """anchor 30 on the device: a ball at its condim-4 rest depth spinning slowly about the vertical is braked by the torsional pyramid pair in closed form"""
    from test_cpu_oracle_anchors import spin_deceleration

    mu, mu_t, m_, rad = 0.6, 0.02, 0.8, 0.1
    r0 = rest_depth(3.0 / (mu * mu * (1 + mu * mu)))
    for w in (0.01, 0.05):          # b mu_t w below the stiffness term: both torsional rows stay on
        w32 = float(np.float32(w))
        qpos, qvel = _settle_on_gpu(SPHERE.format(cd=4, mu=mu, spin=mu_t, mass=m_), 1, state=([0, 0, 0.1 - r0, 1, 0, 0, 0], [0, 0, 0, 0, 0, w32]))
        alpha = (qvel[5] - w32) / 0.001
        assert abs(alpha / spin_deceleration(w32, mu, mu_t, r0, m_, rad) - 1) < 1e-2 and np.abs(qvel[:2]).max() < 1e-6 and np.abs(qvel[3:5]).max() < 1e-5, (w, alpha, qvel)


def test_gpu_more_than_32_contacts_equal_the_oracle(tmp_path):
    """The reference never truncates a contact list (mujoco.mj_step, envs/robot_env.py:340-341; the kitchen scene of franka_env.py:92-110 has 258 geoms).  The large
    tables of the overflow lane hold 64 contacts (one lane each): nine boxes sliding on the floor are 36 contacts / 144 pyramid rows, and the HIP kernel on those
    tables follows the oracle (MAXCON 128), contact list for contact list, step by step; on 32-contact tables the same state raises the overflow flag that sends a
    world of a fast kernel to the large ones."""
    import torch

    from gymnasium_robotics_amd import _native
    from gymnasium_robotics_amd.mjcf import compile_mjcf
    from oracle.oracle_sim import OracleSim

    bodies = "".join(f'<body pos="{0.25 * (i % 4):.2f} {0.25 * (i // 4):.2f} 0.0405"><freejoint/><geom type="box" size="0.1 0.08 0.04" mass="{0.3 + 0.05 * i:.2f}"/></body>' for i in range(9))
    path = os.path.join(tmp_path, "many.xml")
    with open(path, "w") as f:
        f.write(f'<mujoco><option timestep="0.002"/><worldbody><geom name="floor" type="plane" size="3 3 0.1"/>{bodies}</worldbody></mujoco>')
    big = compile_mjcf(path, capacity={"maxcon": 64, "maxefc": 256, "jpool": 4080})
    s = OracleSim(big)
    s.qvel[:] = 0.2 * np.random.default_rng(5).standard_normal(s.qvel.shape)
    L, dev = _native.lib(), torch.device("cuda:0")
    nq, nv = big.dim("nq"), big.dim("nv")

    def make(model):
        H, I, F = model.pack()
        h = ctypes.c_void_p()
        _native.check(L.grx_model_create(H.ctypes.data, H.size, I.ctypes.data, I.size, F.ctypes.data, F.size, 0, ctypes.byref(h)))
        return h

    def gpu_step(h, qpos, qvel, ws):
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)[None]).to(dev)
        z = lambda *sh, dtype=torch.float32: torch.zeros(*sh, dtype=dtype, device=dev)
        bufs = dict(qpos=f32(qpos), qvel=f32(qvel), qacc_ws=f32(ws), goal=z(1, 2), action=z(1, 1), obs=z(1, nq + nv), achieved=z(1, 2), reward=z(1), success=z(1, dtype=torch.uint8),
                    terminated=z(1, dtype=torch.uint8), status=z(1, dtype=torch.int32))
        b = _native.PointBuffersStruct()
        for k, t in bufs.items():
            setattr(b, k, t.data_ptr())
        b.mask = b.packed = None
        task = _native.PointTaskStruct(1, 1, 1, 1, 0.45, 5.0)
        _native.check(L.grx_point_step(h, ctypes.byref(task), ctypes.byref(b), 1, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        torch.cuda.synchronize()
        return bufs["qpos"][0].cpu().numpy().astype(np.float64), bufs["qvel"][0].cpu().numpy().astype(np.float64), int(bufs["status"][0])

    h_big, h_small = make(big), make(big.with_capacity(maxcon=32, maxefc=256, jpool=4080))
    try:
        worst, most = 0.0, 0
        for t in range(40):
            q0, v0, w0 = s.qpos.copy(), s.qvel.copy(), s.qacc_warmstart.copy()
            s.step(1)
            most = max(most, s.ncon)
            qg, vg, st = gpu_step(h_big, q0, v0, w0)
            assert st & 0xFFFF == 0, (t, st)
            worst = max(worst, np.abs(qg - s.qpos).max(), np.abs(vg - s.qvel).max())
        assert most == 36 and worst < 1e-4, (most, worst)
        _, _, st = gpu_step(h_small, q0, v0, w0)
        assert st & 2      # GRX_ST_CON_OVERFLOW
    finally:
        L.grx_model_destroy(h_big); L.grx_model_destroy(h_small)
