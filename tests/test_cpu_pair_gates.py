"""Joint-box gates of the hull pairs (gymnasium_robotics_amd/mjcf/pair_gates.py): the packaged Fetch models carry them, and the checker -- which does not know
about gates and sends every candidate through the bounding-box filter and the portal routine -- finds NO contact for a gated pair at joint values inside its box.
The engine-source side (the emulator runs the gated sweep, the oracle the ungated one) is covered by the teacher-forced fixtures of test_cpu_engine_emu.py."""
import numpy as np
import pytest

from gymnasium_robotics_amd.envs.fetch import load_fetch_model
from oracle.oracle_sim import OracleSim


@pytest.mark.parametrize("task", ["FetchPickAndPlace", "FetchReach"])
def test_packaged_fetch_models_carry_gates(task):
    m = load_fetch_model(task)
    T = m.tables
    gate = np.asarray(T["devpair_gate"]).ravel()
    assert gate.size == np.asarray(T["devpair"]).size and (gate >= 0).sum() >= 30
    qa, box = np.asarray(T["gate_qadr"]).reshape(-1, 3), np.asarray(T["gate_box"]).reshape(-1, 3, 2)
    assert len(qa) == len(box) == gate.max() + 1
    g = m.names["geom"]
    pairs = {(int(T["pair_geom1"][p]), int(T["pair_geom2"][p])): int(gate[k]) for k, p in enumerate(np.asarray(T["devpair"]).ravel())}
    gi = pairs[(g["robot0:torso_lift_link"], g["robot0:shoulder_lift_link"])]      # the pair that passed the bounding-box filter in every substep of every world
    assert gi >= 0
    jq = {int(np.asarray(T["jnt_qposadr"]).ravel()[j]): n for n, j in m.names["joint"].items()}
    named = {jq[int(a)]: tuple(b) for a, b in zip(qa[gi], box[gi]) if a >= 0}
    lo, hi = named["robot0:shoulder_pan_joint"]
    assert lo < -0.9 and hi > 0.8          # covers the pan angles the tasks reach (|pan| < 0.8 in random-action rollouts)
    assert all(b[0] < b[1] for b in box.reshape(-1, 2) if b[0] > -1e29)


def test_no_contact_inside_a_gate_box():
    m = load_fetch_model("FetchPickAndPlace")
    T = m.tables
    sim = OracleSim(m)
    rng = np.random.default_rng(0)
    gate = np.asarray(T["devpair_gate"]).ravel()
    qa, box = np.asarray(T["gate_qadr"]).reshape(-1, 3), np.asarray(T["gate_box"]).reshape(-1, 3, 2)
    dp = np.asarray(T["devpair"]).ravel()
    jr, jl, jq, jt = np.asarray(T["jnt_range"]).reshape(-1, 2), np.asarray(T["jnt_limited"]).ravel(), np.asarray(T["jnt_qposadr"]).ravel(), np.asarray(T["jnt_type"]).ravel()
    q0 = np.asarray(T["qpos0"]).ravel().copy()
    hits = tested = 0
    for trial in range(400):
        q = q0.copy()
        for j in range(len(jt)):                    # every hinge / slide joint anywhere in its range (unlimited hinges: a full turn)
            if jt[j] == 3:
                q[jq[j]] = rng.uniform(*(jr[j] if jl[j] else (-np.pi, np.pi)))
            elif jt[j] == 2 and jl[j]:
                q[jq[j]] = rng.uniform(*jr[j])
        sim.qpos[:] = q
        sim.forward()
        con = sim.contacts()
        touching = {(int(c[7]), int(c[8])) for c in con}
        for k, p in enumerate(dp):
            if gate[k] < 0:
                continue
            inside = all(a < 0 or (b[0] < q[a] < b[1]) for a, b in zip(qa[gate[k]], box[gate[k]]))
            if inside:
                tested += 1
                hits += (int(T["pair_geom1"][p]), int(T["pair_geom2"][p])) in touching
    assert tested > 5000 and hits == 0


def test_gate_prover_on_a_hinged_mesh_cube_with_a_known_collision_angle():
    """compute_pair_gates from first principles: a 10 cm mesh cube orbits a hinge at radius 0.3 m; a second, world-fixed mesh cube sits on the orbit at 40 degrees.
    Elementary geometry: the two cubes cannot touch while the hinge angle is below ~0.36 rad (centre distance 2 r sin(delta / 2) must undercut the largest possible
    extent sum, 2 x 0.0707); they do touch at 0.698 rad.  The proven box must contain the reference pose, reach a useful part of the clear range, stay inside it --
    and the checker must find no contact anywhere inside it, but one beyond."""
    import os
    import tempfile

    from gymnasium_robotics_amd.mjcf import compile_mjcf
    from gymnasium_robotics_amd.mjcf.pair_gates import compute_pair_gates
    from tests.test_cpu_oracle_anchors import _write_cube_stl

    ang = np.deg2rad(40.0)
    with tempfile.TemporaryDirectory() as d:
        _write_cube_stl(os.path.join(d, "cube.stl"), (0.05, 0.05, 0.05))
        xml = f"""<mujoco><option timestep="0.002" gravity="0 0 0"/><asset><mesh name="cube" file="cube.stl"/></asset><worldbody>
        <geom name="post" type="mesh" mesh="cube" pos="{0.3 * np.cos(ang)} {0.3 * np.sin(ang)} 0" euler="0 0 40"/>
        <body pos="0 0 0"><joint name="swing" type="hinge" axis="0 0 1" limited="true" range="-115 115"/><geom name="rider" type="mesh" mesh="cube" pos="0.3 0 0" mass="1"/></body>
        </worldbody></mujoco>"""
        p = os.path.join(d, "m.xml")
        with open(p, "w") as f:
            f.write(xml)
        m = compile_mjcf(p)
    T = m.tables
    gate, qa, box, rep = compute_pair_gates(dict(T), cells2=128)
    assert (gate >= 0).sum() == 1 and len(qa) == 3 and qa[0] == 0 and qa[1] == qa[2] == -1
    lo, hi = box[0], box[1]
    assert lo < -1.5 and 0.2 < hi < 0.55                      # (the range is +-115 degrees) clear all the way down to the limit on the far side; on the near side useful but not beyond the true onset
    sim = OracleSim(m)

    def touching(q):
        sim.qpos[0] = q
        sim.forward()
        return sim.ncon > 0

    assert not any(touching(q) for q in np.linspace(lo + 1e-6, hi - 1e-6, 400))
    assert touching(ang) and touching(ang - 0.2)              # the pair does collide, beyond the box


def test_gated_engine_source_finds_the_ungated_checker_contact_set(fetch_models):
    """The device engine source (lane emulator: joint-box gates evaluated, gated candidates dropped from the sweep) against the checker (no gates: every candidate goes through the
    filter and the portal routine) over random arm configurations ANYWHERE in the joint ranges -- inside the boxes, outside them, next to their faces: the set of geom pairs in contact
    must be the same, configuration by configuration.  (Half of the samples hug the faces of the torso / shoulder gate: shoulder_pan at a face +- 0.02 rad.)"""
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.fetch_spec import make_fetch_task

    m = fetch_models["FetchPickAndPlace"].copy()
    m.tables["eq_data"][:, :7] = [0, 0, 0, 0, 0, 0, 1]
    T = m.tables
    emu, sim = EmuSim(m, make_fetch_task(m, "FetchPickAndPlace")), OracleSim(m)
    rng = np.random.default_rng(1)
    jr, jl, jq, jt = np.asarray(T["jnt_range"]).reshape(-1, 2), np.asarray(T["jnt_limited"]).ravel(), np.asarray(T["jnt_qposadr"]).ravel(), np.asarray(T["jnt_type"]).ravel()
    q0 = np.asarray(T["qpos0"]).ravel().copy()
    names = m.names["joint"]
    pan = int(jq[names["robot0:shoulder_pan_joint"]])
    g = m.names["geom"]
    gate = np.asarray(T["devpair_gate"]).ravel()
    dp = np.asarray(T["devpair"]).ravel()
    k33 = next(k for k, p in enumerate(dp) if (int(T["pair_geom1"][p]), int(T["pair_geom2"][p])) == (g["robot0:torso_lift_link"], g["robot0:shoulder_lift_link"]))
    qa, box = np.asarray(T["gate_qadr"]).reshape(-1, 3)[gate[k33]], np.asarray(T["gate_box"]).reshape(-1, 3, 2)[gate[k33]]
    faces = [b for a, b in zip(qa, box) if a == pan][0]
    with_contacts = differing_pan_side = 0
    for trial in range(120):
        q = q0.copy()
        for j in range(len(jt)):
            if jt[j] == 3:
                q[jq[j]] = rng.uniform(*(jr[j] if jl[j] else (-np.pi, np.pi)))
            elif jt[j] == 2 and jl[j]:
                q[jq[j]] = rng.uniform(*jr[j])
        if trial % 2:
            q[pan] = faces[trial // 2 % 2] + rng.uniform(-0.02, 0.02)
            differing_pan_side += 1
        sim.qpos[:] = q; sim.qvel[:] = 0
        sim.forward()
        want = sorted((int(c[7]), int(c[8])) for c in sim.contacts())
        emu.qpos[:] = q.astype(np.float32); emu.qvel[:] = 0; emu.qacc_ws[:] = 0
        emu.mocap[:] = np.concatenate([T["mocap_pos0"].ravel(), T["mocap_quat0"].ravel()]).astype(np.float32)
        ncon, _ = emu.physics_steps(1)
        pairs = emu.ctx("con_pair", max(ncon, 1), np.int32)[:ncon]
        got = sorted((int(T["pair_geom1"][p]), int(T["pair_geom2"][p])) for p in pairs)
        assert got == want, (trial, got, want)
        with_contacts += len(want) > 0
    assert with_contacts > 60 and differing_pan_side == 60
