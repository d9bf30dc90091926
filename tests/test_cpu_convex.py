"""General convex narrow phase (Minkowski Portal Refinement + analytic normals for smooth geoms): known answers for the oracle,
and the device engine source (through the lane emulator) against the oracle's golden fixtures of the families that need it
(HandManipulateEgg*: ellipsoid vs capsules / boxes; FetchSlide: cylinder vs box is covered in test_cpu_engine_emu.py)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(__file__)


def _contacts(tmp_path, geom_a, geom_b, pos_b, quat_b="1 0 0 0"):
    from gymnasium_robotics_amd.mjcf.compiler import compile_mjcf
    from oracle.oracle_sim import OracleSim

    xml = f"""<mujoco><option timestep="0.002"/><worldbody>
      <body name="a" pos="0 0 0"><geom name="ga" {geom_a}/></body>
      <body name="b" pos="{pos_b}" quat="{quat_b}"><joint type="free"/><geom name="gb" {geom_b}/></body>
    </worldbody></mujoco>"""
    path = os.path.join(tmp_path, "pair.xml")
    with open(path, "w") as f:
        f.write(xml)
    sim = OracleSim(compile_mjcf(path))
    sim.forward()
    return sim.contacts()      # rows: dist, pos[3], normal[3], ...


def test_oracle_convex_known_answers(tmp_path):
    # two ellipsoids that are spheres: the sphere-sphere answer (dist = |c| - r1 - r2, normal along the centres, geom1 -> geom2)
    c = np.array([0.1, 0.08, 0.03])
    r = _contacts(tmp_path, 'type="ellipsoid" size="0.1 0.1 0.1"', 'type="ellipsoid" size="0.05 0.05 0.05"', "0.1 0.08 0.03")
    assert r.shape[0] == 1
    assert abs(r[0, 0] - (np.linalg.norm(c) - 0.15)) < 1e-9 and np.abs(r[0, 4:7] - c / np.linalg.norm(c)).max() < 1e-9
    assert np.abs(r[0, 1:4] - c / np.linalg.norm(c) * (0.1 + 0.5 * r[0, 0])).max() < 1e-9      # midway between the two surfaces
    # the same pair apart by more than the radii: no contact
    assert _contacts(tmp_path, 'type="ellipsoid" size="0.1 0.1 0.1"', 'type="ellipsoid" size="0.05 0.05 0.05"', "0.2 0 0").shape[0] == 0
    # spherical ellipsoid 1 cm into the top face of a box (ellipsoid is geom 1: type order): depth and normal of the face
    r = _contacts(tmp_path, 'type="box" size="0.5 0.5 0.05"', 'type="ellipsoid" size="0.1 0.1 0.1"', "0.02 0.03 0.14")
    assert r.shape[0] == 1 and abs(r[0, 0] + 0.01) < 1e-6 and np.abs(r[0, 4:7] - [0, 0, -1]).max() < 2e-3      # normal = ellipsoid gradient at the (portal) contact position
    # egg-shaped ellipsoid beside a capsule: closest approach along y, 5 mm deep; normal from the capsule (geom 1) to the egg
    r = _contacts(tmp_path, 'type="capsule" size="0.01 0.02"', 'type="ellipsoid" size="0.03 0.03 0.04"', "0.0 0.035 0.0")
    assert r.shape[0] == 1 and abs(r[0, 0] + 0.005) < 1e-6 and np.abs(r[0, 4:7] - [0, 1, 0]).max() < 1e-6
    # cylinder standing 1 mm inside a box: one contact, face normal, exact depth; position somewhere inside the cap
    r = _contacts(tmp_path, 'type="box" size="0.5 0.5 0.05"', 'type="cylinder" size="0.025 0.02"', "0.1 0.0 0.069")
    assert r.shape[0] == 1 and abs(r[0, 0] + 0.001) < 1e-6 and np.abs(r[0, 4:7] - [0, 0, -1]).max() < 1e-6
    assert np.hypot(r[0, 1] - 0.1, r[0, 2]) <= 0.025 + 1e-6 and abs(r[0, 3] - 0.0495) < 1e-3
    # a tilted egg on a plane: the analytic deepest point
    r = _contacts(tmp_path, 'type="plane" size="1 1 1"', 'type="ellipsoid" size="0.03 0.03 0.04"', "0 0 0.03", "0.9238795 0.3826834 0 0")
    hz = np.sqrt((0.03 * np.sin(np.pi / 4)) ** 2 + (0.04 * np.cos(np.pi / 4)) ** 2)      # extent of the rotated ellipsoid along z
    assert r.shape[0] == 1 and abs(r[0, 0] - (0.03 - hz)) < 1e-9 and np.abs(r[0, 4:7] - [0, 0, 1]).max() < 1e-12


def test_oracle_plane_cylinder(tmp_path):
    # upright cylinder 2 mm into the plane: the near rim point + two more corners of the inscribed triangle, all at the cap's depth
    r = _contacts(tmp_path, 'type="plane" size="1 1 1"', 'type="cylinder" size="0.025 0.02"', "0.3 0.1 0.018")
    assert r.shape[0] == 3 and np.abs(r[:, 0] + 0.002).max() < 1e-12 and np.abs(r[:, 4:7] - [0, 0, 1]).max() < 1e-12
    rim = np.hypot(r[:, 1] - 0.3, r[:, 2] - 0.1)
    assert np.abs(rim - 0.025).max() < 1e-9 and np.abs(r[:, 3] - (-0.001)).max() < 1e-12
    cen = r[:, 1:3].mean(axis=0)
    assert np.abs(cen - [0.3, 0.1]).max() < 1e-9                                   # equilateral: the three points balance the cap
    # lying on its side (axis along y), 1 mm deep: the two end points of the lowest generator
    r = _contacts(tmp_path, 'type="plane" size="1 1 1"', 'type="cylinder" size="0.025 0.02"', "0 0 0.024", "0.7071068 0.7071068 0 0")
    assert r.shape[0] == 2 and np.abs(r[:, 0] + 0.001).max() < 1e-6 and np.abs(np.sort(r[:, 2]) - [-0.02, 0.02]).max() < 1e-6
    # tilted 30 degrees: one rim point, the deepest one
    r = _contacts(tmp_path, 'type="plane" size="1 1 1"', 'type="cylinder" size="0.025 0.02"', "0 0 0.029", "0.9659258 0.258819 0 0")
    lowest = 0.029 - (0.025 * np.sin(np.pi / 6) + 0.02 * np.cos(np.pi / 6))
    assert r.shape[0] == 1 and abs(r[0, 0] - lowest) < 1e-6
    # above the margin: nothing
    assert _contacts(tmp_path, 'type="plane" size="1 1 1"', 'type="cylinder" size="0.025 0.02"', "0 0 0.03").shape[0] == 0


def test_oracle_smooth_pair_depth_is_the_extent_along_the_normal(tmp_path):
    """Random egg-capsule poses: dist == centre distance along n - extents along n, and n is the average analytic normal at pos."""
    rng = np.random.default_rng(0)
    seen = 0
    for _ in range(25):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        pos = d * rng.uniform(0.03, 0.047)
        r = _contacts(tmp_path, 'type="capsule" size="0.01 0.0225"', 'type="ellipsoid" size="0.03 0.03 0.04"',
                      " ".join(map(str, pos)), " ".join(map(str, q)))
        if r.shape[0] == 0:
            continue
        seen += 1
        n = r[0, 4:7]
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        ext_cap = 0.01 + 0.0225 * abs(n[2])
        ext_egg = np.linalg.norm(np.array([0.03, 0.03, 0.04]) * (R.T @ n))
        assert abs(r[0, 0] - (pos @ n - ext_cap - ext_egg)) < 1e-9
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and r[0, 0] < 0
    assert seen >= 10


def _emu_hand(obj, golden, touch="off"):
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.hand import load_hand_block_model
    from gymnasium_robotics_amd.envs.manipulate_spec import make_block_task

    model = load_hand_block_model(touch=touch != "off", obj=obj)
    g = np.load(os.path.join(HERE, "golden", golden))
    return model, g, EmuSim(model, make_block_task(model, "ignore" if "Rotate" in golden else "random", "xyz", "sparse", touch, obj=obj))


def test_emulated_egg_step_matches_golden():
    model, g, emu = _emu_hand("egg", "hand_EggRotate_teacher.npz")
    T = model.tables
    assert 4 in T["geom_type"].tolist() and (model.dim("nq"), model.dim("nv")) == (31, 30)
    assert g["ncon"].max() >= 4 and (g["ncon"] > 0).mean() > 0.9          # contact-rich fixture
    pos_err, vel_err = [], []
    for i in range(0, g["obs"].shape[0], 2):
        emu.load_world(g, i, ("qpos", "qvel", "qacc_ws"))
        emu.hand_step(g["action"][i])
        assert emu.status.value == 0
        e = np.abs(emu.hand_obs[:61] - g["obs"][i])
        pe, ve = max(e[:24].max(), e[54:].max()), e[24:54].max()
        pos_err.append(pe); vel_err.append(ve)
        assert (pe < 1e-4 and ve < 1e-4) if g["activation_gap"][i] >= 1e-6 else (pe < 2e-2 and ve < 2.0), (i, pe, ve, g["activation_gap"][i])
    # With the portal search in fp64 (round 4) every well-posed snapshot meets 1e-4 (above); over the whole fixture (HandEgg row of tests/golden/tolerance_table.json:
    # positions p99 1.1e-6, velocities p99 5.5e-5, one snapshot 5e-7 m from a contact switch at 9.9e-2) the bulk sits at rounding level:
    pos_err, vel_err = np.array(pos_err), np.array(vel_err)
    assert np.median(pos_err) < 1e-6 and np.quantile(pos_err, 0.9) < 1e-5 and np.median(vel_err) < 5e-5 and np.quantile(vel_err, 0.9) < 1e-4, (
        np.median(pos_err), np.quantile(pos_err, 0.9), np.median(vel_err), np.quantile(vel_err, 0.9))


def test_emulated_egg_touch_matches_golden():
    model, g, emu = _emu_hand("egg", "hand_Egg_touch_teacher.npz", touch="sensordata")
    assert len(model.tables["touch_body"]) == 92
    rel, bits, same = [], [], 0
    for i in range(g["obs"].shape[0]):
        emu.load_world(g, i, ("qpos", "qvel", "qacc_ws"))
        emu.hand_step(g["action"][i])
        assert emu.status.value == 0
        touch, ref = emu.hand_obs[61:153], g["obs"][i][61:]
        rel.append(np.abs(touch - ref).max() / max(1.0, ref.max()))
        same += np.array_equal(touch > 0, ref > 0)
        bits.append(np.mean((touch > 0) == (ref > 0)))
    # measured: the on / off pattern of the 92 sensors equals the oracle's in 114 of 120 snapshots, 99.90 % of all sensor bits agree (worst snapshot:
    # 90 of 92), relative reading error p50 3.5e-5 / p75 9e-4 (a contact that switches sides of a zone boundary moves a whole reading: max 0.36)
    assert same >= 112 and np.mean(bits) > 0.998 and min(bits) >= 88 / 92, (same, np.mean(bits), min(bits))
    assert np.median(rel) < 1e-4 and np.quantile(rel, 0.75) < 2e-3, (np.median(rel), np.quantile(rel, 0.75))
