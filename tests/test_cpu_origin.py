"""Workspace-centred world frames (compile_mjcf(origin=...), include/grx_model.h GRX_ORIGIN_*).

A model compiled with an origin lives in the MJCF's world frame translated by -origin, so that the fp32 positions of the workspace are small numbers.  Pinned here, on the CPU:
  * the translation touches exactly the tables that hold an absolute position, and CompiledModel.in_mjcf_frame() undoes it (against a compilation with origin 0, when the
    reference's assets are mounted);
  * the oracle -- which always runs the model in the MJCF's own frame (oracle/oracle_sim.py) -- reproduces the fixtures recorded BEFORE the hand models were re-centred, from the
    packaged re-centred blobs: the round trip loses nothing;
  * state rows <-> world frame conversions (the free-joint translation only), and the emulated kernel reports world-frame observations.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
ASSETS = "/root/reference/gymnasium_robotics/envs/assets"


def _hand(obj="block", touch=False):
    from gymnasium_robotics_amd.envs.hand import load_hand_block_model

    return load_hand_block_model(None, touch=touch, obj=obj)


def test_packaged_hand_models_are_palm_centred():
    from gymnasium_robotics_amd.envs.hand import HAND_ORIGIN, load_hand_reach_model

    for m in (load_hand_reach_model(None), _hand(), _hand(touch=True), _hand("egg"), _hand("pen", True)):
        assert np.array_equal(m.origin, np.asarray(HAND_ORIGIN)) and m.info["origin"] == list(HAND_ORIGIN)
        # every root body of the hand's workspace within 0.3 m of the model's origin (fp32 ulp <= 3e-8 m there; 1.2e-7 in the MJCF's frame)
        par = np.asarray(m.tables["body_parent"]).reshape(-1)
        roots = np.asarray(m.tables["body_pos"]).reshape(-1, 3)[1:][par[1:] == 0]
        assert np.abs(roots).max() < 0.3
        back = m.in_mjcf_frame()
        assert not back.origin.any() and np.abs(np.asarray(back.tables["body_pos"]).reshape(-1, 3)[1:][par[1:] == 0]).max() > 0.8


def test_world_columns_and_row_round_trip():
    m = _hand()
    cols, axes = m.world_columns("qpos")
    j = int(m.names["joint"]["object:joint"])
    qa = int(np.asarray(m.tables["jnt_qposadr"]).reshape(-1)[j])
    assert cols.tolist() == [qa, qa + 1, qa + 2] and axes.tolist() == [0, 1, 2]
    assert len(m.world_columns("qvel")[0]) == 0 and len(m.world_columns("mocap")[0]) == 0
    rng = np.random.default_rng(0)
    q = rng.normal(size=(5, m.dim("nq")))
    r = m.rows_from_world("qpos", q)
    assert np.array_equal(np.delete(r, cols, axis=1), np.delete(q, cols, axis=1))      # joint angles and the quaternion are untouched
    assert np.allclose(q[:, cols] - r[:, cols], m.origin) and np.abs(m.rows_to_world("qpos", r) - q).max() < 1e-15
    assert np.array_equal(m.rows_from_world("qvel", q[:, :30]), q[:, :30])


@pytest.mark.skipif(not os.path.isdir(ASSETS), reason="needs the reference's MJCF assets")
@pytest.mark.parametrize("xml, kw", [("hand/reach.xml", "reach"), ("hand/manipulate_block_touch_sensors.xml", "touch")])
def test_translation_touches_only_absolute_positions_and_is_undone(xml, kw):
    from gymnasium_robotics_amd.envs.hand import HAND_MANIP_CAPACITY, HAND_ORIGIN, HAND_REACH_COMPILE
    from gymnasium_robotics_amd.envs.manipulate_spec import drop_target_body, touch_filter
    from gymnasium_robotics_amd.mjcf import compile_mjcf

    args = HAND_REACH_COMPILE if kw == "reach" else dict(mutate=drop_target_body, touch_filter=touch_filter, keep_sites=[], capacity=dict(HAND_MANIP_CAPACITY, jpool=928))
    plain = compile_mjcf(os.path.join(ASSETS, xml), **args)
    moved = compile_mjcf(os.path.join(ASSETS, xml), origin=HAND_ORIGIN, **args)
    changed = {k for k in plain.tables if np.asarray(plain.tables[k]).size and not np.allclose(np.asarray(plain.tables[k], dtype=np.float64), np.asarray(moved.tables[k], dtype=np.float64), rtol=1e-9, atol=1e-12)}
    assert changed <= {"opt", "qpos0", "body_pos", "geom_pos", "site_pos"} and "body_pos" in changed      # (inverse weights etc. are translation invariant: they move by rounding only)
    back = moved.in_mjcf_frame()
    for k in plain.tables:
        a, b = np.asarray(plain.tables[k]), np.asarray(back.tables[k])
        assert a.shape == b.shape, k
        if a.dtype.kind == "f":
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12), k
        else:
            assert np.array_equal(a, b), k


@pytest.mark.parametrize("obj, touch, fixture", [("block", False, "hand_BlockRotateXYZ_teacher.npz"), ("block", True, "hand_BlockRotateXYZ_touch_teacher.npz"), ("egg", False, "hand_EggRotate_teacher.npz")])
def test_oracle_on_the_recentred_blob_reproduces_the_mjcf_frame_fixtures(obj, touch, fixture):
    """The fixtures were written by the oracle on models compiled in the MJCF's frame (round 4 / 5).  The oracle handed today's palm-centred blob translates it back first;
    teacher-forced from the recorded states it must give the recorded observations up to the fp64 rounding of the round trip (the compiled inverse weights differ in
    their last bits, 2e-13 relative; stiff contacts amplify that to ~1e-8 on a velocity) -- the product model IS the reference's model, moved."""
    from oracle.manipulate_oracle import OracleHandBlockEnv

    m = _hand(obj, touch)
    g = np.load(os.path.join(HERE, "golden", fixture))
    orc = OracleHandBlockEnv(m, "ignore", "xyz", "sparse", "sensordata" if touch else "off", obj=obj)
    assert not orc.sim.model.origin.any()
    worst = 0.0
    for i in range(0, g["obs"].shape[0], 7):
        s = orc.sim
        s.qpos[:], s.qvel[:], s.qacc_warmstart[:] = g["qpos"][i], g["qvel"][i], g["qacc_ws"][i]
        o, *_ = orc.step(g["action"][i].astype(np.float64))
        worst = max(worst, float(np.abs(o["observation"] - g["obs"][i])[:61].max()))
    assert worst < 2e-7, worst


def test_emulated_kernel_reports_world_frame_positions():
    """rows in the model's frame, observation / achieved goal in the MJCF's: the object's position leaves the task code with the origin added back in fp64"""
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.manipulate_spec import make_block_task

    m = _hand()
    g = np.load(os.path.join(HERE, "golden", "hand_BlockRotateXYZ_teacher.npz"))
    emu = EmuSim(m, make_block_task(m, "ignore", "xyz", "sparse"))
    emu.load_world(g, 3)
    qa = int(make_block_task(m, "ignore", "xyz", "sparse").obj_qadr)
    assert np.abs(emu.qpos[qa: qa + 3]).max() < 0.3 and np.abs(g["qpos"][3][qa: qa + 3]).max() > 0.8      # palm-centred rows
    emu.hand_step(g["action"][3])
    pos_rows = emu.qpos[qa: qa + 3].astype(np.float64)
    assert np.array_equal(emu.hand_obs[54:57], (pos_rows + m.origin).astype(np.float32)) and np.array_equal(emu.hand_achieved[:3], emu.hand_obs[54:57])
    assert np.abs(emu.hand_obs[54:57] - g["obs"][3][54:57]).max() < 1e-6
    assert np.array_equal(emu.hand_obs[57:61], emu.qpos[qa + 3: qa + 7])      # the quaternion as it is
