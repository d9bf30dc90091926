"""The Adroit hand tasks on the CPU side: registry / spec logic, the oracle's noslip pass against its own definition, and the DEVICE ENGINE SOURCE
(lane emulator, tests/emu) against the fp64 oracle's golden fixtures (tools/make_golden_adroit.py)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def model():
    from gymnasium_robotics_amd.envs.adroit_spec import load_adroit_hammer_model

    return load_adroit_hammer_model()


def test_registry_and_model_dimensions(model):
    import gymnasium_robotics_amd as grx

    assert grx.env_family("AdroitHandHammer-v2") == "adroit" and grx.env_family("AdroitHandPenSparse-v2") == "adroit"
    # door 30 / 30 / 28, pen 30 / 30 / 24, relocate 36 / 36 / 30; every dof carries a friction-loss row (adroit_assets.xml:12)
    from gymnasium_robotics_amd.envs.adroit_spec import load_adroit_model
    for task, dims in (("door", (30, 30, 28)), ("pen", (30, 30, 24)), ("relocate", (36, 36, 30))):
        mm = load_adroit_model(task)
        assert (mm.dim("nq"), mm.dim("nv"), mm.dim("nu")) == dims and int(mm.tables["dims"][15]) == 20 and mm.info["unsupported_pairs"] == 0
        assert (np.asarray(mm.tables["dof_frictionloss"]) > 0).all()
    assert abs(float(load_adroit_model("door").tables["dof_frictionloss"][28]) - 2.0) < 1e-12     # door_hinge (adroit_door.xml:62)
    # 33 / 33 / 26, hammer = 3 slides + 3 hinges (no free joint), 44 limited tendons, noslip_iterations 20, iterations 20
    assert (model.dim("nq"), model.dim("nv"), model.dim("nu")) == (33, 33, 26)
    assert len(model.tables["tendon_adr"]) == 44 and int(model.tables["dims"][15]) == 20 and int(model.tables["dims"][13]) == 20
    assert abs(model.opt("timestep") - 0.002) < 1e-15                      # no timestep in <option>: MuJoCo's default
    assert abs(float(model.tables["dof_frictionloss"][26]) - 2.5) < 1e-12   # nail_dir (adroit_hammer.xml:81)
    # the constructor's actuator rewrite (adroit_hammer.py:234-262) is part of the packaged blob
    A, g, b = model.names["actuator"], model.tables["act_gainprm"].reshape(-1, 3), model.tables["act_biasprm"].reshape(-1, 3)
    assert np.allclose(g[A["A_WRJ1"]], [10, 0, 0]) and np.allclose(b[A["A_WRJ0"]], [0, -10, 0]) and np.allclose(g[A["A_THJ0"]], [1, 0, 0]) and np.allclose(b[A["A_FFJ3"]], [0, -1, 0])
    assert np.allclose(g[A["A_ARRx"]], [500, 0, 0]) and np.allclose(b[A["A_ARRy"]], [0, -200, 0])


def test_oracle_noslip_removes_friction_creep(model):
    """What noslip is for (MuJoCo docs, solver parameters): with the regularised solver a resting object in frictional contact creeps; the
    post-solver re-solves the friction forces without regularisation.  The hammer lying on the table with a small sideways pull on its handle
    slides measurably less with the pass enabled, and the pass terminates by its improvement test well before 20 sweeps in the static case."""
    from oracle.oracle_sim import OracleSim

    def slide(noslip_iterations):
        m = model.copy()
        m.tables["dims"][15] = noslip_iterations
        s = OracleSim(m)
        s.step(200)                          # settle on the table
        x0 = s.qpos[27:30].copy()
        g = s.model_table("opt", 16)
        g[1] = 0.6                           # gravity_x: a sideways pull well inside the friction cone (mu = 1)
        s.step(300)
        return np.linalg.norm(s.qpos[27:30] - x0), s.noslip_iter

    d0, _ = slide(0)
    d1, it = slide(20)
    assert d1 < 0.25 * d0, (d0, d1)
    assert 1 <= it < 20


@pytest.mark.parametrize("task", ["hammer", "door", "pen", "relocate"])
def test_emulated_kernel_matches_golden(task):
    """Teacher-forced env.step() of the engine source (fp32, emulated lanes) against every third of the 420 oracle snapshots of each task.  Joint
    angles and site / body positions hold 1e-4 (tests/adroit_cases.py names the exceptions).  The objects' velocities and Euler angles are
    ill-conditioned whenever a cylinder or capsule rests on another convex geom: the general convex routine returns ONE point of a line / face
    contact (as MuJoCo's does), fp32 and fp64 pick different ones and a light object (hammer: 8.9e-5 kg m^2) turns the difference into angular
    velocity.  The same source compiled in fp64 (tools/emu_fp64_check.py) agrees with the oracle to 5e-6 on every snapshot, so the quantiles
    asserted here measure rounding sensitivity, not logic."""
    from adroit_cases import check
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.adroit_spec import action_scaling, load_adroit_model, make_adroit_task

    model = load_adroit_model(task)
    g = np.load(os.path.join(GOLDEN, f"adroit_{task}_teacher.npz"))
    emu = EmuSim(model, make_adroit_task(model, "dense", task))
    am, ar = action_scaling(model)
    idx = list(range(0, g["obs"].shape[0], 3))
    obs, rew = [], []
    for i in idx:
        emu.qpos[:], emu.qvel[:], emu.qacc_ws[:] = g["qpos"][i], g["qvel"][i], g["qacc_ws"][i]
        o, r, suc = emu.adroit_step(g["action"][i], g["shift"][i], am, ar, target=g["target"][i])
        assert emu.status.value == 0 and suc == int(g["success"][i])
        obs.append(o); rew.append(r)
    print(check(task, np.array(obs), g["obs"][idx], np.array(rew), g["reward"][idx]))


@pytest.mark.parametrize("task", ["hammer", "door", "pen", "relocate"])
def test_emulated_reset_forward_matches_golden(task):
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.adroit_spec import action_scaling, load_adroit_model, make_adroit_task

    model = load_adroit_model(task)
    g = np.load(os.path.join(GOLDEN, f"adroit_{task}_teacher.npz"))
    emu = EmuSim(model, make_adroit_task(model, "dense", task))
    am, ar = action_scaling(model)
    for k in range(len(g["reset_seed"])):
        emu.qpos[:], emu.qvel[:], emu.qacc_ws[:] = model.tables["qpos0"], 0, 0
        obs, _, _ = emu.adroit_step(np.zeros(model.dim("nu"), np.float32), g["reset_shift"][k], am, ar, forward_only=True, target=g["reset_target"][k])
        assert np.abs(obs - g["reset_obs"][k]).max() < 1e-6


def test_reset_draws_follow_the_reference_order():
    """sample_reset: the draws of each reset_model in the reference's order, and the engine state they turn into (shift offset / rotation, target)."""
    from gymnasium_robotics_amd.envs.adroit_spec import load_adroit_model, sample_reset
    from gymnasium_robotics_amd.envs.manipulate_spec import euler2quat

    mk = lambda: np.random.Generator(np.random.PCG64(np.random.SeedSequence(7)))
    m = load_adroit_model("door")
    d, r = sample_reset("door", mk(), m), mk()
    want = np.array([r.uniform(low=-0.3, high=-0.2), r.uniform(low=0.25, high=0.35), r.uniform(low=0.252, high=0.35)])     # adroit_door.py:362-370
    assert np.array_equal(d["edit"], want) and np.allclose(d["shift"][:3], want - np.array(m.info["shift_pos0"])) and np.array_equal(d["shift"][3:], [1, 0, 0, 0])
    m = load_adroit_model("pen")
    d, r = sample_reset("pen", mk(), m), mk()
    q = euler2quat(np.array([r.uniform(low=-1, high=1), r.uniform(low=-1, high=1), 0.0]))                                    # adroit_pen.py:380-383
    assert np.array_equal(d["edit"], q) and np.allclose(d["shift"][3:], q)
    # the rotation is about the target body's origin: the origin itself stays where it is
    from gymnasium_robotics_amd.envs.adroit_spec import _quat2mat
    p0 = np.array(m.info["shift_pos0"])
    assert np.allclose(_quat2mat(d["shift"][3:]) @ p0 + d["shift"][:3], p0)
    m = load_adroit_model("relocate")
    d, r = sample_reset("relocate", mk(), m), mk()
    ox, oy = r.uniform(low=-0.15, high=0.15), r.uniform(low=-0.15, high=0.3)                                                 # adroit_relocate.py:353-372
    tg = np.array([r.uniform(low=-0.2, high=0.2), r.uniform(low=-0.2, high=0.2), r.uniform(low=0.15, high=0.35)])
    assert np.array_equal(d["target"], tg) and np.allclose(d["shift"][:3], [ox - m.info["shift_pos0"][0], oy - m.info["shift_pos0"][1], 0.0])
