"""AdroitHandHammer on the CPU side: registry / spec logic, the oracle's noslip pass against its own definition, and the DEVICE ENGINE SOURCE
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

    assert grx.env_family("AdroitHandHammer-v2") == "adroit_hammer" and grx.env_family("AdroitHandHammerSparse-v2") == "adroit_hammer"
    with pytest.raises(grx.UnsupportedEnvError):
        grx.env_family("AdroitHandDoor-v2")
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


def test_emulated_kernel_matches_golden(model):
    """Teacher-forced env.step() of the engine source (fp32, emulated lanes) against 420 oracle snapshots.  qpos and the site / body positions hold
    1e-4 throughout.  The hammer's velocities and Euler angles are ill-conditioned whenever its cylinder head or capsule handle rests on the
    table: the general convex routine returns ONE point of a line / face contact (as MuJoCo's does), fp32 and fp64 pick different ones and the
    8.9e-5 kg m^2 hammer turns the difference into angular velocity.  The same source compiled in fp64 agrees with the oracle to 1e-6 (median 1e-11),
    so the quantiles asserted below measure rounding sensitivity, not logic."""
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.adroit_spec import action_scaling, board_shift, make_adroit_task

    g = np.load(os.path.join(GOLDEN, "adroit_hammer_teacher.npz"))
    emu = EmuSim(model, make_adroit_task(model, "dense"))
    am, ar = action_scaling(model)
    e_q, e_pos, e_vel, e_rot, e_rew = [], [], [], [], []
    for i in range(0, g["obs"].shape[0], 3):
        emu.qpos[:], emu.qvel[:], emu.qacc_ws[:] = g["qpos"][i], g["qvel"][i], g["qacc_ws"][i]
        obs, rew, suc = emu.adroit_step(g["action"][i], board_shift(model, float(g["board_z"][i])), am, ar)
        assert emu.status.value == 0
        e = np.abs(obs - g["obs"][i])
        e_q.append(e[:27].max()); e_vel.append(e[27:33].max()); e_pos.append(max(e[33:39].max(), e[42:45].max())); e_rot.append(e[39:42].max())
        e_rew.append(abs(rew - g["reward"][i])); assert suc == int(g["success"][i])
        assert e[45] < 1e-3
    e_q, e_pos, e_vel, e_rot = map(np.array, (e_q, e_pos, e_vel, e_rot))
    assert e_q.max() < 1e-4 and e_pos.max() < 2e-4, (e_q.max(), e_pos.max())
    assert np.median(e_vel) < 5e-3 and np.quantile(e_vel, 0.9) < 6e-2 and e_vel.max() < 0.5, (np.median(e_vel), np.quantile(e_vel, 0.9), e_vel.max())
    assert np.median(e_rot) < 1e-4 and e_rot.max() < 5e-3, (np.median(e_rot), e_rot.max())
    assert np.median(e_rew) < 1e-4 and max(e_rew) < 5e-3


def test_emulated_reset_forward_matches_golden(model):
    from emu_sim import EmuSim

    from gymnasium_robotics_amd.envs.adroit_spec import action_scaling, board_shift, make_adroit_task

    g = np.load(os.path.join(GOLDEN, "adroit_hammer_teacher.npz"))
    emu = EmuSim(model, make_adroit_task(model, "dense"))
    am, ar = action_scaling(model)
    for k in range(len(g["reset_seed"])):
        emu.qpos[:], emu.qvel[:], emu.qacc_ws[:] = model.tables["qpos0"], 0, 0
        obs, _, _ = emu.adroit_step(np.zeros(26, np.float32), board_shift(model, float(g["reset_board_z"][k])), am, ar, forward_only=True)
        assert np.abs(obs - g["reset_obs"][k]).max() < 1e-6
