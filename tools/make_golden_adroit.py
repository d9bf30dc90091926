"""Golden fixtures for the Adroit hand tasks (tests/golden/adroit_<task>_teacher.npz): teacher-forcing snapshots from the fp64 oracle --
random-action rollouts and scripted rollouts that drive the hand into its object (hammer handle, door latch, pen, ball) so that finger-object
contacts with the noslip pass active are on the tested path.  `shift` / `target` are the per-world model edits of reset_model as the engine takes them.

    python tools/make_golden_adroit.py [hammer door pen relocate]
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd.envs.adroit_spec import load_adroit_model  # noqa: E402
from oracle.adroit_oracle import OracleAdroitEnv  # noqa: E402


def scripted(task, nu, t):
    """bias of the action towards the object, ramped in over 25 steps (the noise is added by the caller)"""
    ramp = min(1.0, t / 25.0)
    if task == "hammer":     # arm pitch down, fingers close
        return np.concatenate([[-0.6, 0.3], [0.0, -0.3], np.full(22, 0.7)]) * ramp
    if task == "door":       # arm forward / up towards the handle, fingers half closed
        return np.concatenate([[0.6, 0.2, -0.3, 0.4], [0.0, -0.2], np.full(22, 0.4)]) * ramp
    if task == "pen":        # close the hand around the pen that starts above the palm
        return np.concatenate([[0.0, -0.3], np.full(22, 0.6)]) * ramp
    return np.concatenate([[0.0, -0.6, -0.9, 0.0, 0.0, 0.0], [0.0, -0.2], np.full(22, 0.6 if t > 20 else -0.2)]) * ramp   # relocate: reach down onto the ball, then close


def generate(task):
    model = load_adroit_model(task)
    env = OracleAdroitEnv(model, "dense", task)
    nu = model.dim("nu")
    rng = np.random.default_rng(21)
    rec = {k: [] for k in ("qpos", "qvel", "qacc_ws", "shift", "target", "action", "obs", "reward", "success", "qpos_next", "qvel_next", "ncon", "nefc", "noslip_iter",
                           "episode", "activation_gap")}
    resets = {k: [] for k in ("seed", "obs", "edit", "shift", "target")}
    for ep in range(6):
        obs, _ = env.reset(seed=ep)
        resets["seed"].append(ep); resets["obs"].append(obs); resets["edit"].append(env.model_edit.copy()); resets["shift"].append(env.sim.shift.copy())
        resets["target"].append(env.target_pos.copy())
        for t in range(70):
            a = rng.uniform(-1, 1, nu).astype(np.float32)
            if ep >= 3:
                a = np.clip(0.35 * a + scripted(task, nu, t).astype(np.float32), -1, 1)
            s = env.sim
            pre = dict(qpos=s.qpos.copy(), qvel=s.qvel.copy(), qacc_ws=s.qacc_warmstart.copy(), shift=s.shift.copy(), target=env.target_pos.copy(), action=a)
            s.min_activation_gap[0] = 1e30
            obs, r, _, _, info = env.step(a.astype(np.float64))
            for k, v in pre.items():
                rec[k].append(v)
            rec["obs"].append(obs); rec["reward"].append(r); rec["success"].append(info["success"]); rec["qpos_next"].append(s.qpos.copy()); rec["qvel_next"].append(s.qvel.copy())
            rec["ncon"].append(s.ncon); rec["nefc"].append(s.nefc); rec["noslip_iter"].append(s.noslip_iter); rec["episode"].append(ep)
            rec["activation_gap"].append(float(s.min_activation_gap[0]))
            assert s.bad_state == 0 and s.unsupported_hits == 0
    out = {k: np.asarray(v) for k, v in rec.items()}
    out.update({"reset_" + k: np.asarray(v) for k, v in resets.items()})
    if task == "hammer":     # names the round-2 tests use
        out["board_z"] = out["shift"][:, 2] + model.info["shift_pos0"][2]
        out["reset_board_z"] = out["reset_edit"][:, 2]
    path = os.path.join(ROOT, "tests", "golden", f"adroit_{task}_teacher.npz")
    np.savez_compressed(path, **out)
    print(f"{task}: {len(out['obs'])} snapshots, max ncon {out['ncon'].max()}, max nefc {out['nefc'].max()}, noslip sweeps mean {out['noslip_iter'].mean():.1f} max "
          f"{out['noslip_iter'].max()}, snapshots with >= 4 contacts: {(out['ncon'] >= 4).sum()}, successes {int(out['success'].sum())}, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    for task in (sys.argv[1:] or ["hammer", "door", "pen", "relocate"]):
        generate(task)
