"""Golden fixtures for FrankaKitchen-v1 (tests/golden/kitchen_teacher.npz): teacher-forcing snapshots from the fp64 oracle -- random-action rollouts
and scripted ones that drive the arm into the scene (kettle, microwave door, burner knobs), with the default observation noise; the 59 uniform draws of
every observation are recorded so that the device path can be fed the same noise.

    python tools/make_golden_kitchen.py
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd.envs.kitchen_spec import TASKS, completed_mask, load_kitchen_model  # noqa: E402
from oracle.kitchen_oracle import OracleKitchenEnv  # noqa: E402


def draws_of_next_observation(env):
    g = copy.deepcopy(env.np_random)
    return np.concatenate([g.uniform(low=-1.0, high=1.0, size=k) for k in (9, 9, 21, 20)])


if __name__ == "__main__":
    model = load_kitchen_model()
    env = OracleKitchenEnv(model)
    rng = np.random.default_rng(5)
    rec = {k: [] for k in ("qpos", "qvel", "qacc_ws", "last_qpos", "action", "noise", "obs", "completed", "reward", "qpos_next", "qvel_next", "ncon", "nefc", "episode",
                           "activation_gap")}
    resets = {k: [] for k in ("seed", "obs", "noise")}
    bias = [np.zeros(9), np.zeros(9), np.array([-0.6, 0.5, 0.3, 0.6, 0.2, -0.5, 0.3, 0.5, 0.5]), np.array([0.7, 0.6, -0.4, 0.8, -0.3, 0.4, -0.2, -0.5, -0.5]),
            np.array([-0.2, 0.9, 0.6, 0.9, 0.5, 0.3, 0.1, 0.8, 0.8])]
    def rollout(ep, steps, act_bias, scale):
        for t in range(steps):
            a = np.clip(scale * rng.uniform(-1, 1, 9) + act_bias * min(1.0, t / 12.0), -1, 1)
            s = env.sim
            pre = dict(qpos=s.qpos.copy(), qvel=s.qvel.copy(), qacc_ws=s.qacc_warmstart.copy(), last_qpos=env._last_robot_qpos.copy(), action=a.copy(),
                       noise=draws_of_next_observation(env))
            s.min_activation_gap[0] = 1e30
            obs, r, term, trunc, info = env.step(a)
            for k, v in pre.items():
                rec[k].append(v)
            rec["obs"].append(obs["observation"]); rec["completed"].append(int(completed_mask(s.qpos))); rec["reward"].append(r)
            rec["qpos_next"].append(s.qpos.copy()); rec["qvel_next"].append(s.qvel.copy()); rec["ncon"].append(s.ncon); rec["nefc"].append(s.nefc)
            rec["episode"].append(ep); rec["activation_gap"].append(float(s.min_activation_gap[0]))
            assert s.bad_state == 0 and s.unsupported_hits == 0

    for ep in range(3):
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(ep)))
        resets["noise"].append(np.concatenate([g.uniform(low=-1.0, high=1.0, size=k) for k in (9, 9, 21, 20)]))
        obs, _ = env.reset(seed=ep)
        resets["seed"].append(ep); resets["obs"].append(obs["observation"])
        rollout(ep, 40, bias[ep + 1], 0.5)
    # contact-rich starts: arm poses (rejection-sampled inside the joint bounds) that touch the scene without deep penetration, objects displaced
    # (microwave and cabinet doors ajar, a burner knob turned, the kettle lifted 3 cm so that it drops), then 16 steps of random actions from each
    from gymnasium_robotics_amd.envs.kitchen_spec import INIT_QPOS, franka_config
    pb = franka_config(model)["pos_bound"][:9]
    starts = []
    while len(starts) < 8:
        q = INIT_QPOS.copy()
        q[:9] = rng.uniform(pb[:, 0], pb[:, 1])
        q[22] = -0.4 * rng.uniform(); q[21] = 0.5 * rng.uniform(); q[19] = 0.2 * rng.uniform(); q[9] = -0.3 * rng.uniform(); q[25] += 0.03
        env.sim.reset_data(); env.sim.qpos[:] = q; env.sim.forward()
        C = env.sim.contacts()
        robot = [c for c in C if c[0] < 0]
        if 6 <= len(C) <= 14 and all(c[0] > -0.004 for c in C):
            starts.append(q)
    for k, q in enumerate(starts):
        env.reset(seed=100 + k)
        env.sim.qpos[:] = q; env.sim.forward(); env._last_robot_qpos = q[:9].copy()
        rollout(3 + k, 16, np.zeros(9), 0.6)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out.update({"reset_" + k: np.asarray(v) for k, v in resets.items()})
    path = os.path.join(ROOT, "tests", "golden", "kitchen_teacher.npz")
    np.savez_compressed(path, **out)
    moved = np.abs(out["qpos_next"][:, 9:] - out["qpos"][:, 9:]).max(axis=0)
    print(f"{len(out['obs'])} snapshots, max ncon {out['ncon'].max()}, max nefc {out['nefc'].max()}, snapshots with >= 6 contacts: {(out['ncon'] >= 6).sum()}, "
          f"tasks seen complete: {[t for k, t in enumerate(TASKS) if (out['completed'] >> k & 1).any()]}, largest object-joint move per step {moved.max():.3f}, "
          f"{os.path.getsize(path) / 1024:.0f} KiB")
