"""Golden fixtures for the hull-vs-convex narrow phase (mesh-mesh / mesh-box pairs of the Fetch arm, assets/fetch/robot.xml:16-93):
teacher-forcing snapshots of FetchPickAndPlace rollouts whose scripted actions fold the arm into the head / torso or push the wrist and
gripper housing onto the table, so that most snapshots carry hull contacts (random actions produce them in 0.04 % of the substeps).

    python tools/make_golden_hull.py   ->  tests/golden/fetch_hull_teacher.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd.envs.fetch import load_fetch_model  # noqa: E402
from oracle.fetch_oracle import OracleFetchEnv  # noqa: E402

MOTIONS = {"back-up": (-1, 0, 1, 0), "back": (-1, 0, 0, 0), "right-back": (-1, -1, 0.3, 0), "down-back": (-1, 0, -1, -1), "fwd-down": (1, 0, -1, 0),
           "left-up": (-0.3, 1, 1, 0)}

if __name__ == "__main__":
    task = "FetchPickAndPlace"
    model = load_fetch_model(task)
    gt = model.tables["geom_type"].ravel()
    env = OracleFetchEnv(model, task)
    rng = np.random.default_rng(7)
    rec = {k: [] for k in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal", "action", "obs", "achieved", "reward", "success", "qpos_next", "qvel_next",
                           "ncon", "nefc", "hull_contacts", "activation_gap", "motion")}
    for mi, (name, base) in enumerate(MOTIONS.items()):
        env.reset(seed=100 + mi)
        for t in range(90):
            a = np.clip(np.asarray(base, np.float32) + rng.uniform(-0.3, 0.3, 4).astype(np.float32), -1, 1)
            s = env.sim
            p, q = env._gripper_body_pose()
            pre = dict(qpos=s.qpos.copy(), qvel=s.qvel.copy(), qacc_ws=s.qacc_warmstart.copy(), mocap=np.concatenate([s.mocap_pos, s.mocap_quat]),
                       aux=np.concatenate([p, q, [0.0]]), goal=env.goal.copy(), action=a)
            s.min_activation_gap[0] = 1e30
            c0 = s.mesh_contacts
            obs, r, _, _, info = env.step(a.astype(np.float64))
            nh = sum(1 for c in s.contacts() if gt[int(c[8])] == 7 and gt[int(c[7])] != 0)
            if t % 3 == 0 and t >= 6:
                for k, v in pre.items():
                    rec[k].append(v)
                rec["activation_gap"].append(float(s.min_activation_gap[0])); rec["obs"].append(obs["observation"]); rec["achieved"].append(obs["achieved_goal"])
                rec["reward"].append(r); rec["success"].append(info["is_success"]); rec["qpos_next"].append(s.qpos.copy()); rec["qvel_next"].append(s.qvel.copy())
                rec["ncon"].append(s.ncon); rec["nefc"].append(s.nefc); rec["hull_contacts"].append(s.mesh_contacts - c0); rec["motion"].append(mi)
            assert s.bad_state == 0
    out = {k: np.asarray(v) for k, v in rec.items()}
    path = os.path.join(ROOT, "tests", "golden", "fetch_hull_teacher.npz")
    np.savez_compressed(path, **out)
    hc = out["hull_contacts"]
    print(f"{len(hc)} snapshots, {int((hc > 0).sum())} with hull contacts in some substep (max {hc.max()} contact-substeps), max nefc {out['nefc'].max()}, "
          f"max ncon {out['ncon'].max()}, {os.path.getsize(path) / 1024:.0f} KiB")
