"""Generate golden fixtures under tests/golden/ from the fp64 oracle (oracle/), which is the checker of
record in this repo ("parity unpinned": no MuJoCo available).  Each fixture holds
teacher-forcing snapshots: the full pre-step state of a world, the action, and the oracle's post-step
outputs.  The HIP path is stepped from the same pre-step states and compared (tests/test_gpu_fetch.py).

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd.envs.fetch import load_fetch_model  # noqa: E402
from oracle.fetch_oracle import OracleFetchEnv  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def snapshots(task, episodes, bias_down):
    model = load_fetch_model(task)
    env = OracleFetchEnv(model, task)
    env2 = OracleFetchEnv(model, task)  # re-runs each snapshot from fp32-rounded inputs: the oracle's own sensitivity
    prng = np.random.default_rng(99)
    rng = np.random.default_rng(1234)
    rec = {k: [] for k in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal", "action", "obs", "achieved", "reward", "success",
                           "qpos_next", "qvel_next", "ncon", "nefc", "seed", "t", "activation_gap", "sensitivity")}
    resets = {k: [] for k in ("seed", "obs", "achieved", "goal", "qpos")}
    for ep in range(episodes):
        obs, _ = env.reset(seed=ep)
        resets["seed"].append(ep); resets["obs"].append(obs["observation"]); resets["achieved"].append(obs["achieved_goal"])
        resets["goal"].append(obs["desired_goal"]); resets["qpos"].append(env.sim.qpos.copy())
        for t in range(50):
            a = rng.uniform(-1, 1, 4).astype(np.float32)
            if bias_down and ep % 2:
                a[2] = -abs(a[2])
            s = env.sim
            p, q = env._gripper_body_pose()
            rec["qpos"].append(s.qpos.copy()); rec["qvel"].append(s.qvel.copy()); rec["qacc_ws"].append(s.qacc_warmstart.copy())
            rec["mocap"].append(np.concatenate([s.mocap_pos, s.mocap_quat])); rec["aux"].append(np.concatenate([p, q, [0.0]]))
            rec["goal"].append(env.goal.copy()); rec["action"].append(a)
            s.min_activation_gap[0] = 1e30
            obs, r, _, _, info = env.step(a.astype(np.float64))
            rec["activation_gap"].append(float(s.min_activation_gap[0]))
            rec["obs"].append(obs["observation"]); rec["achieved"].append(obs["achieved_goal"]); rec["reward"].append(r)
            rec["success"].append(info["is_success"]); rec["qpos_next"].append(s.qpos.copy()); rec["qvel_next"].append(s.qvel.copy())
            rec["ncon"].append(s.ncon); rec["nefc"].append(s.nefc); rec["seed"].append(ep); rec["t"].append(t)
            # sensitivity: same step from inputs rounded to fp32 and jittered by one more fp32 ulp (what any fp32 engine sees)
            dev = 0.0
            for trial in range(2):
                s2 = env2.sim
                jit = (1.0 + prng.uniform(-1, 1, s2.nq) * 6e-8 * trial)
                s2.qpos[:] = rec["qpos"][-1].astype(np.float32).astype(np.float64) * jit
                s2.qvel[:] = rec["qvel"][-1].astype(np.float32).astype(np.float64)
                s2.qacc_warmstart[:] = rec["qacc_ws"][-1].astype(np.float32).astype(np.float64)
                env2.goal = env.goal.copy()
                o2, _, _, _, _ = env2.step(a.astype(np.float64), aux=rec["aux"][-1].astype(np.float32).astype(np.float64))
                dev = max(dev, float(np.abs(o2["observation"] - obs["observation"]).max()))
            rec["sensitivity"].append(dev)
            assert s.bad_state == 0
    out = {k: np.asarray(v) for k, v in rec.items()}
    out.update({"reset_" + k: np.asarray(v) for k, v in resets.items()})
    out["initial_gripper_xpos"] = env.initial_gripper_xpos
    out["height_offset"] = np.float64(getattr(env, "height_offset", 0.0))
    return out


if __name__ == "__main__":
    for task, eps, bias in (("FetchReach", 4, False), ("FetchPush", 6, True), ("FetchPickAndPlace", 8, True)):
        d = snapshots(task, eps, bias)
        path = os.path.join(OUT, f"fetch_{task}_teacher.npz")
        np.savez_compressed(path, **d)
        print(task, d["obs"].shape, "max nefc", d["nefc"].max(), "max ncon", d["ncon"].max(), f"{os.path.getsize(path)/1024:.0f} KiB")
