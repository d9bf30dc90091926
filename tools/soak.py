"""Long free-running rollouts with autoreset on one MI355X: finite outputs, capacity / bad-state flag rates, success rates.
    python tools/soak.py [steps]   (run on the GPU box; summary to stdout)"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

import gymnasium_robotics_amd as grx  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
CASES = [("FetchPickAndPlace-v4", 4096), ("FetchSlide-v4", 4096), ("HandReach-v3", 4096), ("HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", 4096),
         ("HandManipulateEggRotate-v1", 4096), ("AntMaze_Large_Diverse_GR-v5", 4096), ("PointMaze_Medium-v3", 4096), ("AdroitHandHammer-v2", 4096),
         ("AdroitHandDoor-v2", 4096), ("AdroitHandPen-v2", 4096), ("AdroitHandRelocate-v2", 4096), ("FrankaKitchen-v1", 2048)]
if len(sys.argv) > 2:
    CASES = [c for c in CASES if any(c[0].startswith(p) for p in sys.argv[2:])]
for env_id, n in CASES:
    env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
    obs, _ = env.reset(seed=0)
    na = env.single_action_space.shape[0]
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    bits = torch.tensor([1, 2, 4, 8], device="cuda:0", dtype=torch.int32)
    counts = torch.zeros(4, device="cuda:0", dtype=torch.int64)      # accumulated on the device: no per-step synchronisation
    finite, succ, t0 = True, 0.0, time.time()
    for t in range(STEPS):
        obs, r, term, trunc, info = env.step(torch.rand(n, na, device="cuda:0", generator=g) * 2 - 1)
        counts += ((env.status.unsqueeze(1) & bits) != 0).sum(dim=0)
        if t % 50 == 49:
            o = obs["observation"] if isinstance(obs, dict) else obs
            finite = finite and bool(torch.isfinite(o).all()) and bool(torch.isfinite(torch.as_tensor(r)).all())
            key = "is_success" if "is_success" in info else ("success" if "success" in info else None)
            succ += float(torch.as_tensor(info[key]).float().mean()) if key else 0.0
    torch.cuda.synchronize()
    dt = time.time() - t0
    flags = {int(b): int(c) for b, c in zip(bits.tolist(), counts.tolist())}
    tot = n * STEPS
    print(f"{env_id}: {n} worlds x {STEPS} steps in {dt:.1f} s ({tot / dt:,.0f} env-steps/s incl. resets); finite {finite}; world-steps flagged: "
          f"bad-number {flags[1]} ({100 * flags[1] / tot:.4f} %), contact-capacity {flags[2]} ({100 * flags[2] / tot:.4f} %), row/pool-capacity {flags[4]} "
          f"({100 * flags[4] / tot:.4f} %), factorisation {flags[8]}; mean success rate at sampled steps {succ / max(1, STEPS // 50):.4f}")
    if env_id.startswith("AntMaze"):   # no overflow lane behind the ant kernels, tables below a full contact list (envs/point_maze.py ANT_CAPACITY): an excess would be DROPPED contacts
        assert flags[2] == 0 and flags[4] == 0, "ant worlds exceeded the engine tables: raise ANT_CAPACITY"
    env.close()
