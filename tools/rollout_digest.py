"""Bitwise fingerprint of short seeded rollouts of every family on the library $GRX_HIP_LIB selects (GPU): two builds whose arithmetic is the same print the same digests.
    python tools/rollout_digest.py [steps] [family prefix ...]
Round 6: the per-stage model records (GrxModel::reci_* / recf_*) replaced table walks and must not change a single bit."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

import gymnasium_robotics_amd as grx  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
CASES = [("FetchPickAndPlace-v4", 512), ("FetchSlide-v4", 256), ("FetchReach-v4", 128), ("HandReach-v3", 256), ("HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", 256),
         ("HandManipulateEggRotate-v1", 128), ("HandManipulatePenRotate-v1", 128), ("AntMaze_Large_Diverse_GR-v5", 256), ("PointMaze_Medium-v3", 128), ("AdroitHandHammer-v2", 256),
         ("AdroitHandDoor-v2", 128), ("AdroitHandPen-v2", 128), ("AdroitHandRelocate-v2", 128), ("FrankaKitchen-v1", 128)]
if len(sys.argv) > 2:
    CASES = [c for c in CASES if any(c[0].startswith(p) for p in sys.argv[2:])]
for env_id, n in CASES:
    env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
    obs, _ = env.reset(seed=0)
    na = env.single_action_space.shape[0]
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    h = hashlib.sha256()
    for t in range(STEPS):
        obs, r, term, trunc, info = env.step(torch.rand(n, na, device="cuda:0", generator=g) * 2 - 1)
        parts = list(obs.values()) if isinstance(obs, dict) else [obs]
        for x in parts + [torch.as_tensor(r)]:
            h.update(torch.as_tensor(x).detach().contiguous().cpu().numpy().tobytes())
    h.update(env.status.cpu().numpy().tobytes())
    print(f"{env_id} {n}x{STEPS}: {h.hexdigest()[:24]}", flush=True)
    env.close()
