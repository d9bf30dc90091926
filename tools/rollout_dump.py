"""Dump (or compare) the observations of a short seeded rollout on the library $GRX_HIP_LIB selects: which step / component differs first between two builds.
    python tools/rollout_dump.py dump <out.npz> <env_id> [n] [steps]        python tools/rollout_dump.py cmp <a.npz> <b.npz>"""
import os
import sys

import numpy as np

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        if d.max() > 0:
            t, w, i = np.unravel_index(np.argmax(d > 0), d.shape) if d.ndim == 3 else (np.argmax(d.reshape(d.shape[0], -1).max(1) > 0), -1, -1)
            per_step = d.reshape(d.shape[0], -1).max(1)
            print(f"{k}: first difference at step {t} world {w} component {i}; max |diff| per step: {np.array2string(per_step[:12], precision=3)}")
            if d.ndim == 3:
                comp = (d[t] > 0).sum(0)
                print(f"   components differing at that step (count of worlds): {dict((int(j), int(c)) for j, c in enumerate(comp) if c)}")
        else:
            print(f"{k}: identical")
    sys.exit(0)
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

import gymnasium_robotics_amd as grx  # noqa: E402

out, env_id = sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
obs, _ = env.reset(seed=0)
na = env.single_action_space.shape[0]
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
rec = {}
def put(o):
    for k, v in (o.items() if isinstance(o, dict) else [("obs", o)]):
        rec.setdefault(k, []).append(torch.as_tensor(v).detach().cpu().numpy().copy())
put(obs)
for t in range(steps):
    obs, r, term, trunc, info = env.step(torch.rand(n, na, device="cuda:0", generator=g) * 2 - 1)
    put(obs)
np.savez(out, **{k: np.stack(v) for k, v in rec.items()})
