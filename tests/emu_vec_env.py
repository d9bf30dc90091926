"""TEST INFRASTRUCTURE: a FetchVecEnv-shaped environment whose worlds are stepped by the sequential lane emulator (tests/emu) instead of the
GPU, so that the N > 1 data path (tile-sharding by seed_offset, the per-step packed output rows, the one all-gather) can be exercised with a
real env API object in the GPU-less build container (gloo).  Same host logic as gymnasium_robotics_amd/envs/fetch.py: per-world PCG64 draws
(fetch_spec / sample_fetch_reset), _env_setup on the engine, packed rows [obs | achieved | desired | reward | success]."""
import numpy as np
import torch

from gymnasium_robotics_amd.core import np_random
from gymnasium_robotics_amd.envs.fetch import load_fetch_model, sample_fetch_reset
from gymnasium_robotics_amd.envs.fetch_spec import DISTANCE_THRESHOLD, FETCH_TASKS, N_SUBSTEPS, make_fetch_task, parse_env_id


class EmuFetchVecEnv:
    def __init__(self, env_id="FetchPickAndPlace-v4", num_envs=1, seed_offset=0):
        from emu_sim import EmuSim

        task, self.reward_type = parse_env_id(env_id)
        self.cfg, self.num_envs, self.seed_offset = FETCH_TASKS[task], int(num_envs), int(seed_offset)
        self.model = load_fetch_model(task).copy()
        self.model.tables["eq_data"][self.model.tables["eq_type"] == 1, :7] = [0, 0, 0, 0, 0, 0, 1]
        self.task = make_fetch_task(self.model, task, self.reward_type)
        self.emu = EmuSim(self.model, self.task)
        self.obs_dim = int(self.task.obs_dim)
        T, n = self.model.tables, self.model.names
        self._jq = T["jnt_qposadr"].ravel()
        # _env_setup (fetch_env.py:404-428) on the emulated engine: forward, mocap target, 10 x 20 settle steps
        e = self.emu
        q0 = T["qpos0"].astype(np.float64).copy()
        for name, v in self.cfg["initial_qpos"].items():
            v = np.atleast_1d(np.asarray(v, dtype=np.float64))
            a = int(self._jq[n["joint"][name]])
            q0[a: a + len(v)] = v
        mocap0 = np.concatenate([T["mocap_pos0"].ravel(), T["mocap_quat0"].ravel()]).astype(np.float32)
        e.qpos[:], e.qvel[:], e.qacc_ws[:], e.mocap[:] = q0, 0, 0, mocap0
        e.forward(0)
        target = np.array([-0.498, 0.005, -0.431 + self.cfg["gripper_extra_height"]]) + e.obs[:3].astype(np.float64)
        e.mocap[:] = list(target) + [1.0, 0.0, 1.0, 0.0]
        e.forward(10 * N_SUBSTEPS)
        self.initial_gripper_xpos = e.obs[:3].astype(np.float64).copy()
        self.height_offset = float(e.achieved[2]) if self.cfg["has_object"] else 0.0
        self._init = (e.qpos.copy(), e.qvel.copy(), mocap0)
        self._obj_qadr = int(self._jq[n["joint"]["object0:joint"]]) if self.cfg["has_object"] else -1
        N = self.num_envs
        self.state = [dict(qpos=None, qvel=None, qacc_ws=None, mocap=None, aux=None) for _ in range(N)]
        self.goal = np.zeros((N, 3), np.float32)
        self.packed = torch.zeros(N, self.obs_dim + 8)
        self.np_randoms = [np_random(None)[0] for _ in range(N)]

    def _run(self, w, action=None):
        e, st = self.emu, self.state[w]
        e.qpos[:], e.qvel[:], e.qacc_ws[:], e.mocap[:] = st["qpos"], st["qvel"], st["qacc_ws"], st["mocap"]
        if action is None:
            e.forward(0)
        else:
            e.aux[:] = st["aux"]
            e.step(action)
        st.update(qpos=e.qpos.copy(), qvel=e.qvel.copy(), qacc_ws=e.qacc_ws.copy(), mocap=e.mocap.copy(), aux=e.aux.copy())
        d = np.float32(np.linalg.norm(e.achieved.astype(np.float32) - self.goal[w]))
        r = -np.float32(d > DISTANCE_THRESHOLD) if self.reward_type == "sparse" else -d
        self.packed[w] = torch.from_numpy(np.concatenate([e.obs, e.achieved, self.goal[w], [r, float(d < DISTANCE_THRESHOLD)]]).astype(np.float32))

    def reset(self, seed=None):
        if seed is not None:
            self.np_randoms = [np_random(seed + self.seed_offset + i)[0] for i in range(self.num_envs)]
        for w in range(self.num_envs):
            oxy, goal = sample_fetch_reset(self.cfg, self.np_randoms[w], self.initial_gripper_xpos, self.height_offset)
            q = self._init[0].copy()
            if oxy is not None:
                q[self._obj_qadr: self._obj_qadr + 2] = oxy
            self.state[w].update(qpos=q, qvel=self._init[1].copy(), qacc_ws=np.zeros_like(self._init[1]), mocap=self._init[2].copy(), aux=np.zeros(8, np.float32))
            self.goal[w] = goal
            self._run(w)
        return self.packed.clone()

    def step(self, actions):
        for w in range(self.num_envs):
            self._run(w, np.asarray(actions[w], dtype=np.float32))
        return self.packed.clone()
