"""Single-world fp64 PointMaze environment on the C oracle -- TEST INFRASTRUCTURE ONLY.

Restates /root/reference/gymnasium_robotics/envs/maze/point_maze.py:316-419 (PointMazeEnv), point.py:22-77 (PointEnv:
clip action, clip qvel to +-5, do_simulation(a, 1), obs = qpos|qvel) and maze_v4.py:299-418 (reset / reward / termination).
The reset sampler itself is shared host logic (gymnasium_robotics_amd/envs/maze_spec.py) and is pinned against the
reference's own golden vectors (tests/envs/maze/test_point_maze.py:20-45) in tests/test_cpu_maze.py.
"""
import numpy as np

from gymnasium_robotics_amd.envs.maze_spec import GOAL_RADIUS, Maze, redraw_goal, sample_maze_reset

from .oracle_sim import OracleSim


class OraclePointMazeEnv:
    def __init__(self, model, maze: Maze, reward_type="sparse", continuing_task=True, position_noise_range=0.25, reset_target=False):
        self.reset_target = reset_target
        self.sim, self.maze = OracleSim(model), maze
        self.reward_type, self.continuing_task, self.position_noise_range = reward_type, continuing_task, position_noise_range
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(None)))
        self.goal = np.zeros(2)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        self.goal, reset_pos = sample_maze_reset(self.maze, self.np_random, self.position_noise_range, options)
        s = self.sim
        s.reset_data()
        s.qpos[:2] = reset_pos
        s.qvel[:] = 0
        s.forward()
        obs = self._obs()
        return obs, {"success": bool(np.linalg.norm(obs["achieved_goal"] - self.goal) <= GOAL_RADIUS)}

    def _obs(self):
        o = np.concatenate([self.sim.qpos, self.sim.qvel])
        return {"observation": o, "achieved_goal": o[:2].copy(), "desired_goal": self.goal.copy()}

    def step(self, action):
        s = self.sim
        a = np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0)
        s.qvel[:] = np.clip(s.qvel, -5.0, 5.0)
        s.forward()  # MujocoEnv.set_state runs mj_forward [3P]
        s.ctrl[:] = a
        s.step(1)
        obs = self._obs()
        d = np.linalg.norm(obs["achieved_goal"] - self.goal)
        reward = float(np.exp(-d)) if self.reward_type == "dense" else float(d <= GOAL_RADIUS)
        terminated = (not self.continuing_task) and bool(d <= GOAL_RADIUS)
        if self.continuing_task and self.reset_target and d <= GOAL_RADIUS:   # update_goal (maze_v4.py:400-418), after the outputs were built
            self.goal = redraw_goal(self.maze, self.np_random, obs["achieved_goal"], self.goal, self.position_noise_range)
        return obs, reward, terminated, False, {"success": bool(d <= GOAL_RADIUS)}


class OracleAntMazeEnv(OraclePointMazeEnv):
    """ant_maze_v5.py:221-320 on top of gymnasium's AntEnv-v5 [3P] (frame_skip 5, obs = qpos | qvel with positions kept,
    reset_noise_scale = 0, init_qpos = the model's qpos0)."""

    FRAME_SKIP = 5

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        self.goal, reset_pos = sample_maze_reset(self.maze, self.np_random, self.position_noise_range, options)
        s = self.sim
        s.reset_data()
        s.qpos[:2] = reset_pos
        s.qvel[:] = 0
        s.forward()
        obs = self._obs()
        return obs, {"success": bool(np.linalg.norm(obs["achieved_goal"] - self.goal) <= GOAL_RADIUS)}

    def _obs(self):
        o = np.concatenate([self.sim.qpos, self.sim.qvel])
        return {"observation": o[2:].copy(), "achieved_goal": o[:2].copy(), "desired_goal": self.goal.copy()}

    def step(self, action):
        s = self.sim
        s.ctrl[:] = np.asarray(action, dtype=np.float64)
        s.step(self.FRAME_SKIP)
        obs = self._obs()
        d = np.linalg.norm(obs["achieved_goal"] - self.goal)
        reward = float(np.exp(-d)) if self.reward_type == "dense" else float(d <= GOAL_RADIUS)
        terminated = (not self.continuing_task) and bool(d <= GOAL_RADIUS)
        if self.continuing_task and self.reset_target and d <= GOAL_RADIUS:   # update_goal (maze_v4.py:400-418), after the outputs were built
            self.goal = redraw_goal(self.maze, self.np_random, obs["achieved_goal"], self.goal, self.position_noise_range)
        return obs, reward, terminated, False, {"success": bool(d <= GOAL_RADIUS)}
