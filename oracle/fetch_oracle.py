"""Single-world fp64 Fetch environment on the C oracle -- TEST INFRASTRUCTURE ONLY.

Restates, for one world, the reference's Fetch step/reset path so the batched
HIP path can be checked against it on identical seeds and actions:

* step / reset loop ........ /root/reference/gymnasium_robotics/envs/robot_env.py:114-186
* sim reset ................ robot_env.py:305-316, fetch/fetch_env.py:375-402
* env setup (settle) ....... fetch/fetch_env.py:404-428
* action mapping ........... fetch/fetch_env.py:85-105,305-310;
                             utils/mujoco_utils.py:34-107 (ctrl_set_action, mocap_set_action,
                             reset_mocap_welds, reset_mocap2body_xpos)
* observation .............. fetch/fetch_env.py:107-143,312-360; utils/rotations.py:162-184 (mat2euler)
* goal / reward / success .. fetch/fetch_env.py:16-18,74-80,153-170
* task constants ........... fetch/{reach,push,slide,pick_and_place}.py constructors

The RNG is numpy's PCG64 seeded the way gymnasium.utils.seeding.np_random does
it [3P]: ``np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))``.
"""
import numpy as np

from .oracle_sim import OracleSim

FETCH_TASKS = {
    # kwargs of the four constructors (reach.py:125-147, push.py:167-190, slide.py:166-189, pick_and_place.py:139-162)
    "FetchReach": dict(xml="reach.xml", has_object=False, block_gripper=True, gripper_extra_height=0.2,
                       target_in_the_air=True, target_offset=0.0, obj_range=0.15, target_range=0.15,
                       initial_qpos={"robot0:slide0": 0.4049, "robot0:slide1": 0.48, "robot0:slide2": 0.0}),
    "FetchPush": dict(xml="push.xml", has_object=True, block_gripper=True, gripper_extra_height=0.0,
                      target_in_the_air=False, target_offset=0.0, obj_range=0.15, target_range=0.15,
                      initial_qpos={"robot0:slide0": 0.405, "robot0:slide1": 0.48, "robot0:slide2": 0.0,
                                    "object0:joint": [1.25, 0.53, 0.4, 1.0, 0.0, 0.0, 0.0]}),
    "FetchSlide": dict(xml="slide.xml", has_object=True, block_gripper=True, gripper_extra_height=-0.02,
                       target_in_the_air=False, target_offset=np.array([0.4, 0.0, 0.0]), obj_range=0.1, target_range=0.3,
                       initial_qpos={"robot0:slide0": 0.05, "robot0:slide1": 0.48, "robot0:slide2": 0.0,
                                     "object0:joint": [1.7, 1.1, 0.41, 1.0, 0.0, 0.0, 0.0]}),
    "FetchPickAndPlace": dict(xml="pick_and_place.xml", has_object=True, block_gripper=False, gripper_extra_height=0.2,
                              target_in_the_air=True, target_offset=0.0, obj_range=0.15, target_range=0.15,
                              initial_qpos={"robot0:slide0": 0.405, "robot0:slide1": 0.48, "robot0:slide2": 0.0,
                                            "object0:joint": [1.25, 0.53, 0.4, 1.0, 0.0, 0.0, 0.0]}),
}
N_SUBSTEPS = 20
DISTANCE_THRESHOLD = 0.05


def mat2euler(mat):
    """rotations.py:162-184 restated for one 3x3 matrix."""
    eps4 = np.finfo(np.float64).eps * 4.0
    cy = np.sqrt(mat[2, 2] * mat[2, 2] + mat[1, 2] * mat[1, 2])
    if cy > eps4:
        return np.array([-np.arctan2(mat[1, 2], mat[2, 2]), -np.arctan2(-mat[0, 2], cy), -np.arctan2(mat[0, 1], mat[0, 0])])
    return np.array([0.0, -np.arctan2(-mat[0, 2], cy), -np.arctan2(-mat[1, 0], mat[1, 1])])


def goal_distance(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b), axis=-1)


class OracleFetchEnv:
    def __init__(self, model, task="FetchPickAndPlace", reward_type="sparse", mesh_plane=True):
        cfg = FETCH_TASKS[task]
        self.cfg, self.reward_type, self.model = cfg, reward_type, model
        self.sim = OracleSim(model)
        self.sim.set_option("opt_disable_mesh_plane", 0 if mesh_plane else 1)
        n = model.names
        self.jq = {k: int(model.tables["jnt_qposadr"].ravel()[v]) for k, v in n["joint"].items()}
        self.jd = {k: int(model.tables["jnt_dofadr"].ravel()[v]) for k, v in n["joint"].items()}
        self.robot_joints = [k for k in sorted(n["joint"], key=lambda k: n["joint"][k]) if k.startswith("robot")]
        self.site = n["site"]
        self.grip_body, self.grip_rel = model.info["body_orig"]["robot0:gripper_link"][0], model.info["body_orig"]["robot0:gripper_link"][1:]
        self.has_object, self.block_gripper = cfg["has_object"], cfg["block_gripper"]
        self.dt = N_SUBSTEPS * model.opt("timestep")
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(None)))
        self._env_setup()
        self.initial_time = float(self.sim.time[0])
        self.initial_qpos = self.sim.qpos.copy()
        self.initial_qvel = self.sim.qvel.copy()
        self.goal = np.zeros(3)

    # ---- mujoco_utils restatements ------------------------------------------------
    def _gripper_body_pose(self):
        """data.xpos / data.xquat of robot0:gripper_link (a fused static child of wrist_roll_link)."""
        s, b = self.sim, self.grip_body
        R = s.xmat[9 * b: 9 * b + 9].reshape(3, 3)
        pos = s.xpos[3 * b: 3 * b + 3] + R @ np.array(self.grip_rel[0])
        q, r = s.xquat[4 * b: 4 * b + 4], np.array(self.grip_rel[1])
        quat = np.array([
            q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3], q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2],
            q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1], q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0]])
        return pos, quat

    def _site_xpos(self, name):
        i = self.site[name]
        return self.sim.site_xpos[3 * i: 3 * i + 3]

    def _site_vel(self, name):
        jacp, jacr = self.sim.jac_site(self.site[name])
        return jacp @ self.sim.qvel, jacr @ self.sim.qvel

    # ---- construction (fetch_env.py:404-428) -------------------------------------------
    def _env_setup(self):
        s, cfg = self.sim, self.cfg
        for name, v in cfg["initial_qpos"].items():
            v = np.atleast_1d(v)
            s.qpos[self.jq[name]: self.jq[name] + len(v)] = v
        # reset_mocap_welds (mujoco_utils.py:74-80)
        s.model_table("eq_data", 11 * self.model.dim("neq"))[:7] = [0, 0, 0, 0, 0, 0, 1]
        s.forward()
        s.forward()
        target = np.array([-0.498, 0.005, -0.431 + cfg["gripper_extra_height"]]) + self._site_xpos("robot0:grip")
        s.mocap_pos[:] = target
        s.mocap_quat[:] = [1.0, 0.0, 1.0, 0.0]
        for _ in range(10):
            s.step(N_SUBSTEPS)
        self.initial_gripper_xpos = self._site_xpos("robot0:grip").copy()
        if self.has_object:
            self.height_offset = float(self._site_xpos("object0")[2])

    # ---- reset (robot_env.py:154-186) -------------------------------------------------
    def reset(self, seed=None):
        if seed is not None:
            self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        s, cfg = self.sim, self.cfg
        s.reset_data()
        s.time[0] = self.initial_time
        s.qpos[:] = self.initial_qpos
        s.qvel[:] = self.initial_qvel
        s.forward()
        if self.has_object:
            oxy = self.initial_gripper_xpos[:2]
            while np.linalg.norm(oxy - self.initial_gripper_xpos[:2]) < 0.1:
                oxy = self.initial_gripper_xpos[:2] + self.np_random.uniform(-cfg["obj_range"], cfg["obj_range"], size=2)
            a = self.jq["object0:joint"]
            s.qpos[a: a + 2] = oxy
        s.forward()
        self.goal = self._sample_goal()
        return self._get_obs(), {}

    def _sample_goal(self):
        cfg = self.cfg
        goal = self.initial_gripper_xpos[:3] + self.np_random.uniform(-cfg["target_range"], cfg["target_range"], size=3)
        if self.has_object:
            goal += cfg["target_offset"]
            goal[2] = self.height_offset
            if cfg["target_in_the_air"] and self.np_random.uniform() < 0.5:
                goal[2] += self.np_random.uniform(0, 0.45)
        return goal.copy()

    # ---- step (robot_env.py:114-152) ----------------------------------------------------
    def step(self, action, aux=None):
        """aux: optional (pos3, quat4) of gripper_link to use instead of the sim's last kinematics (teacher forcing)."""
        s = self.sim
        # robot_env.py:132 clips with the float32 bounds of the action space and fetch_env.py:85-93 scales in place: the arithmetic runs in
        # the dtype of the action that was passed in (float32 actions -> float32 product), as in the reference
        action = np.clip(np.asarray(action), np.float32(-1.0), np.float32(1.0))
        pos_ctrl = np.array(action[:3], copy=True)
        pos_ctrl *= 0.05
        pos_ctrl, g = pos_ctrl.astype(np.float64), (0.0 if self.block_gripper else float(action[3]))
        # ctrl_set_action: position servos take targets relative to the current qpos
        for i in range(s.nu):
            jid = int(self.model.tables["act_trnid"][i])
            if int(self.model.tables["act_biastype"][i]) == 0:
                s.ctrl[i] = g
            else:
                s.ctrl[i] = s.qpos[int(self.model.tables["jnt_qposadr"].ravel()[jid])] + g
        # mocap_set_action: snap the mocap onto the welded body, then add the deltas
        p, q = self._gripper_body_pose() if aux is None else (np.asarray(aux[:3]), np.asarray(aux[3:7]))
        s.mocap_pos[:] = p + pos_ctrl
        s.mocap_quat[:] = q + np.array([1.0, 0.0, 1.0, 0.0])
        s.step(N_SUBSTEPS)
        if self.block_gripper:  # _step_callback, fetch_env.py:295-303
            s.qpos[self.jq["robot0:l_gripper_finger_joint"]] = 0.0
            s.qpos[self.jq["robot0:r_gripper_finger_joint"]] = 0.0
            s.forward()
        obs = self._get_obs()
        info = {"is_success": self._is_success(obs["achieved_goal"], self.goal)}
        reward = self.compute_reward(obs["achieved_goal"], self.goal, info)
        return obs, reward, False, False, info

    def _get_obs(self):
        s, dt = self.sim, self.dt
        grip_pos = self._site_xpos("robot0:grip").copy()
        grip_velp = self._site_vel("robot0:grip")[0] * dt
        rq = np.array([s.qpos[self.jq[k]] for k in self.robot_joints])
        rv = np.array([s.qvel[self.jd[k]] for k in self.robot_joints])
        if self.has_object:
            object_pos = self._site_xpos("object0").copy()
            i = self.site["object0"]
            object_rot = mat2euler(s.site_xmat[9 * i: 9 * i + 9].reshape(3, 3))
            vp, vr = self._site_vel("object0")
            object_velp, object_velr = vp * dt, vr * dt
            object_rel_pos = object_pos - grip_pos
            object_velp = object_velp - grip_velp
        else:
            object_pos = object_rot = object_velp = object_velr = object_rel_pos = np.zeros(0)
        obs = np.concatenate([grip_pos, object_pos, object_rel_pos, rq[-2:], object_rot, object_velp, object_velr, grip_velp, rv[-2:] * dt])
        ag = object_pos.copy() if self.has_object else grip_pos.copy()
        return {"observation": obs, "achieved_goal": ag, "desired_goal": self.goal.copy()}

    def compute_reward(self, achieved_goal, goal, info):
        d = goal_distance(achieved_goal, goal)
        if self.reward_type == "sparse":
            return -(d > DISTANCE_THRESHOLD).astype(np.float32)
        return -d

    def _is_success(self, achieved_goal, desired_goal):
        return (goal_distance(achieved_goal, desired_goal) < DISTANCE_THRESHOLD).astype(np.float32)
