"""ctypes harness for tests/emu/libgrx_emu.so (sequential lane emulator of the device engine).
TEST INFRASTRUCTURE ONLY -- lets the kernel source be checked against the oracle without a GPU."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.abspath(os.path.join(_HERE, "..", ".."))


def build(force=False):
    so = os.path.join(_HERE, "libgrx_emu.so")
    import glob

    srcs = [os.path.join(_HERE, "grx_emu.cpp")] + sorted(glob.glob(os.path.join(_ROOT, "gymnasium_robotics_amd", "csrc", "*.h"))) + sorted(
        glob.glob(os.path.join(_ROOT, "include", "*")))
    stale = lambda: force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale():
        import fcntl

        with open(so + ".lock", "w") as lk:      # pytest-xdist workers: one builds, the others wait and find the fresh library
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                tmp = f"{so}.{os.getpid()}.tmp"
                subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-misleading-indentation", "-o", tmp, srcs[0]], cwd=_HERE)
                os.replace(tmp, so)
                force = False
    return so


class EmuSim:
    def __init__(self, model, task_struct):
        self.model, self.task = model, task_struct
        self.L = ctypes.CDLL(build())
        self.L.emu_create.restype = ctypes.c_void_p
        self.L.emu_create.argtypes = [ctypes.c_void_p] * 3
        self.L.emu_ctx_ptr.restype = ctypes.POINTER(ctypes.c_float)
        self.L.emu_ctx_ptr.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        self.L.emu_set_table.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int]
        self.L.emu_ctx_words.argtypes = [ctypes.c_void_p]
        H, I, F = model.pack()
        self._keep = (H, I, F)
        self.h = self.L.emu_create(H.ctypes.data, I.ctypes.data, F.ctypes.data)
        self.nq, self.nv, self.nmocap = model.dim("nq"), model.dim("nv"), model.dim("nmocap")
        self.qpos = np.zeros(self.nq, np.float32)
        self.qvel = np.zeros(self.nv, np.float32)
        self.qacc_ws = np.zeros(self.nv, np.float32)
        self.mocap = np.zeros(7 * self.nmocap, np.float32)
        self.aux = np.zeros(8, np.float32)
        self.obs = np.zeros(getattr(task_struct, "obs_dim", self.nq + self.nv - (2 if getattr(task_struct, "agent", 0) else 0)), np.float32)
        self.achieved = np.zeros(3, np.float32)
        self.status = ctypes.c_int(0)

    def load_world(self, g, i, keys=("qpos", "qvel", "qacc_ws")):
        """state of fixture snapshot i -- recorded by the oracle in the MJCF's world frame -- into the emulated world's rows, which live in the MODEL's frame
        (CompiledModel.rows_from_world: the subtraction in fp64, one rounding to fp32)"""
        for k in keys:
            getattr(self, k)[:] = self.model.rows_from_world(k, g[k][i])

    def set_table(self, name, data):
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert self.L.emu_set_table(ctypes.c_void_p(self.h), name.encode(), data.ctypes.data, data.size) == 0

    def _args(self):
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        return [ctypes.c_void_p(self.h), ctypes.byref(self.task), p(self.qpos), p(self.qvel), p(self.qacc_ws), p(self.mocap), p(self.aux)]

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float32)
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        self.L.emu_fetch_step(*self._args(), p(a), p(self.obs), p(self.achieved), ctypes.byref(self.status))

    def forward(self, nstep=0):
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        self.L.emu_forward(*self._args(), p(self.obs), p(self.achieved), ctypes.byref(self.status), ctypes.c_int(nstep))

    def ctx(self, name, n, dtype=np.float32):
        ptr = self.L.emu_ctx_ptr(ctypes.c_void_p(self.h), name.encode())
        a = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        return a.view(np.int32) if dtype == np.int32 else a


    def physics_steps(self, nstep=1, ctrl=None):
        """nstep raw physics steps of the loaded state (self.qpos / qvel / qacc_ws); returns (ncon, nefc) of the last one"""
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        c = None if ctrl is None else p(np.ascontiguousarray(ctrl, dtype=np.float32))
        ncon, nefc = ctypes.c_int(0), ctypes.c_int(0)
        self.L.emu_physics_steps(ctypes.c_void_p(self.h), p(self.qpos), p(self.qvel), p(self.qacc_ws), c, ctypes.byref(self.status), ctypes.byref(ncon), ctypes.byref(nefc),
                                 ctypes.c_int(nstep))
        return ncon.value, nefc.value

    def kitchen_step(self, action, last_qpos, noise=None, forward_only=False):
        """FrankaKitchen env.step() (or the reset-time forward pass) of one world; last_qpos is updated in place; returns (obs[59], completed mask)"""
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        a = np.ascontiguousarray(action, dtype=np.float32)
        nz = None if noise is None else p(np.ascontiguousarray(noise, dtype=np.float32))
        obs, done = np.zeros(int(self.task.obs_dim), np.float32), ctypes.c_int(0)
        assert last_qpos.dtype == np.float32 and last_qpos.shape == (9,)
        self.L.emu_kitchen_step(ctypes.c_void_p(self.h), ctypes.byref(self.task), p(self.qpos), p(self.qvel), p(self.qacc_ws), p(last_qpos), p(a), nz, p(obs),
                                ctypes.byref(done), ctypes.byref(self.status), ctypes.c_int(int(forward_only)))
        return obs, done.value

    def point_step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float32)
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        self.L.emu_point_step(ctypes.c_void_p(self.h), ctypes.byref(self.task), p(self.qpos), p(self.qvel), p(self.qacc_ws), p(a), p(self.obs),
                              p(self.achieved), ctypes.byref(self.status))

    def hand_step(self, action, forward_only=False):
        a = np.ascontiguousarray(action, dtype=np.float32)
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        if not hasattr(self, "hand_obs"):
            self.hand_obs = np.zeros(256, np.float32)
            self.hand_achieved = np.zeros(15, np.float32)
            self.palm = np.zeros(3, np.float32)
        self.L.emu_hand_step(ctypes.c_void_p(self.h), ctypes.byref(self.task), p(self.qpos), p(self.qvel), p(self.qacc_ws), p(a), p(self.hand_obs),
                             p(self.hand_achieved), p(self.palm), ctypes.byref(self.status), ctypes.c_int(int(forward_only)))

    def adroit_step(self, action, shift, act_mean, act_rng, forward_only=False, target=None):
        """Adroit env.step() (or the reset-time forward pass) of one world; shift = the world's 7-vector group pose; returns (obs, reward, success)"""
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        a, sh = np.ascontiguousarray(action, dtype=np.float32), np.ascontiguousarray(shift, dtype=np.float32)
        assert sh.shape == (7,)
        tg = np.ascontiguousarray(np.zeros(3) if target is None else target, dtype=np.float32)
        am, ar = np.ascontiguousarray(act_mean, dtype=np.float32), np.ascontiguousarray(act_rng, dtype=np.float32)
        obs, rew, suc = np.zeros(int(self.task.obs_dim), np.float32), ctypes.c_float(0), ctypes.c_ubyte(0)
        self.L.emu_adroit_step(ctypes.c_void_p(self.h), ctypes.byref(self.task), p(self.qpos), p(self.qvel), p(self.qacc_ws), p(sh), p(tg), p(a), p(am), p(ar), p(obs),
                               ctypes.byref(rew), ctypes.byref(suc), ctypes.byref(self.status), ctypes.c_int(int(forward_only)))
        return obs, float(rew.value), int(suc.value)
