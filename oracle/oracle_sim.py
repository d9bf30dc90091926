"""ctypes harness around oracle/libgrx_oracle.so -- TEST INFRASTRUCTURE ONLY.

``OracleSim`` plays the role ``(mujoco.MjModel, mujoco.MjData)`` play in the
reference (/root/reference/gymnasium_robotics/envs/robot_env.py:292-303): it
owns one world's fp64 state and exposes ``forward`` / ``step`` / ``reset_data``
/ ``jac_site``.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module (see oracle/grx_oracle.c header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgrx_oracle.so")
    src = os.path.join(_HERE, "grx_oracle.c")
    stale = lambda: force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    if stale():
        import fcntl

        with open(so + ".lock", "w") as lk:      # pytest-xdist workers: one builds, the others wait and find the fresh library
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                tmp = f"libgrx_oracle.{os.getpid()}.tmp.so"
                subprocess.check_call(["make", "-C", _HERE, "-B", tmp, f"OUT={tmp}"], stdout=subprocess.DEVNULL)
                os.replace(os.path.join(_HERE, tmp), so)
                force = False
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_reset_data.argtypes = [ctypes.c_void_p]
        L.orc_forward.argtypes = [ctypes.c_void_p]
        L.orc_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_jac_site.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_ptr.restype = ctypes.POINTER(ctypes.c_double)
        L.orc_ptr.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_model_ptr.restype = ctypes.POINTER(ctypes.c_double)
        L.orc_model_ptr.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_int.restype = ctypes.c_int
        L.orc_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_set_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.orc_contacts.restype = ctypes.c_int
        L.orc_contacts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _LIB = L
    return _LIB


class OracleSim:
    def __init__(self, model):
        """model: gymnasium_robotics_amd.mjcf.CompiledModel.  The oracle always works in the MJCF's own world frame -- the coordinates the reference's
        MjData holds -- whatever workspace-centred frame the product model was compiled in (CompiledModel.in_mjcf_frame: the identity for origin 0)."""
        model = model.in_mjcf_frame()
        self.model = model
        H, I, F = model.pack()
        self._L = lib()
        self._h = self._L.orc_create(H.ctypes.data, H.size, I.ctypes.data, I.size, F.ctypes.data, F.size)
        self.nq, self.nv, self.nu = model.dim("nq"), model.dim("nv"), model.dim("nu")
        self.nbody, self.nsite, self.ngeom, self.nmocap = (model.dim(k) for k in ("nbody", "nsite", "ngeom", "nmocap"))
        sizes = dict(
            qpos=self.nq, qvel=self.nv, ctrl=self.nu, mocap_pos=3 * self.nmocap, mocap_quat=4 * self.nmocap,
            qacc_warmstart=self.nv, xpos=3 * self.nbody, xquat=4 * self.nbody, xmat=9 * self.nbody,
            xipos=3 * self.nbody, geom_xpos=3 * self.ngeom, geom_xmat=9 * self.ngeom, site_xpos=3 * self.nsite,
            site_xmat=9 * self.nsite, subtree_com=3 * self.nbody, cdof=6 * self.nv, M=self.nv * self.nv,
            cvel=6 * self.nbody, qfrc_bias=self.nv, qfrc_passive=self.nv, qfrc_actuator=self.nv, qfrc_smooth=self.nv,
            qacc_smooth=self.nv, qfrc_constraint=self.nv, qacc=self.nv, time=1, min_activation_gap=1, shift=7,
            touch=len(model.tables.get("touch_body", [])),
        )
        for name, n in sizes.items():
            p = self._L.orc_ptr(self._h, name.encode())
            setattr(self, name, np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n])
        self.reset_data()

    def __del__(self):
        try:
            self._L.orc_destroy(self._h)
        except Exception:
            pass

    # views that depend on nefc
    def efc(self, name):
        n = self.nefc
        p = self._L.orc_ptr(self._h, ("efc_" + name).encode())
        if name == "J":
            return np.ctypeslib.as_array(p, shape=(max(n, 1) * self.nv,))[: n * self.nv].reshape(n, self.nv).copy()
        return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()

    def model_table(self, name, n):
        p = self._L.orc_model_ptr(self._h, name.encode())
        return np.ctypeslib.as_array(p, shape=(n,))

    def contacts(self):
        buf = np.zeros((128, 16))
        n = self._L.orc_contacts(self._h, buf.ctypes.data, 128)
        return buf[:n]

    nefc = property(lambda self: self._L.orc_int(self._h, b"nefc"))
    ncon = property(lambda self: self._L.orc_int(self._h, b"ncon"))
    solver_iter = property(lambda self: self._L.orc_int(self._h, b"solver_iter"))
    noslip_iter = property(lambda self: self._L.orc_int(self._h, b"noslip_iter"))
    bad_state = property(lambda self: self._L.orc_int(self._h, b"bad_state"))
    unsupported_hits = property(lambda self: self._L.orc_int(self._h, b"unsupported_hits"))
    mesh_candidates = property(lambda self: self._L.orc_int(self._h, b"mesh_candidates"))
    mesh_contacts = property(lambda self: self._L.orc_int(self._h, b"mesh_contacts"))

    def set_option(self, name, v):
        self._L.orc_set_int(self._h, name.encode(), int(v))

    def reset_data(self):
        self._L.orc_reset_data(self._h)

    def forward(self):
        self._L.orc_forward(self._h)

    def step(self, nstep=1):
        self._L.orc_step(self._h, int(nstep))

    def jac_site(self, site_id):
        jacp = np.zeros((3, self.nv))
        jacr = np.zeros((3, self.nv))
        self._L.orc_jac_site(self._h, int(site_id), jacp.ctypes.data, jacr.ctypes.data)
        return jacp, jacr
