"""Which STAGE's fp32 arithmetic costs the parity?  Two builds of the lane emulator (fp32 and fp64) run the same env-step stage by stage.  The stages listed in
--fp32 run in fp32 arithmetic, every other stage runs in fp64 and hands its results over ROUNDED to fp32 (the storage precision of the device); compared with the
oracle's golden fixtures like tools/emu_tolerances.py.

    python tools/emu_mixed.py --fp32 0,1,2,3,4,5 AdroitHammer       # all stages fp32 (= the fp32 emulator, sanity check)
    python tools/emu_mixed.py --fp32 3 AdroitHammer FetchSlide      # only the constraint stage in fp32
stages: 0 kinematics, 1 inertia / cdof / M, 2 collision, 3 constraint rows, 4 velocity / bias forces, 5 solve + integrate
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu_tolerances as T  # noqa: E402

STATE = {"qpos", "qvel", "qacc_ws", "mocap_pos", "mocap_quat", "ctrl", "shift", "meshcache", "cnt", "ired"}
PERSIST = STATE | {"xpos", "xquat", "xmat", "sxpos", "sxmat", "cdof", "M", "qfrc_smooth", "qacc_smooth", "qfrc_constraint", "qacc", "Jp", "efc_D", "efc_aref", "efc_floss", "efc_kind", "efc_id", "efc_row",
                   "rk_q0", "rk_v0", "rk_Fv", "rk_Fa"}
CON = {"con_pos", "con_frame", "con_pair", "con_efc", "con_nr", "con_dist", "con_span", "con_ioff", "con_b1", "con_b2"}
LIVE = {   # fields that are valid after stage k (the overlays alias: only the live member of a union is handed over)
    0: PERSIST | {"ploc", "qloc", "janchor", "jaxis"},
    1: PERSIST | {"ploc", "qloc", "janchor", "jaxis", "crb", "cinert"},
    2: PERSIST | {"gxpos", "gxmat", "cinert"} | CON,
    3: PERSIST | {"gxpos", "gxmat", "cinert"} | CON,
    4: PERSIST | {"gxpos", "gxmat", "cinert", "cvel", "cacc", "cfrc", "cdof_dot", "qfrc_bias", "qfrc_passive", "qfrc_actuator"} | CON,
    5: PERSIST | {"A", "Ma", "grad", "search", "Mv", "tmpv", "efc_jar", "efc_jv", "efc_force", "efc_quad"} | ({"con_pos", "con_frame", "con_pair", "con_efc", "con_nr"}),
}
LIVE[6] = LIVE[5]


def build(fp64):
    so = f"/tmp/libgrx_emu{'64' if fp64 else '32'}_mixed.so"
    flags = (["-DGRX_EMU_FP64", "-DGRX_MPR_EPS=2.220446049250313e-16"] if fp64 else [a for a in sys.argv if a.startswith("-D")]) + ["-DGRX_EMU_STAGEHOOK"]
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w"] + flags + ["-o", so, os.path.join(ROOT, "tests", "emu", "grx_emu.cpp")])
    L = ctypes.CDLL(so)
    L.emu_create.restype = ctypes.c_void_p; L.emu_create.argtypes = [ctypes.c_void_p] * 3
    L.emu_field_name.restype = ctypes.c_char_p
    L.emu_run_stage.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return L


class Mixed:
    def __init__(self, fp32_stages):
        self.L32, self.L64 = build(False), build(True)
        self.fp32_stages = set(fp32_stages)
        self.h64 = None
        self.HOOK = ctypes.CFUNCTYPE(None, ctypes.c_int)
        self.cb = self.HOOK(self.hook)
        self.L32.emu_set_stage_hook(self.cb)
        self.ready = False

    def attach(self, model):
        H, I, F = model.pack()
        self._keep = (H, I, F)
        self.h64 = self.L64.emu_create(H.ctypes.data, I.ctypes.data, F.ctypes.data)
        self.ready = False

    def _maps(self):
        n = self.L32.emu_nfields()
        assert n == self.L64.emu_nfields() and n > 0
        self.names = [self.L32.emu_field_name(k).decode() for k in range(n)]
        lens = [self.L32.emu_field_len(k) for k in range(n)]
        assert lens == [self.L64.emu_field_len(k) for k in range(n)]
        self.buf = np.zeros(sum(lens))
        self.sel = {k: np.array([nm in v for nm in self.names], np.uint8) for k, v in LIVE.items()}
        self.sel[-1] = np.array([nm in STATE for nm in self.names], np.uint8)
        self.ready = True

    def move(self, src, dst, sel):
        src.emu_export(sel.ctypes.data_as(ctypes.c_void_p), self.buf.ctypes.data_as(ctypes.c_void_p))
        dst.emu_import(sel.ctypes.data_as(ctypes.c_void_p), self.buf.ctypes.data_as(ctypes.c_void_p))

    def hook(self, k):
        if not self.ready:
            self._maps()
        if k == -1:
            self.move(self.L32, self.L64, self.sel[-1])       # the state of the pass (fp32 storage) -> the fp64 build
            return
        if k == 7:      # inside the solve stage: Newton has finished, the Euler stage follows ("N" / "E" of --fp32 split stage 5)
            if 5 in self.fp32_stages or "N" in self.fp32_stages:
                self.move(self.L32, self.L64, self.sel[5])
            else:
                self.L64.emu_run_stage(ctypes.c_void_p(self.h64), 7)
                self.move(self.L64, self.L32, self.sel[5])
            return
        if k == 5 and 5 not in self.fp32_stages:
            if "E" in self.fp32_stages:
                self.move(self.L32, self.L64, self.sel[5])
            else:
                self.L64.emu_run_stage(ctypes.c_void_p(self.h64), 8)
                self.move(self.L64, self.L32, self.sel[5])
            return
        st = min(k, 5)
        if st in self.fp32_stages:
            self.move(self.L32, self.L64, self.sel[k])        # this stage's fp32 results are what the rest of the pass works with
        else:
            self.L64.emu_run_stage(ctypes.c_void_p(self.h64), k)
            self.move(self.L64, self.L32, self.sel[k])        # fp64 arithmetic, handed over rounded to fp32


def main(argv):
    stages = [(x if x in ("N", "E") else int(x)) for x in argv[argv.index("--fp32") + 1].split(",") if x != ""] if "--fp32" in argv else []
    every = int(argv[argv.index("--every") + 1]) if "--every" in argv else 1
    fams = [a for a in argv if a in T.FAMILY_TO_TASK]
    mx = Mixed(stages)
    import emu_fp64_check as E
    for fam in fams:
        m = E._fixture(T.FAMILY_TO_TASK[fam])[0]
        mx.attach(m)
        idx, e, status, comps = T.run_family(mx.L32, fam, False, every)
        for comp, cols in comps.items():
            err = e[:, cols].max(axis=1)
            print(f"fp32 stages {stages!s:14s} {fam:16s} {comp:24s} p50 {np.median(err):.1e} p90 {np.quantile(err, .9):.1e} p99 {np.quantile(err, .99):.1e} max {err.max():.1e} within 1e-4: {100 * np.mean(err < 1e-4):5.1f} % over {int((err >= 1e-4).sum())}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
