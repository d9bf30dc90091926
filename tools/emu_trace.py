"""Where is an fp32-vs-fp64 difference born?  Steps ONE golden snapshot through the lane emulator in fp32 and in fp64 with -DGRX_EMU_TRACE and prints,
substep by substep, the first pass whose contact lists differ (pair set, or dist / pos / normal beyond a threshold) and the growth of |qvel32 - qvel64|.

    python tools/emu_trace.py AdroitHammer 67 [-DFLAG ...]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu_tolerances as T  # noqa: E402


def build(fp64, extra):
    so = f"/tmp/libgrx_emu{'64' if fp64 else '32'}_trace.so"
    flags = (["-DGRX_EMU_FP64", "-DGRX_MPR_EPS=2.220446049250313e-16"] if fp64 else []) + ["-DGRX_EMU_TRACE"] + [e for e in extra if not fp64 or e.startswith("-DGRX_DBG")]
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-misleading-indentation"] + flags + ["-o", so, os.path.join(ROOT, "tests", "emu", "grx_emu.cpp")])
    return so


def parse(path):
    passes, cur = [], None
    for ln in open(path):
        w = ln.split()
        if w[0] == "PASS": cur = {"con": [], "ncon": int(w[2]), "nefc": int(w[4])}; passes.append(cur)
        elif w[0] == "CON": cur["con"].append((int(w[2]), int(w[4]), int(w[5]), float(w[7]), np.array(w[9:12], float), np.array(w[13:16], float), int(w[17])))
        elif w[0] == "QVEL": cur["qvel"] = np.array(w[1:], float)
        elif w[0] == "QPOS": cur["qpos"] = np.array(w[1:], float)
    return passes


def run(fam, snap, fp64, extra):
    # one process per precision: the trace file handle is a static of the library
    code = f"""
import sys, ctypes, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); sys.path.insert(0, {os.path.join(ROOT, 'tools')!r})
import emu_tolerances as T
L = ctypes.CDLL({build(fp64, extra)!r}); L.emu_create.restype = ctypes.c_void_p; L.emu_create.argtypes = [ctypes.c_void_p] * 3
idx, e, st, comps = T.run_family(L, {fam!r}, {fp64}, every=10**9, only={snap})
print('ERR', e.max(), 'cols', list(np.argsort(-e[0])[:6]), 'status', st)
"""
    env = dict(os.environ, GRX_TRACE_FILE=f"/tmp/grx_trace_{'64' if fp64 else '32'}.txt")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    open(env["GRX_TRACE_FILE"] + ".stderr", "w").write(out.stderr)
    print(("fp64" if fp64 else "fp32"), out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else "")
    return parse(env["GRX_TRACE_FILE"])


def main():
    fam, snap = sys.argv[1], int(sys.argv[2])
    extra = [a for a in sys.argv[3:] if a.startswith("-D")]
    a, b = run(fam, snap, False, extra), run(fam, snap, True, extra)
    print(f"{len(a)} / {len(b)} forward passes")
    for i, (pa, pb) in enumerate(zip(a, b)):
        sa, sb = [(c[0]) for c in pa["con"]], [(c[0]) for c in pb["con"]]
        dv = np.abs(pa["qvel"] - pb["qvel"]).max() if "qvel" in pa and "qvel" in pb else float("nan")
        dq = np.abs(pa["qpos"] - pb["qpos"]).max() if "qpos" in pa and "qpos" in pb else float("nan")
        line = f"pass {i:2d} ncon {pa['ncon']:2d}/{pb['ncon']:2d} nefc {pa['nefc']:3d}/{pb['nefc']:3d} |dqvel| {dv:.2e} |dqpos| {dq:.2e}"
        if sa != sb:
            line += f"  PAIR SETS DIFFER: only32 {sorted(set(sa) - set(sb))} only64 {sorted(set(sb) - set(sa))}"
        print(line)
        if sa == sb:
            for ca, cb in zip(pa["con"], pb["con"]):
                dd, dp, dn = abs(ca[3] - cb[3]), np.abs(ca[4] - cb[4]).max(), np.abs(ca[5] - cb[5]).max()
                act = (ca[6] >= 0) != (cb[6] >= 0)
                if dd > 2e-6 or dp > 1e-4 or dn > 1e-4 or act:
                    print(f"      pair {ca[0]} geoms {ca[1]},{ca[2]}: ddist {dd:.2e} (dist {cb[3]:.3e}) dpos {dp:.2e} dnormal {dn:.2e}" + ("  ACTIVE32 %d ACTIVE64 %d" % (ca[6] >= 0, cb[6] >= 0) if act else ""))
        else:
            for c in pa["con"]:
                if c[0] not in sb: print(f"      only32 pair {c[0]} geoms {c[1]},{c[2]} dist {c[3]:.4e}")
            for c in pb["con"]:
                if c[0] not in sa: print(f"      only64 pair {c[0]} geoms {c[1]},{c[2]} dist {c[3]:.4e}")


if __name__ == "__main__":
    main()
