"""CPU proxy of tests/golden/tolerance_table.json: the kernel source compiled by the lane emulator (tests/emu/grx_emu.cpp) in fp32 -- the arithmetic of
the device build -- and, with --fp64, in double precision, stepped from the oracle's golden fixtures, with the SAME per-component split as
tests/tolerance_cases.py.  It is the development loop for conditioning work on the narrow phase (no GPU needed); the committed table itself is
measured on the MI355X (tools/measure_tolerances.py).

    python tools/emu_tolerances.py [--fp64] [--every K] [family ...]      # families as in tests/tolerance_cases.py CASES
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from emu_fp64_check import _fixture, as_fp64_struct  # noqa: E402

FAMILY_TO_TASK = {"FetchReach": "FetchReach", "FetchPush": "FetchPush", "FetchPickAndPlace": "FetchPickAndPlace", "FetchSlide": "FetchSlide", "FetchHullContacts": "hull",
                  "HandReach": "HandReach", "HandBlock": "HandBlock", "HandEgg": "HandEgg", "HandPen": "HandPen", "AdroitHammer": "hammer", "AdroitDoor": "door",
                  "AdroitPen": "pen", "AdroitRelocate": "relocate", "FrankaKitchen": "kitchen", "HandBlockTouch": "HandBlockTouch", "AntMazeLarge": "antlarge"}


_SENS = {}
JITTER = float(os.environ.get("GRX_JITTER", "6e-8"))      # relative amplitude of the --sensitivity perturbation


def build(fp64):
    so = f"/tmp/libgrx_emu{'64' if fp64 else '32'}_tol.so"
    src = os.path.join(ROOT, "tests", "emu", "grx_emu.cpp")
    flags = ["-DGRX_EMU_FP64", "-DGRX_MPR_EPS=2.220446049250313e-16", "-DGRX_EMU_RNDINJ"] if fp64 else []
    flags += [a for a in sys.argv if a.startswith("-D")]      # experiments: extra defines for the emulator build
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-misleading-indentation"] + flags + ["-o", so, src])
    return so


def _float_struct(t64, task_like):
    return task_like


def run_family(L, family, fp64, every=1, only=None, round_inputs=False, ref=None, jitter=0):
    """per-snapshot |obs - golden| rows [n, obs_dim] of the emulated step"""
    from tolerance_cases import CASES
    task = FAMILY_TO_TASK[family]
    m, t64, g, kind = _fixture(task)
    if fp64:
        t = t64
    else:  # rebuild the float task struct (the fixture helper widens it)
        import emu_fp64_check as E
        keep = E.as_fp64_struct
        E.as_fp64_struct = lambda s: s
        try:
            m, t, g, kind = E._fixture(task)
        finally:
            E.as_fp64_struct = keep
    dt = np.float64 if fp64 else np.float32
    cdt = ctypes.c_double if fp64 else ctypes.c_float
    if kind == "adroit":
        from gymnasium_robotics_amd.envs.adroit_spec import action_scaling
        am, ar = action_scaling(m)
    H, I, F = m.pack()
    h = L.emu_create(H.ctypes.data, I.ctypes.data, F.ctypes.data)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    jr = np.random.default_rng(1000 + jitter)
    def f(a):
        a = np.asarray(a, dtype=np.float32).astype(np.float64) if round_inputs else np.asarray(a, dtype=np.float64)
        if jitter:   # one fp32 ulp of relative noise on every state word: what any engine that holds its state in fp32 sees after one substep
            a = a * (1.0 + jr.uniform(-1, 1, a.shape) * JITTER)
        return np.ascontiguousarray(a, dtype=dt).copy()
    idx = list(range(0, g["obs"].shape[0], every)) if only is None else ([only] if np.isscalar(only) else list(only))
    out = np.zeros((len(idx), g["obs"].shape[1]))
    status = np.zeros(len(idx), np.int64)
    for j, i in enumerate(idx):
        qp, qv, qa, a = f(m.rows_from_world("qpos", g["qpos"][i])), f(g["qvel"][i]), f(g["qacc_ws"][i]), f(g["action"][i])      # fixtures hold MJCF-frame states, the rows are in the model's frame
        st = ctypes.c_int(0)
        if kind == "kitchen":
            obs, last, nz, done = np.zeros(g["obs"].shape[1], dt), f(g["last_qpos"][i]), f(g["noise"][i]), ctypes.c_int(0)
            L.emu_kitchen_step(ctypes.c_void_p(h), ctypes.byref(t), p(qp), p(qv), p(qa), p(last), p(a), p(nz), p(obs), ctypes.byref(done), ctypes.byref(st), ctypes.c_int(0))
        elif kind == "adroit":
            obs, sh, tg, rew, suc = np.zeros(g["obs"].shape[1], dt), f(g["shift"][i]), f(g["target"][i]), cdt(0), ctypes.c_ubyte(0)
            L.emu_adroit_step(ctypes.c_void_p(h), ctypes.byref(t), p(qp), p(qv), p(qa), p(sh), p(tg), p(a), p(f(am)), p(f(ar)), p(obs), ctypes.byref(rew),
                              ctypes.byref(suc), ctypes.byref(st), ctypes.c_int(0))
        elif kind == "point":
            obs, ach = np.zeros(g["obs"].shape[1] + 4, dt), np.zeros(2, dt)
            L.emu_point_step(ctypes.c_void_p(h), ctypes.byref(t), p(qp), p(qv), p(qa), p(a), p(obs), p(ach), ctypes.byref(st))
        elif kind == "fetch":
            obs, ach, mocap, aux = np.zeros(g["obs"].shape[1], dt), np.zeros(3, dt), f(m.rows_from_world("mocap", g["mocap"][i])), f(m.rows_from_world("aux", g["aux"][i]))
            L.emu_fetch_step(ctypes.c_void_p(h), ctypes.byref(t), p(qp), p(qv), p(qa), p(mocap), p(aux), p(a), p(obs), p(ach), ctypes.byref(st))
        else:
            obs, ach, palm = np.zeros(256, dt), np.zeros(15, dt), np.zeros(3, dt)
            L.emu_hand_step(ctypes.c_void_p(h), ctypes.byref(t), p(qp), p(qv), p(qa), p(a), p(obs), p(ach), p(palm), ctypes.byref(st), ctypes.c_int(0))
        out[j] = obs[:g["obs"].shape[1]].astype(np.float64) if isinstance(ref, str) else np.abs(obs[:g["obs"].shape[1]].astype(np.float64) - (g["obs"][i] if ref is None else ref[j]))
        status[j] = st.value
    return np.array(idx), out, status, CASES[family][3]


def run_family_horizons(L, family, horizons=(1, 2, 5, 10), every=1, fp64=False, round_inputs=False, jitter=0, ref=None, handoff=None):
    """FREE-RUNNING twin of run_family (tests/tolerance_cases.py::horizon_errors on the CPU): the emulated world starts from the pre-step state of snapshot i and keeps its own
    state for max(horizons) steps on the rollout's recorded actions (and the kitchen's recorded noise); after h steps its observation is compared with the oracle's recorded
    observation of snapshot i + h - 1.  Returns {h: (start indices, |error| rows [n_h, obs_dim], min activation gap over the h steps)}, comps."""
    from tolerance_cases import CASES, episode_runs
    import emu_fp64_check as E
    task = FAMILY_TO_TASK[family]
    if fp64:      # (the --sensitivity modes: L is the fp64 build of the engine source; ref="raw" returns observations, ref=<those> errors against them; jitter: run_family)
        m, t, g, kind = E._fixture(task)
    else:
        keep = E.as_fp64_struct
        E.as_fp64_struct = lambda s: s
        try:
            m, t, g, kind = E._fixture(task)
        finally:
            E.as_fp64_struct = keep
    dt, cdt = (np.float64, ctypes.c_double) if fp64 else (np.float32, ctypes.c_float)
    if kind == "adroit":
        from gymnasium_robotics_amd.envs.adroit_spec import action_scaling
        am, ar = action_scaling(m)
    H, I, F = m.pack()
    h_ = L.emu_create(H.ctypes.data, I.ctypes.data, F.ctypes.data)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    jr = np.random.default_rng(1000 + jitter)
    def f(a, state=False):
        a = np.asarray(a, dtype=np.float32).astype(np.float64) if round_inputs else np.asarray(a, dtype=np.float64)
        if jitter and state:      # one fp32 ulp of relative noise on every word of the START state (actions and recorded noise draws are the rollout's)
            a = a * (1.0 + jr.uniform(-1, 1, a.shape) * JITTER)
        return np.ascontiguousarray(a, dtype=dt).copy()
    fs = lambda a: f(a, True)
    run, gap, nobs, hmax = episode_runs(g), g["activation_gap"], g["obs"].shape[1], max(horizons)
    res = {h: ([], [], []) for h in horizons}
    if handoff is not None:      # the LAST step is run by the fp64 build (handoff) from the state the fp32 build reached: how the reference algorithm propagates an fp32 engine's own error
        t2 = _fixture(task)[1]
        h2 = handoff.emu_create(H.ctypes.data, I.ctypes.data, F.ctypes.data)

    def one(Lx, hx, tx, dtx, cdtx, S, a, nz):
        c = lambda x: np.ascontiguousarray(x, dtype=dtx)
        st = ctypes.c_int(0)
        if kind == "kitchen":
            obs, done = np.zeros(nobs, dtx), ctypes.c_int(0)
            Lx.emu_kitchen_step(ctypes.c_void_p(hx), ctypes.byref(tx), p(S["qp"]), p(S["qv"]), p(S["qa"]), p(S["last"]), p(c(a)), p(c(nz)), p(obs), ctypes.byref(done), ctypes.byref(st), ctypes.c_int(0))
        elif kind == "adroit":
            obs, rew, suc = np.zeros(nobs, dtx), cdtx(0), ctypes.c_ubyte(0)
            Lx.emu_adroit_step(ctypes.c_void_p(hx), ctypes.byref(tx), p(S["qp"]), p(S["qv"]), p(S["qa"]), p(S["sh"]), p(S["tg"]), p(c(a)), p(c(am)), p(c(ar)), p(obs), ctypes.byref(rew),
                               ctypes.byref(suc), ctypes.byref(st), ctypes.c_int(0))
        elif kind == "fetch":
            obs, ach = np.zeros(nobs, dtx), np.zeros(3, dtx)
            Lx.emu_fetch_step(ctypes.c_void_p(hx), ctypes.byref(tx), p(S["qp"]), p(S["qv"]), p(S["qa"]), p(S["mocap"]), p(S["aux"]), p(c(a)), p(obs), p(ach), ctypes.byref(st))
        else:
            obs, ach, palm = np.zeros(256, dtx), np.zeros(15, dtx), np.zeros(3, dtx)
            Lx.emu_hand_step(ctypes.c_void_p(hx), ctypes.byref(tx), p(S["qp"]), p(S["qv"]), p(S["qa"]), p(c(a)), p(obs), p(ach), p(palm), ctypes.byref(st), ctypes.c_int(0))
        return obs

    for i in range(0, g["obs"].shape[0], every):
        S = {"qp": fs(m.rows_from_world("qpos", g["qpos"][i])), "qv": fs(g["qvel"][i]), "qa": fs(g["qacc_ws"][i])}
        if kind == "kitchen":
            S["last"] = fs(g["last_qpos"][i])
        elif kind == "adroit":
            S["sh"], S["tg"] = f(g["shift"][i]), f(g["target"][i])
        elif kind == "fetch":
            S["mocap"], S["aux"] = fs(m.rows_from_world("mocap", g["mocap"][i])), fs(m.rows_from_world("aux", g["aux"][i]))
        for k in range(min(hmax, int(run[i]))):
            a, nz = f(g["action"][i + k]), (f(g["noise"][i + k]) if kind == "kitchen" else None)
            if handoff is not None and k == hmax - 1:
                S = {key: np.ascontiguousarray(v, dtype=np.float64).copy() for key, v in S.items()}
                obs = one(handoff, h2, t2, np.float64, ctypes.c_double, S, a, nz)
            else:
                obs = one(L, h_, t, dt, cdt, S, a, nz)
            if k + 1 in res:
                res[k + 1][0].append(i)
                o64 = obs[:nobs].astype(np.float64)
                res[k + 1][1].append(o64 if isinstance(ref, str) else np.abs(o64 - (g["obs"][i + k] if ref is None else ref[k + 1][1][len(res[k + 1][1])])))
                res[k + 1][2].append(float(gap[i:i + k + 1].min()))
    return {h: (np.array(v[0]), np.array(v[1]), np.array(v[2])) for h, v in res.items()}, CASES[family][3], g


def main(argv):
    fp64 = "--fp64" in argv
    every = int(argv[argv.index("--every") + 1]) if "--every" in argv else 1
    fams = [a for a in argv if a in FAMILY_TO_TASK] or list(FAMILY_TO_TASK)
    L = ctypes.CDLL(build(fp64))
    L.emu_create.restype = ctypes.c_void_p
    L.emu_create.argtypes = [ctypes.c_void_p] * 3
    rounded = "--rounded" in argv      # reference = the fp64 build of the same source stepped from the SAME fp32-rounded states (isolates arithmetic from input rounding)
    if rounded:
        keep = sys.argv[:]
        sys.argv = [a for a in sys.argv if not a.startswith("-D")]
        L64 = ctypes.CDLL(build(True))
        sys.argv = keep
        L64.emu_create.restype = ctypes.c_void_p
        L64.emu_create.argtypes = [ctypes.c_void_p] * 3
    if "--horizons" in argv and "--sensitivity" in argv:
        # How well-posed is each free-running START for any engine that keeps its state in fp32: the fp64 build of the engine source (the oracle's arithmetic to 5e-10,
        # tools/emu_fp64_check.py) against ITSELF, h steps free-running, from start states perturbed by one fp32 ulp (relative JITTER, worst of GRX_JITTER_TRIALS draws).  A start
        # where the reference algorithm's own answer moves by >= ILL (a tenth of north_star's tolerance) under that perturbation is recorded as ill-conditioned at that horizon;
        # tests/tolerance_cases.py::posed_starts leaves exactly those (and the starts within 1e-6 m of an activation switch) out of the strict 1e-4 claim at horizons 1 and 2.
        import json
        from tolerance_cases import CASES
        ILL, PROP, trials, hz = 1e-5, 5e-5, int(os.environ.get("GRX_JITTER_TRIALS", "6")), (1, 2)
        L64 = ctypes.CDLL(build(True)); L64.emu_create.restype = ctypes.c_void_p; L64.emu_create.argtypes = [ctypes.c_void_p] * 3
        sec = {"_what": (f"free-running starts (tests/tolerance_cases.py::horizon_errors) at which the fp64 build of the engine source, run against ITSELF from a start state perturbed by a "
                         f"relative {JITTER:g} on every word (worst of {trials} draws; both runs from the fp32-rounded fixture state), moves by >= {ILL:g} on some component after h steps: "
                         "the reference ALGORITHM's own spread under one fp32 ulp of state -- such a start is ill-conditioned for every fp32-state engine.  Horizon 2 adds the starts at which the fp64 build, "
                         f"continued from the state the fp32 build of the same source reached after step 1, ends >= {PROP:g} from the fp64 build's own two-step answer (propagated_over_threshold): "
                         "the reference algorithm amplifying a one-step fp32 error to half the tolerance"), "threshold": ILL, "jitter": JITTER, "trials": trials}
        for fam in fams:
            if fam in ("FetchHullContacts", "AntMazeLarge"):
                continue
            ref, comps, g = run_family_horizons(L64, fam, hz, every, fp64=True, round_inputs=True, ref="raw")
            worst = {h: None for h in hz}
            for trial in range(1, 1 + trials):
                res = run_family_horizons(L64, fam, hz, every, fp64=True, round_inputs=True, jitter=trial, ref=ref)[0]
                for h in hz:
                    worst[h] = res[h][1] if worst[h] is None else np.maximum(worst[h], res[h][1])
            # second yardstick at horizon 2: the state an fp32 engine carries INTO its second step differs from the reference's by that engine's one-step error (measured: up to 1e-5 on a
            # velocity), not by one ulp.  The fp32 build of the engine source runs step 1, the fp64 build continues from THAT state: where its answer is >= PROP from the fp64 build's own
            # two-step answer, the reference algorithm itself maps a one-step fp32 error to half the tolerance or more (profiles/fetchslide_start169_r06.txt is such a start).
            prop = run_family_horizons(L, fam, (2,), every, fp64=False, handoff=L64, ref={2: ref[2]})[0][2]
            sec[fam] = {}
            for h in hz:
                idx = ref[h][0]
                per = {}
                for comp, cols in comps.items():
                    per[comp] = (worst[h][:, cols] / np.maximum(1.0, np.abs(ref[h][1][:, cols]))).max(axis=1) if comp.endswith("_relative") else worst[h][:, cols].max(axis=1)
                allc = np.max(np.stack(list(per.values())), axis=0)
                bad = allc >= ILL
                extra = {}
                if h == 2:
                    pp = np.max(np.stack([(prop[1][:, cols] / np.maximum(1.0, np.abs(ref[h][1][:, cols]))).max(axis=1) if comp.endswith("_relative") else prop[1][:, cols].max(axis=1)
                                          for comp, cols in comps.items()]), axis=0)
                    extra = {"propagated_threshold": PROP, "propagated_over_threshold": [int(i) for i in idx[pp >= PROP]], "propagated_p50": float(np.median(pp)), "propagated_max": float(pp.max())}
                    print(f"{fam:18s} h=2 one-step fp32 error propagated by the fp64 algorithm: p50 {np.median(pp):.1e} max {pp.max():.1e}; >= {PROP:g}: {list(map(int, idx[pp >= PROP]))}", flush=True)
                    bad = bad | (pp >= PROP)
                ill = idx[bad]
                sec[fam][str(h)] = {"n_starts": int(len(idx)), "ill_conditioned_starts": [int(i) for i in ill], **extra,
                                    **{comp: {"p50": float(np.median(e)), "p99": float(np.quantile(e, .99)), "max": float(e.max()), "n_over_1e-4": int((e >= 1e-4).sum())} for comp, e in per.items()}}
                print(f"{fam:18s} h={h} starts {len(idx):4d} ill-conditioned (spread >= {ILL:g}) {len(ill):3d} {list(map(int, ill))[:30]} | " +
                      "  ".join(f"{comp} p50 {np.median(e):.1e} max {e.max():.1e}" for comp, e in per.items()), flush=True)
        if "--json" in argv:
            path = argv[argv.index("--json") + 1]
            with open(path) as f:
                full = json.load(f)
            full.setdefault("reference_sensitivity_horizons", {}).update(sec)
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
        return
    if "--horizons" in argv:      # free-running parity on the CPU (tests/tolerance_cases.py::horizon_errors is the GPU measurement)
        for fam in fams:
            if fam in ("FetchHullContacts", "AntMazeLarge"):
                continue
            res, comps, g = run_family_horizons(L, fam, every=every)
            for h, (idx, e, gp) in res.items():
                posed = gp >= 1e-6
                for comp, cols in comps.items():
                    err = (e[:, cols] / np.maximum(1.0, np.abs(g["obs"][idx + h - 1][:, cols]))).max(axis=1) if comp.endswith("_relative") else e[:, cols].max(axis=1)
                    print(f"{fam:18s} h={h:2d} {comp:26s} starts {len(err):4d} posed {int(posed.sum()):4d} p50 {np.median(err):.1e} p90 {np.quantile(err, .9):.1e} max {err.max():.1e} within 1e-4: {100 * np.mean(err < 1e-4):5.1f} % | posed: "
                          f"max {err[posed].max():.1e} within {100 * np.mean(err[posed] < 1e-4):5.1f} %", flush=True)
        return
    if "--sensitivity" in argv:      # the fp64 build against ITSELF from states jittered by one fp32 ulp: how well-posed each snapshot is for any fp32-state engine
        L64 = ctypes.CDLL(build(True)); L64.emu_create.restype = ctypes.c_void_p; L64.emu_create.argtypes = [ctypes.c_void_p] * 3
        for fam in fams:
            ref = run_family(L64, fam, True, every, round_inputs=True, ref="raw")[1]
            worst = None
            for trial in range(1, 1 + int(os.environ.get("GRX_JITTER_TRIALS", "3"))):
                idx, e, status, comps = run_family(L64, fam, True, every, round_inputs=True, ref=ref, jitter=trial)
                worst = e if worst is None else np.maximum(worst, e)
            for comp, cols in comps.items():
                err = worst[:, cols].max(axis=1)
                _SENS.setdefault(fam, {})[comp] = {"p50": float(np.median(err)), "p90": float(np.quantile(err, .9)), "p99": float(np.quantile(err, .99)), "max": float(err.max()),
                                                  "frac_within_1e-4": float(np.mean(err < 1e-4)), "n": int(len(err)), "n_over_1e-4": int((err >= 1e-4).sum())}
                print(f"{fam:18s} {comp:26s} SENSITIVITY n={len(err):4d} p50 {np.median(err):.1e} p99 {np.quantile(err, .99):.1e} max {err.max():.1e} over 1e-4: {int((err >= 1e-4).sum())} {list(idx[err >= 1e-4])[:40]}", flush=True)
        if "--json" in argv:      # merged into tests/golden/tolerance_table.json under "reference_sensitivity" (the yardstick beside the GPU-measured errors)
            import json
            path = argv[argv.index("--json") + 1]
            with open(path) as f:
                full = json.load(f)
            sec = full.setdefault("reference_sensitivity", {})
            sec["_what"] = (f"per-snapshot spread of the fp64 build of the engine source (agrees with the oracle to 5e-10: tools/emu_fp64_check.py) against ITSELF when every word of the "
                            f"input state is perturbed by a relative {JITTER:g} (worst of {os.environ.get('GRX_JITTER_TRIALS', '3')} draws), both runs from the fp32-rounded fixture states: how far the reference ALGORITHM's own answer "
                            "moves under a perturbation of the size of one fp32 ulp of the state -- what no fp32-state engine can avoid")
            sec.update(_SENS)
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
        return
    for fam in fams:
        ref = run_family(L64, fam, True, every, round_inputs=True, ref="raw")[1] if rounded else None
        L.emu_newton_stat.restype = ctypes.c_long
        n0 = [L.emu_newton_stat(k) for k in range(6)]
        idx, e, status, comps = run_family(L, fam, fp64, every, round_inputs=rounded, ref=ref)
        n1 = [L.emu_newton_stat(k) - n0[k] for k in range(6)]
        if hasattr(L, "emu_cert_stat"):
            L.emu_cert_stat.restype = ctypes.c_long
            cs = [L.emu_cert_stat(k) for k in range(4)]
            cs0 = getattr(main, "_cs", [0, 0, 0, 0]); main._cs = cs
            d = [a - b for a, b in zip(cs, cs0)]
            print(f"{fam:18s} certified fp32 portal: primitive pairs {d[0]} attempts, {d[1]} undecided ({100 * d[1] / max(d[0], 1):.1f} %); hull pairs {d[2]} attempts, {d[3]} undecided ({100 * d[3] / max(d[2], 1):.1f} %)", flush=True)
        print(f"{fam:18s} Newton: {n1[0]} solves, {n1[1] / max(n1[0], 1):.3f} iterations / solve, {n1[2] / max(n1[0], 1):.3f} Hessian assemblies / solve, {n1[3] / max(n1[0], 1):.3f} incremental updates / solve, object-block refinements eligible {n1[4] / max(n1[0], 1):.3f} / run {n1[5] / max(n1[0], 1):.3f} per solve", flush=True)
        for comp, cols in comps.items():
            err = e[:, cols].max(axis=1)      # absolute, also for the "_relative" components of tests/tolerance_cases.py
            worst = idx[np.argsort(-err)[:6]]
            print(f"{fam:18s} {comp:26s} n={len(err):4d} p50 {np.median(err):.1e} p90 {np.quantile(err, .9):.1e} p99 {np.quantile(err, .99):.1e} max {err.max():.1e} "
                  f"within 1e-4: {100 * np.mean(err < 1e-4):5.1f} %  over: {int((err >= 1e-4).sum())}  worst snapshots {list(worst)}  status!=0: {int((status != 0).sum())}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
