"""Joint-box gates of hull pairs (compile time, cold).

A hull-vs-hull / hull-vs-box candidate pair whose two bodies are separated by at most three hinge / slide joints has a relative pose that is a function
of those joint values alone.  For such a pair this module PROVES, on a grid over the joint values, a box [lo_i, hi_i] of joint values inside which the
two (margin-inflated) geoms are disjoint:

    distance(q) >= L(q_cell) - sum_i R_i * h_i / 2          for every q in the cell of half-widths h_i / 2 around q_cell

where L is a rigorous lower bound of the hull distance at the cell centre (any direction d gives one: min_b b.d - max_a a.d; the direction comes from
Frank-Wolfe iterations on the Minkowski difference) and R_i bounds the speed of any vertex of the distal geom per unit of joint i (triangle inequality
along the chain: no sampling).  The device engine tests the ACTUAL joint values of the world against the box in its candidate sweep
(csrc/grx_engine.h, grx_gate_clear): inside the box the pair cannot produce a contact within its margin, so dropping it there changes nothing -- the
reference (mujoco.mj_collision under envs/robot_env.py:341) would find no contact either; outside the box the pair goes through the filter and
the narrow phase as before.  Nothing is assumed about joint limits.  The checker (oracle/) does not know about gates: the parity tests compare gated
kernels with an ungated restatement.

Why: the Fetch arm has one hull pair (torso_lift_link / shoulder_lift_link, 1.9 cm apart in every arm pose the tasks reach) that passes the bounding-
box filter in EVERY substep of EVERY world; it made the wave-cooperative hull routine a hot path (4 % of a world's time, and the register budget of the
whole step kernel).  With its gate (|shoulder_pan| < 1.3 rad) the routine runs in the 7 % of the world-steps in which the upper arm is near the head.
"""
import numpy as np

GEOM_BOX, GEOM_MESH = 6, 7
JNT_SLIDE, JNT_HINGE = 2, 3
MAX_GATE_DOFS = 3


def _q2m(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def _axis_rot(ax, th):
    ax = ax / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    th = np.asarray(th)[:, None, None]
    return np.eye(3)[None] + np.sin(th) * K[None] + (1 - np.cos(th)) * (K @ K)[None]


class _Tree:
    def __init__(self, T):
        self.T = T
        self.par = np.asarray(T["body_parent"]).ravel()
        self.bp = np.asarray(T["body_pos"], float).reshape(-1, 3)
        self.bq = np.asarray(T["body_quat"], float).reshape(-1, 4)
        self.jadr, self.jnum = np.asarray(T["body_jntadr"]).ravel(), np.asarray(T["body_jntnum"]).ravel()
        self.jt = np.asarray(T["jnt_type"]).ravel()
        self.jp = np.asarray(T["jnt_pos"], float).reshape(-1, 3)
        self.ja = np.asarray(T["jnt_axis"], float).reshape(-1, 3)
        self.jq = np.asarray(T["jnt_qposadr"]).ravel()
        self.q0 = np.asarray(T["qpos0"], float).ravel()

    def path_up(self, b, stop):
        out = []
        while b != stop:
            out.append(int(b))
            b = int(self.par[b])
        return out

    def ancestors(self, b):
        out = [int(b)]
        while b != 0:
            b = int(self.par[b])
            out.append(b)
        return out

    def pose_in(self, anc, b, qs, n):
        """pose of body b in the frame of its ancestor anc, for joint values qs (joint id -> [n] array of qpos values): R [n,3,3], p [n,3]"""
        R = np.tile(np.eye(3), (n, 1, 1))
        p = np.zeros((n, 3))
        for x in reversed(self.path_up(b, anc)):
            p = p + R @ self.bp[x]
            R = R @ _q2m(self.bq[x])
            for j in range(int(self.jadr[x]), int(self.jadr[x]) + int(self.jnum[x])):
                th = np.asarray(qs[j]) - self.q0[self.jq[j]]
                if self.jt[j] == JNT_HINGE:
                    Rj = _axis_rot(self.ja[j], th)
                    anchor = self.jp[j]
                    p = p + np.einsum("nij,j->ni", R, anchor) - np.einsum("nij,njk,k->ni", R, Rj, anchor)
                    R = R @ Rj
                else:
                    p = p + np.einsum("nij,j->ni", R, self.ja[j] / np.linalg.norm(self.ja[j])) * th[:, None]
        return R, p


def _distance_lower_bound(A, B, R, p, iters):
    """A, B: vertex clouds in two frames, (R, p) [n]: pose of B's frame in A's.  Lower bound of the distance between the two hulls per pose: the separation
    along the best direction met by Frank-Wolfe iterations on A - B (every direction gives a valid bound)."""
    n = len(p)
    v = A.mean(0)[None] - (np.einsum("nij,j->ni", R, B.mean(0)) + p)
    lb = np.full(n, -np.inf)
    for _ in range(iters):
        d = -v
        ia = np.argmax(d @ A.T, axis=1)
        ib = np.argmin(np.einsum("nji,nj->ni", R, d) @ B.T, axis=1)
        w = A[ia] - (np.einsum("nij,nj->ni", R, B[ib]) + p)
        nv = np.linalg.norm(v, axis=1)
        lb = np.maximum(lb, np.einsum("ni,ni->n", v, w) / np.maximum(nv, 1e-30))
        dv = w - v
        t = np.clip(-np.einsum("ni,ni->n", v, dv) / np.maximum(np.einsum("ni,ni->n", dv, dv), 1e-30), 0.0, 1.0)
        v = v + t[:, None] * dv
    return lb


def _geom_cloud(T, g):
    """vertices of the geom's convex shape in its BODY frame (hull of a mesh, corners of a box), or None"""
    gt = int(np.asarray(T["geom_type"]).ravel()[g])
    gp = np.asarray(T["geom_pos"], float).reshape(-1, 3)[g]
    gq = np.asarray(T["geom_quat"], float).reshape(-1, 4)[g]
    if gt == GEOM_MESH:
        a, n = int(np.asarray(T["geom_hulladr"]).ravel()[g]), int(np.asarray(T["geom_hullnum"]).ravel()[g])
        if a < 0 or n <= 0:
            return None
        v = np.asarray(T["mesh_vert"], float).reshape(-1, 3)[a:a + n]
    elif gt == GEOM_BOX:
        s = np.asarray(T["geom_size"], float).reshape(-1, 3)[g]
        v = np.array([[sx * s[0], sy * s[1], sz * s[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    else:
        return None
    return v @ _q2m(gq).T + gp


def compute_pair_gates(T, cells2=256, cells3=40, iters2=120, iters3=48, safety=5e-4, max_width=None, verbose=None, max_gates=64):
    """T: the table dict of a compiled model (after `devpair` is final).  Returns (devpair_gate [ndevpair] int32, gate_qadr [3 * ngate] int32, gate_box [6 * ngate] float64,
    report list).  A gate = up to three (qpos address, lo, hi); unused slots have address -1.  At most `max_gates` gates are kept (the engine evaluates them all once
    per collision pass, one lane each): the pairs that are NEAREST at the model's reference configuration -- a far pair fails the bounding-sphere test anyway."""
    tr = _Tree(T)
    dp = np.asarray(T["devpair"]).ravel().astype(int)
    g1s, g2s = np.asarray(T["pair_geom1"]).ravel(), np.asarray(T["pair_geom2"]).ravel()
    gt, gb = np.asarray(T["geom_type"]).ravel(), np.asarray(T["geom_bodyid"]).ravel()
    margin = np.asarray(T["pair_margin"], float).ravel()
    rng = np.asarray(T["jnt_range"], float).reshape(-1, 2)
    lim = np.asarray(T["jnt_limited"]).ravel()
    mocap = np.asarray(T["body_mocapid"]).ravel() if "body_mocapid" in T else -np.ones(len(tr.par), int)
    bshift = np.asarray(T["body_shift"]).ravel() if "body_shift" in T and len(np.asarray(T["body_shift"]).ravel()) else np.zeros(len(tr.par), int)
    gshift = np.asarray(T["geom_shift"]).ravel() if "geom_shift" in T and len(np.asarray(T["geom_shift"]).ravel()) else np.zeros(len(gt), int)
    devpair_gate = -np.ones(len(dp), np.int32)
    gate_qadr, gate_box, report, nearness = [], [], [], []
    clouds = {}
    for k, p in enumerate(dp):
        ga, gbm = int(g1s[p]), int(g2s[p])
        if gt[gbm] != GEOM_MESH or gt[ga] not in (GEOM_MESH, GEOM_BOX):
            continue
        if gshift[ga] or gshift[gbm]:
            continue
        a, b = int(gb[ga]), int(gb[gbm])
        if a == b:      # (a world-fixed geom against a body at most three joints from the world is a pair like any other)
            continue
        anc_a = tr.ancestors(a)
        common = next(x for x in tr.ancestors(b) if x in anc_a)
        side_a, side_b = tr.path_up(a, common), tr.path_up(b, common)      # bodies whose joints move A / B relative to the common ancestor
        if any(mocap[x] >= 0 or bshift[x] for x in side_a + side_b):
            continue
        joints = []        # (joint, moves B?, bodies distal to the joint's body on the way to the geom)
        ok = True
        for side, moves_b in ((side_a, False), (side_b, True)):
            for pos, x in enumerate(side):
                for j in range(int(tr.jadr[x]), int(tr.jadr[x]) + int(tr.jnum[x])):
                    if tr.jt[j] not in (JNT_SLIDE, JNT_HINGE):
                        ok = False
                    joints.append((j, moves_b, side[:pos]))
        if not ok or not joints or len(joints) > MAX_GATE_DOFS:
            continue
        for g in (ga, gbm):
            if g not in clouds:
                clouds[g] = _geom_cloud(T, g)
        A, B = clouds[ga], clouds[gbm]
        if A is None or B is None:
            continue
        # grid of joint values: the joint's range widened by 0.3 (rad / a tenth of that for slides), the full turn for an unlimited hinge, +- 5 cm for an unlimited slide
        axes = []
        ncell = cells2 if len(joints) <= 2 else cells3
        for j, moves_b, distal in joints:
            q0 = tr.q0[tr.jq[j]]
            if tr.jt[j] == JNT_HINGE:
                lo, hi = (rng[j, 0] - 0.3, rng[j, 1] + 0.3) if lim[j] else (q0 - np.pi, q0 + np.pi)
            else:
                lo, hi = (rng[j, 0] - 0.03, rng[j, 1] + 0.03) if lim[j] else (q0 - 0.05, q0 + 0.05)
            if max_width is not None:
                lo, hi = max(lo, q0 - max_width), min(hi, q0 + max_width)
            axes.append(np.linspace(lo, hi, ncell + 1))
        extent = {j: float(np.abs(e - tr.q0[tr.jq[j]]).max()) for (j, _, _), e in zip(joints, axes)}      # slides: largest travel inside the grid
        # speed bound of any vertex of the distal geom per unit of each joint: triangle inequality along the chain from the joint's anchor to the vertex
        Rb = []
        for j, moves_b, distal in joints:
            if tr.jt[j] == JNT_SLIDE:
                Rb.append(1.0)
                continue
            cloud = B if moves_b else A
            reach = np.linalg.norm(tr.jp[j]) + np.linalg.norm(cloud, axis=1).max()
            x = int(np.asarray(T["jnt_bodyid"]).ravel()[j])
            later = [jj for jj in range(j + 1, int(tr.jadr[x]) + int(tr.jnum[x]))]                                   # joints of the same body applied after this one
            between = [jj for y in distal for jj in range(int(tr.jadr[y]), int(tr.jadr[y]) + int(tr.jnum[y]))]
            reach += sum(np.linalg.norm(tr.bp[y]) for y in distal)
            for jj in later + between:
                reach += 2.0 * np.linalg.norm(tr.jp[jj]) if tr.jt[jj] == JNT_HINGE else extent[jj]
            Rb.append(float(reach))
        centres = [0.5 * (e[1:] + e[:-1]) for e in axes]
        widths = [e[1] - e[0] for e in axes]
        mesh = np.meshgrid(*centres, indexing="ij")
        shape = mesh[0].shape
        flat = [x.ravel() for x in mesh]
        n = flat[0].size
        slack = sum(r * w * 0.5 for r, w in zip(Rb, widths))
        clear = np.zeros(n, bool)
        lbs = []
        for c0 in range(0, n, 16384):
            sl = slice(c0, min(n, c0 + 16384))
            qs = {j: f[sl] for (j, _, _), f in zip(joints, flat)}
            m_ = sl.stop - sl.start
            Ra, pa = tr.pose_in(common, a, qs, m_)
            Rbm, pb = tr.pose_in(common, b, qs, m_)
            Rrel = np.einsum("nji,njk->nik", Ra, Rbm)                     # pose of b's frame in a's
            prel = np.einsum("nji,nj->ni", Ra, pb - pa)
            lb = _distance_lower_bound(A, B, Rrel, prel, iters2 if len(joints) <= 2 else iters3)
            lbs.append(lb)
            clear[sl] = lb - slack > margin[p] + safety
        lb_all = np.concatenate(lbs).reshape(shape)
        clear = clear.reshape(shape)
        home = tuple(int(np.clip(np.searchsorted(e, tr.q0[tr.jq[j]]) - 1, 0, len(e) - 2)) for (j, _, _), e in zip(joints, axes))
        if not clear[home]:
            report.append((int(p), ga, gbm, [j for j, _, _ in joints], None, float(slack)))
            continue
        lo_i, hi_i = list(home), list(home)      # grow the box of clear cells around the home cell, one face at a time
        grew = True
        while grew:
            grew = False
            for d in range(len(joints)):
                for sgn in (-1, 1):
                    idx = [slice(l, h + 1) for l, h in zip(lo_i, hi_i)]
                    if sgn < 0 and lo_i[d] > 0:
                        idx[d] = lo_i[d] - 1
                        if clear[tuple(idx)].all():
                            lo_i[d] -= 1
                            grew = True
                    elif sgn > 0 and hi_i[d] < shape[d] - 1:
                        idx[d] = hi_i[d] + 1
                        if clear[tuple(idx)].all():
                            hi_i[d] += 1
                            grew = True
        box = [(float(e[l]) + 1e-6, float(e[h + 1]) - 1e-6) for e, l, h in zip(axes, lo_i, hi_i)]      # (the device compares fp32 copies of the bounds)
        devpair_gate[k] = len(gate_qadr) // 3
        nearness.append(float(lb_all[home]))
        qa = [int(tr.jq[j]) for j, _, _ in joints] + [-1] * (3 - len(joints))
        bx = box + [(-1e30, 1e30)] * (3 - len(joints))
        gate_qadr += qa
        gate_box += [v for lohi in bx for v in lohi]
        report.append((int(p), ga, gbm, [j for j, _, _ in joints], box, float(slack)))
        if verbose:
            verbose(report[-1])
    gate_qadr, gate_box = np.array(gate_qadr, np.int32).reshape(-1, 3), np.array(gate_box, np.float64).reshape(-1, 6)
    if len(gate_qadr) > max_gates:      # keep the gates of the nearest pairs, renumbered
        keep = np.sort(np.argsort(np.array(nearness), kind="stable")[:max_gates])
        renum = -np.ones(len(gate_qadr), np.int32)
        renum[keep] = np.arange(len(keep), dtype=np.int32)
        devpair_gate = np.where(devpair_gate >= 0, renum[np.maximum(devpair_gate, 0)], -1).astype(np.int32)
        gate_qadr, gate_box = gate_qadr[keep], gate_box[keep]
    return devpair_gate, gate_qadr.reshape(-1), gate_box.reshape(-1), report
