"""same-step autoreset checked against next-step autoreset of a second environment with the same seeds and actions (GPU tests).

At the step in which a world reaches its time limit the same-step environment returns the RESET observation and reports the terminal one in
info["final_obs"]; the next-step environment returns the terminal observation and resets the world (ignoring its action) in the following call.
Both draw the reset from the world's own generator, so up to a world's first time limit the two must agree row by row:
    same_step.final_obs[t] == next_step.obs[t],  same_step.obs[t] == next_step.obs[t + 1]   (rows at their limit in step t),
reward / success of step t equal, every other row identical.  This runs the hot path of the benchmark (pinned staging, the reset / commit kernels,
the overlapped settle chains of the manipulate family)."""
import numpy as np


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def check_same_step_against_next_step(make, horizon, steps, act_dim, tol=0.0, output="torch", seed=5, tol_max=None, outlier_rows=0.0, touch_from=None):
    """tol = 0: bit-equal.  tol > 0: every row within tol, except at most a fraction `outlier_rows` of the compared rows, which stay within tol_max (a rolling
    object whose contact set flips under the different warm start of the overlapped settle).  touch_from: observation columns from
    there on are touch-sensor FORCES (newtons, up to tens): compared relative to 100 * max(1, |reading|), i.e. 2e-2 N on a 1 N reading at tol = 2e-4 (the
    two paths stop their Newton solves at different iterates)"""
    A, B = make(autoreset_mode="same_step", max_episode_steps=horizon, output=output), make(autoreset_mode="next_step", max_episode_steps=horizon, output=output)
    n = A.num_envs
    A.reset(seed=seed); B.reset(seed=seed)
    stag = np.arange(n) % horizon
    A._elapsed[:] = stag; B._elapsed[:] = stag
    rng = np.random.default_rng(seed)
    alive, pending, n_checked = np.ones(n, bool), None, 0
    def close(x, y):
        if tol == 0.0:
            return np.array_equal(x, y)
        if len(x) == 0:
            return True
        e = np.abs(np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64)).reshape(len(x), -1)
        if touch_from is not None and e.shape[1] > touch_from:
            e[:, touch_from:] /= 100.0 * np.maximum(1.0, np.abs(np.asarray(y, dtype=np.float64).reshape(len(x), -1)[:, touch_from:]))
        e = e.max(axis=1, initial=0.0)
        return ((e > tol).sum() <= max(outlier_rows * len(e), 1.0 if outlier_rows else 0.0) and e.max() <= (tol_max or tol))
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
        if output == "torch":
            import torch
            a = torch.from_numpy(a).to(A.device)
        oa, ra, ta, tra, ia = A.step(a)
        oa = {k: _np(v).copy() for k, v in oa.items()}; ra, tra = _np(ra).copy(), _np(tra).astype(bool)
        fin = {k: _np(v).copy() for k, v in ia["final_obs"].items()} if "final_obs" in ia else None
        ob, rb, tb, trb, ib = B.step(a)
        ob = {k: _np(v).copy() for k, v in ob.items()}; rb, trb = _np(rb).copy(), _np(trb).astype(bool)
        if pending is not None:                       # rows that hit their limit in the previous step: B has just reset them
            rows, obs_reset = pending
            for k in obs_reset:
                assert close(ob[k][rows], obs_reset[k]), (t, k, "reset observation", float(np.abs(ob[k][rows].astype(np.float64) - obs_reset[k]).max()),
                                                          np.argwhere(np.abs(ob[k][rows].astype(np.float64) - obs_reset[k]) > tol)[:6].tolist())
            n_checked += len(rows)
            pending = None
        assert np.array_equal(tra[alive], trb[alive]) and not _np(ta).any()
        done, cont = alive & tra, alive & ~tra
        for k in oa:
            assert close(oa[k][cont], ob[k][cont]), (t, k, "rows in step")
        assert close(ra[alive], rb[alive]), (t, "reward")
        if done.any():
            rows_all = np.nonzero(tra)[0]             # final_obs lists every row at its limit, in world order
            sel = np.isin(rows_all, np.nonzero(done)[0])
            for k in fin:
                assert close(fin[k][sel], ob[k][done]), (t, k, "final_obs")
            pending = (np.nonzero(done)[0], {k: v[done] for k, v in oa.items()})
            alive = alive & ~done
    assert n_checked >= n // 2, "the rollout must cross the time limit of most worlds"
    return A, B
