"""GPU numerics tests of the two MI355X-specific primitives against plain fp64 numpy:
the register-resident SPD solve (v_readlane elimination) and the f32-MFMA Hessian assembly."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(mode, nv, nefc, A, b, J, D):
    import torch

    from gymnasium_robotics_amd import _native

    L = _native.lib()
    L.grx_debug_primitive.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    tA, tb, tJ, tD = t(A), t(b), t(J), t(D)
    out = torch.zeros(nv * nv + nv if mode else nv, device="cuda")
    _native.check(L.grx_debug_primitive(mode, nv, nefc, tA.data_ptr(), tb.data_ptr(), tJ.data_ptr(), tD.data_ptr(), out.data_ptr(),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("nv", [14, 15, 21, 24, 30, 9])   # 14..30: register-resident instantiations; 9: LDS fallback
def test_spd_solve(nv):
    rng = np.random.default_rng(nv)
    Q = rng.standard_normal((nv, nv))
    A = Q @ Q.T + nv * np.eye(nv)
    A[0, 0] += 2e8  # the Fetch base slides carry h*damping = 2e8 on the diagonal
    b = rng.standard_normal(nv) * 10
    x = _run(0, nv, 0, A, b, np.zeros((1, nv)), np.zeros(1))
    ref = np.linalg.solve(A, b)
    assert np.abs(x - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("nv,nefc", [(21, 47), (15, 7), (21, 144)])
def test_mfma_hessian(nv, nefc):
    rng = np.random.default_rng(nefc)
    Q = rng.standard_normal((nv, nv))
    M = Q @ Q.T + nv * np.eye(nv)
    J = rng.standard_normal((nefc, nv)) * (rng.random((nefc, nv)) < 0.6)
    D = rng.random(nefc) * 100 + 1
    D[rng.random(nefc) < 0.3] *= -1  # negative = inactive row
    out = _run(1, nv, nefc, M, np.zeros(nv), J, D)
    H, jtf = out[: nv * nv].reshape(nv, nv), out[nv * nv:]
    ref = M + J.T @ (np.where(D > 0, D, 0)[:, None] * J)
    assert np.abs(H - ref).max() < 1e-5 * np.abs(ref).max()
    # the same MFMA chain also delivers J' f for the row forces of the last evaluation (here f_r = 0.5 + 0.01 r)
    f = 0.5 + 0.01 * np.arange(nefc)
    assert np.abs(jtf - J.T @ f).max() < 1e-5 * max(1.0, np.abs(J.T @ f).max())


@pytest.mark.parametrize("nv,nefc", [(30, 70), (24, 33)])
def test_mfma_hessian_two_span_rows(nv, nefc):
    """Rows stored as two dof spans (hand models: wrist..finger | object), read back through grx_row_pos in the MFMA operand fetch."""
    rng = np.random.default_rng(nv + nefc)
    Q = rng.standard_normal((nv, nv))
    M = Q @ Q.T + nv * np.eye(nv)
    J = rng.standard_normal((nefc, nv))
    J[:, nv // 3: nv - nv // 3] = 0.0          # the gap between the two spans carries no entries
    D = rng.random(nefc) * 100 + 1
    D[rng.random(nefc) < 0.3] *= -1
    out = _run(2, nv, nefc, M, np.zeros(nv), J, D)
    H, jtf = out[: nv * nv].reshape(nv, nv), out[nv * nv:]
    ref = M + J.T @ (np.where(D > 0, D, 0)[:, None] * J)
    assert np.abs(H - ref).max() < 1e-5 * np.abs(ref).max()
    f = 0.5 + 0.01 * np.arange(nefc)
    assert np.abs(jtf - J.T @ f).max() < 1e-5 * max(1.0, np.abs(J.T @ f).max())
