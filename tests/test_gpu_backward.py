"""GPU parity of the hand-written backward kernels against torch autograd on a plain float32 torch restatement
of the same forward (floating-point kernels: torch reference, per the parity rules).  Tolerances are relative to
the gradient scale and written next to each check."""
import numpy as np
import pytest
import torch

import svs_oracle as orc
import synth

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from svs_hip import ops as _ops
    return _ops


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def torch_composite(z, sdf, rgb, beta_param, ds, beta_min=1e-4):
    """network.py:281-295 + :237-243 in plain torch (float64 for a clean autograd reference)."""
    beta = beta_param.abs() + beta_min
    sigma = (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    fe = dists * sigma
    sfe = torch.cat([torch.zeros_like(fe[:, :1]), fe[:, :-1]], -1)
    w = (1 - torch.exp(-fe)) * torch.exp(-torch.cumsum(sfe, -1))
    rgb_values = (w.unsqueeze(-1) * rgb).sum(1)
    depth_values = ds * ((w * z).sum(1, keepdim=True) / (w.sum(1, keepdim=True) + 1e-8))
    return w, rgb_values, depth_values


@pytest.mark.parametrize("beta", [0.1, 0.02])
def test_composite_backward(dev, ops, beta):
    rng = np.random.default_rng(5)
    R, S = 64, 98
    z = np.sort(rng.uniform(0.5, 5.5, (R, S)), -1).astype(F32)
    sdf = (rng.normal(0.2, 0.4, (R, S)) - np.linspace(0, 0.8, S)[None]).astype(F32)
    rgb = rng.uniform(0, 1, (R, S, 3)).astype(F32)
    ds = rng.uniform(0.8, 1.0, (R, 1)).astype(F32)
    g_rgb = rng.normal(0, 1, (R, 3)).astype(F32)
    g_w = rng.normal(0, 1, (R, S)).astype(F32)
    g_d = rng.normal(0, 1, (R, 1)).astype(F32)
    T = lambda a, rg=False: torch.tensor(a, dtype=torch.float64, requires_grad=rg)
    tz, tsdf, trgb, tb = T(z), T(sdf, True), T(rgb, True), torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    w, rv, dv = torch_composite(tz, tsdf, trgb, tb, T(ds))
    ((rv * T(g_rgb)).sum() + (w * T(g_w)).sum() + (dv * T(g_d)).sum()).backward()
    d_sdf, d_rgb, d_beta = ops.composite_bwd(G(z, dev), G(sdf.reshape(-1, 1), dev), G(rgb.reshape(-1, 3), dev), G(ds, dev),
                                             torch.tensor(beta, device=dev), 1e-4, G(g_rgb, dev), G(g_w, dev), G(g_d, dev))
    assert rel_err(d_rgb.cpu().numpy().reshape(R, S, 3), trgb.grad.numpy()) < 1e-5
    assert rel_err(d_sdf.cpu().numpy().reshape(R, S), tsdf.grad.numpy()) < 2e-5
    assert abs(float(d_beta) - float(tb.grad)) / abs(float(tb.grad)) < 2e-5


@pytest.mark.parametrize("P", [32, 1000, 5000])
def test_wgrad_gemm(dev, P):
    """dW = A B^T over points, with the optional softplus' factor and the second operand pair."""
    import ctypes
    from svs_hip import lib
    L = lib.load()
    rng = np.random.default_rng(P)
    A0, B0 = rng.normal(0, 1, (P, 256)).astype(F32), rng.normal(0, 1, (P, 256)).astype(F32)
    A1, B1 = rng.normal(0, 1, (P, 256)).astype(F32), rng.normal(0, 1, (P, 256)).astype(F32)
    H1 = rng.uniform(0, 0.05, (P, 256)).astype(F32)
    ta0, tb0, ta1, tb1, th1 = (G(synth.rows_to_tiles(x), dev) for x in (A0, B0, A1, B1, H1))
    dW = torch.zeros(256, 256, device=dev)
    db = torch.zeros(256, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    st = 128 * 64
    lib.check(L.svs_wgrad(ptr(ta0), None, ptr(tb0), st, 0, st, ptr(ta1), ptr(th1), ptr(tb1), st, st, st, None, 0, P,
                          ptr(dW), 256, ptr(db), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ref = A0.astype(np.float64).T @ B0 + (A1 * (1 - np.exp(-100.0 * H1.astype(np.float64)))).T @ B1
    assert rel_err(dW.cpu().numpy(), ref) < 2e-5
    assert rel_err(db.cpu().numpy(), A0.astype(np.float64).sum(0)) < 2e-5
