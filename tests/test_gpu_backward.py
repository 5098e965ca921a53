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
    import torch_ref
    return torch_ref.composite(z, sdf, rgb, beta_param, ds, beta_min)


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


def _operands(precision, A0, B0, A1, B1, dev):
    """Device blocks of the two operand pairs in the form the kernels of that precision read, their records, and the values
    those blocks really hold (fp16x2: A of pair 0 and B of pair 1 are SCALED blocks under a per-point scale, A of pair 1 is
    stored unscaled, B of pair 0 is a PAIR block -- csrc/svs_blocks_h2.h; precision 1 reads both fp16 pieces of every
    operand, precision 2 = SVS_MMA_F16X2_HALF the hi pieces)."""
    if precision == 0:
        return [G(synth.rows_to_tiles(x), dev) for x in (A0, B0, A1, B1)], (None, None), (A0, B0, A1, B1)
    pair = precision == 1
    a0, r0, A0q = synth.rows_to_scaled_block(A0, pair=pair)
    b0 = synth.rows_to_pair_block(B0)
    a1, _, A1q = synth.rows_to_scaled_block(A1, scaled=False, pair=pair)
    b1, r1, B1q = synth.rows_to_scaled_block(B1, pair=pair)
    if pair:
        h = B0.astype(np.float16)
        B0q = h.astype(F32) + (B0 - h.astype(F32)).astype(np.float16).astype(F32)
    else:
        B0q = B0.astype(np.float16).astype(F32)          # the weight gradient reads the hi plane of B's pair block
    return [G(x, dev) for x in (a0, b0, a1, b1)], (G(r0, dev), G(r1, dev)), (A0q, B0q, A1q, B1q)


@pytest.mark.parametrize("precision,gscale", [(0, 1.0), (1, 1.0), (1, 3e-9), (1, 7e4), (2, 1.0), (2, 3e-9), (2, 7e4)])
@pytest.mark.parametrize("P", [32, 1000, 5000])
def test_wgrad_gemm(dev, P, precision, gscale):
    """dW = A B^T over points, with the second operand pair.  fp16x2: the gradient-like operands (A of pair 0, B of pair 1)
    at magnitudes far outside fp16's range; the reference is formed from the values the operand blocks hold (what the
    block formats cost is measured at the level of the whole backward, test_mlp_backward_vs_autograd)."""
    import ctypes
    from svs_hip import lib
    L = lib.load()
    rng = np.random.default_rng(P)
    A0, B0 = (gscale * rng.normal(0, 1, (P, 256))).astype(F32), rng.normal(0, 1, (P, 256)).astype(F32)
    A1, B1 = rng.normal(0, 1, (P, 256)).astype(F32), (gscale * rng.normal(0, 1, (P, 256))).astype(F32)
    # a heavy tail: a few points carry gradients 1000x the typical ones
    A0[:: 97] *= 1000.0
    B1[:: 89] *= 1000.0
    H1 = rng.uniform(0, 0.05, (P, 256)).astype(F32)
    absmax = torch.tensor([max(np.abs(A0).max(), np.abs(B1).max())], dtype=torch.float32, device=dev)
    A1 = (A1 * (1 - np.exp(-100.0 * H1.astype(np.float64)))).astype(F32)
    (ta0, tb0, ta1, tb1), (r0, r1), (A0q, B0q, A1q, B1q) = _operands(precision, A0, B0, A1, B1, dev)
    dW = torch.zeros(256, 256, device=dev)
    db = torch.zeros(256, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = 128 * 64
    lib.check(L.svs_wgrad(ptr(ta0), ptr(tb0), st, st, ptr(ta1), ptr(tb1), st, st, None, 0, P,
                          precision, ptr(absmax) if precision else None, ptr(r0), ptr(r1), ptr(dW), 256, ptr(db),
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ref = A0q.astype(np.float64).T @ B0q + A1q.astype(np.float64).T @ B1q
    assert rel_err(dW.cpu().numpy(), ref) < 2e-5
    assert rel_err(db.cpu().numpy(), A0q.astype(np.float64).sum(0)) < 2e-5
    # and the block formats themselves: two pieces are float32-class, one piece is within fp16's 2^-11 of the operands
    exact = A0.astype(np.float64).T @ B0 + A1.astype(np.float64).T @ B1
    assert rel_err(dW.cpu().numpy(), exact) < (1e-3 if precision == 2 else 5e-6)


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_wgrad_multi(dev, precision):
    """Several layers in one call, incl. the 16 extra B rows of the radiance network's first layer and a ragged tile."""
    import ctypes
    from svs_hip import lib
    L = lib.load()
    rng = np.random.default_rng(5)
    st = 128 * 64
    jobs, refs, outs, keep = [], [], [], []
    for j, (P, extra, two) in enumerate([(4000, True, False), (2500, False, True), (37, False, False)]):
        A0, B0 = (1e-6 * rng.normal(0, 1, (P, 256))).astype(F32), rng.normal(0, 1, (P, 256)).astype(F32)
        A1, B1 = rng.normal(0, 1, (P, 256)).astype(F32), (1e-6 * rng.normal(0, 1, (P, 256))).astype(F32)
        X = rng.normal(0, 1, (P, 32)).astype(F32); X[:, 16:] = 0
        (ta0, tb0, ta1, tb1), (r0, r1), (A0q, B0q, A1q, B1q) = _operands(precision, A0, B0, A1, B1, dev)
        # the extras block: one 32-row float32 tile (16 registers x 64 lanes) per 32 points, in both precisions
        xt = synth.rows_to_tiles(np.concatenate([X, np.zeros((P, 224), F32)], 1)).reshape(-1, 128 * 64)[:, :1024].copy()
        tx = G(xt, dev)
        dW = torch.zeros(256, 288, device=dev); db = torch.zeros(256, device=dev)
        am = torch.tensor([max(np.abs(A0).max(), np.abs(B1).max() if two else 0.0)], dtype=torch.float32, device=dev)
        keep += [ta0, tb0, ta1, tb1, tx, am, r0, r1]
        p = lambda t: t.data_ptr() if t is not None else None
        jobs.append(lib.WGradJob(p(ta0), p(tb0), st, st, p(ta1) if two else None, p(tb1) if two else None, st, st,
                                 p(tx) if extra else None, 1024, P, 288, p(dW), p(db), p(am) if precision else None,
                                 p(r0), p(r1) if two else None))
        ref = A0q.astype(np.float64).T @ B0q + (A1q.astype(np.float64).T @ B1q if two else 0.0)
        refx = A0q.astype(np.float64).T @ X[:, :16] if extra else None
        refs.append((ref, refx, A0q.astype(np.float64).sum(0))); outs.append((dW, db))
    arr = (lib.WGradJob * len(jobs))(*jobs)
    lib.check(L.svs_wgrad_multi(ctypes.cast(arr, ctypes.c_void_p), len(jobs), precision,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for (ref, refx, refb), (dW, db) in zip(refs, outs):
        got = dW.cpu().numpy()
        assert rel_err(got[:, :256], ref) < 2e-5
        if refx is not None:
            assert rel_err(got[:, 256:272], refx) < 2e-5
        assert rel_err(db.cpu().numpy(), refb) < 2e-5


# ------------------------------------------------------------------------------------------------------
# full MLP backward (double backward through the normals) against torch autograd in float64
# ------------------------------------------------------------------------------------------------------
import torch_ref as tref   # noqa: E402  (oracle/torch_ref.py: plain torch restatement with autograd)


def _t64(params, requires_grad=True):
    p = tref.to_torch(params, torch.float64, requires_grad)
    p["density.beta"].requires_grad_(False)
    return p


_sdf_mlp, _rgb_mlp = tref.sdf_mlp, tref.rgb_mlp


@pytest.mark.parametrize("precision", ["f16x2", "f16x2_half", "f32"])
def test_mlp_backward_vs_autograd(dev, ops, precision):
    """Forward + full backward of both MLPs (incl. the double backward through the normals) against float64 autograd, on
    the default fp16x2 kernels (gradient blocks stored as single fp16 pieces: bound 2e-3 of a tensor's largest entry) and
    on the exact float32-MFMA kernels behind SVS_MLP_PRECISION=f32 (bound 2e-5)."""
    from svs_hip.train import MlpBackward, TrainStreams
    prec = {"f16x2": ops.F16X2, "f16x2_half": ops.F16X2_HALF, "f32": ops.F32}[precision]
    # per-tensor max error over the tensor's max entry, against float64 autograd: the default fp16x2 path (both pieces of
    # every block) is held to the float32 class like the float32-MFMA kernels; the one-piece mode to fp16's 2^-11 class
    bound = {"f16x2": 3e-5, "f16x2_half": 2e-3, "f32": 2e-5}[precision]
    params = synth.make_params(0)
    rng = np.random.default_rng(77)
    K, pose = synth.make_camera()
    R, S, NE = 8, 32, 96
    uv = synth.make_uv(R, seed=3)
    dirs, cam, _ = orc.rays_from_uv(uv, pose, K)
    z = np.sort(rng.uniform(0.3, 5.9, (R, S)), -1).astype(F32)      # far samples leave the r=3 sphere -> clamp
    x_main = (cam[None, None] + z[:, :, None] * dirs[:, None, :]).astype(F32).reshape(-1, 3)
    x_eik = rng.uniform(-1.5, 1.5, (NE, 3)).astype(F32)
    Wr = rng.normal(0, 1, (R * S, 3)).astype(F32)
    ws = rng.normal(0, 1, (R * S, 1)).astype(F32)
    Wg = rng.normal(0, 0.1, (NE, 3)).astype(F32)

    # ---- torch float64 reference
    p = _t64(params)
    xm = torch.tensor(x_main.astype(np.float64), requires_grad=True)
    out = _sdf_mlp(p, xm)
    sphere = 20.0 * (3.0 - xm.norm(2, 1, keepdim=True))
    sdf_c = torch.minimum(out[:, :1], sphere)
    grad = torch.autograd.grad(sdf_c.sum(), xm, create_graph=True)[0]
    dflat = torch.tensor(np.repeat(dirs[:, None, :], S, 1).reshape(-1, 3).astype(np.float64))
    rgb = _rgb_mlp(p, xm, grad, dflat, out[:, 1:])
    xe = torch.tensor(x_eik.astype(np.float64), requires_grad=True)
    gt = torch.autograd.grad(_sdf_mlp(p, xe)[:, :1].sum(), xe, create_graph=True)[0]
    loss = (rgb * torch.tensor(Wr, dtype=torch.float64)).sum() + (sdf_c * torch.tensor(ws, dtype=torch.float64)).sum() \
        + (gt * torch.tensor(Wg, dtype=torch.float64)).sum()
    loss.backward()
    assert (sphere < out[:, :1]).any() and (sphere > out[:, :1]).any()

    # ---- HIP
    pk = ops.PackedMlp(dev, precision=prec)
    t = lambda k: G(params[k], dev)
    sdf_p = ([t(f"implicit_network.lin{l}.weight_v") for l in range(9)], [t(f"implicit_network.lin{l}.weight_g") for l in range(9)],
             [t(f"implicit_network.lin{l}.bias") for l in range(9)])
    rgb_p = ([t(f"rendering_network.lin{l}.weight_v") for l in range(5)], [t(f"rendering_network.lin{l}.weight_g") for l in range(5)],
             [t(f"rendering_network.lin{l}.bias") for l in range(5)])
    pk.pack_sdf(*sdf_p)
    pk.pack_rgb(*rgb_p)
    keep = {}
    src = ops.PointSource(points=G(x_eik, dev), cam=G(cam, dev), dirs=G(dirs, dev), z=G(z, dev))
    sdf, gradients, feat_tiles, _, _ = ops.sdf_outputs(pk, src, 3.0, 20.0, clamp_n=R * S, keep=keep)
    np.testing.assert_allclose(sdf[:R * S].cpu().numpy(), sdf_c.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(gradients[R * S:].cpu().numpy(), gt.detach().numpy(), atol=2e-4)
    src_main = ops.PointSource(cam=G(cam, dev), dirs=G(dirs, dev), z=G(z, dev))
    rgb_h = ops.rgb_eval(pk, src_main, gradients[:R * S], G(dirs, dev), feat_tiles, keep=keep)
    np.testing.assert_allclose(rgb_h.cpu().numpy(), rgb.detach().numpy(), atol=1e-4)
    bw = MlpBackward(dev, streams=TrainStreams(dev, precision=prec))
    sdf_g, rgb_g = bw.run(sdf_p, rgb_p, keep, G(Wr, dev), G(ws, dev), G(Wg, dev))
    torch.cuda.synchronize()
    worst = 0.0
    for l in range(5):
        for name, got in zip(("weight_v", "weight_g", "bias"), rgb_g[l]):
            ref = p[f"rendering_network.lin{l}.{name}"].grad.numpy()
            e = rel_err(got.cpu().numpy().reshape(ref.shape), ref)
            worst = max(worst, e)
            assert e < bound, (f"rendering lin{l}.{name}", e)
    for l in range(9):
        for name, got in zip(("weight_v", "weight_g", "bias"), sdf_g[l]):
            ref = p[f"implicit_network.lin{l}.{name}"].grad.numpy()
            e = rel_err(got.cpu().numpy().reshape(ref.shape), ref)
            worst = max(worst, e)
            assert e < bound, (f"implicit lin{l}.{name}", e)
    print(precision, "worst relative gradient error", worst)
