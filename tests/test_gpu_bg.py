"""The fg + inverted-sphere background model (VolSDFNetworkBG, config 4) on the HIP kernels against the reference's
fixtures (tests/golden/forward_bg_*.npz) and the oracle."""
import os

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
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _params():
    p = dict(synth.make_params(0))
    p.update(synth.make_bg_params(0))
    return p


def _model(dev, beta):
    from volsdf.utils.conf import bmvs_model_conf
    from volsdf.model.network_bg import VolSDFNetworkBG
    m = VolSDFNetworkBG(bmvs_model_conf())
    sd = {k: torch.from_numpy(v) for k, v in _params().items()}
    sd["density.beta"] = torch.tensor(beta, dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    return m.to(dev)


@pytest.mark.parametrize("precision", [None, "f32"])
@pytest.mark.parametrize("tag", ["eval_b0.1", "train"])
def test_bg_pieces(dev, golden_dir, tag, precision, monkeypatch):
    """inverse-sphere points, bg implicit / radiance networks and the fg/bg compositing, each on the reference's inputs
    (precision f32: the float32-MFMA forms of the two networks, csrc/svs_bg_f32.hip)"""
    if precision:
        monkeypatch.setenv("SVS_MLP_PRECISION", precision)
    from svs_hip import ops
    g = dict(np.load(os.path.join(golden_dir, "forward_bg_" + tag + ".npz")))
    params = _params()
    R, Nb = g["z_bg"].shape
    dirs, cam, ds = orc.rays_from_uv(g["uv"], g["pose"], g["K"])
    jit = synth.make_train_rng(R, seed=int(g["rng_seed"]), bg=True)["jitter_bg"] if tag == "train" else None
    z_bg, pts, depth = ops.bg_points(G(cam, dev), G(dirs, dev), Nb, 3.0, jitter=G(jit, dev) if jit is not None else None)
    np.testing.assert_allclose(z_bg.cpu().numpy(), g["z_bg"], atol=3e-8)
    np.testing.assert_allclose(pts.cpu().numpy().reshape(R, Nb, 4), g["bg_points"], atol=3e-6)
    np.testing.assert_allclose(depth.cpu().numpy(), g["bg_depth"], rtol=3e-5)
    m = _model(dev, float(g["beta_param"]))
    pkb = m.packed_bg()
    out0, feat = ops.bg_sdf_eval(pkb, G(g["bg_points"].reshape(-1, 4), dev))
    np.testing.assert_allclose(out0.cpu().numpy(), g["bg_sdf"], atol=1e-4)
    ref_out = orc.sdf_mlp_forward(orc.effective_weights(params, "bg_implicit_network", 9), g["bg_points"].reshape(-1, 4), multires=10)
    rows = torch.empty(R * Nb, 256, device=dev)
    from svs_hip import lib
    import ctypes
    lib.check(lib.load().svs_tiles_to_rows(ctypes.c_void_p(feat.data_ptr()), R * Nb, pkb.precision, ctypes.c_void_p(rows.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))   # pair block / float32 block
    np.testing.assert_allclose(rows.cpu().numpy(), ref_out[:, 1:], atol=1e-4)
    view = orc.rays_from_uv(g["uv"], g["near_pose"], g["K"])[0] if tag != "train" else dirs
    rgb = ops.bg_rgb_eval(pkb, G(view, dev), Nb, feat, R * Nb)
    ref_rgb = orc.rgb_mlp_forward(orc.effective_weights(params, "bg_rendering_network", 2), None, None,
                                  np.repeat(view[:, None], Nb, 1).reshape(-1, 3), ref_out[:, 1:], mode="nerf", multires_view=4)
    np.testing.assert_allclose(rgb.cpu().numpy(), ref_rgb, atol=1e-4)
    # compositing on the fixture's arrays
    S = g["z_vals"].shape[1]
    rs = np.random.default_rng(1)
    sdf = rs.normal(0, 0.2, (R, S)).astype(F32)
    frgb = rs.uniform(0, 1, (R, S, 3)).astype(F32)
    beta = orc.get_beta(g["beta_param"])
    w_ref, t_ref, _ = orc.fg_weights_bg_model(g["z_vals"], g["z_max"], sdf, beta)
    bw_ref = orc.bg_weights(g["z_bg"], np.abs(g["bg_sdf"]).reshape(R, Nb))
    comp = ops.composite_bg(G(g["z_vals"], dev), G(g["z_max"], dev), G(sdf.reshape(-1, 1), dev), G(frgb.reshape(-1, 3), dev),
                            G(ds, dev), G(g["beta_param"], dev), 1e-4, G(g["z_bg"], dev), G(g["bg_sdf"], dev),
                            G(ref_rgb, dev), G(g["bg_depth"], dev))
    np.testing.assert_array_equal(comp["weights"].cpu().numpy(), w_ref)
    np.testing.assert_array_equal(comp["bg_transmittance"].cpu().numpy(), t_ref)
    np.testing.assert_array_equal(comp["bg_weights"].cpu().numpy(), bw_ref)
    rgb_ref = (w_ref[:, :, None] * frgb).sum(1) + t_ref[:, None] * (bw_ref[:, :, None] * ref_rgb.reshape(R, Nb, 3)).sum(1)
    np.testing.assert_allclose(comp["rgb_values"].cpu().numpy(), rgb_ref, atol=2e-6)


@pytest.mark.parametrize("tag,precision", [("eval_b0.1", None), ("eval_b0.01", None), ("train", None), ("eval_b0.1", "f32"),
                                           ("train", "f32")])
def test_bg_model_forward_golden(dev, golden_dir, tag, precision, monkeypatch):
    """VolSDFNetworkBG.forward (HIP) against the reference's outputs (f32: all four networks on the float32-MFMA kernels)."""
    if precision:
        monkeypatch.setenv("SVS_MLP_PRECISION", precision)
    from rng_inject import inject_rng
    g = dict(np.load(os.path.join(golden_dir, "forward_bg_" + tag + ".npz")))
    m = _model(dev, float(g["beta_param"]))
    training = tag == "train"
    m.train(training)
    R = g["uv"].shape[0]
    inp = {"intrinsics": G(g["K"], dev)[None], "uv": G(g["uv"], dev)[None], "pose": G(g["pose"], dev)[None],
           "near_pose": G(g["near_pose"], dev)[None]}
    with torch.no_grad():
        if training:
            with inject_rng(synth.make_train_rng(R, seed=int(g["rng_seed"]), bg=True)):
                out = m(inp, fast=int(g["fast"]))
        else:
            out = m(inp, fast=int(g["fast"]))
    out = {k: v.detach().cpu().numpy() for k, v in out.items()}
    # every ray; per-sample arrays where the sample did not move (tests/test_gpu_parity.py::_moved)
    moved = np.abs(out["depth_vals"] - g["depth_vals"]) > 3e-4
    assert moved.mean() < 0.06
    np.testing.assert_allclose(out["xyz"][~moved], g["xyz"][~moved], atol=3e-4)
    np.testing.assert_allclose(out["rgb_values"], g["rgb_values"], atol=1e-4)
    assert np.abs(out["weights"][~moved] - g["weights"][~moved]).mean() < 5e-5        # (a moved neighbour re-weights its interval)
    wsum = g["weights"].sum(1, keepdims=True)
    assert (np.abs(out["depth_values"] - g["depth_values"]) <= 3e-4 / np.maximum(wsum, 1e-3)).all()
    np.testing.assert_allclose(out["depth_values_all"], g["depth_values_all"], rtol=3e-3)
    if training:
        np.testing.assert_allclose(out["grad_theta"][:R], g["grad_theta"][:R], atol=2e-4)
    else:
        np.testing.assert_allclose(out["normal_map"], g["normal_map"], atol=2e-4)


@pytest.mark.parametrize("name", ["forward256_bg_eval_b0.01", "forward256_bg_train_b0.05"])
def test_bg_model_forward_r256(dev, golden_dir, name):
    """VolSDFNetworkBG.forward (HIP) on 256 rays against the reference, eval (fast = -1, near_pose) and train mode: colours to
    1e-4 on every ray, fg depths to 2e-4 / fg weight sum, all-depths to 3e-3 relative."""
    from rng_inject import inject_rng
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    training = "train" in name
    m = _model(dev, float(g["beta_param"]))
    m.train(training)
    inp = {"intrinsics": G(g["K"], dev)[None], "uv": G(g["uv"], dev)[None], "pose": G(g["pose"], dev)[None],
           "near_pose": G(g["near_pose"], dev)[None]}
    with torch.no_grad():
        if training:
            with inject_rng(synth.make_train_rng(256, seed=int(g["rng_seed"]), bg=True)):
                out = m(inp, fast=int(g["fast"]))
        else:
            out = m(inp, fast=int(g["fast"]))
    out = {k: v.detach().cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
    print(f"{name}: rgb max err on all 256 rays {np.abs(out['rgb_values'] - g['rgb_values']).max():.2e}")
    np.testing.assert_allclose(out["rgb_values"], g["rgb_values"], atol=1e-4)
    wsum = g["weights"].sum(1, keepdims=True)
    assert (np.abs(out["depth_values"] - g["depth_values"]) <= 3e-4 / np.maximum(wsum, 1e-3)).all()
    np.testing.assert_allclose(out["depth_values_all"], g["depth_values_all"], rtol=3e-3)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))


def test_composite_bg_backward(dev):
    """svs_composite_bg_bwd against float64 autograd of oracle/torch_ref.composite_bg"""
    import torch_ref as tref
    from svs_hip import ops
    rs = np.random.default_rng(4)
    R, S, Nb = 40, 97, 32
    z = np.sort(rs.uniform(0.2, 5.0, (R, S)), -1).astype(F32)
    z_max = (z[:, -1] + rs.uniform(0.05, 0.5, R)).astype(F32)
    sdf = rs.normal(0.05, 0.15, (R, S)).astype(F32)
    rgb = rs.uniform(0, 1, (R, S, 3)).astype(F32)
    ds = rs.uniform(0.8, 1.0, (R, 1)).astype(F32)
    z_bg = np.sort(rs.uniform(0, 1 / 3, (R, Nb)), -1)[:, ::-1].astype(F32).copy()
    bo = rs.normal(0, 2.0, (R, Nb)).astype(F32)
    brgb = rs.uniform(0, 1, (R, Nb, 3)).astype(F32)
    g_rgb, g_w, g_d = rs.normal(0, 1, (R, 3)).astype(F32), rs.normal(0, 0.1, (R, S)).astype(F32), rs.normal(0, 0.5, (R, 1)).astype(F32)
    beta = 0.07
    T = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tz, tsdf, trgb, tbo, tbrgb, tb = torch.tensor(z, dtype=torch.float64), T(sdf), T(rgb), T(bo), T(brgb), T(np.asarray(beta))
    w, tbg, rv, dv = tref.composite_bg(tz, torch.tensor(z_max, dtype=torch.float64), tsdf, trgb, tb, torch.tensor(ds, dtype=torch.float64),
                                       torch.tensor(z_bg, dtype=torch.float64), tbo, tbrgb)
    loss = (rv * torch.tensor(g_rgb)).sum() + (w * torch.tensor(g_w)).sum() + (dv * torch.tensor(g_d)).sum()
    loss.backward()
    d_sdf, d_rgb, d_bo, d_brgb, d_beta = ops.composite_bg_bwd(G(z, dev), G(z_max, dev), G(sdf.reshape(-1, 1), dev), G(rgb.reshape(-1, 3), dev),
                                                              G(ds, dev), torch.tensor(beta, device=dev), 1e-4, G(z_bg, dev),
                                                              G(bo.reshape(-1, 1), dev), G(brgb.reshape(-1, 3), dev), G(g_rgb, dev),
                                                              G(g_w, dev), G(g_d, dev))
    assert _rel(d_sdf.cpu().numpy().reshape(R, S), tsdf.grad.numpy()) < 2e-5
    assert _rel(d_rgb.cpu().numpy().reshape(R, S, 3), trgb.grad.numpy()) < 2e-5
    assert _rel(d_bo.cpu().numpy().reshape(R, Nb), tbo.grad.numpy()) < 2e-5
    assert _rel(d_brgb.cpu().numpy().reshape(R, Nb, 3), tbrgb.grad.numpy()) < 2e-5
    assert abs(float(d_beta) - float(tb.grad)) / abs(float(tb.grad)) < 5e-5


@pytest.mark.parametrize("fixture", ["train_step_bg", "train_step_bg_sparse"])
def test_bg_train_steps_fused(dev, golden_dir, fixture):
    """TrainStep with VolSDFNetworkBG (forward incl. background, lookup, loss, fg + bg backward, clip + Adam) against the
    reference's optimisation steps.  train_step_bg: 2 steps past the annealing (the L1 colour term reaches the
    background); train_step_bg_sparse: 1 step inside it with every ray unsupported by the prior -- the loss is eikonal
    + sparsity only, and the sparsity term reads depth_values_all, whose gradient reaches the background networks."""
    from rng_inject import inject_rng
    from svs_hip.trainer import TrainStep
    from volsdf.model.loss import VolSDFLoss
    g = dict(np.load(os.path.join(golden_dir, fixture + ".npz")))
    m = _model(dev, 0.1)
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                      anneal_rgb=200, gce=0.5, confi=float(g["confi"]))
    loss.iter_step = int(g["loss_iter_step"])
    ts = TrainStep(m, loss, lr=5e-4)
    views = synth.make_mvs_views(int(g["mvs_seed"]))
    dv = [dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views]
    mvs = dict(views=dv, same_view=0, img_res=(576, 768), inverse_depth=False)
    R = g["uv"].shape[0]
    inp = {"intrinsics": G(views[0]["K"], dev)[None], "uv": G(g["uv"], dev)[None], "pose": G(views[0]["c2w"], dev)[None]}
    gt = {"rgb": G(g["rgb"], dev), "rgb_smooth": G(g["rgb_smooth"], dev)}
    n_steps = 1 + max(int(k[1]) for k in g if k.startswith("s") and k[1].isdigit() and k[2] == "_")

    def tensor_rel(step, kind, named):
        out = {}
        for name, t in named:
            idx, ref = g[f"s{step}_{kind}_idx/{name}"], g[f"s{step}_{kind}/{name}"]
            got = t.detach().cpu().numpy().reshape(-1)[idx]
            out[name] = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
        return out

    for step in range(n_steps):
        with inject_rng(synth.make_train_rng(R, seed=100 + step, bg=True)):
            lo, out = ts(inp, gt, mvs=mvs)
        torch.cuda.synchronize()
        for k in ("rgb_loss", "eikonal_loss", "mvs_loss", "sparse_loss", "loss"):
            print(f"step {step} {k}: {float(lo[k]):.7f} ref {float(g[f's{step}_{k}']):.7f}")
            np.testing.assert_allclose(float(lo[k]), float(g[f"s{step}_{k}"]), rtol=3e-4 if step == 0 else 5e-3, atol=2e-6,
                                       err_msg=f"step {step} {k}")
        norm = float(ts.opt.info[0])
        np.testing.assert_allclose(norm, float(g[f"s{step}_grad_norm"]), rtol=3e-4 if step == 0 else 5e-3,
                                   err_msg=f"gradient norm, step {step}")
        coef = min(1.0, 1.0 / (float(g[f"s{step}_grad_norm"]) + 1e-6))
        rel = tensor_rel(step, "grad", [(n, p.grad / coef) for n, p in m.named_parameters()])
        worst = max(rel, key=rel.get)
        print(f"step {step}: worst per-tensor gradient error {rel[worst]:.3e} {worst}")
        bg_rel = {k: v for k, v in rel.items() if k.startswith("bg_")}
        print(f"step {step}: worst background-tensor gradient error {max(bg_rel.values()):.3e}")
        # (sparse fixture: no colour term, so only the background DENSITY network receives gradients -- through
        # depth_values_all; the sampled entries of its last layer's weight are all feature rows, which feed the colour net)
        dead = ("bg_rendering", "bg_implicit_network.lin8.weight") if fixture.endswith("sparse") else ()
        live = [k for k in bg_rel if not k.startswith(dead)] if dead else list(bg_rel)
        assert all(np.abs(g[f"s{step}_grad/{k}"]).max() > 0 for k in live), "the fixture must exercise the background nets"
        assert max(rel.values()) < (3e-3 if step == 0 else 3e-2), rel
        prel = tensor_rel(step, "param", list(m.named_parameters()))
        assert max(prel.values()) < 2e-2, prel
    if fixture.endswith("sparse"):
        assert float(g["s0_sparse_loss"]) > 0 and float(g["s0_rgb_loss"]) == 0.0


@pytest.mark.parametrize("fixture", ["train_step_bg", "train_step_bg_sparse"])
def test_bg_train_step_autograd_bridge(dev, golden_dir, fixture):
    """The reference's own sequence (volsdf/vsdf.py:196-219 with config/vol/bmvs.yaml's model class) -- model(...),
    cost_mapping, loss(...), loss.backward(), clip_grad_norm_, torch Adam -- driving the HIP kernels of the fg + bg model
    through its autograd bridge, against the reference's step."""
    from rng_inject import inject_rng
    from svs_hip import ops
    from volsdf.model.loss import VolSDFLoss
    g = dict(np.load(os.path.join(golden_dir, fixture + ".npz")))
    m = _model(dev, 0.1)
    m.train()
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                      anneal_rgb=200, gce=0.5, confi=float(g["confi"]))
    loss.iter_step = int(g["loss_iter_step"])
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    views = synth.make_mvs_views(int(g["mvs_seed"]))
    dv = [dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views]
    R = g["uv"].shape[0]
    inp = {"intrinsics": G(views[0]["K"], dev)[None], "uv": G(g["uv"], dev)[None], "pose": G(views[0]["c2w"], dev)[None]}
    gt = {"rgb": G(g["rgb"], dev), "rgb_smooth": G(g["rgb_smooth"], dev)}
    with inject_rng(synth.make_train_rng(R, seed=100, bg=True)):
        out = m(inp, fast=1)
    assert out["rgb_values"].requires_grad and out["depth_values_all"].requires_grad
    with torch.no_grad():
        out['pj'], out['pi'], _ = ops.cost_lookup(dv, 0, (576, 768), xyz=out['xyz'])
    lo = loss(out, gt)
    opt.zero_grad()
    lo['loss'].backward()
    np.testing.assert_allclose(float(lo['loss'].detach()), float(g["s0_loss"]), rtol=3e-4)
    worst, name = 0.0, ""
    for n, p in m.named_parameters():
        idx, ref = g[f"s0_grad_idx/{n}"], g[f"s0_grad/{n}"]
        e = float(np.abs(p.grad.cpu().numpy().reshape(-1)[idx] - ref).max() / (np.abs(ref).max() + 1e-30))
        if e > worst:
            worst, name = e, n
    print(f"{fixture}: bridge, worst per-tensor gradient error {worst:.3e} {name}")
    assert worst < 3e-3, (worst, name)
    norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    np.testing.assert_allclose(float(norm), float(g["s0_grad_norm"]), rtol=3e-4)
    opt.step()


def test_bg_ray_groups_do_not_change_the_step(dev):
    """VolSDFNetworkBG step as two ray groups on concurrent streams == the ungrouped step (first step: per-ray outputs
    bit-identical, gradients up to the float-atomic order / per-launch operand scaling)."""
    from svs_hip.trainer import TrainStep
    from volsdf.model.loss import VolSDFLoss
    R = 256
    K, pose = synth.make_camera()
    inp = {"intrinsics": G(K, dev)[None], "uv": G(synth.make_uv(R, seed=4), dev)[None], "pose": G(pose, dev)[None]}
    rs = np.random.default_rng(6)
    gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev)}
    views = synth.make_mvs_views(2)
    mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views], same_view=0,
               img_res=(576, 768), inverse_depth=False)
    runs = []
    for gr in (None, [(0, 160), (160, 256)]):
        m = _model(dev, 0.1)
        loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                          anneal_rgb=200, gce=0.5, confi=1e-3)
        loss.iter_step = 250                                        # past the rgb annealing: the background nets get gradients
        ts = TrainStep(m, loss, groups=gr)
        torch.manual_seed(13)
        lo, out = ts(inp, gt, mvs=mvs)
        runs.append(({k: float(v) for k, v in lo.items()}, ts.fp.grad.clone(), out["rgb_values"].clone(), out["weights"].clone()))
    (la, ga, ra, wa), (lb, gb, rb, wb) = runs
    for k in la:
        assert la[k] == pytest.approx(lb[k], rel=1e-5, abs=1e-8), k
    assert torch.equal(ra, rb) and torch.equal(wa, wb)
    assert float((ga - gb).abs().max()) <= 1e-4 * float(ga.abs().max())
    assert float(ga.abs().max()) > 0


@pytest.mark.parametrize("R,it,precision", [(256, 50, None), (1024, 50, None), (2048, 50, None), (256, 250, None), (1024, 250, None),
                                            (1024, 250, "f16x2_half"), (1000, 250, None), (90, 50, None), (256, 250, "f32"),
                                            (1024, 50, "f32")])
def test_bg_step_gradient_at_bench_geometry(dev, R, it, precision, monkeypatch):
    """The flat gradient of ONE fused step of the fg + background model at the benchmarked geometry -- 1024 rays, and
    256 rays = the per-GPU share of config 4's 2048-ray batch over 8 GPUs -- against float64 torch autograd
    (oracle/torch_ref.py: forward_differentiable_bg) on the very sample positions, background points, prior look-ups and
    targets the step used.  it = 250: past the colour annealing, every ray carries a colour term and the background
    networks receive gradients; it = 50: the annealed phase, where the sparse term (which reads depth_values_all) is live.
    2048 rays: config 4's whole batch on one GPU.  1000 / 90 rays: not multiples of the kernels' 32-ray granularity -- the step
    pads the batch and leaves the padding out of the loss; the reference gradient is taken over the caller's rays.  Per tensor: max |err| <= 3e-5 of the tensor's largest entry (the float32
    class) on the default fp16x2 path and with SVS_MLP_PRECISION=f32 (float32 MFMA kernels for all four networks); 2e-3 with
    SVS_MLP_PRECISION=f16x2_half (one-piece gradient blocks)."""
    if precision:
        monkeypatch.setenv("SVS_MLP_PRECISION", precision)
    bound = 2e-3 if precision == "f16x2_half" else 3e-5
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import torch_ref as tref
    from svs_hip.trainer import TrainStep
    from volsdf.model.loss import VolSDFLoss
    m = _model(dev, 0.1)
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                      anneal_rgb=200, gce=0.5, confi=1e-3)
    loss.iter_step = it
    K, pose = synth.make_camera()
    inp = {"intrinsics": G(K, dev)[None], "uv": G(synth.make_uv(R, seed=17), dev)[None], "pose": G(pose, dev)[None]}
    rs = np.random.default_rng(5)
    gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev),
          "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev)}
    views = synth.make_mvs_views(2)
    mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views], same_view=0,
               img_res=(576, 768), inverse_depth=False)
    ts = TrainStep(m, loss, lr=5e-4, groups="auto", graph=False)
    p0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(3)
    ts(inp, gt, mvs=mvs)
    torch.cuda.synchronize()
    norm = float(ts.opt.info[0])
    coef = min(1.0, 1.0 / (norm + 1e-6))
    got = {n: (p.grad / coef).double().cpu() for n, p in m.named_parameters()}
    keeps = [h[0] for h in ts._hold]
    outs = [r[1] for r in ts._results]
    cat = lambda xs: torch.cat(xs, 0).double()[:R]                  # [:R]: without the padding rays (if any)
    z, dirs, ds = (cat([k[n] for k in keeps]) for n in ("z_vals", "ray_dirs", "depth_scale"))
    z_max, z_bg, bg_depth = (cat([k[n] for k in keeps]) for n in ("z_max", "z_bg", "bg_depth"))
    Nb = z_bg.shape[1]
    bg_pts = cat([k["bg_pts"].reshape(-1, Nb, 4) for k in keeps])
    eik, lo_ray = [], 0
    for k in keeps:                                  # a group's eikonal points: [uniform of its rays, near-surface of its rays]
        rg = k["z_vals"].shape[0]
        v = max(0, min(lo_ray + rg, R) - lo_ray)
        pts = k["src"].points.double()
        eik += [pts[:v], pts[rg:rg + v]]
        lo_ray += rg
    eik = torch.cat(eik, 0)
    assert lo_ray % ts.ray_multiple() == 0 and 0 <= lo_ray - R < ts.ray_multiple()
    pj, pi = cat([o["pj"] for o in outs]), cat([o["pi"] for o in outs])
    def autograd(dt, pp=None):
        c = lambda t: t.to(dt)
        p = {k: (pp or p0)[k].detach().to(dt).clone().requires_grad_(True) for k in p0}
        out = tref.forward_differentiable_bg(p, c(keeps[0]["cam_loc"]), c(dirs), c(z), c(z_max), c(eik), c(ds), c(z_bg), c(bg_pts),
                                             bg_depth=c(bg_depth), device=dev)
        out["pj"], out["pi"] = c(pj), c(pi)
        out["depth_values"] = out["depth_values_all"]            # loss.py:72-73: the sparse term reads depth_values_all
        tref.loss_fn(out, c(gt["rgb"].reshape(-1, 3)), c(gt["rgb_smooth"].reshape(-1, 3)), it).backward()
        return {k: (v.grad.cpu() if v.grad is not None else torch.zeros(v.shape, dtype=dt)) for k, v in p.items()}
    ref = autograd(torch.float64)
    ref_norm = float(torch.sqrt(sum((v ** 2).sum() for v in ref.values())))
    live_bg = sum(int(n.startswith("bg_") and float(ref[n].abs().max()) > 0) for n in got)
    what = (f"bmvs, {R} rays, step {it}, {precision or 'fp16x2'}, {len(keeps)} group(s), gradient norm {norm:.6f} vs {ref_norm:.6f}, "
            f"{live_bg} background tensors with gradients")
    if it >= 200:
        assert live_bg >= 18, live_bg
    from grad_class import assert_f32_class, f32_yardstick, per_tensor_errors
    if precision == "f16x2_half":
        errs = per_tensor_errors(got, ref)
        worst = max(errs, key=lambda n: errs[n][0])
        print(f"{what}: worst per-tensor gradient error vs float64 autograd {errs[worst][0]:.2e} ({worst})")
        assert norm == pytest.approx(ref_norm, rel=1e-3), (norm, ref_norm)
        assert errs[worst][0] < bound, (worst, errs[worst])
    else:
        assert norm == pytest.approx(ref_norm, rel=1e-5), (norm, ref_norm)
        yard = f32_yardstick(autograd, p0)
        assert_f32_class(per_tensor_errors(got, ref, yard), what, floor=bound, floors={"density.beta": 1e-4})


def test_degenerate_ray_poisons_the_step_like_the_reference(dev):
    """A ray through the centre of the bounding sphere has no rotation axis in the inverted-sphere parametrisation
    (network_bg.py:196-197: rot_axis = cross(ray_o, p_sphere) / 0): the reference's background points, colour and loss are NaN,
    `loss.backward()` makes EVERY gradient NaN (L1's sign(NaN) is NaN, the weight-gradient sums run over all points) and
    on_after_backward (vsdf.py:454-463) drops the step.  Same here: NaN loss, the guard zeroes the whole gradient (info[1] =
    1) and a fresh optimiser leaves the parameters where they were."""
    from svs_hip.trainer import TrainStep
    from volsdf.model.loss import VolSDFLoss
    R = 64
    K, pose = synth.make_camera()                       # looks at the origin: the principal point's ray hits the sphere centre
    uv = synth.make_uv(R, seed=8).astype(F32)
    uv[5] = (K[0, 2], K[1, 2])
    inp = {"intrinsics": G(K, dev)[None], "uv": G(uv, dev)[None], "pose": G(pose, dev)[None]}
    rs = np.random.default_rng(6)
    gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(F32), dev), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(F32), dev)}
    m = _model(dev, 0.1)
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=0.0, sparse_weight=0.0,
                      anneal_rgb=0, gce=0.5, confi=1e-3)
    ts = TrainStep(m, loss)
    before = ts.fp.flat.clone()
    torch.manual_seed(3)
    lo, out = ts(inp, gt)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(out["rgb_values"]).all(-1)
    assert bad.nonzero().flatten().tolist() == [5] and not np.isfinite(float(lo["rgb_loss"]))
    assert float(ts.opt.info[1]) == 1.0                         # the guard dropped the gradient
    assert torch.equal(ts.fp.flat, before)                      # zero gradient + fresh Adam moments: no parameter moved
    # the same batch without that ray is an ordinary step
    uv[5] = uv[6] + 1
    ts2 = TrainStep(_model(dev, 0.1), loss)
    torch.manual_seed(3)
    lo2, _ = ts2(dict(inp, uv=G(uv, dev)[None]), gt)
    assert np.isfinite(float(lo2["rgb_loss"])) and float(ts2.opt.info[1]) == 0.0


@pytest.mark.parametrize("model_kind,R", [("bg", 256), ("dtu", 1024)])
def test_deterministic_mode_repeats_bit_for_bit(dev, model_kind, R):
    """svs_set_deterministic(1) (SVS_DETERMINISTIC=1): three optimisation steps from the same state and the same draws give the
    SAME parameters, bit for bit, in two runs -- the weight-gradient workgroups add their partial sums in launch order
    (csrc/svs_ticket.h) and the step runs as one ray group on one stream.  The default mode (float atomics in arrival order,
    concurrent streams) agrees with it to the float-atomic noise, and is what the other tests and the bench measure."""
    from svs_hip import lib as _lib
    from svs_hip.trainer import TrainStep
    from volsdf.model.loss import VolSDFLoss
    K, pose = synth.make_camera()
    inp = {"intrinsics": G(K, dev)[None], "uv": G(synth.make_uv(R, seed=4), dev)[None], "pose": G(pose, dev)[None]}
    rs = np.random.default_rng(6)
    gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev)}
    views = synth.make_mvs_views(2)
    mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views], same_view=0,
               img_res=(576, 768), inverse_depth=False)

    def make_model():
        if model_kind == "bg":
            return _model(dev, 0.1)
        from volsdf.utils.conf import dtu_model_conf
        from volsdf.model.network import VolSDFNetwork
        m = VolSDFNetwork(dtu_model_conf())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}, strict=True)
        return m.to(dev).train()

    def run(det):
        L = _lib.load()
        was = L.svs_set_deterministic(1 if det else 0)
        try:
            m = make_model()
            loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                              anneal_rgb=200, gce=0.5, confi=1e-3)
            loss.iter_step = 250
            ts = TrainStep(m, loss)
            assert ts.deterministic == det
            torch.manual_seed(13)
            grads = []
            for _ in range(3):
                ts(inp, gt, mvs=mvs)
                grads.append(ts.fp.grad.clone())
            torch.cuda.synchronize()
            return grads, ts.fp.flat.clone()
        finally:
            L.svs_set_deterministic(was)

    (ga, pa), (gb, pb) = run(True), run(True)
    for i, (x, y) in enumerate(zip(ga, gb)):
        assert torch.equal(x, y), f"step {i}: gradients of two deterministic runs differ in {int((x != y).sum())} entries"
    assert torch.equal(pa, pb)
    assert float(ga[0].abs().max()) > 0 and torch.isfinite(pa).all()
    gd, pd = run(False)                                    # default mode: same sums in another order
    assert float((ga[0] - gd[0]).abs().max()) <= 1e-4 * float(ga[0].abs().max())
