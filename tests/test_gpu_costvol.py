"""GPU parity of the CasMVSNet cost-volume kernels against the oracle and
the reference-generated fixtures.  Tolerances: 2e-4 abs on warped features / variance (bilinear sampling at
|coordinate| ~ 1e2 px), 2e-3 abs / 5e-5 mean on the 11-layer 3-D U-Net output, exact regression index."""
import os

import numpy as np
import pytest
import torch

import casmvs_oracle as corc
import synth

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def test_homo_warping_golden(dev, golden_dir):
    from models.CasMVSNet import homo_warping
    g = load(golden_dir, "homo_warp")
    w = homo_warping(G(g["src"], dev)[None], G(g["src_proj"], dev)[None], G(g["ref_proj"], dev)[None],
                     G(g["depth_values"], dev)[None])[0].cpu().numpy()
    np.testing.assert_allclose(w, g["warped"], atol=2e-4)
    assert np.abs(w - g["warped"]).mean() < 2e-6
    assert np.array_equal(w == 0, g["warped"] == 0)           # same off-image / behind-camera voxels


def test_tail_d192_golden(dev, golden_dir):
    from svs_hip import costvol
    g = load(golden_dir, "depthnet_tail_d192")
    prob, depth, conf, idx = costvol.prob_depth_conf(G(g["reg"], dev), G(g["depth_values"], dev))
    np.testing.assert_allclose(prob.cpu().numpy(), g["prob"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(depth.cpu().numpy(), g["depth"], rtol=3e-6)
    assert np.array_equal(idx.cpu().numpy(), g["idx"])
    np.testing.assert_allclose(conf.cpu().numpy(), g["conf"], atol=1e-6)


def _model(dev, ndepths):
    from models.CasMVSNet import CascadeMVSNet
    m = CascadeMVSNet(refine=False, ndepths=ndepths, depth_interals_ratio=[1.0, 0.5, 0.5], share_cr=False,
                      cr_base_chs=[8, 8, 8], grad_method="detach")
    for st, cin in enumerate((32, 16, 8)):
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_costreg_params(100 + st, cin).items()}
        m.cost_regularization[st].load_state_dict(sd, strict=True)
    return m.to(dev).eval()


def test_three_stage_forward_golden(dev, golden_dir):
    """CascadeMVSNet.forward x 3 stages through the reference's call surface, stage-1 depth overridden as in
    runner.py:240-243."""
    g = load(golden_dir, "casmvs_3stage")
    feats, proj, depth_values = synth.make_mvs_sample(int(g["seed"]), img_hw=(64, 96))
    m = _model(dev, [int(x) for x in g["ndepths"]])
    sample = dict(imgs=torch.zeros(1, 3, 3, 64, 96, device=dev), depth_values=G(depth_values, dev)[None],
                  proj_matrices={k: G(v, dev)[None] for k, v in proj.items()})
    features = [{k: G(v, dev)[None] for k, v in f.items()} for f in feats]
    outputs = None
    for st in range(3):
        cr = m.cost_regularization[st]
        cap = {}
        orig = cr.forward
        cr.forward = lambda x, _o=orig, _c=cap: _c.setdefault("reg", _o(_c.setdefault("var", x)))
        outputs, _ = m(st, sample, features=features, extra=None, outputs=outputs, int_r=m.depth_interals_ratio[st])
        cr.forward = orig
        o = outputs[f"stage{st + 1}"]
        np.testing.assert_allclose(o["depth_values"][0].cpu().numpy(), g[f"s{st}_depth_values"], rtol=5e-6,
                                   err_msg=f"hypotheses stage {st + 1}")
        var = cap["var"]            # conv0's input travels as a SplitVolume (fp16 hi + mid pieces)
        var = (var.float() if hasattr(var, "buf") else var[0]).cpu().numpy().reshape(-1)
        np.testing.assert_allclose(var[g[f"s{st}_variance_idx"]], g[f"s{st}_variance_val"], atol=2e-4)
        reg = cap["reg"][0, 0].cpu().numpy()
        np.testing.assert_allclose(reg, g[f"s{st}_reg"], atol=2e-3, err_msg=f"reg stage {st + 1}")
        assert np.abs(reg - g[f"s{st}_reg"]).mean() < 5e-5
        # depth / confidence: continuous in reg except where the truncated index flips
        np.testing.assert_allclose(o["depth"][0].cpu().numpy(), g[f"s{st}_depth"], rtol=2e-5)
        dconf = np.abs(o["photometric_confidence"][0].cpu().numpy() - g[f"s{st}_conf"])
        assert (dconf > 1e-4).mean() < 0.01
        # the probability volume -- what VolOpt.cost_mapping looks up -- at every stage
        np.testing.assert_allclose(o["prob_volume"][0].cpu().numpy(), g[f"s{st}_prob"], atol=2e-5, err_msg=f"prob stage {st + 1}")
        if st == 0:
            ov = G(g["stage1_depth_override"], dev)[None]
            outputs["stage1"]["depth"] = ov
            outputs["depth"] = ov


@pytest.mark.parametrize("C,n_views,D,hw", [(32, 3, 5, (21, 37)), (16, 2, 70, (13, 36)), (8, 5, 6, (10, 131)),
                                             (32, 4, 9, (6, 200)), (16, 3, 3, (5, 330))])
def test_warp_variance_ragged(dev, C, n_views, D, hw):
    """Fused warp + variance on shapes that exercise every tail of the tiled kernel (widths that are neither a
    multiple of 4 nor of the x tile, depth counts that do not fill the planes-per-workgroup loop, 1..4 source
    views, several x tiles per row) against the oracle's per-view homo_warp + variance."""
    from svs_hip import costvol
    rng = np.random.default_rng(C + D)
    H, W = hw
    feats = [rng.normal(0, 1, (C, H, W)).astype(F32) for _ in range(n_views)]
    projs = np.zeros((n_views, 2, 4, 4), F32)
    for v in range(n_views):
        ext = np.eye(4, dtype=F32)
        ang = 0.03 * v
        ext[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], F32)
        ext[0, 3], ext[1, 3] = 12.0 * v * (-1) ** v, 2.0 * v
        K = np.eye(4, dtype=F32)
        K[0, 0] = K[1, 1] = 1.5 * W
        K[0, 2], K[1, 2] = W / 2.0, H / 2.0
        projs[v, 0], projs[v, 1] = ext, K
    dv = np.broadcast_to((400.0 + 7.0 * np.arange(D, dtype=F32))[:, None, None], (D, H, W)).copy()
    dv += rng.uniform(0, 3, dv.shape).astype(F32)
    got = costvol.warp_variance([G(f, dev)[None] for f in feats], G(projs, dev)[None], G(dv, dev)[None])[0].cpu().numpy()
    ref = corc.variance_volume(feats, projs, dv)
    # white-noise features are the worst case for the sampling position: at |coordinate| ~ 300 px the float32
    # rounding of rot @ [x,y,1] * depth (numpy's matmul order vs the kernel's) moves a sample by ~3e-5 px and the
    # features change by O(1) per pixel
    np.testing.assert_allclose(got, ref, atol=5e-4)
    assert np.abs(got - ref).mean() < 5e-6
    assert np.abs(ref).max() > 0.1


def _mvs_inputs(rng, C, n_views, D, H, W):
    feats = [rng.normal(0, 1, (C, H, W)).astype(F32) for _ in range(n_views)]
    projs = np.zeros((n_views, 2, 4, 4), F32)
    for v in range(n_views):
        ext = np.eye(4, dtype=F32)
        ang = 0.03 * v
        ext[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], F32)
        ext[0, 3], ext[1, 3] = 12.0 * v * (-1) ** v, 2.0 * v
        K = np.eye(4, dtype=F32)
        K[0, 0] = K[1, 1] = 1.5 * W
        K[0, 2], K[1, 2] = W / 2.0, H / 2.0
        projs[v, 0], projs[v, 1] = ext, K
    dv = np.broadcast_to((400.0 + 7.0 * np.arange(D, dtype=F32))[:, None, None], (D, H, W)).copy()
    dv += rng.uniform(0, 3, dv.shape).astype(F32)
    return feats, projs, dv


@pytest.mark.parametrize("C,n_views,D,hw", [(32, 3, 5, (21, 37)), (16, 2, 70, (13, 36)), (8, 5, 6, (10, 131)),
                                             (32, 4, 9, (6, 200))])
def test_warp_variance_split_volume(dev, C, n_views, D, hw):
    """The split-volume output of the fused warp + variance (the input form of the fused conv0) holds the values of
    the float32 output as fp16 hi + mid: equal to 2^-21 relative (two 11-bit pieces), zero outside the interior."""
    from svs_hip import costvol
    H, W = hw
    feats, projs, dv = _mvs_inputs(np.random.default_rng(C + D), C, n_views, D, H, W)
    args = ([G(f, dev)[None] for f in feats], G(projs, dev)[None], G(dv, dev)[None])
    ref = costvol.warp_variance(*args)[0]
    sv = costvol.warp_variance(*args, split=True)
    got = sv.float()
    err = (got - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -21).max()) <= 1e-7         # 1e-7: fp16 underflow of the mid piece
    v6 = sv.buf.view(D + 2, -1, 2, C // 8, 32 * ((W + 31) // 32) + 4, 8)
    assert v6.shape[1] == 4 * ((H + 3) // 4) + 2
    border = v6.clone()
    border[1:D + 1, 1:H + 1, :, :, 1:W + 1] = 0
    assert not bool(border.any())                                      # nothing outside the interior is written
    # packing the float32 volume gives the same pieces bit for bit
    a = sv.buf.clone()
    inner = sv.buf.view(D + 2, -1, 2, C // 8, 32 * ((W + 31) // 32) + 4, 8)[1:D + 1, 1:H + 1, :, :, 1:W + 1]
    inner.add_(1.0)                                                    # a pack that wrote nothing would leave this behind
    sv2 = costvol.SplitVolume.pack(ref)
    assert sv2.buf.data_ptr() == sv.buf.data_ptr() and torch.equal(a, sv2.buf)
    # the buffer is shared per shape: the older object is stale now and says so
    with pytest.raises(RuntimeError, match="reused"):
        sv.float()


@pytest.mark.parametrize("cin,cout,shape", [(32, 8, (5, 7, 37)), (16, 8, (3, 9, 70)), (8, 8, (9, 6, 33)), (32, 5, (2, 4, 32)),
                                            (32, 8, (20, 8, 64))])
def test_conv3d_pair_vs_float64(dev, cin, cout, shape):
    """The fused conv0 (x-pair rows, split-volume input) against a float64 torch convolution on ragged shapes (H not a
    multiple of 4, W not a multiple of 32, D shorter than the ring) and against the channel-row kernel it replaces.
    Tolerance 3e-6 of the output scale: float32-class (fp16x2 operands carry 22 bits, float32 accumulation)."""
    from svs_hip import costvol
    rng = np.random.default_rng(cin + shape[2])
    x = rng.normal(0, 1, (cin,) + shape).astype(F32)
    w = (rng.normal(0, 1, (cin, 27, cout)) / np.sqrt(27 * cin)).astype(F32)
    b = rng.normal(0, 1, cout).astype(F32)
    for relu in (True, False):
        got = costvol.conv3d(costvol.SplitVolume.pack(G(x, dev)), G(w, dev), G(b, dev), relu=relu).cpu().numpy()
        wt = torch.from_numpy(w).double().permute(2, 0, 1).reshape(cout, cin, 3, 3, 3)
        ref = torch.nn.functional.conv3d(torch.from_numpy(x).double()[None], wt, torch.from_numpy(b).double(), padding=1)[0]
        ref = (ref.clamp(min=0) if relu else ref).numpy()
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=3e-6 * np.abs(ref).max())
        old = costvol.conv3d(G(x, dev), G(w, dev), G(b, dev), relu=relu).cpu().numpy()
        np.testing.assert_allclose(got, old, atol=3e-6 * np.abs(ref).max())


@pytest.mark.parametrize("cout,shape", [(16, (5, 7, 38)), (16, (6, 9, 70)), (12, (8, 8, 32)), (16, (16, 24, 96))])
def test_conv3d_stride2_from_8_channels(dev, cout, shape):
    """conv1's kernel (8 -> Cout <= 16, stride 2; whole input rows per lane group, tap kx = 0 from the neighbour lane)
    against a float64 torch convolution: odd D / H, output widths that do not fill a 16-column tile, several tiles per
    row (the tile's left edge column).  Tolerance 3e-6 of the output scale (fp16x2 operands)."""
    from svs_hip import costvol
    rng = np.random.default_rng(cout + shape[2])
    x = rng.normal(0, 1, (8,) + shape).astype(F32)
    w = (rng.normal(0, 1, (8, 27, cout)) / np.sqrt(27 * 8)).astype(F32)
    b = rng.normal(0, 1, cout).astype(F32)
    got = costvol.conv3d(G(x, dev), G(w, dev), G(b, dev), stride=2, relu=True).cpu().numpy()
    wt = torch.from_numpy(w).double().permute(2, 0, 1).reshape(cout, 8, 3, 3, 3)
    ref = torch.nn.functional.conv3d(torch.from_numpy(x).double()[None], wt, torch.from_numpy(b).double(), stride=2,
                                     padding=1)[0].clamp(min=0).numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=3e-6 * np.abs(ref).max())


@pytest.mark.parametrize("cout,shape", [(16, (5, 7, 37)), (16, (3, 9, 70)), (11, (9, 6, 33)), (16, (20, 8, 64))])
def test_conv3d_rows_from_split_volume(dev, cout, shape):
    """conv2's kernel (16 -> Cout <= 16, stride 1, split-volume input, output channels as MFMA rows) against a float64 torch
    convolution on ragged shapes and against the kernel it replaces; and conv1's split-volume output (the producer side) holds
    the pieces of its float32 output bit for bit.  Tolerance 3e-6 of the output scale (fp16x2 operands)."""
    from svs_hip import costvol
    rng = np.random.default_rng(cout + shape[2])
    x = rng.normal(0, 1, (16,) + shape).astype(F32)
    w = (rng.normal(0, 1, (16, 27, cout)) / np.sqrt(27 * 16)).astype(F32)
    b = rng.normal(0, 1, cout).astype(F32)
    got = costvol.conv3d(costvol.SplitVolume.pack(G(x, dev)), G(w, dev), G(b, dev), relu=True).cpu().numpy()
    wt = torch.from_numpy(w).double().permute(2, 0, 1).reshape(cout, 16, 3, 3, 3)
    ref = torch.nn.functional.conv3d(torch.from_numpy(x).double()[None], wt, torch.from_numpy(b).double(), padding=1)[0]
    ref = ref.clamp(min=0).numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=3e-6 * np.abs(ref).max())
    old = costvol.conv3d(G(x, dev), G(w, dev), G(b, dev), relu=True).cpu().numpy()
    np.testing.assert_allclose(got, old, atol=3e-6 * np.abs(ref).max())
    # the producer: conv1 (8 -> 16, stride 2) writing the split form
    D, H, W = 2 * shape[0], 2 * shape[1], 2 * shape[2]
    x1 = rng.normal(0, 1, (8, D, H, W)).astype(F32)
    w1 = (rng.normal(0, 1, (8, 27, 16)) / np.sqrt(27 * 8)).astype(F32)
    b1 = rng.normal(0, 1, 16).astype(F32)
    f = costvol.conv3d(G(x1, dev), G(w1, dev), G(b1, dev), stride=2, relu=True)
    sv = costvol.conv3d(G(x1, dev), G(w1, dev), G(b1, dev), stride=2, relu=True, split_out=True)
    assert isinstance(sv, costvol.SplitVolume) and sv.shape == (1, 16) + tuple(f.shape[1:])
    a = sv.buf.clone()
    assert torch.equal(a, costvol.SplitVolume.pack(f).buf)


@pytest.mark.parametrize("cin,shape", [(8, (5, 7, 37)), (8, (11, 33, 40)), (16, (3, 5, 6)), (8, (24, 40, 64))])
def test_conv3d_one_output_channel(dev, cin, shape):
    """The `prob` layer's kernel (Cout = 1, float32 FMAs on the vector ALUs) against a float64 torch convolution:
    widths that are / are not multiples of 4, depths that do not fill a thread's z run, with and without skip and ReLU.
    Tolerance 2e-6 of the output scale (216-term float32 sums)."""
    from svs_hip import costvol
    rng = np.random.default_rng(cin + shape[2])
    x = rng.normal(0, 1, (cin,) + shape).astype(F32)
    w = (rng.normal(0, 1, (cin, 27, 1)) / np.sqrt(27 * cin)).astype(F32)
    skip = rng.normal(0, 1, (1,) + shape).astype(F32)
    wt = torch.from_numpy(w).double().permute(2, 0, 1).reshape(1, cin, 3, 3, 3)
    ref = torch.nn.functional.conv3d(torch.from_numpy(x).double()[None], wt, padding=1)[0].numpy()
    got = costvol.conv3d(G(x, dev), G(w, dev), None, relu=False).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=2e-6 * np.abs(ref).max())
    got = costvol.conv3d(G(x, dev), G(w, dev), G(np.array([0.25], F32), dev), skip=G(skip, dev), relu=True).cpu().numpy()
    np.testing.assert_allclose(got, np.maximum(ref + 0.25, 0) + skip, atol=2e-6 * np.abs(ref).max())


@pytest.mark.parametrize("cin,shape", [(32, (16, 16, 24)), (16, (8, 24, 16)), (8, (8, 8, 8))])
def test_costreg_vs_torch_reference(dev, cin, shape):
    """3-D U-Net on random volumes (incl. non-cubic shapes) against the plain torch float32 reference."""
    from models.CasMVSNet import CostRegNet
    params = synth.make_costreg_params(7 + cin, cin)
    net = CostRegNet(cin, 8)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}, strict=True)
    net.to(dev).eval()
    x = np.random.default_rng(cin).normal(0, 1, (cin,) + shape).astype(F32)
    y = net(G(x, dev)[None])[0, 0].cpu().numpy()
    ref = corc.cost_reg_net_torch(params, x)
    np.testing.assert_allclose(y, ref, atol=2e-3)
    assert np.abs(y - ref).mean() < 5e-5
    # the same network fed with the split volume (conv0 fused with its producer, the path DepthNet takes)
    from svs_hip import costvol
    y2 = net(costvol.SplitVolume.pack(G(x, dev)))[0, 0].cpu().numpy()
    np.testing.assert_allclose(y2, ref, atol=2e-3)
    assert np.abs(y2 - ref).mean() < 5e-5


def test_config3_sizes_run(dev):
    """BASELINE config 3 geometry (640x512 image, D = 192/32/8): shapes, finiteness and linearity of the fused
    warp+variance in the feature scale (size-independent property)."""
    from svs_hip import costvol
    feats, proj, depth_values = synth.make_mvs_sample(3, img_hw=(512, 640))
    m = _model(dev, [192, 32, 8])
    sample = dict(imgs=torch.zeros(1, 3, 3, 512, 640, device=dev), depth_values=G(depth_values, dev)[None],
                  proj_matrices={k: G(v, dev)[None] for k, v in proj.items()})
    features = [{k: G(v, dev)[None] for k, v in f.items()} for f in feats]
    outputs = None
    for st in range(3):
        outputs, _ = m(st, sample, features=features, extra=None, outputs=outputs, int_r=m.depth_interals_ratio[st])
        o = outputs[f"stage{st + 1}"]
        s = (4, 2, 1)[st]
        assert o["depth"].shape == (1, 512 // s, 640 // s)
        assert o["prob_volume"].shape == (1, [192, 32, 8][st], 512 // s, 640 // s)
        assert torch.isfinite(o["depth"]).all() and torch.isfinite(o["photometric_confidence"]).all()
        np.testing.assert_allclose(o["prob_volume"].sum(1).cpu().numpy(), 1.0, atol=1e-5)
    f1 = [f["stage1"] for f in features]
    dv = outputs["stage1"]["depth_values"]
    v1 = costvol.warp_variance(f1, sample["proj_matrices"]["stage1"], dv)
    v2 = costvol.warp_variance([2.0 * f for f in f1], sample["proj_matrices"]["stage1"], dv)
    np.testing.assert_allclose(v2.cpu().numpy(), 4.0 * v1.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_conv0_full_size_pair_vs_channel_rows(dev):
    """BASELINE config 3, stage 1 (C = 32, 192 x 128 x 160): the fused conv0 (x-pair rows, K split over wave pairs, split-volume
    input) and the channel-row kernel it replaces are two independent float32-class evaluations of the same layer; at
    full size they agree to 3e-6 of the output scale, and the fused kernel is linear in its input."""
    from svs_hip import costvol
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(32, 192, 128, 160, device=dev, generator=g)
    w = torch.randn(32, 27, 8, device=dev, generator=g) / (27 * 32) ** 0.5
    b = torch.randn(8, device=dev, generator=g)
    a = costvol.conv3d(costvol.SplitVolume.pack(x), w, b, relu=False)
    ref = costvol.conv3d(x, w, b, relu=False)
    scale = float(ref.abs().max())
    assert float((a - ref).abs().max()) <= 3e-6 * scale
    a2 = costvol.conv3d(costvol.SplitVolume.pack(2.0 * x), w, None, relu=False)
    assert float((a2 - 2.0 * (a - b.view(8, 1, 1, 1))).abs().max()) <= 3e-6 * scale


def test_reference_resolution_sizes_run(dev):
    """The resolution the reference's MVS dataset actually feeds (1152 x 1536 after its x2 up-scaling, general_eval.py:225-229;
    D = 192 / 32 / 8): every fused path of the cost volume at its largest shapes -- 2.8 GB split volume at stage 1 -- runs,
    probabilities sum to one, depths stay inside the hypothesis range."""
    feats, proj, depth_values = synth.make_mvs_sample(5, img_hw=(1152, 1536))
    m = _model(dev, [192, 32, 8])
    sample = dict(imgs=torch.zeros(1, 3, 3, 1152, 1536, device=dev), depth_values=G(depth_values, dev)[None],
                  proj_matrices={k: G(v, dev)[None] for k, v in proj.items()})
    features = [{k: G(v, dev)[None] for k, v in f.items()} for f in feats]
    outputs = None
    for st in range(3):
        outputs, _ = m(st, sample, features=features, extra=None, outputs=outputs, int_r=m.depth_interals_ratio[st])
        o = outputs[f"stage{st + 1}"]
        sc = (4, 2, 1)[st]
        assert o["depth"].shape == (1, 1152 // sc, 1536 // sc)
        assert torch.isfinite(o["depth"]).all() and torch.isfinite(o["photometric_confidence"]).all()
        assert float((o["prob_volume"].sum(1) - 1.0).abs().max()) < 1e-5
        dv = o["depth_values"]
        assert bool((o["depth"] >= dv.min(1)[0] - 1e-3).all()) and bool((o["depth"] <= dv.max(1)[0] + 1e-3).all())
    torch.cuda.empty_cache()


def test_stage_loop_feature_cache(dev):
    """runner.py:178-243 through svs_hip.stage_loop.StageLoop: three reference views x three stages with the images'
    features extracted once each (3 calls instead of 27), identical outputs to the uncached loop, and the depth
    hand-off between stages."""
    from svs_hip.stage_loop import StageLoop
    torch.manual_seed(0)
    m = _model(dev, [16, 8, 8])
    H, W = 64, 96
    rng = np.random.default_rng(3)
    images = [torch.from_numpy(rng.uniform(0, 1, (1, 3, H, W)).astype(F32)).to(dev) for _ in range(3)]
    _, proj, depth_values = synth.make_mvs_sample(11, img_hw=(H, W))
    samples = []
    for ref in range(3):
        order = [ref] + [v for v in range(3) if v != ref]
        # fresh tensors every time, as a data loader would hand them out
        samples.append(dict(imgs=torch.stack([images[v].clone() for v in order], 1), depth_values=G(depth_values, dev)[None],
                            proj_matrices={k: G(v, dev)[None] for k, v in proj.items()}))
    results = {}
    for cached in (True, False):
        loop = StageLoop(m, cache_features=cached)
        outs = [None] * 3
        for st in range(3):
            outs, _ = loop.cost_volumes(st, samples, outs)
            depths = [o[f"stage{st + 1}"]["depth"] * 1.01 for o in outs]      # stands in for the rendered depths
            outs = StageLoop.hand_off_depth(outs, st, depths)
            assert all(o["depth"] is d for o, d in zip(outs, depths))
        results[cached] = (loop.feature_calls, [o["stage3"]["depth"].cpu().numpy() for o in outs],
                           [o["stage3"]["photometric_confidence"].cpu().numpy() for o in outs])
    assert results[True][0] == 3 and results[False][0] == 27
    for a, b in zip(results[True][1] + results[True][2], results[False][1] + results[False][2]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("hw", [None, (64, 80), (20, 12)])
def test_feature_net_hip(dev, golden_dir, hw):
    """FeatureNet on csrc/svs_conv2d.hip (BatchNorm folded, FPN up-sampling fused into the lateral convolutions) against
    the reference's outputs (fixture) and the torch-functional oracle on other image sizes; batch of 2."""
    from models.CasMVSNet import FeatureNet
    g = dict(np.load(os.path.join(golden_dir, "featurenet.npz")))
    params = synth.make_featurenet_params(int(g["seed"]))
    net = FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode="fpn")
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}, strict=True)
    net.to(dev).eval()
    if hw is None:
        img, want = g["img"][0], {k: g[k] for k in ("stage1", "stage2", "stage3")}
    else:
        img = np.random.default_rng(hw[0]).uniform(0, 1, (3,) + hw).astype(F32)
        want = corc.feature_net_torch(params, img)
    x = G(np.stack([img, img[:, ::-1].copy()]), dev)                         # second batch entry: the image flipped
    with torch.no_grad():
        out = net(x)
    want2 = corc.feature_net_torch(params, img[:, ::-1].copy())
    for k in ("stage1", "stage2", "stage3"):
        assert out[k].shape == (2,) + want[k].shape
        np.testing.assert_allclose(out[k][0].cpu().numpy(), want[k], atol=1e-5)
        np.testing.assert_allclose(out[k][1].cpu().numpy(), want2[k], atol=1e-5)
    assert np.abs(want["stage3"]).max() > 0.1
    net.train()                                                             # train mode: the torch modules (autograd), not the HIP path
    assert net(x)["stage1"].requires_grad


@pytest.mark.parametrize("k,stride,Cin,Cout,hw", [(3, 1, 16, 16, (37, 70)), (3, 1, 32, 32, (24, 132)), (3, 1, 32, 16, (16, 64)),
                                                   (3, 1, 32, 8, (9, 33)), (3, 1, 8, 8, (21, 40)), (3, 1, 32, 21, (13, 47)),
                                                   (5, 2, 16, 32, (38, 70)), (5, 2, 8, 16, (25, 131)), (5, 2, 16, 9, (12, 36)),
                                                   (3, 1, 32, 8, (262, 530)), (3, 1, 16, 16, (259, 545))])      # (8 x 32 windows)
def test_conv2d_mfma_single_layer(dev, k, stride, Cin, Cout, hw):
    """svs_conv2d_mfma (FeatureNet's wide layers on the fp16x2 matrix-core path, csrc/svs_conv2d_mfma.hip) against a float64
    torch convolution on ragged sizes (windows cut by both image edges, odd sizes under stride 2, Cout not a multiple of 16):
    3e-6 of the output scale -- the float32 class (two 11-bit pieces per operand, float32 accumulation) -- and against the
    float32 vector kernel it replaces in the pyramid."""
    from svs_hip import costvol
    H, W = hw
    rs = np.random.default_rng(100 * k + Cin + Cout)
    x = rs.standard_normal((Cin, H, W)).astype(F32)
    w = (rs.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(F32)
    b = rs.standard_normal(Cout).astype(F32)
    assert costvol.conv2d_mfma_supported(Cin, Cout, k, stride)
    for relu in (True, False):
        ref = torch.nn.functional.conv2d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                         stride=stride, padding=k // 2)[0]
        ref = (ref.clamp(min=0) if relu else ref).numpy()
        got = costvol.conv2d_mfma(G(x, dev), G(w, dev), G(b, dev), stride=stride, relu=relu).cpu().numpy()
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=3e-6 * np.abs(ref).max())
        old = costvol.conv2d(G(x, dev), G(w, dev), G(b, dev), stride=stride, relu=relu).cpu().numpy()
        np.testing.assert_allclose(got, old, atol=2e-5)
    got = costvol.conv2d_mfma(G(x, dev), G(w, dev), None, stride=stride, relu=False).cpu().numpy()      # no bias
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), None, stride=stride,
                                     padding=k // 2)[0].numpy()
    np.testing.assert_allclose(got, ref, atol=3e-6 * np.abs(ref).max())
    assert not costvol.conv2d_mfma_supported(3, 8, 3, 1) and not costvol.conv2d_mfma_supported(32, 40, 3, 1)


@pytest.mark.parametrize("hw,Cout", [((38, 70), 8), ((12, 34), 16), ((264, 528), 8)])
def test_conv2d_mfma_lateral_fusion(dev, hw, Cout):
    """svs_conv2d_mfma_lateral (the FPN's inner2 + up-sampled addend formed inside out3's window conversion) == the two
    launches it replaces, bit for bit: the float32 1x1 kernel with the x2 up-sampled addend, then svs_conv2d_mfma -- and both
    against float64 torch."""
    from svs_hip import costvol
    H, W = hw
    rs = np.random.default_rng(H + Cout)
    c0 = rs.standard_normal((8, H, W)).astype(F32)
    f1 = rs.standard_normal((32, H // 2, W // 2)).astype(F32)
    w1 = (rs.standard_normal((32, 8, 1, 1)) / np.sqrt(8)).astype(F32)
    b1 = rs.standard_normal(32).astype(F32)
    w3 = (rs.standard_normal((Cout, 32, 3, 3)) / np.sqrt(288)).astype(F32)
    b3 = rs.standard_normal(Cout).astype(F32)
    f2 = costvol.conv2d(G(c0, dev), G(w1, dev), G(b1, dev), add=G(f1, dev), add_upsample2=True, relu=False)
    two = costvol.conv2d_mfma(f2, G(w3, dev), G(b3, dev), relu=False)
    one = costvol.conv2d_mfma_lateral(G(c0, dev), G(w1, dev), G(b1, dev), G(f1, dev), G(w3, dev), G(b3, dev), relu=False)
    assert torch.equal(one, two)
    x = torch.nn.functional.conv2d(torch.from_numpy(c0).double()[None], torch.from_numpy(w1).double(), torch.from_numpy(b1).double())
    x = x + torch.nn.functional.interpolate(torch.from_numpy(f1).double()[None], scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(x, torch.from_numpy(w3).double(), torch.from_numpy(b3).double(), padding=1)[0].numpy()
    np.testing.assert_allclose(one.cpu().numpy(), ref, atol=4e-6 * np.abs(ref).max())


@pytest.mark.parametrize("k,stride", [(1, 1), (1, 2), (3, 1), (3, 2), (5, 1), (5, 2)])
def test_conv2d_single_layer(dev, k, stride):
    """svs_conv2d alone (the entry the fused pyramid call is built from) against torch.nn.functional.conv2d on the CPU:
    ragged sizes (tiles cut by both image edges), channel counts that are not multiples of 8, bias / ReLU / both kinds
    of addend.  float32 FMA chains in a different order than torch's: 2e-5 absolute on O(1) outputs."""
    from svs_hip import costvol
    rs = np.random.default_rng(10 * k + stride)
    for (Cin, Cout, H, W) in ((3, 8, 37, 70), (11, 13, 24, 132), (32, 8, 16, 64)):
        x = rs.standard_normal((Cin, H, W)).astype(F32)
        w = (rs.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(F32)
        b = rs.standard_normal(Cout).astype(F32)
        ref = torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), stride=stride,
                                         padding=k // 2)[0]
        got = costvol.conv2d(G(x, dev), G(w, dev), G(b, dev), stride=stride, relu=False)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=2e-5)
        Ho, Wo = ref.shape[1:]
        add = rs.standard_normal((Cout, Ho, Wo)).astype(F32)
        got = costvol.conv2d(G(x, dev), G(w, dev), None, add=G(add, dev), stride=stride, relu=True)
        ref2 = torch.relu(ref - torch.from_numpy(b)[:, None, None]) + torch.from_numpy(add)
        np.testing.assert_allclose(got.cpu().numpy(), ref2.numpy(), atol=2e-5)
        if Ho % 2 == 0 and Wo % 2 == 0:
            half = rs.standard_normal((Cout, Ho // 2, Wo // 2)).astype(F32)
            got = costvol.conv2d(G(x, dev), G(w, dev), G(b, dev), add=G(half, dev), add_upsample2=True, stride=stride)
            up = torch.nn.functional.interpolate(torch.from_numpy(half)[None], scale_factor=2, mode="nearest")[0]
            np.testing.assert_allclose(got.cpu().numpy(), (ref + up).numpy(), atol=2e-5)
