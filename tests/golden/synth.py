"""Seeded synthetic inputs shared by the fixture generator, the tests, smoke() and bench.py.

Everything is drawn with numpy's PCG64 (bit-reproducible on any platform), so golden
fixtures only need to store inputs that are not re-derivable and the expected outputs.
Shapes follow the reference's DTU configuration (config/vol/dtu.yaml:27-56,
config/ours.yaml:22-24); parameter names are the reference's state-dict keys
(SURVEY.md section 8a "Parameter shapes").
"""
import numpy as np

F32 = np.float32

SDF_DIMS = [39, 256, 256, 256, 256, 256, 256, 256, 256, 257]   # lin3 emits 217 (skip at 4)
RGB_DIMS = [271, 256, 256, 256, 256, 3]


def make_params(seed=0, trained=True):
    """State-dict-named float32 arrays for the fg DTU model (797 883 parameters).

    Follows the distributions of the geometric initialisation (volsdf/model/network.py:46-62)
    so the SDF is roughly a sphere of radius 0.6; `trained=True` adds a seeded perturbation to
    every tensor (weights, gains, biases) so that no layer is in a degenerate state.
    """
    rng = np.random.default_rng(seed)
    p = {}
    for l in range(9):
        d_in = SDF_DIMS[l]
        d_out = SDF_DIMS[l + 1] - (39 if l + 1 == 4 else 0)
        if l == 8:
            w = rng.normal(np.sqrt(np.pi) / np.sqrt(d_in), 1e-4, (d_out, d_in))
            b = np.full(d_out, -0.6)
        else:
            w = rng.normal(0.0, np.sqrt(2) / np.sqrt(d_out), (d_out, d_in))
            b = np.zeros(d_out)
            if l == 0:
                w[:, 3:] = 0.0
            if l == 4:
                w[:, -36:] = 0.0
        if trained:
            w = w + rng.normal(0.0, 0.02 if l < 8 else 0.01, w.shape) * (0.3 if l in (0, 4) else 1.0)
            b = b + rng.normal(0.0, 0.02, b.shape)
        g = np.sqrt((w ** 2).sum(1, keepdims=True))
        if trained:
            g = g * (1.0 + rng.normal(0.0, 0.02, g.shape))
        p[f"implicit_network.lin{l}.weight_v"] = w.astype(F32)
        p[f"implicit_network.lin{l}.weight_g"] = g.astype(F32)
        p[f"implicit_network.lin{l}.bias"] = b.astype(F32)
    for l in range(5):
        d_in, d_out = RGB_DIMS[l], RGB_DIMS[l + 1]
        k = 1.0 / np.sqrt(d_in)
        w = rng.uniform(-k, k, (d_out, d_in))
        b = rng.uniform(-k, k, d_out)
        g = np.sqrt((w ** 2).sum(1, keepdims=True))
        p[f"rendering_network.lin{l}.weight_v"] = w.astype(F32)
        p[f"rendering_network.lin{l}.weight_g"] = g.astype(F32)
        p[f"rendering_network.lin{l}.bias"] = b.astype(F32)
    p["density.beta"] = np.asarray(0.1, F32)
    return p


def make_trained_params(seed=1):
    """A second weight set at the scale of a TRAINED network (make_params stays near the geometric initialisation):
    log-normal spread of the weight-norm gains (up to ~3x), dense perturbations of the directions incl. the
    positional-encoding columns of lin0, larger biases, a radiance network with 1.35x gains, beta = 0.005.  A fifth to a
    third of the pre-activations lie beyond +-0.2 (softplus(beta=100) is the identity / zero there), activations reach
    ~15, d sdf/dx up to ~20.  The sdf bias is re-centred so that the zero level set stays inside the scene (most rays
    cross a surface).  Everything seeded: fixtures store only the seed."""
    rng = np.random.default_rng(1000 + seed)
    p = dict(make_params(seed))
    for l in range(9):
        v = p[f"implicit_network.lin{l}.weight_v"].astype(np.float64)
        g = p[f"implicit_network.lin{l}.weight_g"].astype(np.float64)
        b = p[f"implicit_network.lin{l}.bias"].astype(np.float64)
        if l < 8:
            v = v + rng.normal(0, 0.02, v.shape)
            if l == 0:
                v[:, 3:] += rng.normal(0, 0.01, v[:, 3:].shape)
            g = g * np.exp(rng.normal(0.05, 0.2, g.shape))
            b = b + rng.normal(0, 0.05, b.shape)
        else:
            v[1:] = v[1:] + rng.normal(0, 0.05, v[1:].shape)
            g[1:] = g[1:] * np.exp(rng.normal(0, 0.3, g[1:].shape))
            b[1:] += rng.normal(0, 0.1, b[1:].shape)
        p[f"implicit_network.lin{l}.weight_v"] = v.astype(F32)
        p[f"implicit_network.lin{l}.weight_g"] = g.astype(F32)
        p[f"implicit_network.lin{l}.bias"] = b.astype(F32)
    for l in range(5):
        g = p[f"rendering_network.lin{l}.weight_g"].astype(np.float64)
        p[f"rendering_network.lin{l}.weight_g"] = (g * np.exp(rng.normal(0.3, 0.3, g.shape))).astype(F32)
    p["density.beta"] = np.asarray(0.005, F32)
    # re-centre: sdf = 0 on average over the sphere |x| = 0.7 (float64 forward of the weight-normed network)
    d = rng.normal(0, 1, (2048, 3))
    x = 0.7 * d / np.linalg.norm(d, axis=1, keepdims=True)
    pe = [x] + [f(x * 2.0 ** k) for k in range(6) for f in (np.sin, np.cos)]
    pe = np.concatenate(pe, 1)
    h = pe
    for l in range(9):
        v = p[f"implicit_network.lin{l}.weight_v"].astype(np.float64)
        w = p[f"implicit_network.lin{l}.weight_g"].astype(np.float64) * v / np.linalg.norm(v, axis=1, keepdims=True)
        if l == 4:
            h = np.concatenate([h, pe], 1) / np.sqrt(2.0)
        a = h @ w.T + p[f"implicit_network.lin{l}.bias"].astype(np.float64)
        h = np.maximum(a, 0) + np.log1p(np.exp(-np.abs(100 * a))) / 100 if l < 8 else a
    b8 = p["implicit_network.lin8.bias"].astype(np.float64)
    b8[0] -= h[:, 0].mean()
    p["implicit_network.lin8.bias"] = b8.astype(F32)
    return p


WEIGHT_SETS = {"w0": lambda: make_params(0), "w1": lambda: make_trained_params(1)}


BG_SDF_DIMS = [84, 256, 256, 256, 256, 256, 256, 256, 256, 257]     # lin3 emits 256 - 84 = 172 rows (skip at 4)


def make_bg_params(seed=0):
    """State-dict-named float32 arrays of the inverted-sphere background networks of VolSDFNetworkBG (563 504
    parameters): bg_implicit_network (4-D points, PE-10, no weight-norm, no geometric init) and
    bg_rendering_network (mode 'nerf': PE-4 view dirs + feature -> 128 -> 3).  nn.Linear's default initialisation
    (uniform +-1/sqrt(fan_in)), with the last implicit layer scaled up so that densities and features are O(1)."""
    rng = np.random.default_rng(seed + 77)
    p = {}
    for l in range(9):
        d_in = BG_SDF_DIMS[l]
        d_out = BG_SDF_DIMS[l + 1] - (84 if l + 1 == 4 else 0)
        k = 1.0 / np.sqrt(d_in)
        gain = np.sqrt(6.0) if l < 8 else 4.0
        p[f"bg_implicit_network.lin{l}.weight"] = (gain * rng.uniform(-k, k, (d_out, d_in))).astype(F32)
        p[f"bg_implicit_network.lin{l}.bias"] = rng.uniform(-k, k, d_out).astype(F32)
    for l, (d_in, d_out) in enumerate(((283, 128), (128, 3))):
        k = 1.0 / np.sqrt(d_in)
        p[f"bg_rendering_network.lin{l}.weight"] = (np.sqrt(3.0) * rng.uniform(-k, k, (d_out, d_in))).astype(F32)
        p[f"bg_rendering_network.lin{l}.bias"] = rng.uniform(-k, k, d_out).astype(F32)
    return p


def make_camera(res_hw=(576, 768), skew=0.0, center=(0.0, 0.0, -2.5), tilt=0.0):
    """Pin-hole K (4,4) and camera-to-world pose (4,4) looking at the origin."""
    h, w = res_hw
    f = 700.0 * (w / 768.0)
    K = np.eye(4, dtype=F32)
    K[0, 0] = K[1, 1] = f
    K[0, 1] = skew
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    c, s = np.cos(tilt), np.sin(tilt)
    pose = np.eye(4, dtype=F32)
    pose[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], F32)
    pose[:3, 3] = np.asarray(center, F32)
    return K, pose


def make_uv(n, seed=0, res_hw=(576, 768), margin=0.25):
    """n distinct integer pixels (x=col, y=row) from the central part of the image."""
    rng = np.random.default_rng(seed + 1000)
    h, w = res_hw
    xs = rng.integers(int(w * margin), int(w * (1 - margin)), size=4 * n + 16)
    ys = rng.integers(int(h * margin), int(h * (1 - margin)), size=4 * n + 16)
    uv = np.unique(np.stack([xs, ys], -1), axis=0)
    rng.shuffle(uv)
    assert uv.shape[0] >= n
    return uv[:n].astype(F32)


def make_train_rng(R, seed=0, n_bins=128, n_final=98, bg=False):
    """The train-mode random draws of SURVEY.md note R as explicit arrays (numpy, seeded).  bg: adds the jitter of the
    inverse-sphere sampler (rand(R,32), drawn after the eikonal-sample pick, ray_sampler.py:215)."""
    rng = np.random.default_rng(seed + 2000)
    d = _train_rng(rng, R, n_bins, n_final)
    if bg:
        d["jitter_bg"] = rng.random((R, 32), dtype=F32)
    return d


def _train_rng(rng, R, n_bins, n_final):
    return {
        "jitter": rng.random((R, 128), dtype=F32),
        "u": rng.random((R, 64), dtype=F32),
        "perm": rng.permutation(n_bins).astype(np.int64),
        "eik_idx": rng.integers(0, n_final, size=R).astype(np.int64),
        "eik_points": rng.uniform(-3.0, 3.0, (R, 3)).astype(F32),
    }


def make_mvs_views(seed, D=48, Hc=36, Wc=48, n_views=3):
    """Synthetic MVS prior: softmax(N(0,1)) probability volumes, per-pixel depth hypotheses,
    three cameras with x offsets 0, +-0.3 looking at the origin."""
    rng = np.random.default_rng(seed)
    views = []
    for j, dx in enumerate((0.0, 0.3, -0.3)[:n_views]):
        K, pose = make_camera(center=(dx, 0.02 * j, -2.5), tilt=-0.12 * dx / 0.3, skew=0.4 * j)
        logits = rng.normal(0, 1, (D, Hc, Wc))
        prob = np.exp(logits) / np.exp(logits).sum(0, keepdims=True)
        base = np.linspace(1.5, 3.5, D)[:, None, None] * (1.0 + 0.05 * rng.uniform(-1, 1, (1, Hc, Wc)))
        if j == 2:
            base[:, :4, :] = 0.0            # a band of invalid hypotheses (near < 1e-5)
        views.append(dict(K=K, c2w=pose, cost=prob.astype(F32), z_mvs=base.astype(F32)))
    return views


# ---------------------------------------------------------------------------------------------------------
# CasMVSNet cost-volume inputs (SURVEY.md 8d "C3"), scaled down for fixtures
# ---------------------------------------------------------------------------------------------------------
COSTREG_LAYERS = [("conv0", None, 1), ("conv1", 1, 2), ("conv2", 2, 2), ("conv3", 2, 4), ("conv4", 4, 4),
                  ("conv5", 4, 8), ("conv6", 8, 8)]      # (name, in multiple of base (None = in_channels), out multiple)
COSTREG_DECONV = [("conv7", 8, 4), ("conv9", 4, 2), ("conv11", 2, 1)]


def make_costreg_params(seed, in_channels, base=8):
    """State-dict-named float32 arrays of one CostRegNet (models/CasMVSNet.py:441-472), BN in eval form."""
    rng = np.random.default_rng(seed)
    p = {}

    def bn(name, c):
        p[f"{name}.bn.weight"] = rng.uniform(0.6, 1.4, c).astype(F32)
        p[f"{name}.bn.bias"] = rng.normal(0, 0.1, c).astype(F32)
        p[f"{name}.bn.running_mean"] = rng.normal(0, 0.1, c).astype(F32)
        p[f"{name}.bn.running_var"] = rng.uniform(0.5, 1.5, c).astype(F32)
        p[f"{name}.bn.num_batches_tracked"] = np.asarray(1, np.int64)

    for name, ci, co in COSTREG_LAYERS:
        cin = in_channels if ci is None else ci * base
        cout = co * base
        p[f"{name}.conv.weight"] = rng.normal(0, np.sqrt(2.0 / (27 * cin)), (cout, cin, 3, 3, 3)).astype(F32)
        bn(name, cout)
    for name, ci, co in COSTREG_DECONV:
        cin, cout = ci * base, co * base
        p[f"{name}.conv.weight"] = rng.normal(0, np.sqrt(2.0 / (27 * cin / 8)), (cin, cout, 3, 3, 3)).astype(F32)
        bn(name, cout)
    p["prob.weight"] = rng.normal(0, np.sqrt(2.0 / (27 * base)), (1, base, 3, 3, 3)).astype(F32)
    return p


def make_mvs_sample(seed, img_hw=(64, 96), n_views=3, numdepth=192):
    """Synthetic MVSDataset item (datasets/general_eval.py:178-273 layout): per-stage features, projection
    matrices (V,2,4,4) per stage ([v,0] = extrinsic, [v,1,:3,:3] = stage intrinsics) and depth_values (numdepth,)."""
    rng = np.random.default_rng(seed)
    H, W = img_hw
    feats = []
    chans = {1: 32, 2: 16, 3: 8}
    for v in range(n_views):
        f = {}
        for st, sc in ((1, 4), (2, 2), (3, 1)):
            h, w = H // sc, W // sc
            yy, xx = np.meshgrid(np.arange(h) / h, np.arange(w) / w, indexing="ij")
            base = np.stack([np.sin(2 * np.pi * (k % 5 + 1) * xx + k + 0.3 * v) * np.cos(2 * np.pi * (k % 3 + 1) * yy - k)
                             for k in range(chans[st])], 0)
            f[f"stage{st}"] = (base + 0.2 * rng.normal(0, 1, base.shape)).astype(F32)
        feats.append(f)
    proj = {}
    for st, sc in ((1, 4), (2, 2), (3, 1)):
        P = np.zeros((n_views, 2, 4, 4), F32)
        for v, tx in enumerate((0.0, 30.0, -30.0)[:n_views]):
            ext = np.eye(4, dtype=F32)
            ang = 0.02 * v
            ext[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], F32)
            ext[0, 3] = tx
            ext[1, 3] = 3.0 * v
            Kst = np.eye(4, dtype=F32)
            f_img = 300.0 * (W / 160.0) * 4.0
            Kst[0, 0] = Kst[1, 1] = f_img / sc
            Kst[0, 2], Kst[1, 2] = (W / 2.0) / sc, (H / 2.0) / sc
            P[v, 0], P[v, 1] = ext, Kst
        proj[f"stage{st}"] = P
    depth_values = (425.0 + 2.5 * 1.06 * np.arange(numdepth)).astype(F32)
    return feats, proj, depth_values


def rows_to_tiles(x):
    """(P, F) row-major features -> wave-tile activation blocks (ceil(P/32), 128*64): the layout of
    svs_mlp.hip (block float index ((i//4)*64 + lane)*4 + i%4, i = 16*tile + r, feature = 32*tile + rho(r) + 4*half,
    lane = point%32 + 32*half).  F <= 256; missing rows / points are zero."""
    P, Fdim = x.shape
    nt = (P + 31) // 32
    out = np.zeros((nt, 128 * 64), F32)
    f = np.arange(Fdim)
    t, local = f // 32, f % 32
    half = (local >> 2) & 1
    r = (local & 3) + 4 * (local >> 3)
    i = 16 * t + r
    p = np.arange(P)
    wt, col = p // 32, p % 32
    lane = col[:, None] + 32 * half[None, :]
    idx = ((i[None, :] // 4) * 64 + lane) * 4 + (i[None, :] % 4)
    out[wt[:, None].repeat(Fdim, 1), idx] = x
    return out


def _fragment_index(Fdim):
    """feature f -> (k-step s, element j, lane half) of the fp16 fragment layout of csrc/svs_blocks_h2.h:
    f = 16 s + 8 (j >> 2) + 4 half + (j & 3)"""
    f = np.arange(Fdim)
    s, w = f // 16, f % 16
    half = (w >> 2) & 1
    j = (w & 3) + 4 * (w >> 3)
    return s, j, half


def _piece_slot(s, lane):
    """piece_slot() of csrc/svs_blocks_h2.h"""
    b, a, q = lane >> 5, (lane >> 2) & 7, lane & 3
    return 16 * (a >> 1) + 8 * ((a & 1) ^ (s & 1)) + 4 * b + q


def rows_to_pair_block(x):
    """(P, F<=256) float32 rows -> PAIR blocks of the fp16x2 kernels (csrc/svs_blocks_h2.h), (ceil(P/32), 128*64)
    float32 words: hi plane [16 k-steps][64 lanes][8 fp16], then the mid plane, value = hi + mid."""
    P, Fdim = x.shape
    nt = (P + 31) // 32
    hi = x.astype(np.float16)
    mid = (x - hi.astype(F32)).astype(np.float16)
    out = np.zeros((nt, 2, 16, 64, 8), np.float16)
    s, j, half = _fragment_index(Fdim)
    p = np.arange(P)
    wt, lane = (p // 32)[:, None], (p % 32)[:, None] + 32 * half[None, :]
    slot = _piece_slot(s[None, :], lane)
    out[wt, 0, s[None, :], slot, j[None, :]] = hi
    out[wt, 1, s[None, :], slot, j[None, :]] = mid
    return out.reshape(nt, -1).view(F32)


def rows_to_scaled_block(x, scaled=True, pair=False):
    """(P, F<=256) float32 rows -> SCALED blocks (csrc/svs_blocks_h2.h): x * s_p with s_p the power of two that puts the
    point's largest magnitude in [2^4, 2^5) (1 when not `scaled`), as the hi plane alone (pair=False, the
    SVS_MMA_F16X2_HALF format) or as hi plane + mid plane (pair=True, SVS_MMA_F16X2).  Returns (blocks (ceil(P/32), 128*64)
    float32 words, records (ceil(P/32), 64) = [scale(32), max(32)], the values the block actually holds (P, F))."""
    P, Fdim = x.shape
    nt = (P + 31) // 32
    mx = np.abs(x).max(1)
    if scaled:
        e = np.floor(np.log2(np.maximum(mx, 2.0 ** -102)))
        sp = (2.0 ** (4 - e)).astype(F32)
    else:
        sp = np.ones(P, F32)
    xs = (x * sp[:, None]).astype(F32)
    hi = xs.astype(np.float16)
    mid = (xs - hi.astype(F32)).astype(np.float16)
    out = np.zeros((nt, 2, 16, 64, 8), np.float16)
    s, j, half = _fragment_index(Fdim)
    p = np.arange(P)
    wt, lane = (p // 32)[:, None], (p % 32)[:, None] + 32 * half[None, :]
    slot = _piece_slot(s[None, :], lane)
    out[wt, 0, s[None, :], slot, j[None, :]] = hi
    if pair:
        out[wt, 1, s[None, :], slot, j[None, :]] = mid
    words = out.reshape(nt, -1).view(F32)
    rec = np.ones((nt, 64), F32)
    rec[:, 32:] = 0.0
    rec.reshape(-1)[(p // 32) * 64 + p % 32] = sp
    rec.reshape(-1)[(p // 32) * 64 + 32 + p % 32] = mx
    held = hi.astype(F32) + (mid.astype(F32) if pair else 0.0)
    return words, rec, held / sp[:, None]


def make_fusion_views(seed, hw=(48, 64), n_views=3, noise=2e-3):
    """Synthetic input of the depth-fusion filter (runner.py:301-332): n_views cameras on an arc looking at a unit
    sphere in front of a plane; per view K (3,3), E (4,4) world->camera (float32, as read_camera_parameters returns),
    the analytic z-depth map with multiplicative noise, a few zero holes and outliers, a confidence map and an image."""
    rng = np.random.default_rng(seed)
    H, W = hw
    views = {}
    for v in range(n_views):
        ang = 0.25 * (v - (n_views - 1) / 2.0)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float64)
        c = np.array([3.2 * np.sin(ang), 0.1 * v, -3.2 * np.cos(ang)])            # camera centre, looking at the origin
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, -R @ c
        K = np.array([[1.1 * W, 0.3, W / 2.0 + 0.5 * v], [0, 1.1 * W, H / 2.0 - 0.25 * v], [0, 0, 1]], np.float64)
        u, w_ = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        d_cam = np.linalg.inv(K) @ np.stack([u.ravel(), w_.ravel(), np.ones(H * W)])
        d_w = R.T @ d_cam                                                         # (3, HW), z_cam = 1 per unit t
        b = (d_w * c[:, None]).sum(0)
        a = (d_w * d_w).sum(0)
        disc = b * b - a * (c @ c - 1.0)
        t_sph = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / a, np.inf)
        t_pl = (1.5 - c[2]) / d_w[2]                                              # plane z_world = 1.5 behind the sphere
        depth = np.minimum(t_sph, t_pl).reshape(H, W)
        depth = depth * (1.0 + noise * rng.normal(0, 1, (H, W)))
        holes = rng.uniform(0, 1, (H, W)) < 0.02
        depth[holes] = 0.0
        outl = rng.uniform(0, 1, (H, W)) < 0.03
        depth[outl] *= rng.uniform(0.8, 1.25, outl.sum())
        views[v] = dict(K=K.astype(F32), E=E.astype(F32), depth=depth.astype(F32),
                        confidence=rng.uniform(0, 1, (H, W)).astype(F32),
                        img=(rng.integers(0, 256, (H, W, 3)).astype(F32) / F32(255.)))
    return views


def make_dtu_scan(seed, n_pred=12000, n_stl=10000):
    """Synthetic stand-in for one DTU evaluation scan (evals/eval_dtu.py inputs, millimetres): a ground-truth cloud on
    a bumpy sphere, a predicted cloud covering part of it with noise, near-duplicate clusters (so that the 0.2 mm
    down-sampling has work to do), far outliers and points outside the padded bounding box; ObsMask / BB / Res as in
    ObsMask*_10.mat and the ground plane of Plane*.mat."""
    rng = np.random.default_rng(seed)
    c = np.array([5.0, -3.0, 2.0])

    def surface(n):
        d = rng.normal(0, 1, (n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = 40.0 + 1.5 * np.sin(5 * d[:, :1]) * np.cos(4 * d[:, 1:2])
        return c + r * d, d

    stl, _ = surface(n_stl)
    n_base = n_pred * 6 // 10
    base, nrm = surface(n_base)
    vis = nrm[:, 2] > -0.3                                       # the prediction misses the bottom cap
    base, nrm = base[vis], nrm[vis]
    base = base + nrm * rng.normal(0, 0.3, (len(base), 1))
    n_dup = n_pred * 3 // 10
    dup = base[rng.integers(0, len(base), n_dup)] + rng.normal(0, 0.06, (n_dup, 3))
    n_out = n_pred - len(base) - n_dup
    far = c + rng.normal(0, 1, (n_out, 3)) * rng.uniform(45, 140, (n_out, 1))
    data_pcd = np.concatenate([base, dup, far], 0)
    BB = np.array([[-45.3, -50.1, -40.7], [55.2, 45.9, 45.4]], F32)
    Res = np.array([[2.0]])
    dims = tuple(int(v) for v in np.ceil((BB[1] - BB[0]) / 2.0).astype(int) + 1)
    ObsMask = (rng.uniform(0, 1, dims) > 0.1).astype(np.uint8)
    ObsMask[:, :, :4] = 0
    P = np.array([[0.05, -0.02, 1.0, 25.0]])
    return dict(data_pcd=data_pcd, stl=stl, ObsMask=ObsMask, BB=BB, Res=Res, P=P)


def make_dtu_mesh(seed, rows=9, cols=13):
    """Synthetic predicted MESH for the evaluator's mesh mode (evals/eval_dtu.py:62-90), in the frame of make_dtu_scan: a
    (rows x cols)-cell patch of the bumpy sphere's upper side with jittered vertices, two triangles per cell with
    alternating diagonals, plus three degenerate triangles (a repeated corner twice, three equal corners once) that the
    script's area test has to drop.  Returns (vertices (n,3) float64, triangles (m,3) int64)."""
    rng = np.random.default_rng(seed)
    c = np.array([5.0, -3.0, 2.0])
    th = np.linspace(0.35, 1.15, rows + 1)[:, None] + rng.normal(0, 0.004, (rows + 1, cols + 1))      # polar angle
    ph = np.linspace(-0.6, 0.7, cols + 1)[None, :] + rng.normal(0, 0.004, (rows + 1, cols + 1))
    d = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], -1).reshape(-1, 3)
    r = 40.0 + 1.5 * np.sin(5 * d[:, :1]) * np.cos(4 * d[:, 1:2]) + rng.normal(0, 0.15, (len(d), 1))
    vertices = c + r * d
    idx = np.arange((rows + 1) * (cols + 1)).reshape(rows + 1, cols + 1)
    tris = []
    for i in range(rows):
        for j in range(cols):
            a, b, e, f = idx[i, j], idx[i, j + 1], idx[i + 1, j], idx[i + 1, j + 1]
            tris += [(a, b, f), (a, f, e)] if (i + j) % 2 == 0 else [(a, b, e), (b, f, e)]
    tris.insert(7, (idx[0, 0], idx[0, 0], idx[1, 1]))
    tris.insert(40, (idx[2, 3], idx[3, 3], idx[2, 3]))
    tris.append((idx[4, 4], idx[4, 4], idx[4, 4]))
    return vertices, np.asarray(tris, np.int64)


def make_featurenet_params(seed, base=8):
    """State-dict-named float32 arrays of FeatureNet, arch_mode 'fpn' (models/CasMVSNet.py:338-399), BN in eval form."""
    rng = np.random.default_rng(seed)
    p = {}

    def block(name, cin, cout, k):
        p[f"{name}.conv.weight"] = rng.normal(0, np.sqrt(2.0 / (k * k * cin)), (cout, cin, k, k)).astype(F32)
        p[f"{name}.bn.weight"] = rng.uniform(0.6, 1.4, cout).astype(F32)
        p[f"{name}.bn.bias"] = rng.normal(0, 0.1, cout).astype(F32)
        p[f"{name}.bn.running_mean"] = rng.normal(0, 0.1, cout).astype(F32)
        p[f"{name}.bn.running_var"] = rng.uniform(0.5, 1.5, cout).astype(F32)
        p[f"{name}.bn.num_batches_tracked"] = np.asarray(1, np.int64)

    b = base
    for name, cin, cout, k in (("conv0.0", 3, b, 3), ("conv0.1", b, b, 3), ("conv1.0", b, 2 * b, 5), ("conv1.1", 2 * b, 2 * b, 3),
                               ("conv1.2", 2 * b, 2 * b, 3), ("conv2.0", 2 * b, 4 * b, 5), ("conv2.1", 4 * b, 4 * b, 3),
                               ("conv2.2", 4 * b, 4 * b, 3)):
        block(name, cin, cout, k)
    for name, cin, cout, k, bias in (("out1", 4 * b, 4 * b, 1, False), ("inner1", 2 * b, 4 * b, 1, True), ("inner2", b, 4 * b, 1, True),
                                     ("out2", 4 * b, 2 * b, 3, False), ("out3", 4 * b, b, 3, False)):
        p[f"{name}.weight"] = rng.normal(0, np.sqrt(1.0 / (k * k * cin)), (cout, cin, k, k)).astype(F32)
        if bias:
            p[f"{name}.bias"] = rng.normal(0, 0.1, cout).astype(F32)
    return p


# ---------------------------------------------------------------------------------------------------------------------------
# An analytic scene with known geometry (round 5: Chamfer parity, tools/chamfer_parity.py).  A sphere and a box, Lambertian
# shading under one directional light with a smooth albedo pattern; a numpy sphere tracer renders any pin-hole view of it.
# Normalised scene units (inside VolSDF's bounding sphere of radius 3; the geometric initialisation is a sphere of radius 0.6).
# ---------------------------------------------------------------------------------------------------------------------------
ANALYTIC = dict(sphere_c=(-0.22, 0.05, 0.0), sphere_r=0.42, box_c=(0.33, -0.08, 0.05), box_h=(0.26, 0.34, 0.3),
                light=(-0.35, 0.45, 0.82))          # direction the light travels (the cameras look along +z)


def analytic_sdf(x):
    """signed distance of the union sphere + axis-aligned box; x (..., 3) float64 -> (...)"""
    x = np.asarray(x, np.float64)
    ds = np.linalg.norm(x - np.asarray(ANALYTIC["sphere_c"]), axis=-1) - ANALYTIC["sphere_r"]
    q = np.abs(x - np.asarray(ANALYTIC["box_c"])) - np.asarray(ANALYTIC["box_h"])
    db = np.linalg.norm(np.maximum(q, 0.0), axis=-1) + np.minimum(q.max(-1), 0.0)
    return np.minimum(ds, db)


def analytic_normal(x, h=1e-5):
    g = np.stack([analytic_sdf(x + h * e) - analytic_sdf(x - h * e) for e in np.eye(3)], -1)
    return g / np.maximum(np.linalg.norm(g, axis=-1, keepdims=True), 1e-12)


def analytic_albedo(x):
    x = np.asarray(x, np.float64)
    base = np.stack([0.55 + 0.35 * np.sin(7.0 * x[..., 0] + 1.0), 0.55 + 0.35 * np.sin(6.0 * x[..., 1] - 0.5),
                     0.55 + 0.35 * np.sin(8.0 * x[..., 2] + 2.0)], -1)
    return np.clip(base, 0.05, 0.95)


def render_analytic_view(K, pose, hw, t_max=6.0, n_iter=96):
    """Sphere tracing from the camera (K: 4x4 or 3x3 intrinsics of an hw image, pose: camera-to-world).  -> dict(rgb (H,W,3)
    float32 in [0,1], black background; mask (H,W) bool; depth (H,W) float64 z-depth in the camera frame, 0 off the object;
    points (n,3) float64 world points of the hit pixels)."""
    H, W = hw
    K = np.asarray(K, np.float64)
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    y = (v - K[1, 2]) / K[1, 1]
    xcam = (u - K[0, 2] - K[0, 1] * y) / K[0, 0]
    d_cam = np.stack([xcam, y, np.ones_like(xcam)], -1)
    R, c = np.asarray(pose, np.float64)[:3, :3], np.asarray(pose, np.float64)[:3, 3]
    d = d_cam @ R.T
    scale = np.linalg.norm(d, axis=-1, keepdims=True)
    d = d / scale
    t = np.zeros((H, W))
    alive = np.ones((H, W), bool)
    for _ in range(n_iter):
        s = analytic_sdf(c + t[..., None] * d)
        t = np.where(alive, t + s, t)
        alive &= t < t_max
    hit = alive & (np.abs(analytic_sdf(c + t[..., None] * d)) < 1e-6)
    p = c + t[..., None] * d
    n = analytic_normal(p)
    l = np.asarray(ANALYTIC["light"], np.float64); l = l / np.linalg.norm(l)
    shade = 0.25 + 0.75 * np.maximum(0.0, -(n @ l))
    rgb = np.where(hit[..., None], analytic_albedo(p) * shade[..., None], 0.0).astype(F32)
    depth = np.where(hit, t / scale[..., 0], 0.0)
    return dict(rgb=rgb, mask=hit, depth=depth, points=p[hit])
