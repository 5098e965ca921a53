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


def make_train_rng(R, seed=0, n_bins=128, n_final=98):
    """The train-mode random draws of SURVEY.md note R as explicit arrays (numpy, seeded)."""
    rng = np.random.default_rng(seed + 2000)
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
