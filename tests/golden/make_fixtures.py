"""Golden-vector generator: imports the REFERENCE (read-only, /root/reference) on CPU in the build
container and stores arrays only (inputs + the reference's outputs) as small .npz fixtures.

Run:  python tests/golden/make_fixtures.py [name ...]
The reference cannot travel to the GPU box; the fixtures can.  No reference source is stored.
"""
import contextlib
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import synth     # noqa: E402

ref_shim.install()
import torch  # noqa: E402

# torch.exp of the reference bound to torch's open-source exp (Sleef_expf8_u10) instead of the host-dependent,
# closed-source MKL VML routine -- see ref_shim.pin_open_exp; fx_sampler_hostexp lifts the pin for its cross-check
UNPIN_EXP = ref_shim.pin_open_exp()

torch.set_num_threads(4)
F32 = np.float32


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build_model(params, beta=None, near=1e-4, bg=False):
    from volsdf.model.network import VolSDFNetwork
    m = VolSDFNetwork(ref_shim.dtu_model_conf(near=near))
    sd = {k: T(v).clone() for k, v in params.items()}
    if beta is not None:
        sd["density.beta"] = torch.tensor(beta, dtype=torch.float32)
    missing = m.load_state_dict(sd, strict=True)
    return m


@contextlib.contextmanager
def capture_sampler():
    """Record searchsorted / sort results and per-round sdf while the reference sampler runs."""
    rec = SimpleNamespace(inds=[], sort_vals=[], sort_idx=[], cdf=[], u=[])
    o_ss, o_sort = torch.searchsorted, torch.sort

    def ss(cdf, u, right=False, **kw):
        r = o_ss(cdf, u, right=right, **kw)
        rec.inds.append(r.numpy().copy()); rec.cdf.append(cdf.detach().numpy().copy()); rec.u.append(u.numpy().copy())
        return r

    def srt(x, *a, **kw):
        r = o_sort(x, *a, **kw)
        rec.sort_vals.append(r[0].detach().numpy().copy()); rec.sort_idx.append(r[1].numpy().copy())
        return r

    torch.searchsorted, torch.sort = ss, srt
    try:
        yield rec
    finally:
        torch.searchsorted, torch.sort = o_ss, o_sort


@contextlib.contextmanager
def inject_rng(draws):
    """Feed the reference's CPU RNG calls from explicit arrays (SURVEY note R order)."""
    o_rand, o_perm, o_randint, o_uniform = torch.rand, torch.randperm, torch.randint, torch.Tensor.uniform_
    q_rand = [draws["jitter"], draws["u"]] + ([draws["jitter_bg"]] if "jitter_bg" in draws else [])

    def rand(*shape, **kw):
        a = q_rand.pop(0)
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(a.shape) == shp, (a.shape, shp)
        return T(a).clone()

    def randperm(n, **kw):
        assert n == len(draws["perm"])
        return T(draws["perm"]).clone()

    def randint(high, size, **kw):
        return T(draws["eik_idx"]).clone()

    def uniform_(self, a, b):
        self.copy_(T(draws["eik_points"]))
        return self

    torch.rand, torch.randperm, torch.randint, torch.Tensor.uniform_ = rand, randperm, randint, uniform_
    try:
        yield
    finally:
        torch.rand, torch.randperm, torch.randint, torch.Tensor.uniform_ = o_rand, o_perm, o_randint, o_uniform


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------
def fx_rays():
    from volsdf.utils import rend_util
    out = {}
    for tag, skew, tilt in (("a", 0.0, 0.0), ("b", 3.5, 0.3)):
        K, pose = synth.make_camera(skew=skew, tilt=tilt, center=(0.3, -0.2, -2.5))
        uv = synth.make_uv(32, seed=1)
        dirs, cam = rend_util.get_camera_params(T(uv)[None], T(pose)[None], T(K)[None])
        tmp, _ = rend_util.get_camera_params(T(uv)[None], torch.eye(4)[None], T(K)[None])
        out.update({f"{tag}_K": K, f"{tag}_pose": pose, f"{tag}_uv": uv, f"{tag}_dirs": dirs[0].numpy(),
                    f"{tag}_cam": cam[0].numpy(), f"{tag}_depth_scale": tmp[0, :, 2:].numpy()})
    save("rays", **out)


def sample_points(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.2, 1.2, (n, 3))
    x[: n // 8] *= 3.0                      # some outside the r=3 sphere clamp region
    x[n // 8: n // 4] *= 0.05               # near the origin
    return x.astype(F32)


def fx_sdf_mlp(wset="w0"):
    params = synth.WEIGHT_SETS[wset]()
    m = build_model(params)
    x = sample_points(96, 3)
    net = m.implicit_network
    with torch.no_grad():
        out = net(T(x)).numpy()
        sdfv = net.get_sdf_vals(T(x)).numpy()
    sdf, feat, grad = net.get_outputs(T(x).clone())
    g2 = net.gradient(T(x).clone())
    save("sdf_mlp" if wset == "w0" else f"sdf_mlp_{wset}", seed=0, x=x, out=out, sdf_vals=sdfv, sdf=sdf.detach().numpy(), feat=feat.detach().numpy(),
         grad=grad.detach().numpy(), grad_raw=g2.detach().numpy())


def fx_rgb_mlp():
    params = synth.make_params(seed=0)
    m = build_model(params)
    rng = np.random.default_rng(5)
    P = 64
    pts = sample_points(P, 7)
    nrm = rng.normal(0, 1, (P, 3)).astype(F32)
    d = rng.normal(0, 1, (P, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F32)
    feat = rng.normal(0, 0.5, (P, 256)).astype(F32)
    with torch.no_grad():
        rgb = m.rendering_network(T(pts), T(nrm), T(d), T(feat)).numpy()
    save("rgb_mlp", seed=0, points=pts, normals=nrm, dirs=d, feat=feat, rgb=rgb)


def fx_density():
    from volsdf.model.density import LaplaceDensity
    rng = np.random.default_rng(9)
    sdf = np.concatenate([rng.normal(0, 0.3, 200), [0.0, 1e-8, -1e-8, 1e-3, -1e-3, 5.0, -5.0]]).astype(F32)
    out = {"sdf": sdf}
    for i, b in enumerate((0.1, 0.01, 0.001)):
        dn = LaplaceDensity(params_init={"beta": b})
        with torch.no_grad():
            out[f"sigma_{i}"] = dn(T(sdf)).numpy()
            out[f"beta_{i}"] = dn.get_beta().numpy()
    dn = LaplaceDensity(params_init={"beta": 0.1})
    br = rng.uniform(0.002, 0.2, (23, 1)).astype(F32)
    s2 = rng.normal(0, 0.2, (23, 9)).astype(F32)
    with torch.no_grad():
        out["sigma_ray"] = dn(T(s2), beta=T(br)).numpy()
    out["beta_ray"], out["sdf_ray"] = br, s2
    save("density", **out)


def run_sampler(m, dirs, cam, fast, training, draws=None):
    sdfs = []
    orig = m.implicit_network.get_sdf_vals
    betas = []
    orig_density = m.density.forward

    def density_fwd(sdf, beta=None):
        # the per-ray call at ray_sampler.py:126 ("Upsample more points") sees the round's final beta
        if beta is not None and beta.dim() == 2 and sdf.dim() == 2:
            betas.append(beta.detach().numpy().copy()[:, 0])
        return orig_density(sdf, beta=beta)

    m.density.forward = density_fwd

    def wrapped(p):
        r = orig(p)
        sdfs.append(r.detach().numpy().copy())
        return r

    m.implicit_network.get_sdf_vals = wrapped
    m.train(training)
    ctx = inject_rng(draws) if training else contextlib.nullcontext()
    with capture_sampler() as rec, ctx:
        z, z_eik = m.ray_sampler.get_z_vals(T(dirs), T(cam), m, fast=fast)
    m.implicit_network.get_sdf_vals = orig
    m.density.forward = orig_density
    # 11 per-ray density calls per round (10 bisection + the final one): keep the final one
    rec.betas = betas[10::11]
    return z, z_eik, rec, sdfs


def fx_sampler():
    params = synth.make_params(seed=0)
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1)
    R = 12
    uv = synth.make_uv(R, seed=2, margin=0.1)
    import oracle_path  # noqa
    from svs_oracle import rays_from_uv
    dirs, cam, _ = rays_from_uv(uv, pose, K)
    cam_r = np.repeat(cam[None], R, 0).astype(F32)
    for beta in (0.1, 0.01, 0.001):
        for fast in (-1, 0, 1, 2):
            if beta != 0.01 and fast in (0, 2):
                continue
            m = build_model(params, beta=beta)
            z, z_eik, rec, sdfs = run_sampler(m, dirs, cam_r, fast, False)
            arr = dict(dirs=dirs, cam=cam_r, beta_param=F32(beta), fast=fast, z=z.numpy(), n_rounds=len(sdfs),
                       inv_4log=(1.0 / (4.0 * torch.log(torch.tensor(0.1 + 1.0)))).numpy())
            for i, s in enumerate(sdfs):
                arr[f"sdf_{i}"] = s
                arr[f"beta_{i}"] = rec.betas[i]
            for i, a in enumerate(rec.inds):
                arr[f"inds_{i}"] = a; arr[f"cdf_{i}"] = rec.cdf[i]
            n_merge = len(rec.sort_idx) - 1
            for i in range(n_merge):
                arr[f"samples_idx_{i}"] = rec.sort_idx[i]; arr[f"zmerged_{i}"] = rec.sort_vals[i]
            save(f"sampler_eval_b{beta}_f{fast}", **arr)
    # train mode
    m = build_model(params, beta=0.05)
    draws = synth.make_train_rng(R, seed=4)
    z, z_eik, rec, sdfs = run_sampler(m, dirs, cam_r, 1, True, draws)
    save("sampler_train", dirs=dirs, cam=cam_r, beta_param=F32(0.05), z=z.numpy(), z_eik=z_eik.numpy(),
         sdf_0=sdfs[0], inds_0=rec.inds[0], cdf_0=rec.cdf[0], beta_0=rec.betas[0],
         inv_4log=(1.0 / (4.0 * torch.log(torch.tensor(0.1 + 1.0)))).numpy())


def fx_sampler_r256():
    """The eval sampler of the reference at R = 256 rays (fast = -1) for beta in {0.1, 0.01, 0.001}, stored compactly: per
    round the sdf values the reference evaluated, its carried beta, its searchsorted indices (uint16), the two cdf entries that
    bracket every u (what a near-tie check needs), and -- so that every round can be replayed from the REFERENCE's state --
    its merged bins and gather indices (uint16); plus the final z.  And the whole forward on the same rays."""
    params = synth.make_params(seed=0)
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1)
    R = 256
    uv = synth.make_uv(R, seed=21, margin=0.05)
    import oracle_path  # noqa
    from svs_oracle import rays_from_uv
    dirs, cam, _ = rays_from_uv(uv, pose, K)
    cam_r = np.repeat(cam[None], R, 0).astype(F32)
    inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None]}
    for beta in (0.1, 0.01, 0.001):
        m = build_model(params, beta=beta)
        z, z_eik, rec, sdfs = run_sampler(m, dirs, cam_r, -1, False)
        arr = dict(dirs=dirs, cam=cam_r, beta_param=F32(beta), fast=-1, z=z.numpy(), n_rounds=len(sdfs),
                   inv_4log=(1.0 / (4.0 * torch.log(torch.tensor(0.1 + 1.0)))).numpy())
        for i, sd in enumerate(sdfs):
            arr[f"sdf_{i}"] = sd.reshape(R, -1)
            arr[f"beta_{i}"] = rec.betas[i]
        for i, a in enumerate(rec.inds):
            cdf = rec.cdf[i]
            n = cdf.shape[1]
            assert a.max() <= n and n < 65536
            arr[f"inds_{i}"] = a.astype(np.uint16)
            arr[f"cdf_lo_{i}"] = np.take_along_axis(cdf, np.maximum(a - 1, 0), 1)
            arr[f"cdf_hi_{i}"] = np.take_along_axis(cdf, np.minimum(a, n - 1), 1)
        for i in range(len(rec.sort_idx) - 1):
            arr[f"samples_idx_{i}"] = rec.sort_idx[i].astype(np.uint16); arr[f"zmerged_{i}"] = rec.sort_vals[i]
        save(f"sampler256_b{beta}", **arr)
        m.eval()
        out = m(inp, fast=-1)
        save(f"forward256_b{beta}", K=K, pose=pose, uv=uv, beta_param=F32(beta), fast=-1,
             **{k: out[k].detach().numpy() for k in ("rgb_values", "depth_values", "normal_map", "depth_vals", "weights")})


def fx_primitives():
    """What torch's own OPEN routines return for the three primitives the sampler's bit-exactness hangs on, so that the
    restatements in oracle/svs_oracle.py (and the device functions behind `svs_test_primitives`) are pinned on any host:
    Sleef_expf8_u10 and Sleef_expm1f8_u10 as exported by libtorch_cpu.so, and torch.sum(dim=-1) on float32 rows of every
    length 1..160 plus the sampler's lengths up to 640 and two long rows that exercise the cascade levels."""
    rng = np.random.default_rng(17)
    n = 1 << 13
    x = np.concatenate([-rng.random(n) * 20, rng.random(n // 2) * 14, -np.exp(rng.random(n) * 30 - 25),
                        rng.standard_normal(n // 2) * 1e-3, -rng.random(n // 4) * 110, rng.random(n // 4) * 95,
                        [0, -0.0, 1e-45, -1e-45, 88.7, 88.73, 100, 100.5, -104, -104.5, -103.9, -87.5, -16.635,
                         -16.636, 1e-30, -1e-30, np.inf, -np.inf, 16.0, -1.0]]).astype(F32)
    arr = dict(x=x, expf=ref_shim.torch_vec8("Sleef_expf8_u10avx2")(x), expm1f=ref_shim.torch_vec8("Sleef_expm1f8_u10avx2")(x))
    assert np.array_equal(arr["expm1f"].view(np.uint32), torch.expm1(T(x)).numpy().view(np.uint32))   # torch.expm1 IS that routine
    lens = list(range(1, 161)) + [254, 255, 256, 382, 383, 384, 510, 511, 512, 513, 638, 639, 640, 641, 1023, 1024, 1025,
                                  2047, 4100, 20001]
    rows, sums = [], []
    for m in lens:
        nr = 8 if m <= 160 else (2 if m <= 641 else 1)
        r = (rng.random((nr, m)) * np.exp(rng.random((nr, 1)) * 12 - 8)).astype(F32)
        rows.append(r.ravel()); sums.append(torch.sum(T(r), -1).numpy())
    arr.update(sum_lens=np.asarray(lens), sum_rows=np.concatenate(rows), sum_out=np.concatenate(sums))
    save("primitives", **arr)


def fx_sampler_hostexp():
    """The UNPINNED reference (torch.exp = this host's MKL VML kernel) on 64 rays, beta = 0.01, fast = -1, in the compact
    layout of fx_sampler_r256, plus a probe of the host's exp (inputs and outputs): a test that finds the same exp on its
    host binds it into the oracle and must then reproduce these indices and cdf entries exactly -- the proof that the
    restated sum and expm1 are the reference's and that exp is the only host-dependent primitive."""
    global UNPIN_EXP
    UNPIN_EXP()
    try:
        params = synth.make_params(seed=0)
        K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1)
        R = 64
        uv = synth.make_uv(R, seed=22, margin=0.05)
        import oracle_path  # noqa
        from svs_oracle import rays_from_uv
        dirs, cam, _ = rays_from_uv(uv, pose, K)
        cam_r = np.repeat(cam[None], R, 0).astype(F32)
        rng = np.random.default_rng(5)
        probe = np.concatenate([-rng.random(4096) * 20, rng.random(2048) * 14, -np.exp(rng.random(2048) * 20 - 15)]).astype(F32)
        for beta in (0.01,):
            m = build_model(params, beta=beta)
            z, z_eik, rec, sdfs = run_sampler(m, dirs, cam_r, -1, False)
            arr = dict(dirs=dirs, cam=cam_r, beta_param=F32(beta), fast=-1, z=z.numpy(), n_rounds=len(sdfs),
                       inv_4log=(1.0 / (4.0 * torch.log(torch.tensor(0.1 + 1.0)))).numpy(),
                       exp_probe_in=probe, exp_probe_out=torch.exp(T(probe)).numpy(),
                       sqrt_probe_in=np.abs(probe), sqrt_probe_out=torch.sqrt(T(np.abs(probe))).numpy())
            for i, sd in enumerate(sdfs):
                arr[f"sdf_{i}"] = sd.reshape(R, -1)
                arr[f"beta_{i}"] = rec.betas[i]
            for i, a in enumerate(rec.inds):
                cdf = rec.cdf[i]
                n = cdf.shape[1]
                arr[f"inds_{i}"] = a.astype(np.uint16)
                arr[f"cdf_lo_{i}"] = np.take_along_axis(cdf, np.maximum(a - 1, 0), 1)
                arr[f"cdf_hi_{i}"] = np.take_along_axis(cdf, np.minimum(a, n - 1), 1)
            for i in range(len(rec.sort_idx) - 1):
                arr[f"samples_idx_{i}"] = rec.sort_idx[i].astype(np.uint16); arr[f"zmerged_{i}"] = rec.sort_vals[i]
            save(f"sampler64_hostexp_b{beta}", **arr)
    finally:
        UNPIN_EXP = ref_shim.pin_open_exp()


def fx_sampler_r256_more():
    """Two more 256-ray sampler runs of the reference in the compact layout of fx_sampler_r256:
    `sampler256_train_b0.05`: TRAIN mode (fast = 1: one round; stratified jitter, random u, randperm extras, eikonal pick --
    the draws of synth.make_train_rng fed through torch's CPU RNG call sites in the reference's order);
    `sampler256_bg_b0.01`: the fg + inverted-sphere background model in eval mode (fast = -1): far = sphere exit per ray,
    add_tiny = 1e-6 in the up-sampling pdf, the inverse-sphere depths returned beside the fg samples."""
    params = dict(synth.make_params(seed=0))
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1)
    R = 256
    uv = synth.make_uv(R, seed=23, margin=0.05)
    import oracle_path  # noqa
    from svs_oracle import rays_from_uv
    dirs, cam, _ = rays_from_uv(uv, pose, K)
    cam_r = np.repeat(cam[None], R, 0).astype(F32)
    inv4 = (1.0 / (4.0 * torch.log(torch.tensor(0.1 + 1.0)))).numpy()

    def pack(arr, rec, sdfs):
        for i, sd in enumerate(sdfs):
            arr[f"sdf_{i}"] = sd.reshape(R, -1)
            arr[f"beta_{i}"] = rec.betas[i]
        for i, a in enumerate(rec.inds):
            cdf = rec.cdf[i]
            n = cdf.shape[1]
            arr[f"inds_{i}"] = a.astype(np.uint16)
            arr[f"cdf_lo_{i}"] = np.take_along_axis(cdf, np.maximum(a - 1, 0), 1)
            arr[f"cdf_hi_{i}"] = np.take_along_axis(cdf, np.minimum(a, n - 1), 1)
        for i in range(len(rec.sort_idx) - 1):
            arr[f"samples_idx_{i}"] = rec.sort_idx[i].astype(np.uint16); arr[f"zmerged_{i}"] = rec.sort_vals[i]
        return arr

    m = build_model(params, beta=0.05)
    draws = synth.make_train_rng(R, seed=31)
    z, z_eik, rec, sdfs = run_sampler(m, dirs, cam_r, 1, True, draws)
    save("sampler256_train_b0.05", **pack(dict(dirs=dirs, cam=cam_r, beta_param=F32(0.05), fast=1, rng_seed=31, z=z.numpy(),
                                               z_eik=z_eik.numpy(), n_rounds=len(sdfs), inv_4log=inv4), rec, sdfs))
    bgp = dict(params); bgp.update(synth.make_bg_params(seed=0))
    m = build_bg_model(bgp, 0.01)
    (z, z_bg), z_eik, rec, sdfs = run_sampler(m, dirs, cam_r, -1, False)
    save("sampler256_bg_b0.01", **pack(dict(dirs=dirs, cam=cam_r, beta_param=F32(0.01), fast=-1, z=z.numpy(), z_bg=z_bg.numpy(),
                                            n_rounds=len(sdfs), inv_4log=inv4), rec, sdfs))


def fx_forward_r256_more():
    """Whole forwards of the reference at 256 rays beyond the eval runs of fx_sampler_r256: the DTU model in TRAIN mode
    (fast = 1, the draws of synth.make_train_rng) and the fg + inverted-sphere background model in eval (fast = -1, near_pose)
    and train mode -- integrated outputs, per-sample weights / depths, eikonal gradients."""
    params = dict(synth.make_params(seed=0))
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1)
    _, near_pose = synth.make_camera(center=(0.25, 0.0, -2.45), tilt=0.05)
    R = 256
    uv = synth.make_uv(R, seed=27, margin=0.05)
    inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None]}
    keep = ("rgb_values", "depth_values", "depth_values_all", "depth_vals", "weights", "grad_theta", "normal_map")
    m = build_model(params, beta=0.05)
    m.train()
    with inject_rng(synth.make_train_rng(R, seed=33)):
        out = m(inp, fast=1)
    save("forward256_train_b0.05", K=K, pose=pose, uv=uv, beta_param=F32(0.05), fast=1, rng_seed=33,
         **{k: out[k].detach().numpy() for k in keep if k in out})
    bgp = dict(params); bgp.update(synth.make_bg_params(seed=0))
    inp_bg = dict(inp, near_pose=T(near_pose)[None])
    m = build_bg_model(bgp, 0.01)
    m.eval()
    out = m(inp_bg, fast=-1)
    save("forward256_bg_eval_b0.01", K=K, pose=pose, near_pose=near_pose, uv=uv, beta_param=F32(0.01), fast=-1,
         **{k: out[k].detach().numpy() for k in keep if k in out})
    m = build_bg_model(bgp, 0.05)
    m.train()
    with inject_rng(synth.make_train_rng(R, seed=35, n_final=98, bg=True)):
        out = m(inp_bg, fast=1)
    save("forward256_bg_train_b0.05", K=K, pose=pose, near_pose=near_pose, uv=uv, beta_param=F32(0.05), fast=1, rng_seed=35,
         **{k: out[k].detach().numpy() for k in keep if k in out})


def fx_forward_r1024_train():
    """The DTU model's TRAIN-mode forward (fast = 1) of the reference at the bench batch size, 1024 rays (configs[1]):
    integrated outputs and eikonal gradients of every ray, the per-sample arrays of every 8th ray (size)."""
    params = dict(synth.make_params(seed=0))
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1)
    R = 1024
    uv = synth.make_uv(R, seed=29, margin=0.05)
    inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None]}
    m = build_model(params, beta=0.05)
    m.train()
    with inject_rng(synth.make_train_rng(R, seed=37)):
        out = m(inp, fast=1)
    o = {k: out[k].detach().numpy() for k in ("rgb_values", "depth_values", "depth_vals", "weights", "grad_theta")}
    save("forward1024_train_b0.05", K=K, pose=pose, uv=uv, beta_param=F32(0.05), fast=1, rng_seed=37, every=8,
         rgb_values=o["rgb_values"], depth_values=o["depth_values"], grad_theta=o["grad_theta"],
         depth_vals=o["depth_vals"][::8], weights=o["weights"][::8])


def fx_composite():
    params = synth.make_params(seed=0)
    m = build_model(params, beta=0.03)
    rng = np.random.default_rng(11)
    R, S = 16, 98
    z = np.sort(rng.uniform(0.5, 5.5, (R, S)), -1).astype(F32)
    z[:, 5] = z[:, 4]                         # duplicate sample -> zero-length interval
    sdf = (rng.normal(0.3, 0.5, (R, S)) - np.linspace(0, 1.0, S)[None]).astype(F32)
    rgb = rng.uniform(0, 1, (R, S, 3)).astype(F32)
    with torch.no_grad():
        w, dists = m.volume_rendering(T(z), T(sdf).reshape(-1, 1))
    save("composite", z=z, sdf=sdf, rgb=rgb, beta_param=F32(0.03), weights=w.numpy(), dists=dists.numpy())


def fx_forward_w1():
    """The trained-scale weight set (synth.make_trained_params): eval render with the full sampler at the set's own
    beta = 0.005, and a train-mode forward."""
    params = synth.WEIGHT_SETS["w1"]()
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1, skew=0.7)
    R = 12
    uv = synth.make_uv(R, seed=3, margin=0.1)
    inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None]}
    m = build_model(params, beta=float(params["density.beta"]))
    m.eval()
    out = m(inp, fast=-1)
    save("forward_w1_eval", K=K, pose=pose, uv=uv, beta_param=params["density.beta"], fast=-1,
         **{k: v.detach().numpy() for k, v in out.items()})
    m.train()
    draws = synth.make_train_rng(R, seed=6)
    with inject_rng(draws):
        out = m(inp, fast=1)
    save("forward_w1_train", K=K, pose=pose, uv=uv, beta_param=params["density.beta"], fast=1, rng_seed=6,
         **{k: v.detach().numpy() for k, v in out.items()})


def fx_forward():
    params = synth.make_params(seed=0)
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1, skew=0.7)
    R = 12
    uv = synth.make_uv(R, seed=3, margin=0.1)
    inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None]}
    for tag, beta, fast in (("eval_b0.1", 0.1, -1), ("eval_b0.01", 0.01, -1), ("eval_b0.01_f1", 0.01, 1)):
        m = build_model(params, beta=beta)
        m.eval()
        out = m(inp, fast=fast)
        save(f"forward_{tag}", K=K, pose=pose, uv=uv, beta_param=F32(beta), fast=fast,
             **{k: v.detach().numpy() for k, v in out.items()})
    m = build_model(params, beta=0.05)
    m.train()
    draws = synth.make_train_rng(R, seed=6)
    with inject_rng(draws):
        out = m(inp, fast=1)
    save("forward_train", K=K, pose=pose, uv=uv, beta_param=F32(0.05), fast=1, rng_seed=6,
         **{k: v.detach().numpy() for k, v in out.items()})


def build_bg_model(params, beta):
    from volsdf.model.network_bg import VolSDFNetworkBG
    m = VolSDFNetworkBG(ref_shim.bmvs_model_conf())
    sd = {k: T(v).clone() for k, v in params.items()}
    sd["density.beta"] = torch.tensor(beta, dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    return m


def fx_forward_bg():
    """VolSDFNetworkBG.forward (config 4: fg + inverted-sphere bg), eval (with near_pose) and train mode, with the
    intermediate background quantities captured."""
    params = dict(synth.make_params(seed=0))
    params.update(synth.make_bg_params(seed=0))
    K, pose = synth.make_camera(center=(0.1, 0.05, -2.5), tilt=0.1, skew=0.7)
    _, near_pose = synth.make_camera(center=(0.25, 0.0, -2.45), tilt=0.05)
    R = 12
    uv = synth.make_uv(R, seed=3, margin=0.1)
    for tag, beta, fast, training in (("eval_b0.1", 0.1, -1, False), ("eval_b0.01", 0.01, -1, False), ("train", 0.05, 1, True)):
        m = build_bg_model(params, beta)
        m.train(training)
        cap = {}
        o_d2p, o_bgvr, o_vr = m.depth2pts_outside, m.bg_volume_rendering, m.volume_rendering

        def d2p(o, d, depth):
            r = o_d2p(o, d, depth); cap["z_bg"] = depth.detach().numpy().copy()
            cap["bg_points"], cap["bg_depth"] = r[0].detach().numpy().copy(), r[1].detach().numpy().copy(); return r

        def bgvr(z, s):
            r = o_bgvr(z, s); cap["bg_sdf"] = s.detach().numpy().copy(); cap["bg_weights"] = r.detach().numpy().copy(); return r

        def vr(z, zmax, sdf):
            r = o_vr(z, zmax, sdf); cap["z_vals"], cap["z_max"] = z.detach().numpy().copy(), zmax.detach().numpy().copy()
            cap["bg_transmittance"] = r[1].detach().numpy().copy(); return r

        m.depth2pts_outside, m.bg_volume_rendering, m.volume_rendering = d2p, bgvr, vr
        inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None], "near_pose": T(near_pose)[None]}
        if training:
            draws = synth.make_train_rng(R, seed=6, n_final=98, bg=True)
            with inject_rng(draws):
                out = m(inp, fast=fast)
        else:
            out = m(inp, fast=fast)
        save(f"forward_bg_{tag}", K=K, pose=pose, near_pose=near_pose, uv=uv, beta_param=F32(beta), fast=fast, rng_seed=6,
             **{k: v.detach().numpy() for k, v in out.items()}, **cap)


def fx_cost_mapping():
    from volsdf.vsdf import VolOpt
    rng = np.random.default_rng(21)
    R, S = 24, 98
    views = synth.make_mvs_views(5)
    K, pose = views[0]["K"], views[0]["c2w"]
    import oracle_path  # noqa
    from svs_oracle import rays_from_uv
    uv = synth.make_uv(R, seed=8, margin=0.02)
    dirs, cam, _ = rays_from_uv(uv, pose, K)
    z = np.sort(rng.uniform(0.2, 5.5, (R, S)), -1).astype(F32)
    xyz = (cam[None, None] + z[:, :, None] * dirs[:, None, :]).astype(F32)
    ids = [25, 22, 28]
    for inv in (False, True):
        for vi in (0, 2):
            ds = SimpleNamespace(img_res=[576, 768], intrinsics_all={ids[j]: T(views[j]["K"]) for j in range(3)},
                                 pose_all={ids[j]: T(views[j]["c2w"]) for j in range(3)})
            me = SimpleNamespace(trains_i=ids, costs={j: T(views[j]["cost"])[None] for j in range(3)},
                                 z_mvs={j: T(views[j]["z_mvs"])[None] for j in range(3)}, train_dataset=ds,
                                 hparams=SimpleNamespace(inverse_depth=inv), stg=0)
            pj, pi, valid = VolOpt.cost_mapping(me, z_vals=T(z), ts=torch.tensor([ids[vi]]), xyz_raw=T(xyz))
            save(f"cost_mapping_inv{int(inv)}_v{vi}", xyz=xyz, view_index=vi, inverse_depth=inv, seed=5,
                 pj=pj.numpy(), pi=pi.numpy(), valid=valid.numpy())


def fx_loss():
    from volsdf.model.loss import VolSDFLoss
    rng = np.random.default_rng(31)
    R, S = 32, 98
    out = dict(rgb_values=rng.uniform(0, 1, (R, 3)).astype(F32), grad_theta=rng.normal(0, 1, (2 * R, 3)).astype(F32),
               weights=(rng.uniform(0, 1, (R, S)) ** 4).astype(F32), pi=(rng.uniform(0, 0.2, (R, S)) ** 2).astype(F32),
               pj=(rng.uniform(0, 0.3, (R, S)) ** 2).astype(F32), depth_values=rng.uniform(0.5, 4, (R, 1)).astype(F32))
    out["pi"][:6] = 0.0
    gt = dict(rgb=rng.uniform(0, 1, (1, R, 3)).astype(F32), rgb_smooth=rng.uniform(0, 1, (1, R, 3)).astype(F32))
    arr = dict(**out, rgb=gt["rgb"], rgb_smooth=gt["rgb_smooth"])
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                      anneal_rgb=200, gce=0.5, confi=1e-3)
    for it in (0, 100, 250):
        loss.iter_step = it
        res = loss({k: T(v) for k, v in out.items()}, {k: T(v) for k, v in gt.items()})
        for k, v in res.items():
            arr[f"it{it}_{k}"] = np.asarray(float(v), F32)
    save("loss", **arr)


def _load_costreg(model, stage, params):
    sd = {k: T(v) for k, v in params.items()}
    model.cost_regularization[stage].load_state_dict(sd, strict=True)


def fx_casmvs():
    from models.CasMVSNet import CascadeMVSNet, homo_warping
    import oracle_path  # noqa
    # ---- homo_warping alone: per-pixel hypotheses, off-image and behind-camera projections
    rng = np.random.default_rng(3)
    C, H, W, D = 8, 12, 16, 6
    src = rng.normal(0, 1, (C, H, W)).astype(F32)
    Kc = np.eye(4, dtype=F32); Kc[0, 0] = Kc[1, 1] = 20.0; Kc[0, 2], Kc[1, 2] = 8.0, 6.0
    def P(tx, ang):
        E = np.eye(4, dtype=F32)
        E[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], F32)
        E[0, 3] = tx
        out = E.copy(); out[:3, :4] = Kc[:3, :3] @ E[:3, :4]
        return out
    ref_p, src_p = P(0.0, 0.0), P(1.5, 0.35)
    dv = (rng.uniform(1.0, 12.0, (D, H, W))).astype(F32)
    dv[0, :3, :3] = 0.2                                 # close points -> projection behind/near the src camera
    warped = homo_warping(T(src)[None], T(src_p)[None], T(ref_p)[None], T(dv)[None])[0].numpy()
    save("homo_warp", src=src, src_proj=src_p, ref_proj=ref_p, depth_values=dv, warped=warped)

    # ---- DepthNet / CascadeMVSNet, 3 stages on a 64x96 image
    feats, proj, depth_values = synth.make_mvs_sample(7, img_hw=(64, 96))
    ndepths = [48, 32, 8]
    model = CascadeMVSNet(refine=False, ndepths=ndepths, depth_interals_ratio=[1.0, 0.5, 0.5], share_cr=False,
                          cr_base_chs=[8, 8, 8], grad_method="detach")
    model.eval()
    for st, cin in enumerate((32, 16, 8)):
        _load_costreg(model, st, synth.make_costreg_params(100 + st, cin))
    sample = dict(imgs=torch.zeros(1, 3, 3, 64, 96), depth_values=T(depth_values)[None],
                  proj_matrices={k: T(v)[None] for k, v in proj.items()})
    features = [{k: T(v)[None] for k, v in f.items()} for f in feats]
    outputs = None
    arr = {}
    for st in range(3):
        cap = {}
        cr = model.cost_regularization[st]
        orig = cr.forward
        def wrapped(x, _o=orig, _c=cap):
            _c["variance"] = x.detach().numpy().copy()
            y = _o(x)
            _c["reg"] = y.detach().numpy().copy()
            return y
        cr.forward = wrapped
        outputs, _ = model(st, sample, features=features, extra=None, outputs=outputs,
                           int_r=model.depth_interals_ratio[st])
        cr.forward = orig
        if st == 0:
            # what runner.py:240-243 does: the rendered depth replaces the MVS depth that seeds stage 2
            smooth = outputs["depth"] * 0.98 + 4.0
            outputs["stage1"]["depth"] = smooth
            outputs["depth"] = smooth
            arr["stage1_depth_override"] = smooth[0].numpy()
        o = outputs[f"stage{st + 1}"]
        var = cap["variance"][0]
        pick = np.random.default_rng(50 + st).choice(var.size, 4000, replace=False)
        arr[f"s{st}_variance_idx"] = pick                    # the full volume is MBs: pin 4000 random voxels
        arr[f"s{st}_variance_val"] = var.reshape(-1)[pick]
        arr[f"s{st}_reg"] = cap["reg"][0, 0]
        arr[f"s{st}_depth"] = o["depth"][0].numpy() if st else None
        arr[f"s{st}_conf"] = o["photometric_confidence"][0].numpy()
        arr[f"s{st}_prob"] = o["prob_volume"][0].numpy()       # what cost_mapping consumes, at every stage
        arr[f"s{st}_depth_values"] = o["depth_values"][0].numpy()
    # stage-1 depth before the override
    arr["s0_depth"] = (arr["stage1_depth_override"] - 4.0) / 0.98
    save("casmvs_3stage", seed=7, ndepths=np.asarray(ndepths), **{k: v for k, v in arr.items() if v is not None})
    # one D = 192 case on a tiny map (depth regression / confidence window at full depth count)
    rng = np.random.default_rng(12)
    reg = rng.normal(0, 2, (192, 9, 12)).astype(F32)
    dv = np.broadcast_to((425 + 2.65 * np.arange(192, dtype=F32)).reshape(-1, 1, 1), reg.shape).astype(F32).copy()
    import torch.nn.functional as Fn
    from models.CasMVSNet import depth_regression
    prob = Fn.softmax(T(reg)[None], dim=1)
    depth = depth_regression(prob, depth_values=T(dv)[None])
    s4 = 4 * Fn.avg_pool3d(Fn.pad(prob.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1, padding=0).squeeze(1)
    di = depth_regression(prob, depth_values=torch.arange(192, dtype=torch.float)).long().clamp(min=0, max=191)
    conf = torch.gather(s4, 1, di.unsqueeze(1)).squeeze(1)
    save("depthnet_tail_d192", reg=reg, depth_values=dv, prob=prob[0].numpy(), depth=depth[0].numpy(),
         conf=conf[0].numpy(), idx=di[0].numpy())


def param_digest(named, seed=0, k=48):
    """Small fingerprint of a parameter set: per tensor a fixed random subset of entries."""
    out = {}
    rng = np.random.default_rng(seed)
    for name, t in named:
        a = np.asarray(t).reshape(-1)
        idx = rng.choice(a.size, min(k, a.size), replace=False)
        out[name] = (idx, a[idx].copy())
    return out


def fx_train_step(R=16, n_steps=3, name="train_step", wset="w0"):
    """Three optimisation steps of the reference: VolSDFNetwork + cost_mapping + VolSDFLoss + clip + guard + Adam
    (volsdf/vsdf.py:196-219), 16 rays, injected random draws (seed = 100 + step).  `train_step_r32`: two steps with 32
    rays, the smallest batch that splits into two ray groups (a group needs rays * 98 samples to be a multiple of 32)."""
    from volsdf.vsdf import VolOpt
    from volsdf.model.loss import VolSDFLoss
    params = synth.WEIGHT_SETS[wset]()
    m = build_model(params, beta=float(params["density.beta"]))
    m.train()
    views = synth.make_mvs_views(5)
    K, pose = views[0]["K"], views[0]["c2w"]
    uv = synth.make_uv(R, seed=12, margin=0.2)
    rng = np.random.default_rng(3)
    gt = dict(rgb=rng.uniform(0, 1, (1, R, 3)).astype(F32), rgb_smooth=rng.uniform(0, 1, (1, R, 3)).astype(F32))
    ids = [25, 22, 28]
    ds = SimpleNamespace(img_res=[576, 768], intrinsics_all={ids[j]: T(views[j]["K"]) for j in range(3)},
                         pose_all={ids[j]: T(views[j]["c2w"]) for j in range(3)})
    me = SimpleNamespace(trains_i=ids, costs={j: T(views[j]["cost"])[None] for j in range(3)},
                         z_mvs={j: T(views[j]["z_mvs"])[None] for j in range(3)}, train_dataset=ds,
                         hparams=SimpleNamespace(inverse_depth=False), stg=0)
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                      anneal_rgb=200, gce=0.5, confi=1e-3)
    loss.iter_step = 0          # VolOpt calls loss.set_stg(0) before training (runner.py:220)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None]}
    arr = dict(uv=uv, rgb=gt["rgb"], rgb_smooth=gt["rgb_smooth"], mvs_seed=5)
    for step in range(n_steps):
        draws = synth.make_train_rng(R, seed=100 + step)
        with inject_rng(draws):
            out = m(inp, fast=1)
        out['pj'], out['pi'], _ = VolOpt.cost_mapping(me, z_vals=out['depth_vals'], ts=torch.tensor([ids[0]]),
                                                      xyz_raw=out['xyz'])
        lo = loss(out, {k: T(v) for k, v in gt.items()})
        opt.zero_grad()
        lo['loss'].backward()
        raw = param_digest([(n, p.grad.detach().numpy()) for n, p in m.named_parameters()], seed=step)
        norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        for k, v in lo.items():
            arr[f"s{step}_{k}"] = np.asarray(float(v), F32)
        arr[f"s{step}_grad_norm"] = np.asarray(float(norm), F32)
        for n, (idx, val) in raw.items():
            arr[f"s{step}_grad_idx/{n}"] = idx.astype(np.int32); arr[f"s{step}_grad/{n}"] = val
        for n, (idx, val) in param_digest([(n, p.detach().numpy()) for n, p in m.named_parameters()], seed=10 + step).items():
            arr[f"s{step}_param_idx/{n}"] = idx.astype(np.int32); arr[f"s{step}_param/{n}"] = val
    save(name, **arr)


def fx_train_step_bg(name="train_step_bg", iter_step=250, confi=1e-3, n_steps=2):
    """Two optimisation steps of the reference with the fg + background model (VolSDFNetworkBG, config 4): forward,
    cost_mapping, VolSDFLoss, clip, Adam (volsdf/vsdf.py:196-219); 32 rays (32 x 97 fg + 32 x 32 bg samples)."""
    from volsdf.vsdf import VolOpt
    from volsdf.model.loss import VolSDFLoss
    params = dict(synth.make_params(seed=0)); params.update(synth.make_bg_params(seed=0))
    m = build_bg_model(params, beta=0.1)
    m.train()
    R = 32
    views = synth.make_mvs_views(5)
    K, pose = views[0]["K"], views[0]["c2w"]
    uv = synth.make_uv(R, seed=12, margin=0.2)
    rng = np.random.default_rng(3)
    gt = dict(rgb=rng.uniform(0, 1, (1, R, 3)).astype(F32), rgb_smooth=rng.uniform(0, 1, (1, R, 3)).astype(F32))
    ids = [25, 22, 28]
    ds = SimpleNamespace(img_res=[576, 768], intrinsics_all={ids[j]: T(views[j]["K"]) for j in range(3)},
                         pose_all={ids[j]: T(views[j]["c2w"]) for j in range(3)})
    me = SimpleNamespace(trains_i=ids, costs={j: T(views[j]["cost"])[None] for j in range(3)},
                         z_mvs={j: T(views[j]["z_mvs"])[None] for j in range(3)}, train_dataset=ds,
                         hparams=SimpleNamespace(inverse_depth=False), stg=0)
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                      anneal_rgb=200, gce=0.5, confi=confi)
    # 250: past the rgb annealing: the plain L1 colour term is active and reaches the background.  train_step_bg_sparse:
    # iteration 50 with confi = 1e3 -- every ray counts as unsupported by the MVS prior, so the sparsity term
    # 1 / (depth_values_all + 1e-3) is live on all rays and its gradient reaches the background through the fg + bg depth
    loss.iter_step = iter_step
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    inp = {"intrinsics": T(K)[None], "uv": T(uv)[None], "pose": T(pose)[None]}
    arr = dict(uv=uv, rgb=gt["rgb"], rgb_smooth=gt["rgb_smooth"], mvs_seed=5, loss_iter_step=iter_step, confi=np.asarray(confi, F32))
    for step in range(n_steps):
        draws = synth.make_train_rng(R, seed=100 + step, bg=True)
        with inject_rng(draws):
            out = m(inp, fast=1)
        out['pj'], out['pi'], _ = VolOpt.cost_mapping(me, z_vals=out['depth_vals'], ts=torch.tensor([ids[0]]),
                                                      xyz_raw=out['xyz'])
        lo = loss(out, {k: T(v) for k, v in gt.items()})
        opt.zero_grad()
        lo['loss'].backward()
        raw = param_digest([(n, p.grad.detach().numpy()) for n, p in m.named_parameters()], seed=step)
        norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        for k, v in lo.items():
            arr[f"s{step}_{k}"] = np.asarray(float(v), F32)
        arr[f"s{step}_grad_norm"] = np.asarray(float(norm), F32)
        for n, (idx, val) in raw.items():
            arr[f"s{step}_grad_idx/{n}"] = idx.astype(np.int32); arr[f"s{step}_grad/{n}"] = val
        for n, (idx, val) in param_digest([(n, p.detach().numpy()) for n, p in m.named_parameters()], seed=10 + step).items():
            arr[f"s{step}_param_idx/{n}"] = idx.astype(np.int32); arr[f"s{step}_param/{n}"] = val
    save(name, **arr)


def fx_fusion():
    """helpers/utils.py check_geometric_consistency of the reference on a synthetic 3-view scene.  cv2 is not
    installed: the reference's single cv2 call (cv2.remap, INTER_LINEAR) is bound to the oracle's restatement
    `fusion_oracle.remap_linear`, so this fixture pins the reference's geometry, masks and type promotions GIVEN that
    sampler; the sampler itself stays unpinned (oracle/fusion_oracle.py header)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import fusion_oracle
    import cv2
    cv2.INTER_LINEAR = 1
    cv2.remap = lambda src, mx, my, interpolation: fusion_oracle.remap_linear(src, mx, my)
    if not hasattr(np, "bool"):
        np.bool = np.bool_
    from helpers.utils import check_geometric_consistency
    arr = {"seed": np.asarray(21), "hw": np.asarray((40, 56))}
    views = synth.make_fusion_views(21, hw=(40, 56), n_views=3)
    for ref, src in ((0, 1), (0, 2), (1, 2), (2, 0)):
        for fd, fr in ((1, 0.01), (0.5, 0.003)):
            m, d, x, y = check_geometric_consistency(views[ref]["depth"], views[ref]["K"], views[ref]["E"],
                                                     views[src]["depth"], views[src]["K"], views[src]["E"], fd, fr)
            tag = f"r{ref}s{src}_{fd}_{fr}"
            arr[tag + "/mask"], arr[tag + "/depth"], arr[tag + "/x"], arr[tag + "/y"] = m, d, x, y
    save("fusion_geo", **arr)


def fx_filter_depth():
    """`filter_depth` of the reference (runner.py:301-404) run END TO END on a synthetic scan folder.  runner.py cannot be
    imported (hydra's get_config() runs at import), so the function's own source is taken from the file with `ast`, compiled
    and executed -- unmodified -- in a namespace that binds the names it uses: the reference's own helpers
    (read_camera_parameters, read_img, save_mask, check_geometric_consistency, read_pfm, get_trains_ids), numpy / os / Path,
    `args` as a namespace, a silent logger, and two stand-ins for packages the image lacks: cv2.remap -> the oracle's
    restatement of OpenCV's fixed-point INTER_LINEAR remap (as in fx_fusion) and plyfile's PlyData / PlyElement -> a stub that
    CAPTURES what the function hands to PlyElement.describe (the structured vertex array: field order, dtypes, values).
    Inputs (cams/*.txt written by the reference's write_cam, images/*.jpg, depth_est / confidence PFMs written by the
    reference's save_pfm) and outputs (vertex array, the three masks per view) are stored; the test replays the files."""
    import ast
    import io
    import shutil
    import tempfile
    from pathlib import Path
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import fusion_oracle
    import cv2
    cv2.INTER_LINEAR = 1
    cv2.remap = lambda src, mx, my, interpolation: fusion_oracle.remap_linear(src, mx, my)
    if not hasattr(np, "bool"):
        np.bool = np.bool_
    import helpers.utils as hu
    from datasets.data_io import read_pfm, save_pfm
    from volsdf.datasets.scene_dataset import get_trains_ids

    src = open(os.path.join(ref_shim.REFERENCE_ROOT, "runner.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "filter_depth")
    captured = {}

    class PlyElement:
        @staticmethod
        def describe(arr, name):
            captured["vertex"], captured["element"] = arr.copy(), name
            return ("element", name)

    class PlyData:
        def __init__(self, els):
            captured["n_elements"] = len(els)

        def write(self, filename):
            captured["plyfile"] = filename

    conf = dict(conf=0.3, filter_dist=1, filter_diff=0.01, thres_view=1)
    args = SimpleNamespace(vol=SimpleNamespace(dataset=SimpleNamespace(data_dir="DTU")), num_view=3, eval_mask=False,
                           data_dir_root="unused", **conf)
    ns = dict(np=np, os=os, Path=Path, args=args, logger=ref_shim._NoLog(), get_trains_ids=get_trains_ids,
              read_camera_parameters=hu.read_camera_parameters, read_img=hu.read_img, read_pfm=read_pfm,
              check_geometric_consistency=hu.check_geometric_consistency, save_mask=hu.save_mask, cv2=cv2,
              PlyData=PlyData, PlyElement=PlyElement)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "runner.py:filter_depth", "exec"), ns)

    ids = get_trains_ids("DTU", "scan24", 3)                               # [25, 22, 28]
    views = synth.make_fusion_views(33, hw=(40, 56), n_views=3)
    root = tempfile.mkdtemp(prefix="svs_fd_")
    try:
        scan, out = os.path.join(root, "scan24"), os.path.join(root, "out", "scan24")
        for d in ("cams", "images"):
            os.makedirs(os.path.join(scan, d))
        for d in ("depth_est", "confidence"):
            os.makedirs(os.path.join(out, d))
        arr = {"view_ids": np.asarray(ids), "hw": np.asarray((40, 56)), **{k: np.asarray(v) for k, v in conf.items()}}
        for vid, v in zip(ids, views.values()):
            K4 = np.eye(4, dtype=F32); K4[:3, :3] = v["K"]
            hu.write_cam(os.path.join(scan, "cams/{:0>8}_cam.txt".format(vid)), [v["E"], K4], cam_near_far=(1.0, 0.01, 192, 3.0))
            Image.fromarray((v["img"] * 255).astype(np.uint8)).save(os.path.join(scan, "images/{:0>8}.jpg".format(vid)), quality=95)
            save_pfm(os.path.join(out, "depth_est/{:0>8}.pfm".format(vid)), v["depth"])
            save_pfm(os.path.join(out, "confidence/{:0>8}.pfm".format(vid)), v["confidence"])
            arr[f"cam_{vid}"] = np.frombuffer(open(os.path.join(scan, "cams/{:0>8}_cam.txt".format(vid)), "rb").read(), np.uint8)
            arr[f"jpg_{vid}"] = np.frombuffer(open(os.path.join(scan, "images/{:0>8}.jpg".format(vid)), "rb").read(), np.uint8)
            arr[f"depth_{vid}"], arr[f"confidence_{vid}"] = v["depth"], v["confidence"]
            arr[f"img_{vid}"] = hu.read_img(os.path.join(scan, "images/{:0>8}.jpg".format(vid)))      # what the function sees
            K, E = hu.read_camera_parameters(os.path.join(scan, "cams/{:0>8}_cam.txt".format(vid)))
            arr[f"K_{vid}"], arr[f"E_{vid}"] = K, E
        ns["filter_depth"](scan, out, os.path.join(root, "scan24.ply"))
        vtx = captured["vertex"]
        assert captured["element"] == "vertex" and captured["n_elements"] == 1
        arr["vertex_descr"] = np.asarray(repr(vtx.dtype.descr))
        arr["vertex_xyz"] = np.stack([vtx[k] for k in "xyz"], 1)
        arr["vertex_rgb"] = np.stack([vtx[k] for k in ("red", "green", "blue")], 1)
        for vid in ids:
            for tag in ("photo", "geo", "final"):
                arr[f"mask_{vid}_{tag}"] = np.array(Image.open(os.path.join(out, "mask/{:0>8}_{}.png".format(vid, tag)))) > 0
        print(f"   filter_depth: {len(vtx)} vertices, dtype {vtx.dtype.descr}")
        save("filter_depth", **arr)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def fx_pfm():
    """datasets/data_io.py save_pfm / read_pfm of the reference: the bytes it writes for a grey and a colour image and
    what it reads back."""
    import tempfile
    from datasets.data_io import read_pfm, save_pfm
    rng = np.random.default_rng(4)
    arr = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, img, scale in (("grey", rng.normal(0, 100, (5, 7)).astype(F32), 1), ("colour", rng.uniform(0, 1, (4, 6, 3)).astype(F32), 2.5),
                                ("hw1", rng.normal(0, 1, (3, 2, 1)).astype(F32), 1)):
            fn = os.path.join(td, tag + ".pfm")
            save_pfm(fn, img, scale)
            arr[tag + "/image"] = img
            arr[tag + "/scale"] = np.asarray(float(scale))
            arr[tag + "/bytes"] = np.frombuffer(open(fn, "rb").read(), np.uint8)
            if tag != "hw1":                      # the reference cannot read back its own (H,W,1) files as (H,W,1)
                back, sc = read_pfm(fn)
                arr[tag + "/read"], arr[tag + "/read_scale"] = np.ascontiguousarray(back), np.asarray(float(sc))
    save("pfm_codec", **arr)


def fx_chamfer():
    """The reference's evaluation script itself (evals/eval_dtu.py, run as __main__ through runpy) on a synthetic
    scan laid out in the DTU directory structure.  Two substitutions, neither in the algorithm: open3d (not
    installed) is a stub whose read_point_cloud parses the binary PLYs written below with numpy, and the script's
    unseeded np.random.default_rng() is seeded so that the shuffled order can be stored."""
    import runpy
    import tempfile
    import types
    from scipy.io import savemat
    scan = 24
    sc = synth.make_dtu_scan(31)

    def write_ply(fn, pts):
        with open(fn, "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                     "property double z\nend_header\n" % len(pts)).encode())
            np.ascontiguousarray(pts, "<f8").tofile(f)

    def read_point_cloud(fn):
        with open(fn, "rb") as f:
            while f.readline().strip() != b"end_header":
                pass
            pts = np.fromfile(f, "<f8").reshape(-1, 3)
        return types.SimpleNamespace(points=pts)

    o3d = types.ModuleType("open3d")
    o3d.io = types.SimpleNamespace(read_point_cloud=read_point_cloud)
    sys.modules["open3d"] = o3d
    real_rng = np.random.default_rng
    with tempfile.TemporaryDirectory() as td:
        ds = os.path.join(td, "root", "DTU", "DTU_MVS_Data")
        os.makedirs(os.path.join(ds, "ObsMask")); os.makedirs(os.path.join(ds, "Points", "stl")); os.makedirs(os.path.join(td, "pred"))
        savemat(os.path.join(ds, "ObsMask", f"ObsMask{scan}_10.mat"), dict(ObsMask=sc["ObsMask"], BB=sc["BB"], Res=sc["Res"]))
        savemat(os.path.join(ds, "ObsMask", f"Plane{scan}.mat"), dict(P=sc["P"]))
        write_ply(os.path.join(ds, "Points", "stl", f"stl{scan:03}_total.ply"), sc["stl"])
        write_ply(os.path.join(td, "pred", f"mvsnet{scan:03}_l3.ply"), sc["data_pcd"])
        argv = sys.argv
        sys.argv = ["eval_dtu.py", "--data_dir_root", os.path.join(td, "root"), "--datadir", os.path.join(td, "pred"), "--scan", str(scan)]
        np.random.default_rng = lambda *a: real_rng(*a) if a else real_rng(77)
        try:
            g = runpy.run_path(os.path.join(ref_shim.REFERENCE_ROOT, "evals", "eval_dtu.py"), run_name="__main__")
        finally:
            np.random.default_rng = real_rng
            sys.argv = argv
    save("chamfer_ref", seed=np.asarray(31), shuffle_seed=np.asarray(77), data_pcd_shuffled_head=g["data_pcd"][:64],
         keep=np.packbits(g["mask"]), n_down=np.asarray(len(g["data_down"])), n_in=np.asarray(len(g["data_in"])),
         n_in_obs=np.asarray(len(g["data_in_obs"])), n_stl_above=np.asarray(len(g["stl_above"])),
         dist_d2s=g["dist_d2s"][:, 0], dist_s2d=g["dist_s2d"][:, 0], mean_d2s=np.asarray(g["mean_d2s"]),
         mean_s2d=np.asarray(g["mean_s2d"]), over_all=np.asarray(g["over_all"]))
    print("   acc %.4f comp %.4f, kept %d of %d, in_obs %d" % (g["mean_d2s"], g["mean_s2d"], len(g["data_down"]), len(g["data_pcd"]),
                                                            len(g["data_in_obs"])))


def fx_chamfer_mesh():
    """The reference's evaluation script in --mode mesh (evals/eval_dtu.py:62-90 and on), run as __main__ through runpy
    on the synthetic scan of fx_chamfer with a predicted triangle mesh.  Substitutions, none in the algorithm: the open3d
    stub's read_triangle_mesh / read_point_cloud parse the PLYs written below with numpy; multiprocessing.Pool is a
    serial map (the script's worker function lives in the runpy module, which a forked worker cannot import); the
    unseeded np.random.default_rng() is seeded."""
    import multiprocessing
    import runpy
    import tempfile
    import types
    from scipy.io import savemat
    scan = 24
    sc = synth.make_dtu_scan(31)
    vertices, triangles = synth.make_dtu_mesh(53)

    def write_ply(fn, pts):
        with open(fn, "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                     "property double z\nend_header\n" % len(pts)).encode())
            np.ascontiguousarray(pts, "<f8").tofile(f)

    def read_point_cloud(fn):
        with open(fn, "rb") as f:
            while f.readline().strip() != b"end_header":
                pass
            pts = np.fromfile(f, "<f8").reshape(-1, 3)
        return types.SimpleNamespace(points=pts)

    class SerialPool:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def map(self, fn, it, chunksize=None):
            return [fn(x) for x in it]

    o3d = types.ModuleType("open3d")
    o3d.io = types.SimpleNamespace(read_point_cloud=read_point_cloud,
                                   read_triangle_mesh=lambda fn: types.SimpleNamespace(vertices=vertices.copy(), triangles=triangles.copy()))
    sys.modules["open3d"] = o3d
    real_rng, real_pool = np.random.default_rng, multiprocessing.Pool
    with tempfile.TemporaryDirectory() as td:
        ds = os.path.join(td, "root", "DTU", "DTU_MVS_Data")
        os.makedirs(os.path.join(ds, "ObsMask")); os.makedirs(os.path.join(ds, "Points", "stl")); os.makedirs(os.path.join(td, "pred"))
        savemat(os.path.join(ds, "ObsMask", f"ObsMask{scan}_10.mat"), dict(ObsMask=sc["ObsMask"], BB=sc["BB"], Res=sc["Res"]))
        savemat(os.path.join(ds, "ObsMask", f"Plane{scan}.mat"), dict(P=sc["P"]))
        write_ply(os.path.join(ds, "Points", "stl", f"stl{scan:03}_total.ply"), sc["stl"])
        argv = sys.argv
        sys.argv = ["eval_dtu.py", "--data_dir_root", os.path.join(td, "root"), "--datadir", os.path.join(td, "pred"), "--scan", str(scan),
                    "--mode", "mesh"]
        np.random.default_rng = lambda *a: real_rng(*a) if a else real_rng(78)
        multiprocessing.Pool = SerialPool
        try:
            g = runpy.run_path(os.path.join(ref_shim.REFERENCE_ROOT, "evals", "eval_dtu.py"), run_name="__main__")
        finally:
            np.random.default_rng = real_rng
            multiprocessing.Pool = real_pool
            sys.argv = argv
    new_pts = g["new_pts"]
    save("chamfer_mesh_ref", scan_seed=np.asarray(31), mesh_seed=np.asarray(53), shuffle_seed=np.asarray(78),
         n_new_pts=np.asarray(len(new_pts)), per_tri=np.asarray([len(q) for q in g["mp_pool"].map(g["sample_single_tri"], (
             (g["n1"][i, 0], g["n2"][i, 0], g["v1"][i:i + 1], g["v2"][i:i + 1], g["tri_vert"][i:i + 1, 0]) for i in range(len(g["n1"]))))], np.int32),
         new_pts_every_61=new_pts[::61], new_pts_sum=new_pts.sum(0), data_pcd_shuffled_head=g["data_pcd"][:64],
         keep=np.packbits(g["mask"]), n_down=np.asarray(len(g["data_down"])), n_in=np.asarray(len(g["data_in"])),
         n_in_obs=np.asarray(len(g["data_in_obs"])), n_stl_above=np.asarray(len(g["stl_above"])),
         dist_d2s=g["dist_d2s"][:, 0], dist_s2d=g["dist_s2d"][:, 0], mean_d2s=np.asarray(g["mean_d2s"]),
         mean_s2d=np.asarray(g["mean_s2d"]), over_all=np.asarray(g["over_all"]))
    print("   mesh: %d triangles (%d with area), %d sampled points; acc %.4f comp %.4f, kept %d of %d" % (
        len(triangles), len(g["n1"]), len(new_pts), g["mean_d2s"], g["mean_s2d"], len(g["data_down"]), len(g["data_pcd"])))


def fx_featurenet():
    """FeatureNet (arch_mode 'fpn', base 8) of the reference on a 3 x 36 x 52 image (sizes that are multiples of 4 but
    not of 8, so that the strided 5x5 layers see odd intermediate sizes): the three pyramid outputs."""
    from models.CasMVSNet import FeatureNet
    params = synth.make_featurenet_params(41)
    net = FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode="fpn")
    net.load_state_dict({k: T(v) for k, v in params.items()}, strict=True)
    net.eval()
    img = np.random.default_rng(42).uniform(0, 1, (1, 3, 36, 52)).astype(F32)
    with torch.no_grad():
        out = net(T(img))
    save("featurenet", seed=np.asarray(41), img=img, **{k: v[0].numpy() for k, v in out.items()})


ALL = dict(fusion=fx_fusion, filter_depth=fx_filter_depth, pfm=fx_pfm, chamfer=fx_chamfer, chamfer_mesh=fx_chamfer_mesh, featurenet=fx_featurenet, rays=fx_rays, sdf_mlp=fx_sdf_mlp, rgb_mlp=fx_rgb_mlp, density=fx_density, sampler=fx_sampler,
           composite=fx_composite, forward=fx_forward, forward_bg=fx_forward_bg, cost_mapping=fx_cost_mapping, loss=fx_loss, casmvs=fx_casmvs, train_step=fx_train_step, train_step_r32=lambda: fx_train_step(32, 2, "train_step_r32"),
           sdf_mlp_w1=lambda: fx_sdf_mlp("w1"), forward_w1=fx_forward_w1,
           train_step_w1=lambda: fx_train_step(16, 2, "train_step_w1", "w1"),
           train_step_bg=fx_train_step_bg,
           train_step_bg_sparse=lambda: fx_train_step_bg("train_step_bg_sparse", 50, 1e3, 1),
           sampler_r256=fx_sampler_r256, sampler_r256_more=fx_sampler_r256_more, forward_r256_more=fx_forward_r256_more, forward_r1024_train=fx_forward_r1024_train, sampler_hostexp=fx_sampler_hostexp, primitives=fx_primitives)

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    for n in names:
        print(f"[{n}]")
        ALL[n]()
