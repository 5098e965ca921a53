"""Pins oracle/svs_oracle.py against golden vectors produced by the imported reference
(tests/golden/make_fixtures.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

import svs_oracle as orc
import synth

F32 = np.float32


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def test_exp_restatements_within_one_ulp_of_libm():
    x = np.concatenate([np.linspace(-104, 89, 20001), [-1e-8, 0.0, 1e-8, -200.0, 100.5, np.inf, -np.inf]]).astype(F32)
    y = orc.ref_exp(x)
    ref = np.exp(x.astype(np.float64))
    with np.errstate(over="ignore"):
        ref32 = ref.astype(F32)
    fin = np.isfinite(ref32) & (ref32 > 1e-37)
    ulp = np.abs(y[fin].astype(np.float64) - ref[fin]) / np.spacing(ref32[fin]).astype(np.float64)
    assert ulp.max() <= 1.0                     # Sleef's u10 bound
    assert y[-1] == 0.0 and np.isinf(y[-2]) and y[-4] == 0.0 and np.isinf(y[-3])
    xm = np.concatenate([-np.logspace(-12, 2, 4000), np.logspace(-12, 1, 500)]).astype(F32)
    ym = orc.ref_expm1(xm)
    refm = np.expm1(xm.astype(np.float64))
    assert np.max(np.abs(ym - refm) / np.abs(refm)) < 1.2e-7


def test_primitives_bit_equal_torch_routines(golden_dir):
    """The three primitives the sampler's bit-exactness hangs on, against what torch's own routines returned in the
    build container (fixture `primitives`, make_fixtures.py::fx_primitives): Sleef_expf8_u10 / Sleef_expm1f8_u10 as
    exported by libtorch_cpu.so (torch.expm1 IS the latter; torch.exp is pinned to the former, see svs_oracle's
    docstring) and torch.sum(dim=-1) for rows of every length 1..160, the sampler's lengths and rows long enough to
    fold two cascade levels."""
    g = load(golden_dir, "primitives")
    x = g["x"]
    for mine, ref in ((orc.sleef_expf(x), g["expf"]), (orc.sleef_expm1f(x), g["expm1f"])):
        both_nan = np.isnan(mine) & np.isnan(ref)
        assert np.array_equal(mine.view(np.uint32)[~both_nan], ref.view(np.uint32)[~both_nan])
    off = o = 0
    for m in g["sum_lens"]:
        nr = 8 if m <= 160 else (2 if m <= 641 else 1)
        rows = g["sum_rows"][off:off + nr * m].reshape(nr, m)
        assert np.array_equal(orc.aten_sum(rows)[:, 0], g["sum_out"][o:o + nr]), m
        off += nr * m
        o += nr


def test_primitives_against_this_hosts_torch():
    """Same, live, where torch is importable: torch.expm1 and torch.sum are the restated routines on every x86-64 host
    (expm1: Sleef; sum: the AVX2 kernel also serves AVX-512 hosts).  torch.exp / torch.sqrt are NOT compared -- in an
    MKL build they are host-dependent closed-source routines (test_sampler_hostexp covers them)."""
    import torch
    rng = np.random.default_rng(3)
    x = np.concatenate([-rng.random(20000) * 30, rng.random(5000) * 20, rng.standard_normal(5000) * 1e-4]).astype(F32)
    assert np.array_equal(orc.sleef_expm1f(x).view(np.uint32), torch.expm1(torch.from_numpy(x)).numpy().view(np.uint32))
    for m in list(range(1, 130)) + [255, 383, 511, 639, 640, 5000]:
        r = (rng.random((16, m)) * np.exp(rng.random((16, 1)) * 12 - 8)).astype(F32)
        assert np.array_equal(orc.aten_sum(r)[:, 0], torch.sum(torch.from_numpy(r), -1).numpy()), m


def test_canon_cumsum_is_f64_cumsum():
    rng = np.random.default_rng(0)
    for m in (1, 63, 64, 127, 128, 255, 639, 640):
        x = rng.uniform(0, 1, (5, m)).astype(F32) * rng.choice([1e-6, 1.0, 1e4], (5, m)).astype(F32)
        a = orc.canon_cumsum(x)
        b = np.cumsum(x.astype(np.float64), -1).astype(F32)
        assert (a != b).mean() < 1e-3            # equal up to float64 re-association
        np.testing.assert_allclose(a, b, rtol=1e-7)


def test_linspace_matches_torch():
    import torch
    for n in (32, 64, 128):
        assert np.array_equal(orc.linspace01(n), torch.linspace(0., 1., n).numpy())
    for n, k in ((128, 32), (256, 32), (640, 32), (384, 32)):
        assert np.array_equal(orc.extras_index_eval(n, k), torch.linspace(0, n - 1, k).long().numpy())


def test_rays(golden_dir):
    g = load(golden_dir, "rays")
    for t in "ab":
        dirs, cam, ds = orc.rays_from_uv(g[t + "_uv"], g[t + "_pose"], g[t + "_K"])
        # bit for bit since round 4 (F.normalize's norm is sqrt(fma(z, z, fma(y, y, x*x))) in torch's reduce kernel: norm3)
        assert np.array_equal(dirs, g[t + "_dirs"]) and np.array_equal(cam, g[t + "_cam"])
        assert np.array_equal(ds, g[t + "_depth_scale"])


def test_density(golden_dir):
    g = load(golden_dir, "density")
    for i, b in enumerate((0.1, 0.01, 0.001)):
        beta = orc.get_beta(b)
        assert beta == g[f"beta_{i}"]
        # bit for bit since round 4: expm1 is the restated Sleef routine (svs_oracle.sleef_expm1f)
        assert np.array_equal(orc.laplace_density(g["sdf"], beta), g[f"sigma_{i}"])
    assert np.array_equal(orc.laplace_density(g["sdf_ray"], g["beta_ray"]), g["sigma_ray"])


@pytest.mark.parametrize("name,wset", [("sdf_mlp", "w0"), ("sdf_mlp_w1", "w1")])
def test_sdf_mlp(golden_dir, name, wset):
    """w1: the trained-scale weight set (synth.make_trained_params): activations up to ~15, gradients up to ~20 -- the
    float32 rounding of the reference itself scales with them (rtol)."""
    g = load(golden_dir, name)
    layers = orc.effective_weights(synth.WEIGHT_SETS[wset](), "implicit_network", 9)
    out = orc.sdf_mlp_forward(layers, g["x"])
    np.testing.assert_allclose(out, g["out"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(orc.sdf_vals(layers, g["x"]), g["sdf_vals"], atol=2e-5, rtol=2e-5)
    sdf, feat, grad = orc.sdf_outputs(layers, g["x"])
    np.testing.assert_allclose(sdf, g["sdf"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(feat, g["feat"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(grad, g["grad"], atol=1e-4, rtol=1e-4)
    graw = orc.sdf_outputs(layers, g["x"], clamp=False)[2]
    np.testing.assert_allclose(graw, g["grad_raw"], atol=1e-4, rtol=1e-4)
    assert (np.abs(g["sdf"] - g["out"][:, :1]) > 1e-3).any(), "fixture must exercise the sphere clamp"


def test_rgb_mlp(golden_dir):
    g = load(golden_dir, "rgb_mlp")
    layers = orc.effective_weights(synth.make_params(int(g["seed"])), "rendering_network", 5)
    rgb = orc.rgb_mlp_forward(layers, g["points"], g["normals"], g["dirs"], g["feat"])
    np.testing.assert_allclose(rgb, g["rgb"], atol=2e-6)


def test_composite(golden_dir):
    g = load(golden_dir, "composite")
    w, _, dists = orc.ray_weights(g["z"], g["sdf"], orc.get_beta(g["beta_param"]))
    assert np.array_equal(dists, g["dists"])
    assert np.array_equal(w, g["weights"])          # bit for bit (exp = the pinned Sleef routine, cumsum in float64)


SAMPLER_FX = sorted(os.path.basename(p)[:-4] for p in
                    glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sampler_eval_*.npz")))


def _replay_rounds(g, training=False, rng=None):
    """Yield (round index, oracle record, reference arrays) with every round restarted from the
    REFERENCE's own state (bins, merged sdf, carried beta), so one near-tie cannot cascade."""
    nr = int(g["n_rounds"]) if "n_rounds" in g else 1
    fast = int(g["fast"]) if "fast" in g else 1
    max_iters = fast if fast >= 0 else 5
    R = g["dirs"].shape[0]
    beta0 = orc.get_beta(g["beta_param"])
    z = orc.uniform_z(F32(1e-4), F32(6.0), 128, rng["jitter"] if training else None)
    if z.shape[0] == 1:
        z = np.repeat(z, R, 0)
    d = z[:, 1:] - z[:, :-1]
    beta_in = orc.ref_sqrt((F32(g["inv_4log"]) * orc.ref_sum(d * d)[:, 0]).astype(F32))
    sdf = None
    for i in range(nr):
        new_sdf = g[f"sdf_{i}"].reshape(R, -1)
        sdf = new_sdf if i == 0 else np.take_along_axis(np.concatenate([sdf, new_sdf], -1), g[f"samples_idx_{i-1}"], 1)
        rec = orc.sampler_round(z, sdf, beta_in, beta0, upsample_allowed=(i + 1 < max_iters), training=training,
                                u_final=None if rng is None else rng["u"])
        yield i, rec
        beta_in = g[f"beta_{i}"]
        if f"zmerged_{i}" in g:
            z = g[f"zmerged_{i}"]


def _check_inds(rec, g, i):
    """searchsorted indices and the whole cdf equal the reference's bit for bit (the reference's primitives are restated
    exactly: svs_oracle's docstring)."""
    assert np.array_equal(rec["cdf"], g[f"cdf_{i}"]), i
    assert np.array_equal(rec["inds"], g[f"inds_{i}"]), i
    return 0


@pytest.mark.parametrize("name", SAMPLER_FX)
def test_sampler_eval_rounds(golden_dir, name):
    """Every round of the reference's eval sampler replayed from the reference's state: beta, cdf, indices and the merged
    bins are the reference's bit for bit."""
    g = load(golden_dir, name)
    for i, rec in _replay_rounds(g):
        assert np.array_equal(rec["beta"], g[f"beta_{i}"]), i
        _check_inds(rec, g, i)
        if rec["upsample"]:
            assert f"samples_idx_{i}" in g
            ref_idx = g[f"samples_idx_{i}"]
            assert np.array_equal(rec["z_next"], g[f"zmerged_{i}"]), i
            # torch.sort is not stable: the two gather orders may differ only inside runs of equal keys
            cat = np.concatenate([rec["z"], rec["samples"]], -1)
            assert np.array_equal(np.sort(ref_idx, -1), np.sort(rec["samples_idx"], -1))
            assert np.array_equal(np.take_along_axis(cat, ref_idx, -1), g[f"zmerged_{i}"])


R256 = ["sampler256_b0.1", "sampler256_b0.01", "sampler256_b0.001"]


def load256(golden_dir, name):
    """the compact R = 256 sampler fixtures (make_fixtures.py::fx_sampler_r256): indices come as uint16"""
    g = load(golden_dir, name)
    for k in list(g):
        if k.startswith(("inds_", "samples_idx_")):
            g[k] = g[k].astype(np.int64)
    return g


def _assert_rounds_exact(g):
    n_idx = 0
    for i, rec in _replay_rounds(g):
        ref = g[f"inds_{i}"]
        n = rec["cdf"].shape[1]
        assert np.array_equal(rec["beta"], g[f"beta_{i}"]), i
        assert np.array_equal(np.take_along_axis(rec["cdf"], np.maximum(ref - 1, 0), 1), g[f"cdf_lo_{i}"]), i
        assert np.array_equal(np.take_along_axis(rec["cdf"], np.minimum(ref, n - 1), 1), g[f"cdf_hi_{i}"]), i
        assert np.array_equal(rec["inds"], ref), (i, int((rec["inds"] != ref).sum()))
        if rec["upsample"]:
            assert np.array_equal(rec["z_next"], g[f"zmerged_{i}"]), i
        n_idx += ref.size
    return n_idx


@pytest.mark.parametrize("name", R256)
def test_sampler_r256_indices(golden_dir, name):
    """Index-level identity with the reference sampler on 256 rays, all rounds (5 at beta <= 0.01), every round replayed
    from the reference's own state: ZERO differing searchsorted indices (49 152 / 147 456 / 147 456), the bracketing cdf
    entries, beta and the merged bins bit-equal.  (Round 3 accepted 0.14 % near-tie flips; they were the float64 row sum
    and the correctly-rounded exp / expm1 of that round's contract, not properties of the reference.)"""
    g = load256(golden_dir, name)
    n_idx = _assert_rounds_exact(g)
    print(f"{name}: 0 differing indices of {n_idx} ({int(g['n_rounds'])} rounds)")


@pytest.mark.parametrize("name", ["forward256_train_b0.05", "forward256_bg_eval_b0.01", "forward256_bg_train_b0.05"])
def test_forward_r256_train_and_background(golden_dir, name):
    """Whole forwards at 256 rays beyond the eval runs: the DTU model in TRAIN mode and the fg + background model in eval and
    train mode -- colours to 5e-6 on every ray (north star: 1e-4), depths to their scale."""
    g = load(golden_dir, name)
    bg, training = "_bg_" in name, "train" in name
    params = dict(synth.make_params(0))
    rng = synth.make_train_rng(256, seed=int(g["rng_seed"]), bg=bg) if training else None
    if bg:
        params.update(synth.make_bg_params(0))
        out = orc.render_forward_bg(params, g["uv"], g["pose"], g["K"], beta_param=g["beta_param"], fast=int(g["fast"]),
                                    training=training, rng=rng, near_pose=g["near_pose"])
        np.testing.assert_allclose(out["depth_values_all"], g["depth_values_all"], rtol=1e-5)
    else:
        out = orc.render_forward(params, g["uv"], g["pose"], g["K"], beta_param=g["beta_param"], fast=int(g["fast"]),
                                 training=training, rng=rng)
    np.testing.assert_allclose(out["rgb_values"], g["rgb_values"], atol=5e-6)
    np.testing.assert_allclose(out["depth_values"], g["depth_values"], atol=1e-4)
    if training:
        np.testing.assert_allclose(out["grad_theta"], g["grad_theta"], atol=2e-3)
        np.testing.assert_allclose(out["grad_theta"][:256], g["grad_theta"][:256], atol=1e-4)
    else:
        np.testing.assert_allclose(out["normal_map"], g["normal_map"], atol=1e-5)


def test_forward_r1024_train(golden_dir):
    """The bench batch size: the reference's TRAIN-mode forward of the DTU model on 1024 rays (forward1024_train_b0.05) --
    colours to the north star's 1e-4 on every ray and to 5e-6 on all but a few (the numpy MLP's sdf values differ from
    torch's in their last bits; at a near-tie of a random u with a cdf entry a sample sits in the neighbouring bin: one ray
    of 1024 here, 2.8e-5), depths to 1e-4, the uniform eikonal points' gradients to 1e-4 on every ray."""
    g = load(golden_dir, "forward1024_train_b0.05")
    rng = synth.make_train_rng(1024, seed=int(g["rng_seed"]))
    out = orc.render_forward(dict(synth.make_params(0)), g["uv"], g["pose"], g["K"], beta_param=g["beta_param"], fast=1,
                             training=True, rng=rng)
    np.testing.assert_allclose(out["rgb_values"], g["rgb_values"], atol=1e-4)
    assert (np.abs(out["rgb_values"] - g["rgb_values"]).max(-1) > 5e-6).sum() <= 4
    dd = np.abs(out["depth_values"] - g["depth_values"]).reshape(-1)      # (that ray misses the surface: depth 5, 4e-3 off)
    assert (dd > 1e-4).sum() <= 2 and dd.max() < 1e-2, (np.sort(dd)[-4:])
    # ... and ONLY on rays that miss the surface (the reference's depth there is the far bound, 5): no ray that hits it is excused
    assert np.all(g["depth_values"].reshape(-1)[dd > 1e-4] > 4.5), g["depth_values"].reshape(-1)[dd > 1e-4]
    np.testing.assert_allclose(out["grad_theta"][:1024], g["grad_theta"][:1024], atol=1e-4)
    np.testing.assert_allclose(out["grad_theta"], g["grad_theta"], atol=2e-3)
    ev = int(g["every"])
    same = np.abs(out["depth_vals"][::ev] - g["depth_vals"]) < 3e-4
    assert same.mean() > 0.99
    rays = same.all(1)           # (a moved sample changes its neighbours' interval lengths, hence their weights)
    assert rays.mean() > 0.9
    np.testing.assert_allclose(out["weights"][::ev][rays], g["weights"][rays], atol=1e-4)


def test_sampler_r256_train_mode(golden_dir):
    """The reference's TRAIN-mode sampler on 256 rays (fast = 1; stratified jitter, random u, randperm extras, eikonal pick):
    index, cdf, beta and final z / z_eik identity, bit for bit."""
    g = load256(golden_dir, "sampler256_train_b0.05")
    rng = synth.make_train_rng(256, seed=int(g["rng_seed"]))
    for i, rec in _replay_rounds(g, training=True, rng=rng):
        ref = g[f"inds_{i}"]
        assert np.array_equal(rec["beta"], g[f"beta_{i}"]) and np.array_equal(rec["inds"], ref)
        assert np.array_equal(np.take_along_axis(rec["cdf"], np.maximum(ref - 1, 0), 1), g[f"cdf_lo_{i}"])
    z, z_eik = orc.error_bound_sampler(None, g["dirs"], g["cam"], orc.get_beta(g["beta_param"]), fast=1, training=True, rng=rng,
                                       inv_4log=g["inv_4log"], sdf_override=[g["sdf_0"]])
    assert np.array_equal(z, g["z"]) and np.array_equal(z_eik, g["z_eik"])


def test_sampler_r256_background_model(golden_dir):
    """The sampler of the fg + inverted-sphere background model on 256 rays (far = sphere exit per ray -- through
    `cam_loc.norm(2, 1) ** 2`, the square of the rounded norm --, near = 0, add_tiny = 1e-6): the fg samples and the
    inverse-sphere depths are the reference's bit for bit."""
    g = load256(golden_dir, "sampler256_bg_b0.01")
    nr = int(g["n_rounds"])
    (z, z_bg), _ = orc.error_bound_sampler(None, g["dirs"], g["cam"], orc.get_beta(g["beta_param"]), near=0.0, fast=-1,
                                           inv_4log=g["inv_4log"], inverse_sphere_bg=True, N_samples_inverse_sphere=32,
                                           add_tiny=1e-6, sdf_override=[g[f"sdf_{i}"] for i in range(nr)])
    assert np.array_equal(z, g["z"]) and np.array_equal(z_bg, g["z_bg"])


def _host_matches(g):
    import torch
    e = torch.exp(torch.from_numpy(g["exp_probe_in"])).numpy()
    q = torch.sqrt(torch.from_numpy(g["sqrt_probe_in"])).numpy()
    return np.array_equal(e, g["exp_probe_out"]) and np.array_equal(q, g["sqrt_probe_out"])


def test_sampler_hostexp(golden_dir):
    """The UNPINNED reference (torch.exp / torch.sqrt = the generating host's MKL VML kernels; 64 rays, beta = 0.01, 5
    rounds).  On a host whose torch returns the same exp and sqrt (probe vectors in the fixture) the oracle with those
    two host routines bound in reproduces every index, cdf entry, beta and merged bin of the unmodified reference -- so
    the restated sum / expm1 / cumsum and every elementary operation are the reference's, and exp / sqrt are the only
    host-dependent primitives.  With the pinned (open) exp / sqrt the same fixture shows what that host dependence
    amounts to: a fraction of a percent of near-tie indices."""
    import torch
    g = load256(golden_dir, "sampler64_hostexp_b0.01")
    flips = total = 0
    for i, rec in _replay_rounds(g):
        flips += int((rec["inds"] != g[f"inds_{i}"]).sum())
        total += rec["inds"].size
    print(f"pinned exp/sqrt vs the generating host's MKL kernels: {flips} of {total} indices differ")
    assert flips <= 0.003 * total
    if not _host_matches(g):
        pytest.skip("this host's torch.exp / torch.sqrt (MKL VML dispatch) differ from the generating host's")
    saved = orc.ref_exp, orc.ref_sqrt
    orc.ref_exp = lambda x: torch.exp(torch.from_numpy(np.ascontiguousarray(x, F32))).numpy()
    orc.ref_sqrt = lambda x: torch.sqrt(torch.from_numpy(np.ascontiguousarray(x, F32))).numpy()
    try:
        _assert_rounds_exact(g)
    finally:
        orc.ref_exp, orc.ref_sqrt = saved


@pytest.mark.parametrize("beta", ["0.1", "0.01", "0.001"])
def test_forward_r256(golden_dir, beta):
    """The whole eval forward (fast = -1, up to 5 sampler rounds) on 256 rays against the reference, EVERY ray: colours,
    depths and normals far inside the north star's 1e-4 (observed 3e-7 / 5e-5 / 1e-6).  Sample POSITIONS are compared
    where they carry weight: the oracle's SDF values differ from the reference's in the last bits (MKL sgemm vs a numpy
    matmul -- any two evaluation orders do), and where the cdf is flat (transmittance ~ 0 behind the surface, denom
    clamped to 1e-5) the inverse-CDF map turns one ulp of cdf into up to 1e-2 of a bin; those samples have weight
    < 1e-5 and do not reach any output."""
    g = load(golden_dir, "forward256_b" + beta)
    params = synth.make_params(0)
    out = orc.render_forward(params, g["uv"], g["pose"], g["K"], beta_param=g["beta_param"], fast=-1)
    print(f"forward256_b{beta}: max err on ALL 256 rays: rgb {np.abs(out['rgb_values'] - g['rgb_values']).max():.2e}, "
          f"depth {np.abs(out['depth_values'] - g['depth_values']).max():.2e}, "
          f"normal {np.abs(out['normal_map'] - g['normal_map']).max():.2e}")
    np.testing.assert_allclose(out["rgb_values"], g["rgb_values"], atol=5e-6)
    np.testing.assert_allclose(out["depth_values"], g["depth_values"], atol=2e-4)
    np.testing.assert_allclose(out["normal_map"], g["normal_map"], atol=1e-5)
    moved = np.abs(out["depth_vals"] - g["depth_vals"]) > 5e-3
    assert g["weights"][moved].max(initial=0.0) < 1e-5 and out["weights"][moved].max(initial=0.0) < 1e-5


@pytest.mark.parametrize("name", SAMPLER_FX + R256)
def test_sampler_eval_chain(golden_dir, name):
    """Whole sampler, all rounds CHAINED (no restart from the reference's state), on the reference's per-round sdf values:
    the final sample set is the reference's bit for bit, on every ray (12 / 256 rays, 1 - 5 rounds, beta 0.1 ... 0.001)."""
    g = load(golden_dir, name)
    R, nr = g["dirs"].shape[0], int(g["n_rounds"])
    z, _ = orc.error_bound_sampler(None, g["dirs"], g["cam"], orc.get_beta(g["beta_param"]), fast=int(g["fast"]),
                                   inv_4log=g["inv_4log"], sdf_override=[g[f"sdf_{i}"].reshape(R, -1) for i in range(nr)])
    assert np.array_equal(z, g["z"])
    if int(g["fast"]) == 0:
        assert z.shape[1] == 128 + 34


def test_sampler_train_round(golden_dir):
    g = load(golden_dir, "sampler_train")
    R = g["dirs"].shape[0]
    rng = synth.make_train_rng(R, seed=4)
    for i, rec in _replay_rounds(g, training=True, rng=rng):
        assert _check_inds(rec, g, i) == 0
        assert not rec["upsample"]
        z, z_eik = orc.sampler_finalize(rec["z"], rec["samples"], near=F32(1e-4), far=F32(6.0), training=True, rng=rng)
        assert np.array_equal(z, g["z"]) and np.array_equal(z_eik, g["z_eik"])


def test_torch_sampler_port_vs_reference(golden_dir):
    """oracle/torch_ref.error_bound_sampler_train (the plain-torch sampler bench.py's same-GPU comparator runs) against
    the reference's own train-mode output on the `sampler_train` fixture: same draws, the reference's sdf values fed
    back in.  Same tolerance as the restatement's finalize (conditioning of (u - cdf_b) / denom)."""
    import torch
    import torch_ref as tref
    g = load(golden_dir, "sampler_train")
    R = g["dirs"].shape[0]
    rng = synth.make_train_rng(R, seed=4)
    trng = {k: torch.from_numpy(np.asarray(v)) for k, v in rng.items() if k in ("jitter", "u", "perm", "eik_idx")}
    sdf0 = torch.from_numpy(g["sdf_0"].astype(F32)).reshape(-1)
    z, z_eik = tref.error_bound_sampler_train(lambda pts: sdf0, torch.zeros(3), torch.from_numpy(g["dirs"].astype(F32)),
                                              float(orc.get_beta(g["beta_param"])), trng, fast=1)
    np.testing.assert_allclose(z.numpy(), g["z"], atol=2e-4)
    assert (np.abs(z.numpy() - g["z"]) > 2e-6).mean() < 0.01
    np.testing.assert_allclose(z_eik.numpy(), g["z_eik"], atol=2e-4)


@pytest.mark.parametrize("tag", ["eval_b0.1", "eval_b0.01", "eval_b0.01_f1", "train", "w1_eval", "w1_train"])
def test_forward(golden_dir, tag):
    """Whole forward (oracle MLPs + oracle sampler) against VolSDFNetwork.forward of the reference, every ray: integrated
    outputs to 5e-6 (colours) / 1e-5 (depths, normals).  Per-sample arrays are compared where the sample did not move:
    the oracle's SDF values differ from the reference's in their last bits, which the inverse-CDF map amplifies where
    the cdf is flat (test_forward_r256); such samples must carry no weight (< 1e-5).
    w1_*: the trained-scale weight set at its own beta = 0.005."""
    g = load(golden_dir, "forward_" + tag)
    params = synth.WEIGHT_SETS["w1" if tag.startswith("w1") else "w0"]()
    training = tag.endswith("train")
    R = g["uv"].shape[0]
    rng = synth.make_train_rng(R, seed=6) if training else None
    out = orc.render_forward(params, g["uv"], g["pose"], g["K"], beta_param=g["beta_param"], fast=int(g["fast"]),
                             training=training, rng=rng)
    moved = np.abs(out["depth_vals"] - g["depth_vals"]) > 3e-4
    assert moved.mean() < 0.05
    assert g["weights"][moved].max(initial=0.0) < 1e-5 and out["weights"][moved].max(initial=0.0) < 1e-5
    np.testing.assert_allclose(out["xyz"][~moved], g["xyz"][~moved], atol=3e-4)
    np.testing.assert_allclose(out["weights"][~moved], g["weights"][~moved], atol=2e-3)
    assert np.abs(out["weights"][~moved] - g["weights"][~moved]).mean() < 2e-5
    np.testing.assert_allclose(out["rgb_values"], g["rgb_values"], atol=5e-6)
    np.testing.assert_allclose(out["depth_values"], g["depth_values"], atol=1e-5)
    if training:
        np.testing.assert_allclose(out["grad_theta"][:R], g["grad_theta"][:R], atol=1e-4)
        np.testing.assert_allclose(out["grad_theta"][R:], g["grad_theta"][R:], atol=2e-3)
    else:
        np.testing.assert_allclose(out["normal_map"], g["normal_map"], atol=1e-5)


@pytest.mark.parametrize("tag", ["eval_b0.1", "eval_b0.01", "train"])
def test_forward_bg(golden_dir, tag):
    """VolSDFNetworkBG.forward (fg + inverted-sphere background, config 4) of the reference against the oracle:
    the background pieces on the reference's own inputs, then the whole forward."""
    g = load(golden_dir, "forward_bg_" + tag)
    params = dict(synth.make_params(0)); params.update(synth.make_bg_params(0))
    training = tag == "train"
    R = g["uv"].shape[0]
    # background pieces on the fixture's inverse depths
    dirs, cam, _ = orc.rays_from_uv(g["uv"], g["pose"], g["K"])
    Nb = g["z_bg"].shape[1]
    pts, dreal = orc.depth2pts_outside(np.repeat(cam[None, None], R, 0).repeat(Nb, 1), np.repeat(dirs[:, None], Nb, 1),
                                       g["z_bg"], 3.0)
    np.testing.assert_allclose(pts, g["bg_points"], atol=2e-6)
    np.testing.assert_allclose(dreal, g["bg_depth"], rtol=2e-5)
    bg_out = orc.sdf_mlp_forward(orc.effective_weights(params, "bg_implicit_network", 9), g["bg_points"].reshape(-1, 4),
                                 multires=10)
    np.testing.assert_allclose(bg_out[:, :1], g["bg_sdf"], atol=2e-5)
    bw = orc.bg_weights(g["z_bg"], np.abs(g["bg_sdf"]).reshape(R, Nb))
    np.testing.assert_allclose(bw, g["bg_weights"], atol=2e-6)
    # whole forward
    rng = synth.make_train_rng(R, seed=int(g["rng_seed"]), bg=True) if training else None
    out = orc.render_forward_bg(params, g["uv"], g["pose"], g["K"], beta_param=g["beta_param"], fast=int(g["fast"]),
                                training=training, rng=rng, near_pose=g["near_pose"])
    np.testing.assert_allclose(out["z_bg"], g["z_bg"], atol=1e-7)
    np.testing.assert_allclose(out["z_max"], g["z_max"], atol=2e-6)
    # every ray; per-sample weights where the sample did not move (see test_forward)
    moved = np.abs(out["depth_vals"] - g["depth_vals"]) > 3e-4
    assert moved.mean() < 0.05 and max(out["weights"][moved].max(initial=0.0), g["weights"][moved].max(initial=0.0)) < 1e-5
    np.testing.assert_allclose(out["weights"][~moved], g["weights"][~moved], atol=2e-3)
    np.testing.assert_allclose(out["bg_transmittance"], g["bg_transmittance"], atol=2e-5)
    np.testing.assert_allclose(out["rgb_values"], g["rgb_values"], atol=5e-6)
    # depth_values is normalised by the fg weight sum, which is small for rays that mostly see the background
    wsum = g["weights"].sum(1, keepdims=True)
    assert (np.abs(out["depth_values"] - g["depth_values"]) <= 3e-5 / np.maximum(wsum, 1e-3)).all()
    # the background depths reach 1e6 (1 / (depth + 1e-6)): small weight differences move this mean visibly
    np.testing.assert_allclose(out["depth_values_all"], g["depth_values_all"], rtol=3e-3)
    if training:
        np.testing.assert_allclose(out["grad_theta"][:R], g["grad_theta"][:R], atol=1e-4)
    else:
        np.testing.assert_allclose(out["normal_map"], g["normal_map"], atol=2e-5)


def test_torch_ref_forward_bg(golden_dir):
    """oracle/torch_ref.py's differentiable fg + background forward (the float64 reference of the GPU gradient tests at
    the benchmarked geometry) on the reference's own sample positions and background points == the reference's outputs
    (forward_bg_train fixture), including depth_values_all."""
    import torch
    import torch_ref as tref
    g = load(golden_dir, "forward_bg_train")
    params = dict(synth.make_params(0)); params.update(synth.make_bg_params(0))
    params["density.beta"] = np.asarray(g["beta_param"], F32)
    dirs, cam, ds = orc.rays_from_uv(g["uv"], g["pose"], g["K"])
    R = g["uv"].shape[0]
    p = tref.to_torch(params, torch.float64)
    with torch.enable_grad():
        out = tref.forward_differentiable_bg(p, cam, dirs, g["z_vals"], g["z_max"], np.zeros((2, 3), F32), ds, g["z_bg"],
                                             g["bg_points"], bg_depth=g["bg_depth"])
    n = lambda t: t.detach().numpy()
    np.testing.assert_allclose(n(out["weights"]), g["weights"], atol=2e-5)
    np.testing.assert_allclose(n(out["bg_transmittance"]), g["bg_transmittance"], atol=2e-5)
    np.testing.assert_allclose(n(out["rgb_values"]), g["rgb_values"], atol=2e-5)
    # depth_values = sum(w z) / (sum(w) + 1e-8): rays that see only the background have sum(w) ~ 1e-7 in float32 (or an
    # exact 0) and the quotient means nothing there
    fg = g["weights"].sum(1) > 1e-3
    assert fg.sum() >= 3
    np.testing.assert_allclose(n(out["depth_values"])[fg], g["depth_values"][fg], rtol=2e-4)
    np.testing.assert_allclose(n(out["depth_values_all"]), g["depth_values_all"], rtol=2e-4)
    np.testing.assert_allclose(n(out["bg_out0"]), g["bg_sdf"], atol=2e-5)
    # and it is differentiable down to the background parameters
    out["depth_values_all"].sum().backward()
    assert float(p["bg_implicit_network.lin0.weight"].grad.abs().max()) > 0


@pytest.mark.parametrize("name", ["cost_mapping_inv0_v0", "cost_mapping_inv0_v2", "cost_mapping_inv1_v0",
                                  "cost_mapping_inv1_v2"])
def test_cost_mapping(golden_dir, name):
    g = load(golden_dir, name)
    views = synth.make_mvs_views(int(g["seed"]))
    pj, pi, valid = orc.cost_mapping(g["xyz"], int(g["view_index"]), views, (576, 768), bool(g["inverse_depth"]))
    assert np.array_equal(valid, g["valid"])
    assert 0.05 < valid.mean() < 0.95, "fixture must mix valid and invalid samples"
    np.testing.assert_allclose(pj, g["pj"], atol=2e-6)
    np.testing.assert_allclose(pi, g["pi"], atol=2e-6)


@pytest.mark.parametrize("name", ["cost_mapping_inv0_v0", "cost_mapping_inv0_v2", "cost_mapping_inv1_v0",
                                  "cost_mapping_inv1_v2"])
def test_torch_cost_mapping_vs_reference(golden_dir, name):
    """oracle/torch_ref.cost_mapping (the prior look-up of bench.py's same-GPU comparator) against the reference's own
    outputs, same fixtures and bars as the numpy restatement."""
    import torch
    import torch_ref as tref
    g = load(golden_dir, name)
    views = synth.make_mvs_views(int(g["seed"]))
    pj, pi, valid = tref.cost_mapping(torch.from_numpy(g["xyz"].astype(F32)), int(g["view_index"]), views, (576, 768),
                                      bool(g["inverse_depth"]))
    assert np.array_equal(valid.numpy(), g["valid"])
    np.testing.assert_allclose(pj.numpy(), g["pj"], atol=2e-6)
    np.testing.assert_allclose(pi.numpy(), g["pi"], atol=2e-6)


def test_loss(golden_dir):
    g = load(golden_dir, "loss")
    out = {k: g[k] for k in ("rgb_values", "grad_theta", "weights", "pi", "pj", "depth_values")}
    for it in (0, 100, 250):
        res = orc.volsdf_loss(out, g["rgb"], g["rgb_smooth"], it)
        for k, v in res.items():
            np.testing.assert_allclose(v, g[f"it{it}_{k}"], rtol=2e-6, atol=1e-7, err_msg=f"it{it} {k}")


# ------------------------------------------------------------------------------------------------------
# CasMVSNet cost volume (a13-a16)
# ------------------------------------------------------------------------------------------------------
import casmvs_oracle as corc   # noqa: E402


def test_homo_warp(golden_dir):
    g = load(golden_dir, "homo_warp")
    w = corc.homo_warp(g["src"], g["src_proj"], g["ref_proj"], g["depth_values"])
    assert (g["warped"] == 0).mean() > 0.02, "fixture must contain off-image samples"
    np.testing.assert_allclose(w, g["warped"], atol=2e-4)
    assert np.abs(w - g["warped"]).mean() < 2e-6


def test_depthnet_tail_d192(golden_dir):
    g = load(golden_dir, "depthnet_tail_d192")
    prob, depth, conf, idx = corc.depthnet_tail(g["reg"], g["depth_values"])
    np.testing.assert_allclose(prob, g["prob"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(depth, g["depth"], rtol=2e-6)
    assert np.array_equal(idx, g["idx"])
    np.testing.assert_allclose(conf, g["conf"], atol=3e-7)


def test_casmvs_three_stages(golden_dir):
    g = load(golden_dir, "casmvs_3stage")
    feats, proj, depth_values = synth.make_mvs_sample(int(g["seed"]), img_hw=(64, 96))
    ndepths = [int(x) for x in g["ndepths"]]
    int_r = [1.0, 0.5, 0.5]
    prev = None
    for st in range(3):
        key = f"stage{st + 1}"
        dv = corc.depth_hypotheses(st, depth_values, (64, 96), ndepths[st], (4, 2, 1)[st], int_r[st], prev_depth=prev)
        np.testing.assert_allclose(dv, g[f"s{st}_depth_values"], rtol=3e-6, err_msg=f"depth hypotheses {key}")
        out = corc.depthnet_forward([f[key] for f in feats], list(proj[key]), g[f"s{st}_depth_values"],
                                    synth.make_costreg_params(100 + st, (32, 16, 8)[st]))
        np.testing.assert_allclose(out["variance"].reshape(-1)[g[f"s{st}_variance_idx"]], g[f"s{st}_variance_val"],
                                   atol=2e-4, err_msg=f"variance {key}")
        np.testing.assert_allclose(out["reg"], g[f"s{st}_reg"], atol=2e-3, err_msg=f"reg {key}")
        assert np.abs(out["reg"] - g[f"s{st}_reg"]).mean() < 5e-5
        # the tail is pinned on the reference's own regularised volume (argmax-like index is discontinuous)
        prob, depth, conf, idx = corc.depthnet_tail(g[f"s{st}_reg"], g[f"s{st}_depth_values"])
        np.testing.assert_allclose(depth, g[f"s{st}_depth"], rtol=3e-6, err_msg=f"depth {key}")
        np.testing.assert_allclose(conf, g[f"s{st}_conf"], atol=5e-7, err_msg=f"conf {key}")
        prev = g["stage1_depth_override"] if st == 0 else g[f"s{st}_depth"]


# ------------------------------------------------------------------------------------------------------
# oracle/torch_ref.py (autograd reference of the differentiable part) pinned on the reference's fixtures
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,wset", [("train_step", "w0"), ("train_step_w1", "w1")])
def test_torch_ref_forward_and_first_step_gradients(golden_dir, name, wset):
    """torch_ref forward on the reference's own sample positions == the reference's outputs, and its gradients of
    the reference loss == the reference's gradients (train_step fixtures, step 0; both weight sets)."""
    import torch
    import torch_ref as tref
    g = load(golden_dir, name)
    params = synth.WEIGHT_SETS[wset]()
    views = synth.make_mvs_views(int(g["mvs_seed"]))
    K, pose = views[0]["K"], views[0]["c2w"]
    R = g["uv"].shape[0]
    rng = synth.make_train_rng(R, seed=100)
    layers = orc.effective_weights(params, "implicit_network", 9)
    dirs, cam, ds = orc.rays_from_uv(g["uv"], pose, K)
    z, z_eik = orc.error_bound_sampler(lambda x: orc.sdf_vals(layers, x), dirs, cam, orc.get_beta(params["density.beta"]),
                                       fast=1, training=True, rng=rng)
    eik = np.concatenate([rng["eik_points"], (cam[None] + z_eik * dirs).astype(F32)], 0)
    p = tref.to_torch(params, torch.float64)
    out = tref.forward_differentiable(p, cam, dirs, z, eik, ds)
    xyz = (cam[None, None] + z[:, :, None] * dirs[:, None, :]).astype(F32)
    pj, pi, _ = orc.cost_mapping(xyz, 0, views, (576, 768))
    out["pi"], out["pj"] = torch.tensor(pi, dtype=torch.float64), torch.tensor(pj, dtype=torch.float64)
    total = tref.loss_fn(out, torch.tensor(g["rgb"].reshape(-1, 3), dtype=torch.float64),
                         torch.tensor(g["rgb_smooth"].reshape(-1, 3), dtype=torch.float64), 0)
    np.testing.assert_allclose(float(total), float(g["s0_loss"]), rtol=2e-4)
    total.backward()
    worst = 0.0
    for name in params:
        if f"s0_grad/{name}" not in g:
            continue
        got = p[name].grad.numpy().reshape(-1)[g[f"s0_grad_idx/{name}"]]
        ref = g[f"s0_grad/{name}"]
        worst = max(worst, float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)))
    assert worst < 2e-3, worst


def test_feature_net(golden_dir):
    """The torch-functional restatement of FeatureNet == the reference's module (fixture featurenet.npz)."""
    g = dict(np.load(os.path.join(golden_dir, "featurenet.npz")))
    out = corc.feature_net_torch(synth.make_featurenet_params(int(g["seed"])), g["img"][0])
    for k in ("stage1", "stage2", "stage3"):
        assert out[k].shape == g[k].shape
        np.testing.assert_allclose(out[k], g[k], atol=2e-6)
    assert g["stage1"].shape == (32, 9, 13) and g["stage3"].shape == (8, 36, 52)
