"""Plain torch restatement (CPU, any float dtype) of the differentiable part of the hot path, with autograd.

TEST INFRASTRUCTURE ONLY (see oracle/svs_oracle.py).  Floating-point kernels may be checked against a plain torch
reference; this file is that reference for everything that needs gradients: the SDF / radiance MLPs
(volsdf/model/network.py:71-131,170-190), compositing (:281-295,237-243), the loss (volsdf/model/loss.py:80-114) and one
optimisation step (volsdf/vsdf.py:214-219).  The sampler (no gradients flow through it) comes from svs_oracle.
Pinned by tests/test_oracle_golden.py::test_torch_ref_* against the reference-generated fixtures.
"""
import math

import numpy as np
import torch


def to_torch(params, dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in params.items()}


def weightnorm(p, prefix, l):
    v, g = p[f"{prefix}.lin{l}.weight_v"], p[f"{prefix}.lin{l}.weight_g"]
    return g * v / v.norm(dim=1, keepdim=True)


def posenc(x, L):
    out = [x]
    for k in range(L):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def sdf_mlp(p, x):
    """ImplicitNetwork.forward (network.py:71-88) -> (P,257)"""
    inp = posenc(x, 6)
    h = inp
    for l in range(9):
        if l == 4:
            h = torch.cat([h, inp], 1) / np.sqrt(2)
        h = h @ weightnorm(p, "implicit_network", l).T + p[f"implicit_network.lin{l}.bias"]
        if l < 8:
            h = torch.nn.functional.softplus(h, beta=100)
    return h


def sdf_outputs(p, x, radius=3.0, scale=20.0, clamp=True):
    """get_outputs (network.py:105-123) / gradient (:90-103) with create_graph=True."""
    x = x.requires_grad_(True)
    out = sdf_mlp(p, x)
    sdf = out[:, :1]
    if clamp and radius > 0:
        sdf = torch.minimum(sdf, scale * (radius - x.norm(2, 1, keepdim=True)))
    grad = torch.autograd.grad(sdf.sum(), x, create_graph=True)[0]
    return sdf, out[:, 1:], grad


def rgb_mlp(p, x, n, d, feat):
    """RenderingNetwork.forward, mode idr (network.py:170-190)"""
    h = torch.cat([x, posenc(d, 1), n, feat], -1)
    for l in range(5):
        h = h @ weightnorm(p, "rendering_network", l).T + p[f"rendering_network.lin{l}.bias"]
        if l < 4:
            h = torch.relu(h)
    return torch.sigmoid(h)


def composite(z, sdf, rgb, beta_param, ds, beta_min=1e-4):
    """network.py:281-295 + :237-243 -> weights, rgb_values, depth_values"""
    beta = beta_param.abs() + beta_min
    sigma = (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    fe = dists * sigma
    sfe = torch.cat([torch.zeros_like(fe[:, :1]), fe[:, :-1]], -1)
    w = (1 - torch.exp(-fe)) * torch.exp(-torch.cumsum(sfe, -1))
    rgb_values = (w.unsqueeze(-1) * rgb).sum(1)
    depth_values = ds * ((w * z).sum(1, keepdim=True) / (w.sum(1, keepdim=True) + 1e-8))
    return w, rgb_values, depth_values


def loss_fn(out, rgb, rgb_smooth, it, *, eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
            gce=0.5, confi=1e-3, anneal_rgb=200, norm=None):
    """VolSDFLoss.forward (loss.py:80-114) -> total.  norm = (n_rays, n_eik): denominators of the means when `out` holds
    only a shard of a larger batch (the shards' totals then add up to the batch's total); None: the sizes of `out`."""
    n_rays, n_eik = norm if norm is not None else (out["rgb_values"].shape[0], out["grad_theta"].shape[0])
    over_rays = lambda per_ray: per_ray.sum() / n_rays
    rgb_loss = over_rays((out["rgb_values"] - rgb).abs().mean(-1))
    eik = ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).sum() / n_eik
    total = eikonal_weight * eik
    on = sparse_weight > 0 and anneal_rgb > 0 and it < anneal_rgb
    if "pi" in out:
        pw = out["pi"] * out["pj"]
        w = out["weights"]
        l = (-pw * w.detach() ** gce * torch.log(w + 1e-8)).sum(1)
        total = total + mvs_weight * over_rays(1. * (pw.sum(1) > confi) * l)
        if on:
            conf = pw.sum(-1)
            sparse = over_rays((1. / (out["depth_values"].squeeze() + 1e-3)) * (conf < confi))
            total = total + sparse_weight * (1.0 - it / anneal_rgb) * sparse
            rgb_loss = over_rays((out["rgb_values"] - rgb_smooth).abs().mean(-1) * (conf < 1e-8))
    return total + rgb_weight * rgb_loss


def forward_differentiable(p, cam, dirs, z, eik_points, depth_scale, radius=3.0, scale=20.0, device=None):
    """VolSDFNetwork.forward after the sampler (network.py:226-268), train mode.  device: where the parameters live
    (None: CPU; inputs may then be numpy arrays or tensors)."""
    R, S = z.shape
    dt = p["density.beta"].dtype
    conv = lambda a: (a.detach().to(device=device, dtype=dt) if torch.is_tensor(a)
                      else torch.tensor(np.asarray(a), dtype=dt, device=device))
    cam_t, dirs_t, z_t = conv(cam), conv(dirs), conv(z)
    pts = (cam_t.view(1, 1, 3) + z_t.unsqueeze(2) * dirs_t.unsqueeze(1)).reshape(-1, 3)
    sdf, feat, grad = sdf_outputs(p, pts, radius, scale)
    dflat = dirs_t.unsqueeze(1).repeat(1, S, 1).reshape(-1, 3)
    rgb = rgb_mlp(p, pts, grad, dflat, feat).reshape(R, S, 3)
    w, rgb_values, depth_values = composite(z_t, sdf.reshape(R, S), rgb, p["density.beta"], conv(depth_scale))
    _, _, gt = sdf_outputs(p, conv(eik_points), clamp=False)
    return dict(rgb_values=rgb_values, depth_values=depth_values, weights=w, grad_theta=gt, sdf=sdf, rgb=rgb)


# ---------------------------------------------------------------------------------------------------------
# VolSDFNetworkBG (volsdf/model/network_bg.py), train mode, after the sampler
# ---------------------------------------------------------------------------------------------------------
def bg_sdf_mlp(p, x):
    """bg_implicit_network: ImplicitNetwork with d_in 4, multires 10, no weight-norm (network.py:71-88) -> (P,257)"""
    inp = posenc(x, 10)
    h = inp
    for l in range(9):
        if l == 4:
            h = torch.cat([h, inp], 1) / np.sqrt(2)
        h = h @ p[f"bg_implicit_network.lin{l}.weight"].T + p[f"bg_implicit_network.lin{l}.bias"]
        if l < 8:
            h = torch.nn.functional.softplus(h, beta=100)
    return h


def bg_rgb_mlp(p, d, feat):
    """bg_rendering_network, mode nerf (network.py:176): cat[PE4(view), feature] -> 128 -> 3"""
    h = torch.cat([posenc(d, 4), feat], -1)
    h = torch.relu(h @ p["bg_rendering_network.lin0.weight"].T + p["bg_rendering_network.lin0.bias"])
    return torch.sigmoid(h @ p["bg_rendering_network.lin1.weight"].T + p["bg_rendering_network.lin1.bias"])


def composite_bg(z, z_max, sdf, rgb, beta_param, ds, z_bg, bg_out0, bg_rgb, beta_min=1e-4, bg_depth=None):
    """network_bg.py:76-125,147-180 -> weights, bg_transmittance, rgb_values, depth_values[, depth_values_all when the
    background samples' conventional depths bg_depth (R,Nb) are given: network_bg.py:105-107]"""
    beta = beta_param.abs() + beta_min
    sigma = (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))
    dists = torch.cat([z[:, 1:] - z[:, :-1], z_max.unsqueeze(-1) - z[:, -1:]], -1)
    fe = dists * sigma
    sfe = torch.cat([torch.zeros_like(fe[:, :1]), fe], -1)
    trans = torch.exp(-torch.cumsum(sfe, -1))
    w = (1 - torch.exp(-fe)) * trans[:, :-1]
    t_bg = trans[:, -1]
    bdists = torch.cat([z_bg[:, :-1] - z_bg[:, 1:], torch.full_like(z_bg[:, :1], 1e10)], -1)
    bfe = bdists * bg_out0.abs()
    bsfe = torch.cat([torch.zeros_like(bfe[:, :1]), bfe[:, :-1]], -1)
    bw = (1 - torch.exp(-bfe)) * torch.exp(-torch.cumsum(bsfe, -1))
    rgb_values = (w.unsqueeze(-1) * rgb).sum(1) + t_bg.unsqueeze(-1) * (bw.unsqueeze(-1) * bg_rgb).sum(1)
    dv = z * ds
    depth_values = (w * dv).sum(1, keepdim=True) / (w.sum(1, keepdim=True) + 1e-8)
    if bg_depth is None:
        return w, t_bg, rgb_values, depth_values
    w_all = torch.cat([w, t_bg.unsqueeze(-1) * bw], 1)
    d_all = ds * torch.cat([z, bg_depth], 1)
    depth_values_all = (w_all * d_all).sum(1, keepdim=True) / (w_all.sum(1, keepdim=True) + 1e-8)
    return w, t_bg, rgb_values, depth_values, depth_values_all


def forward_differentiable_bg(p, cam, dirs, z, z_max, eik_points, depth_scale, z_bg, bg_pts, bg_depth=None, device=None):
    """VolSDFNetworkBG.forward after the sampler and depth2pts_outside, train mode (network_bg.py:60-134).  bg_depth: the
    background samples' conventional depths (adds depth_values_all); device: where the parameters live (None: CPU)."""
    R, S = z.shape
    Nb = z_bg.shape[1]
    dt = p["density.beta"].dtype
    T = lambda a: (a.detach().to(device=device, dtype=dt) if torch.is_tensor(a)
                   else torch.tensor(np.asarray(a), dtype=dt, device=device))
    cam_t, dirs_t, z_t = T(cam), T(dirs), T(z)
    pts = (cam_t.view(1, 1, 3) + z_t.unsqueeze(2) * dirs_t.unsqueeze(1)).reshape(-1, 3)
    sdf, feat, grad = sdf_outputs(p, pts, 0.0, 1.0)
    rgb = rgb_mlp(p, pts, grad, dirs_t.unsqueeze(1).repeat(1, S, 1).reshape(-1, 3), feat).reshape(R, S, 3)
    bg_out = bg_sdf_mlp(p, T(bg_pts).reshape(-1, 4))
    bg_rgb = bg_rgb_mlp(p, dirs_t.unsqueeze(1).repeat(1, Nb, 1).reshape(-1, 3), bg_out[:, 1:]).reshape(R, Nb, 3)
    comp = composite_bg(z_t, T(z_max), sdf.reshape(R, S), rgb, p["density.beta"], T(depth_scale), T(z_bg),
                        bg_out[:, 0].reshape(R, Nb), bg_rgb, bg_depth=None if bg_depth is None else T(bg_depth))
    w, t_bg, rgb_values, depth_values = comp[:4]
    _, _, gt = sdf_outputs(p, T(eik_points), clamp=False)
    out = dict(rgb_values=rgb_values, depth_values=depth_values, weights=w, grad_theta=gt, sdf=sdf, rgb=rgb,
               bg_out0=bg_out[:, :1], bg_rgb=bg_rgb, bg_transmittance=t_bg)
    if bg_depth is not None:
        out["depth_values_all"] = comp[4]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The error-bounded sampler in plain torch (train mode), for the same-GPU comparator of bench.py: the reference runs
# ErrorBoundSampler.get_z_vals (volsdf/model/ray_sampler.py:67-219) in torch under no_grad in every train step.  This
# is the same sequence of tensor operations in float32 -- NOT the bit-exact restatement (that is oracle/svs_oracle.py,
# numpy, with the reference's reduction orders); tests/test_oracle_golden.py holds it to the restatement to 1e-4.
# ---------------------------------------------------------------------------------------------------------------------
def _density(sdf, beta):
    alpha = 1.0 / beta
    return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


def _error_bound(beta, sdf, dists, d_star):
    density = _density(sdf, beta)
    sfe = torch.cat([torch.zeros_like(dists[:, :1]), dists * density[:, :-1]], -1)
    integral = torch.cumsum(sfe, -1)
    err = torch.exp(-d_star / beta) * (dists ** 2.0) / (4 * beta ** 2)
    err_int = torch.cumsum(err, -1)
    bound = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * torch.exp(-integral[:, :-1])
    return bound.max(-1)[0]


def error_bound_sampler_train(sdf_fn, cam, dirs, beta0, rng, *, near=1e-4, radius=3.0, N_samples=64, N_samples_eval=128,
                              N_samples_extra=32, eps=0.1, beta_iters=10, max_total_iters=5, fast=1):
    """sdf_fn (P,3) -> (P,) ; cam (3,), dirs (R,3); rng: 'jitter' (R,128), 'u' (R,64), 'perm' (n,), 'eik_idx' (R,) tensors.
    -> z_vals (R, N_samples + N_samples_extra + 2), z_eik (R,1)."""
    R, dev = dirs.shape[0], dirs.device
    far = 2.0 * radius
    max_iters = fast if fast >= 0 else max_total_iters
    t = torch.linspace(0.0, 1.0, N_samples_eval, device=dev)
    z = (near * (1.0 - t) + far * t).expand(R, -1)
    mids = 0.5 * (z[:, 1:] + z[:, :-1])
    upper, lower = torch.cat([mids, z[:, -1:]], -1), torch.cat([z[:, :1], mids], -1)
    z = lower + (upper - lower) * rng["jitter"]
    samples, samples_idx = z, None
    dists = z[:, 1:] - z[:, :-1]
    bound = (1.0 / (4.0 * math.log(eps + 1.0))) * (dists ** 2.0).sum(-1)
    beta = torch.sqrt(bound)
    beta0 = torch.as_tensor(beta0, dtype=torch.float32, device=dev)
    total, not_converge, sdf = 0, True, None
    while not_converge and total < max_iters:
        pts = cam.view(1, 1, 3) + samples.unsqueeze(2) * dirs.unsqueeze(1)
        with torch.no_grad():
            samples_sdf = sdf_fn(pts.reshape(-1, 3)).reshape(R, -1)
        if samples_idx is not None:
            sdf = torch.gather(torch.cat([sdf, samples_sdf], -1), 1, samples_idx)
        else:
            sdf = samples_sdf
        dists = z[:, 1:] - z[:, :-1]
        a, b, c = dists, sdf[:, :-1].abs(), sdf[:, 1:].abs()
        first, second = a.pow(2) + b.pow(2) <= c.pow(2), a.pow(2) + c.pow(2) <= b.pow(2)
        d_star = torch.zeros_like(dists)
        d_star[first] = b[first]
        d_star[second] = c[second]
        s = (a + b + c) / 2.0
        area = s * (s - a) * (s - b) * (s - c)
        mask = ~first & ~second & (b + c - a > 0)
        d_star[mask] = (2.0 * torch.sqrt(area[mask])) / a[mask]
        d_star = (sdf[:, 1:].sign() * sdf[:, :-1].sign() == 1) * d_star
        curr = _error_bound(beta0, sdf, dists, d_star)
        beta = torch.where(curr <= eps, beta0.expand_as(beta), beta)
        beta_min, beta_max = beta0.expand_as(beta).clone(), beta
        for _ in range(beta_iters):
            mid = (beta_min + beta_max) / 2.0
            curr = _error_bound(mid.unsqueeze(-1), sdf, dists, d_star)
            beta_max = torch.where(curr <= eps, mid, beta_max)
            beta_min = torch.where(curr > eps, mid, beta_min)
        beta = beta_max
        density = _density(sdf, beta.unsqueeze(-1))
        d1 = torch.cat([dists, torch.full_like(dists[:, :1], 1e10)], -1)
        fe = d1 * density
        sfe = torch.cat([torch.zeros_like(fe[:, :1]), fe[:, :-1]], -1)
        alpha = 1 - torch.exp(-fe)
        trans = torch.exp(-torch.cumsum(sfe, -1))
        weights = alpha * trans
        total += 1
        not_converge = bool((beta.max() > beta0).item())            # the reference's host synchronisation (:136)
        if not_converge and total < max_iters:
            N = N_samples_eval
            bins = z
            err = torch.exp(-d_star / beta.unsqueeze(-1)) * (dists ** 2.0) / (4 * beta.unsqueeze(-1) ** 2)
            err_int = torch.cumsum(err, -1)
            pdf = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * trans[:, :-1]
            N, u = N_samples_eval, torch.linspace(0.0, 1.0, N_samples_eval, device=dev).expand(R, -1)
        else:
            bins = z
            pdf = weights[..., :-1] + 1e-5
            N, u = N_samples, rng["u"]
        pdf = pdf / pdf.sum(-1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
        inds = torch.searchsorted(cdf, u.contiguous(), right=True)
        below, above = (inds - 1).clamp(min=0), inds.clamp(max=cdf.shape[-1] - 1)
        cb, ca = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
        bb, ba = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
        denom = ca - cb
        denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
        samples = bb + (u - cb) / denom * (ba - bb)
        if not_converge and total < max_iters:
            z, samples_idx = torch.sort(torch.cat([z, samples], -1), -1)
    z_samples = samples
    nearc, farc = torch.full((R, 1), near, device=dev), torch.full((R, 1), far, device=dev)
    idx = rng["perm"][:N_samples_extra]
    z_final = torch.sort(torch.cat([z_samples, nearc, farc, z[:, idx]], -1), -1)[0]
    z_eik = torch.gather(z_final, 1, rng["eik_idx"].view(-1, 1))
    return z_final, z_eik


def cost_mapping(xyz, view_index, views, img_res, inverse_depth=False):
    """VolOpt.cost_mapping (vsdf.py:382-452) in torch: per training view a rigid transform, the skew-aware projection, two
    bilinear look-ups of the near / far hypotheses, the depth normalisation and one trilinear look-up of the probability
    volume (grid_sample, zeros padding, align_corners=True, as the reference calls it).  Torch counterpart of
    svs_oracle.cost_mapping (same arguments; tensors on xyz's device) for bench.py's same-GPU comparator; held to the
    reference's outputs by tests/test_oracle_golden.py::test_torch_cost_mapping_vs_reference."""
    from torch.nn.functional import grid_sample
    dev = xyz.device
    R, S, _ = xyz.shape
    _h, _w = img_res
    as_t = lambda a: torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a), dtype=torch.float32, device=dev)
    pj = torch.zeros(R, S, device=dev)
    pi = torch.zeros(R, S, device=dev)
    valid = torch.zeros(R, S, dtype=torch.bool, device=dev)
    with torch.no_grad():
        for i, v in enumerate(views):
            K, c2w = as_t(v["K"]), as_t(v["c2w"])[:3]
            cost = v["cost"] if torch.is_tensor(v["cost"]) else as_t(v["cost"])
            if "z_mvs" in v:
                z_mvs = v["z_mvs"] if torch.is_tensor(v["z_mvs"]) else as_t(v["z_mvs"])
            else:                                                   # only the first and the last hypothesis plane are read
                z_mvs = torch.stack([as_t(v["z_near"]), as_t(v["z_far"])])
            cost, z_mvs = cost.reshape(cost.shape[-3:]).to(dev), z_mvs.reshape(z_mvs.shape[-3:]).to(dev)
            fx, fy, cx, cy, sk = K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1]
            # same sequence as svs_oracle.cost_mapping, one torch operation per numpy operation
            p = (xyz - c2w[:, 3].view(1, 1, 3)) @ c2w[:, :3]
            z = p[..., 2]
            y = (p[..., 1] / z) * fy + cy
            x = (p[..., 0] / z) * fx + cx + (y - cy) * sk / fy
            x = x / ((_w - 1) / 2) - 1
            y = y / ((_h - 1) / 2) - 1
            inval = (z < 1e-5) | (x > 1.001) | (x < -1.001) | (y > 1.001) | (y < -1.001)
            off = torch.full_like(x, -99.)
            x, y, z = torch.where(inval, off, x), torch.where(inval, off, y), torch.where(inval, off, z)
            xy = torch.stack([x, y], -1).view(1, R, S, 2)
            near = grid_sample(z_mvs[None, :1], xy, mode='bilinear', padding_mode='zeros', align_corners=True)[0, 0]
            far = grid_sample(z_mvs[None, -1:], xy, mode='bilinear', padding_mode='zeros', align_corners=True)[0, 0]
            if inverse_depth:
                far = torch.where(inval, torch.full_like(far, 1e-8), far)
                zn = 2 * (1. - near / z) / (1. - near / far) - 1
            else:
                zn = 2 * (z - near) / (far - near) - 1
            inval = (near < 1e-5) | (far < 1e-5) | (zn > 1.01) | (zn < -1.01) | inval
            x, y, zn = torch.where(inval, off, x), torch.where(inval, off, y), torch.where(inval, off, zn)
            g = torch.stack([x, y, zn], -1).permute(1, 0, 2).reshape(1, S, R, 1, 3)
            c = grid_sample(cost[None, None], g, mode='bilinear', padding_mode='zeros', align_corners=True)
            c = c.reshape(S, R).permute(1, 0)
            if i == view_index:
                pi = c
            else:
                pj = pj + c
                valid = valid | ~inval
        pi = torch.where(valid, pi, torch.zeros_like(pi))
    return pj, pi, valid
