"""CPU oracle for the CasMVSNet cost-volume build.

TEST INFRASTRUCTURE ONLY (see oracle/svs_oracle.py).  numpy restatement of models/CasMVSNet.py:280-315
(homo_warping), :601-663 (DepthNet.forward), :519-595 and :705-761 (depth hypotheses, CascadeMVSNet.forward).
The 3-D U-Net (:441-472) is a floating-point contraction: its reference here is a plain torch float32
restatement (`cost_reg_net_torch`), as allowed for floating-point kernels.
Parity status: PINNED by tests/test_oracle_golden.py against fixtures generated from the imported reference.
"""
import numpy as np

F32 = np.float32
F64 = np.float64


def combine_proj(P):
    """DepthNet.forward :622-625: P (2,4,4) -> K[:3,:3] @ E[:3,:4] inside a copy of the extrinsic."""
    out = np.asarray(P[0], F32).copy()
    out[:3, :4] = (np.asarray(P[1], F32)[:3, :3] @ np.asarray(P[0], F32)[:3, :4]).astype(F32)
    return out


def relative_proj(src_proj, ref_proj):
    """:290-292: proj = src @ inv(ref) -> rot (3,3), trans (3,).  The inverse is taken in float64."""
    proj = (np.asarray(src_proj, F64) @ np.linalg.inv(np.asarray(ref_proj, F64))).astype(F32)
    return proj[:3, :3], proj[:3, 3]


def homo_warp(src_fea, src_proj, ref_proj, depth_values):
    """homo_warping, CasMVSNet.py:280-315.  src_fea (C,H,W), depth_values (D,H,W) -> (C,D,H,W).
    grid_sample bilinear / zeros / align_corners=False on coordinates normalised with the (W-1)/2 formula,
    i.e. sample position x_s = px*W/(W-1) - 0.5."""
    C, H, W = src_fea.shape
    D = depth_values.shape[0]
    rot, trans = relative_proj(src_proj, ref_proj)
    y, x = np.meshgrid(np.arange(H, dtype=F32), np.arange(W, dtype=F32), indexing="ij")
    xyz = np.stack([x, y, np.ones_like(x)], 0).reshape(3, -1)                      # (3, H*W)
    rot_xyz = (rot @ xyz).astype(F32)                                              # (3, H*W)
    p = (rot_xyz[:, None, :] * depth_values.reshape(1, D, -1)).astype(F32) + trans.reshape(3, 1, 1)
    with np.errstate(all="ignore"):
        px = (p[0] / p[2]).astype(F32)
        py = (p[1] / p[2]).astype(F32)
        gx = (px / F32((W - 1) / 2) - F32(1.0)).astype(F32)
        gy = (py / F32((H - 1) / 2) - F32(1.0)).astype(F32)
        ix = (((gx + F32(1.0)) * F32(W) - F32(1.0)) / F32(2.0)).astype(F32)          # align_corners=False unnormalise
        iy = (((gy + F32(1.0)) * F32(H) - F32(1.0)) / F32(2.0)).astype(F32)
    x0, y0 = np.floor(ix), np.floor(iy)
    tx, ty = (ix - x0).astype(F32), (iy - y0).astype(F32)
    out = np.zeros((C, D, H * W), F32)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            w = ((tx if dx else F32(1.0) - tx) * (ty if dy else F32(1.0) - ty)).astype(F32)
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            xi_c = np.clip(np.nan_to_num(xi, nan=0, posinf=0, neginf=0), 0, W - 1).astype(np.int64)
            yi_c = np.clip(np.nan_to_num(yi, nan=0, posinf=0, neginf=0), 0, H - 1).astype(np.int64)
            vals = src_fea[:, yi_c, xi_c]                                          # (C, D, H*W)
            out += np.where(ok[None], w[None] * vals, F32(0.0)).astype(F32)
    return out.reshape(C, D, H, W)


def variance_volume(features, projs, depth_values):
    """DepthNet.forward :611-642.  features: list of (C,H,W) (ref first), projs: list of (2,4,4)."""
    ref = features[0]
    D = depth_values.shape[0]
    V = len(features)
    vol_sum = np.repeat(ref[:, None], D, 1).astype(F32)
    vol_sq = (vol_sum ** 2).astype(F32)
    ref_proj = combine_proj(projs[0])
    for fea, P in zip(features[1:], projs[1:]):
        warped = homo_warp(fea, combine_proj(P), ref_proj, depth_values)
        vol_sum = vol_sum + warped
        vol_sq = vol_sq + warped ** 2
    return (vol_sq / F32(V) - (vol_sum / F32(V)) ** 2).astype(F32)


def depthnet_tail(reg, depth_values):
    """:648-663.  reg (D,H,W) regularised cost, depth_values (D,H,W) -> prob (D,H,W), depth (H,W), conf (H,W), idx."""
    D = reg.shape[0]
    m = reg.max(0, keepdims=True)
    e = np.exp((reg - m).astype(F64))
    prob = (e / e.sum(0, keepdims=True)).astype(F32)
    depth = (prob * depth_values).sum(0, dtype=F32)
    padded = np.concatenate([np.zeros((1,) + prob.shape[1:], F32), prob, np.zeros((2,) + prob.shape[1:], F32)], 0)
    sum4 = (padded[0:D] + padded[1:D + 1] + padded[2:D + 2] + padded[3:D + 3]).astype(F32)
    idx_f = (prob * np.arange(D, dtype=F32).reshape(-1, 1, 1)).sum(0, dtype=F32)
    idx = np.clip(idx_f.astype(np.int64), 0, D - 1)
    conf = np.take_along_axis(sum4, idx[None], 0)[0]
    return prob, depth, conf, idx


def _linear_resize_axis(x, out_size, axis):
    """F.interpolate(mode=(bi|tri)linear, align_corners=False) along one axis."""
    n = x.shape[axis]
    if out_size == n:
        return x
    scale = n / out_size
    src = np.maximum((np.arange(out_size) + 0.5) * scale - 0.5, 0.0).astype(F32)
    i0 = np.minimum(np.floor(src).astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    t = (src - i0).astype(F32)
    shp = [1] * x.ndim
    shp[axis] = out_size
    a, b = np.take(x, i0, axis), np.take(x, i1, axis)
    return ((F32(1.0) - t.reshape(shp)) * a + t.reshape(shp) * b).astype(F32)


def resize_linear(x, out_shape):
    """separable linear resize of the trailing len(out_shape) axes (align_corners=False)."""
    for k, s in enumerate(out_shape):
        x = _linear_resize_axis(x, s, x.ndim - len(out_shape) + k)
    return x


def depth_hypotheses(stage_idx, depth_values_1d, img_hw, ndepth, stage_scale, int_r, prev_depth=None,
                     inverse_depth=False):
    """CascadeMVSNet.forward :712-751 with get_depth_range_samples(:579-595) / _inverse (:538-547) /
    get_cur_depth_range_samples (:519-536) -> (D, H/scale, W/scale)."""
    H, W = img_hw
    dmin, dmax = float(depth_values_1d[0]), float(depth_values_1d[-1])
    interval = (dmax - dmin) / depth_values_1d.shape[0]
    if prev_depth is None:
        lo, hi = F32(depth_values_1d[0]), F32(depth_values_1d[-1])
        if inverse_depth:
            from svs_oracle import linspace32
            t = linspace32(0.0, 1.0, ndepth)
            samples = (F32(1.0) / (F32(1.0) / lo * (F32(1.0) - t) + F32(1.0) / hi * t)).astype(F32)
        else:
            new_int = (hi - lo) / F32(ndepth - 1)
            samples = (lo + np.arange(ndepth, dtype=F32) * new_int).astype(F32)
        vol = np.broadcast_to(samples.reshape(-1, 1, 1), (ndepth, H, W)).astype(F32)
    else:
        cur = resize_linear(np.asarray(prev_depth, F32), (H, W))
        pix = F32(int_r * interval)
        cmin = (cur - F32(ndepth / 2) * pix).astype(F32)
        cmax = (cur + F32(ndepth / 2) * pix).astype(F32)
        new_int = ((cmax - cmin) / F32(ndepth - 1)).astype(F32)
        vol = (cmin[None] + np.arange(ndepth, dtype=F32).reshape(-1, 1, 1) * new_int[None]).astype(F32)
    return resize_linear(vol, (ndepth, H // int(stage_scale), W // int(stage_scale)))


def cost_reg_net_torch(params, x):
    """CostRegNet.forward (:441-472) as a plain torch float32 reference (floating-point kernel).
    params: dict from synth.make_costreg_params; x (C,D,H,W) numpy -> (D,H,W) numpy."""
    import torch
    import torch.nn.functional as Fn
    T = lambda k: torch.from_numpy(np.asarray(params[k]))

    def bn_relu(y, name):
        y = Fn.batch_norm(y, T(f"{name}.bn.running_mean"), T(f"{name}.bn.running_var"), T(f"{name}.bn.weight"),
                          T(f"{name}.bn.bias"), training=False, eps=1e-5)
        return Fn.relu(y)

    def conv(y, name, stride):
        return bn_relu(Fn.conv3d(y, T(f"{name}.conv.weight"), stride=stride, padding=1), name)

    def deconv(y, name):
        return bn_relu(Fn.conv_transpose3d(y, T(f"{name}.conv.weight"), stride=2, padding=1, output_padding=1), name)

    with torch.no_grad():
        x = torch.from_numpy(np.asarray(x, F32))[None]
        c0 = conv(x, "conv0", 1)
        c2 = conv(conv(c0, "conv1", 2), "conv2", 1)
        c4 = conv(conv(c2, "conv3", 2), "conv4", 1)
        y = conv(conv(c4, "conv5", 2), "conv6", 1)
        y = c4 + deconv(y, "conv7")
        y = c2 + deconv(y, "conv9")
        y = c0 + deconv(y, "conv11")
        y = Fn.conv3d(y, T("prob.weight"), padding=1)
    return y[0, 0].numpy()


def depthnet_forward(features, projs, depth_values, costreg_params):
    """DepthNet.forward :601-663 for one stage."""
    var = variance_volume(features, projs, depth_values)
    reg = cost_reg_net_torch(costreg_params, var)
    prob, depth, conf, idx = depthnet_tail(reg, depth_values)
    return dict(variance=var, reg=reg, prob_volume=prob, depth=depth, photometric_confidence=conf, depth_index=idx,
                depth_values=depth_values)


def feature_net_torch(params, img):
    """FeatureNet, arch_mode 'fpn' (models/CasMVSNet.py:338-439) with plain torch float32 functional ops on the CPU.
    params: state-dict-named arrays (synth.make_featurenet_params), img (3,H,W) -> {'stage1','stage2','stage3'} arrays."""
    import torch
    import torch.nn.functional as F
    P = {k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}

    def block(name, x, stride, pad):
        x = F.conv2d(x, P[f"{name}.conv.weight"], None, stride=stride, padding=pad)
        x = F.batch_norm(x, P[f"{name}.bn.running_mean"], P[f"{name}.bn.running_var"], P[f"{name}.bn.weight"], P[f"{name}.bn.bias"],
                         training=False, eps=1e-5)
        return F.relu(x)

    x = torch.from_numpy(np.asarray(img, np.float32))[None]
    c0 = block("conv0.1", block("conv0.0", x, 1, 1), 1, 1)
    c1 = block("conv1.2", block("conv1.1", block("conv1.0", c0, 2, 2), 1, 1), 1, 1)
    c2 = block("conv2.2", block("conv2.1", block("conv2.0", c1, 2, 2), 1, 1), 1, 1)
    out = {"stage1": F.conv2d(c2, P["out1.weight"])}
    f = F.interpolate(c2, scale_factor=2, mode="nearest") + F.conv2d(c1, P["inner1.weight"], P["inner1.bias"])
    out["stage2"] = F.conv2d(f, P["out2.weight"], padding=1)
    f = F.interpolate(f, scale_factor=2, mode="nearest") + F.conv2d(c0, P["inner2.weight"], P["inner2.bias"])
    out["stage3"] = F.conv2d(f, P["out3.weight"], padding=1)
    return {k: v[0].numpy() for k, v in out.items()}
