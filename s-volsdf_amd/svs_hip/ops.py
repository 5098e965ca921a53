"""torch-tensor wrappers over the C-ABI (device pointers + the current HIP stream).

PyTorch is used only for device memory and streams; every computation is a hand-written HIP kernel.
Shapes/semantics follow the reference functions named in each docstring (paths relative to the
reference root).
"""
import ctypes
import os

import torch

from . import lib as _lib


def _ptr(t):
    """device address of a contiguous device tensor as a plain int (every entry point declares its argument types, so ctypes
    converts it; ~400 of these per train step: no c_void_p object each)"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current HIP stream of the current device as a hipStream_t (torch.cuda.current_stream() builds a Stream object and
    resolves the device three times: 10 us a call, ~30 calls per train step)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _f32(t):
    if t.dtype is torch.float32 and not t.requires_grad and t.is_contiguous():
        return t                      # (the common case: ~150 calls per train step)
    return t.detach().to(dtype=torch.float32).contiguous()


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


# ---------------------------------------------------------------------------------------------------
def rays_from_uv(uv, pose, intrinsics):
    """rend_util.get_camera_params (volsdf/utils/rend_util.py:60-95) + the depth_scale of network.py:216-217.

    uv (R,2), pose (4,4), intrinsics (4,4) -> ray_dirs (R,3), cam_loc (3,), depth_scale (R,1)
    """
    L = _lib.load()
    uv, pose, K = _f32(uv), _f32(pose), _f32(intrinsics)
    R = uv.shape[0]
    dirs = torch.empty(R, 3, device=uv.device)
    cam = torch.empty(3, device=uv.device)
    ds = torch.empty(R, 1, device=uv.device)
    _lib.check(L.svs_rays_from_uv(_ptr(uv), _ptr(pose), _ptr(K), R, _ptr(dirs), _ptr(cam), _ptr(ds), _stream()),
               "svs_rays_from_uv")
    return dirs, cam, ds


# MFMA precision of the MLP kernels (include/svolsdf_hip.h: SVS_MMA_F32 / SVS_MMA_F16X2 / SVS_MMA_F16X2_HALF)
F32, F16X2, F16X2_HALF = 0, 1, 2
_PRECISIONS = {"f32": F32, "f16x2": F16X2, "f16x2_half": F16X2_HALF}


def is_h2(precision):
    """the fp16x2 kernels (either format of the backward's gradient-only blocks)"""
    return precision in (F16X2, F16X2_HALF)


def default_precision():
    """SVS_MLP_PRECISION: f16x2 (default: two-piece fp16 operands on the 16-bit matrix cores, forward AND backward in the
    float32 accuracy class), f16x2_half (the same kernels with the backward's gradient-only blocks stored as one fp16 piece:
    a faster, mixed-precision training step whose parameter gradients are 2e-4 ... 8e-4 off), f32 (float32 MFMA)."""
    import os
    v = os.environ.get("SVS_MLP_PRECISION", "f16x2").lower()
    if v not in _PRECISIONS:
        raise ValueError(f"SVS_MLP_PRECISION must be one of {sorted(_PRECISIONS)}, not {v!r}")
    return _PRECISIONS[v]


class PackedMlp:
    """Packed weight streams of one ImplicitNetwork / RenderingNetwork pair (rebuilt after every optimiser step)."""

    def __init__(self, device, precision=None, sdf_kernel=None):
        L = _lib.load()
        self.device = device
        self.precision = default_precision() if precision is None else int(precision)
        self.sdf_stream = torch.empty(L.svs_stream_bytes(1) // 4, device=device)
        self.rgb_stream = torch.empty(L.svs_stream_bytes(3) // 4, device=device)
        # sdf_kernel / SVS_SDF_KERNEL: which kernel runs the sampler's sdf-only evaluations -- "32" (default: one wave per SIMD,
        # 32 points per wave), "16" (two waves per SIMD, 16-point waves, csrc/svs_mlp_w16.hip: its own encoding of the forward
        # stream) or "pair" (two waves per SIMD on the same 32 points, csrc/svs_mlp_h2p.hip).  The two-wave variants are
        # experiments kept for A/B runs: same speed / slower
        import os
        self.sdf_kernel = (sdf_kernel or os.environ.get("SVS_SDF_KERNEL", "32")) if is_h2(self.precision) else "32"
        if self.sdf_kernel not in ("32", "16", "pair"):
            raise ValueError(f"SVS_SDF_KERNEL must be 32, 16 or pair, not {self.sdf_kernel!r}")
        if self.sdf_kernel != "32" and not hasattr(L, "svs_sdf_vals_pair"):
            raise _lib.SvsError(f"SVS_SDF_KERNEL={self.sdf_kernel}: the experimental two-waves-per-SIMD kernels are not in this "
                                "library; rebuild with SVS_BUILD_EXPERIMENTS=1 python s-volsdf_amd/build.py --force")
        self.w16 = self.sdf_kernel == "16"
        self.sdf_stream16 = torch.empty(L.svs_stream_bytes(9) // 4, device=device) if self.w16 else None
        self._ws16 = torch.empty(L.svs_pack_workspace_bytes() // 4, device=device) if self.w16 else None
        # one row-norm workspace per stream: the fused train step packs the two on different HIP streams at the same time
        self._ws = torch.empty(L.svs_pack_workspace_bytes() // 4, device=device)
        self._ws_rgb = torch.empty(L.svs_pack_workspace_bytes() // 4, device=device)

    def pack_sdf(self, weight_v, weight_g, bias):
        """weight_v/bias: 9 tensors; weight_g: 9 tensors or None (no weight-norm). network.py:64-65."""
        L = _lib.load()
        v = [_f32(t) for t in weight_v]
        b = [_f32(t) for t in bias]
        g = [_f32(t) for t in weight_g] if weight_g is not None else None
        self._keep = (v, b, g)
        _lib.check(L.svs_pack_stream(1, self.precision, _ptr_array(v), _ptr_array(g) if g else None, _ptr_array(b),
                                     _ptr(self._ws), _ptr(self.sdf_stream), _stream()), "svs_pack_stream(sdf)")
        if self.w16:
            _lib.check(L.svs_pack_stream(9, self.precision, _ptr_array(v), _ptr_array(g) if g else None, _ptr_array(b),
                                         _ptr(self._ws16), _ptr(self.sdf_stream16), _stream()), "svs_pack_stream(sdf w16)")

    def pack_rgb(self, weight_v, weight_g, bias):
        L = _lib.load()
        v = [_f32(t) for t in weight_v]
        b = [_f32(t) for t in bias]
        g = [_f32(t) for t in weight_g] if weight_g is not None else None
        self._keep_rgb = (v, b, g)
        _lib.check(L.svs_pack_stream(3, self.precision, _ptr_array(v), _ptr_array(g) if g else None, _ptr_array(b),
                                     _ptr(self._ws_rgb), _ptr(self.rgb_stream), _stream()), "svs_pack_stream(rgb)")


class PointSource:
    """Sample positions of one MLP launch: ray samples cam + z * dir (R*S of them) followed by explicit points.
    Either part may be absent."""

    def __init__(self, points=None, cam=None, dirs=None, z=None):
        self.points = _f32(points) if points is not None else None
        self.n_points = self.points.shape[0] if points is not None else 0
        if z is not None:
            self.cam, self.dirs, self.z = _f32(cam), _f32(dirs), _f32(z)
            self.n_rays, self.S = self.z.shape
            self.cam_stride = 0 if self.cam.numel() == 3 else 3
        else:
            self.cam = self.dirs = self.z = None
            self.n_rays, self.S, self.cam_stride = 0, 0, 0
        self.n_ray_points = self.n_rays * self.S
        self.n = self.n_ray_points + self.n_points

    def args(self):
        return (_ptr(self.points), self.n_points, _ptr(self.cam), self.cam_stride, _ptr(self.dirs), _ptr(self.z),
                self.S, self.n_rays)

    @property
    def device(self):
        return (self.points if self.points is not None else self.z).device


def sdf_vals(packed, src, sphere_radius, sphere_scale, out=None, gate=None, clamp_n=-1, gate_points=0, gate_stride=0):
    """ImplicitNetwork.get_sdf_vals (network.py:125-131) -> (P,1).  gate: optional address of device int flags, one
    per group of gate_points points (0: one group), gate_stride ints apart: groups whose flag is 0 are skipped."""
    L = _lib.load()
    sdf = out if out is not None else torch.empty(src.n, 1, device=src.device)
    if getattr(packed, "sdf_kernel", "32") == "pair":
        _lib.check(L.svs_sdf_vals_pair(*src.args(), _ptr(packed.sdf_stream), float(sphere_radius), float(sphere_scale),
                                       int(clamp_n), _ptr(sdf), ctypes.c_void_p(gate) if gate else None, int(gate_points),
                                       int(gate_stride), _stream()), "svs_sdf_vals_pair")
        return sdf
    if getattr(packed, "w16", False):
        _lib.check(L.svs_sdf_vals16(*src.args(), _ptr(packed.sdf_stream16), float(sphere_radius), float(sphere_scale),
                                    int(clamp_n), _ptr(sdf), ctypes.c_void_p(gate) if gate else None, int(gate_points),
                                    int(gate_stride), _stream()), "svs_sdf_vals16")
        return sdf
    _lib.check(L.svs_sdf_vals(*src.args(), _ptr(packed.sdf_stream), packed.precision, float(sphere_radius), float(sphere_scale),
                              int(clamp_n), _ptr(sdf), ctypes.c_void_p(gate) if gate else None, int(gate_points),
                              int(gate_stride), _stream()), "svs_sdf_vals")
    return sdf


def sdf_outputs(packed, src, sphere_radius, sphere_scale, want_feature_rows=False, clamp_n=-1, keep=None):
    """ImplicitNetwork.get_outputs (network.py:105-123): sdf (P,1), d sdf/dx (P,3), feature tiles, hbuf.

    The sphere clamp (network.py:110-112) applies to the first clamp_n points (-1: all); the others, and all
    of them when sphere_radius <= 0, differentiate the raw output (ImplicitNetwork.gradient, :90-103).
    """
    L = _lib.load()
    dev = src.device
    sdf = torch.empty(src.n, 1, device=dev)
    grad = torch.empty(src.n, 3, device=dev)
    feat = torch.empty(L.svs_feat_tiles_bytes(src.n) // 4, device=dev)
    hbuf = torch.empty(L.svs_sdf_hbuf_bytes(src.n) // 4, device=dev)
    gbuf = mask = None
    if keep is not None:          # training: keep the gradient-pass state for the backward kernels
        gbuf = torch.empty(L.svs_sdf_gbuf_bytes(src.n) // 4, device=dev)      # 8 blocks + their records (max |ghat_l| per point)
        mask = torch.empty(src.n, dtype=torch.uint8, device=dev)
        keep.update(hbuf=hbuf, gbuf=gbuf, clamp_mask=mask, src=src)
    _lib.check(L.svs_sdf_outputs(*src.args(), _ptr(packed.sdf_stream), packed.precision, float(sphere_radius), float(sphere_scale),
                                 int(clamp_n), _ptr(sdf), _ptr(grad), _ptr(feat), _ptr(hbuf), _ptr(gbuf), _ptr(mask),
                                 _stream()), "svs_sdf_outputs")
    rows = None
    if want_feature_rows:
        rows = torch.empty(src.n, 256, device=dev)
        _lib.check(L.svs_tiles_to_rows(_ptr(feat), src.n, packed.precision, _ptr(rows), _stream()), "svs_tiles_to_rows")
    return sdf, grad, feat, hbuf, rows


def rgb_eval(packed, src, normals, view_dirs, feat_tiles, keep=None):
    """RenderingNetwork.forward, mode 'idr' (network.py:170-190) -> (P,3).
    view_dirs: (R,3) with src in ray mode (one direction per ray), or (P,3)."""
    L = _lib.load()
    normals, view_dirs = _f32(normals), _f32(view_dirs)
    view_S = src.S if (src.n_points == 0 and view_dirs.shape[0] * src.S == src.n) else 0
    if view_S == 0:
        assert view_dirs.shape[0] == src.n
    rgb = torch.empty(src.n, 3, device=src.device)
    rbuf = None
    if keep is not None:
        rbuf = torch.empty(L.svs_rgb_rbuf_bytes(src.n) // 4, device=src.device)
        keep.update(rbuf=rbuf, feat_tiles=feat_tiles, rgb=rgb)
    _lib.check(L.svs_rgb_eval(*src.args(), _ptr(normals), _ptr(view_dirs), view_S, _ptr(feat_tiles),
                              _ptr(packed.rgb_stream), packed.precision, _ptr(rgb), _ptr(rbuf), _stream()), "svs_rgb_eval")
    return rgb


def composite(z, sdf, rgb, depth_scale, beta_param, beta_min, normals=None):
    """VolSDFNetwork.volume_rendering + reductions (network.py:281-295, :237-256, :270-276)."""
    L = _lib.load()
    z = _f32(z)
    R, S = z.shape
    dev = z.device
    sdf, rgb, depth_scale = _f32(sdf), _f32(rgb), _f32(depth_scale)
    beta_param = _f32(beta_param).reshape(1)
    weights = torch.empty(R, S, device=dev)
    rgb_values = torch.empty(R, 3, device=dev)
    depth_values = torch.empty(R, 1, device=dev)
    depth_vals = torch.empty(R, S, device=dev)
    normal_map = torch.empty(R, 3, device=dev) if normals is not None else None
    nrm = _f32(normals) if normals is not None else None
    _lib.check(L.svs_composite(R, S, _ptr(z), _ptr(sdf), _ptr(rgb), _ptr(nrm), _ptr(depth_scale), _ptr(beta_param),
                               float(beta_min), _ptr(weights), _ptr(rgb_values), _ptr(depth_values), _ptr(depth_vals),
                               _ptr(normal_map), _stream()), "svs_composite")
    return dict(weights=weights, rgb_values=rgb_values, depth_values=depth_values, depth_vals=depth_vals,
                normal_map=normal_map)


# ---------------------------------------------------------------------------------------------------
# inverted-sphere background model (volsdf/model/network_bg.py): kernels of svs_bg_h2.hip (fp16x2) / svs_bg_f32.hip (float32 MFMA)
# ---------------------------------------------------------------------------------------------------
class PackedBg:
    """Packed weight streams of bg_implicit_network / bg_rendering_network (no weight-norm)."""

    def __init__(self, device, precision=None):
        self.precision = default_precision() if precision is None else int(precision)
        L = _lib.load()
        self.device = device
        self.sdf_stream = torch.empty(L.svs_stream_bytes(5) // 4, device=device)
        self.rgb_stream = torch.empty(L.svs_stream_bytes(7) // 4, device=device)
        self._ws = torch.empty(L.svs_pack_workspace_bytes() // 4, device=device)

    def pack(self, sdf_wb, rgb_wb):
        """sdf_wb / rgb_wb: (weights, biases) lists of 9 / 2 tensors."""
        L = _lib.load()
        self._keep = []
        for which, (w, b), stream in ((5, sdf_wb, self.sdf_stream), (7, rgb_wb, self.rgb_stream)):
            w, b = [_f32(t) for t in w], [_f32(t) for t in b]
            self._keep += [w, b]
            _lib.check(L.svs_pack_stream(which, self.precision, _ptr_array(w), None, _ptr_array(b), _ptr(self._ws), _ptr(stream),
                                         _stream()), "svs_pack_stream(bg)")


def bg_points(cam, dirs, n_bg, radius, jitter=None):
    """Inverse-sphere samples (ray_sampler.py:215-216, flipped as network_bg.py:82) and depth2pts_outside
    (network_bg.py:182-214) -> z_bg (R,n_bg) descending, pts (R*n_bg,4), depth_real (R,n_bg)."""
    L = _lib.load()
    cam, dirs = _f32(cam), _f32(dirs)
    R, dev = dirs.shape[0], dirs.device
    z_bg = torch.empty(R, n_bg, device=dev)
    pts = torch.empty(R * n_bg, 4, device=dev)
    depth = torch.empty(R, n_bg, device=dev)
    jitter = _f32(jitter) if jitter is not None else None
    _lib.check(L.svs_bg_points(_ptr(cam), 0 if cam.numel() == 3 else 3, _ptr(dirs), R, n_bg, _ptr(jitter), float(radius),
                               _ptr(z_bg), _ptr(pts), _ptr(depth), _stream()), "svs_bg_points")
    return z_bg, pts, depth


def bg_sdf_eval(packed, pts, keep=None):
    """bg_implicit_network (network_bg.py:85-88): pts (P,4) -> out0 (P,1) = output[:, :1], feature tiles."""
    L = _lib.load()
    pts = _f32(pts)
    P, dev = pts.shape[0], pts.device
    out0 = torch.empty(P, 1, device=dev)
    feat = torch.empty(L.svs_feat_tiles_bytes(P) // 4, device=dev)
    hbuf = ghat7 = pebuf = None
    if keep is not None:
        hbuf = torch.empty(L.svs_sdf_hbuf_bytes(P) // 4, device=dev)
        ghat7 = torch.empty(L.svs_block_bytes(P, 1) // 4, device=dev)
        pebuf = torch.empty(L.svs_block_bytes(P, 1) // 4, device=dev)
        keep.update(bg_hbuf=hbuf, bg_ghat7=ghat7, bg_pebuf=pebuf, bg_pts=pts)
    _lib.check(L.svs_bg_sdf_eval(_ptr(pts), P, _ptr(packed.sdf_stream), packed.precision, _ptr(out0), _ptr(feat), _ptr(hbuf),
                                 _ptr(ghat7), _ptr(pebuf), _stream()), "svs_bg_sdf_eval")
    return out0, feat


def bg_rgb_eval(packed, view_dirs, n_bg, feat_tiles, n_points, keep=None):
    """bg_rendering_network, mode 'nerf' (network_bg.py:91-93): one view direction per ray -> rgb (P,3)."""
    L = _lib.load()
    view_dirs = _f32(view_dirs)
    dev = view_dirs.device
    rgb = torch.empty(n_points, 3, device=dev)
    rbuf = None
    if keep is not None:
        rbuf = torch.empty(L.svs_bg_rbuf_bytes(n_points) // 4, device=dev)
        keep.update(bg_rbuf=rbuf, bg_feat=feat_tiles, bg_rgb=rgb)
    _lib.check(L.svs_bg_rgb_eval(n_points, _ptr(view_dirs), n_bg, _ptr(feat_tiles), _ptr(packed.rgb_stream), packed.precision,
                                 _ptr(rgb), _ptr(rbuf), _stream()), "svs_bg_rgb_eval")
    return rgb


def composite_bg(z, z_max, sdf, rgb, depth_scale, beta_param, beta_min, z_bg, bg_out0, bg_rgb, bg_depth, normals=None):
    """volume_rendering / bg_volume_rendering and the composition of VolSDFNetworkBG.forward (network_bg.py:76-125)."""
    L = _lib.load()
    z = _f32(z)
    R, S = z.shape
    Nb = z_bg.shape[1]
    dev = z.device
    beta_param = _f32(beta_param).reshape(1)
    f = lambda *s: torch.empty(*s, device=dev)
    weights, bg_trans, bg_w = f(R, S), f(R), f(R, Nb)
    rgb_values, depth_values, depth_all, depth_vals = f(R, 3), f(R, 1), f(R, 1), f(R, S)
    normal_map = f(R, 3) if normals is not None else None
    _lib.check(L.svs_composite_bg(R, S, Nb, _ptr(z), _ptr(_f32(z_max)), _ptr(_f32(sdf)), _ptr(_f32(rgb)),
                                  _ptr(_f32(normals)) if normals is not None else None, _ptr(_f32(depth_scale)),
                                  _ptr(beta_param), float(beta_min), _ptr(_f32(z_bg)), _ptr(_f32(bg_out0)), _ptr(_f32(bg_rgb)),
                                  _ptr(_f32(bg_depth)), _ptr(weights), _ptr(bg_trans), _ptr(bg_w), _ptr(rgb_values),
                                  _ptr(depth_values), _ptr(depth_all), _ptr(depth_vals), _ptr(normal_map), _stream()),
               "svs_composite_bg")
    return dict(weights=weights, bg_transmittance=bg_trans, bg_weights=bg_w, rgb_values=rgb_values,
                depth_values=depth_values, depth_values_all=depth_all, depth_vals=depth_vals, normal_map=normal_map)


_STAGE_IN = os.environ.get("SVS_STAGE_IN", "1") != "0"      # A/B switch: 0 = hipMemcpyAsync (tensor.copy_)


def stage_in(dev_tensor, pinned_tensor):
    """pinned host tensor -> device tensor of the same byte size, by a kernel on the current stream (svs_stage_in); falls back to
    a non-blocking copy_ for buffers the kernel does not take (alignment, sizes that are not a multiple of 4 bytes)."""
    nbytes = pinned_tensor.numel() * pinned_tensor.element_size()
    ok = (_STAGE_IN and pinned_tensor.is_pinned() and pinned_tensor.is_contiguous() and dev_tensor.is_contiguous()
          and nbytes == dev_tensor.numel() * dev_tensor.element_size() and nbytes % 4 == 0
          and pinned_tensor.data_ptr() % 16 == 0 and dev_tensor.data_ptr() % 16 == 0 and nbytes > 0)
    if not ok:
        dev_tensor.copy_(pinned_tensor, non_blocking=True)
        return dev_tensor
    _lib.check(_lib.load().svs_stage_in(pinned_tensor.data_ptr(), dev_tensor.data_ptr(), nbytes, _stream()), "svs_stage_in")
    return dev_tensor


def split_last(z):
    """(R, n) -> (z[:, :-1] dense, z[:, -1]) in one launch (network_bg.py:60-62)."""
    z = _f32(z)
    R, n = z.shape
    head, last = torch.empty(R, n - 1, device=z.device), torch.empty(R, device=z.device)
    _lib.check(_lib.load().svs_split_last(_ptr(z), R, n, _ptr(head), _ptr(last), _stream()), "svs_split_last")
    return head, last


def eikonal_points(uniform_points, cam_loc, z_eik, ray_dirs):
    """network.py:258-266: (2R,3) = [the uniform draws ; cam + z_eik * dirs] in one launch."""
    R = ray_dirs.shape[0]
    out = torch.empty(2 * R, 3, device=ray_dirs.device)
    _lib.check(_lib.load().svs_eikonal_points(_ptr(_f32(uniform_points)), _ptr(_f32(cam_loc)), _ptr(_f32(z_eik)),
                                              _ptr(_f32(ray_dirs)), R, _ptr(out), _stream()), "svs_eikonal_points")
    return out


def composite_bg_bwd(z, z_max, sdf, rgb, depth_scale, beta_param, beta_min, z_bg, bg_out0, bg_rgb, d_rgb_values,
                     d_weights=None, d_depth_values=None, d_depth_values_all=None, bg_depth=None, d_sdf_out=None,
                     d_beta_out=None):
    """backward of composite_bg -> d_sdf (R*S,1), d_rgb (R*S,3), d_bg_out0 (R*Nb,1), d_bg_rgb (R*Nb,3), d_beta (1,).
    d_depth_values: gradient of the foreground depth; d_depth_values_all (with bg_depth (R,Nb)): gradient of
    depth_values_all, the fg + bg depth the loss's sparsity term reads (network_bg.py:105-110, loss.py:72-73).
    d_sdf_out / d_beta_out: optional contiguous float32 device tensors the two results are written into (as composite_bwd)."""
    L = _lib.load()
    z = _f32(z)
    R, S = z.shape
    Nb = z_bg.shape[1]
    dev = z.device
    f = lambda *s: torch.empty(*s, device=dev)
    for t, shape in ((d_sdf_out, (R * S, 1)), (d_beta_out, (1,))):
        if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev):
            raise ValueError(f"output tensor must be contiguous float32 {shape} on {dev}")
    d_sdf = d_sdf_out if d_sdf_out is not None else f(R * S, 1)
    d_rgb, d_bo, d_brgb = f(R * S, 3), f(R * Nb, 1), f(R * Nb, 3)
    d_beta_ray = f(R)
    d_beta = d_beta_out if d_beta_out is not None else f(1)
    opt = lambda t: _ptr(_f32(t)) if t is not None else None
    _lib.check(L.svs_composite_bg_bwd(R, S, Nb, _ptr(z), _ptr(_f32(z_max)), _ptr(_f32(sdf)), _ptr(_f32(rgb)),
                                      _ptr(_f32(depth_scale)), _ptr(_f32(beta_param).reshape(1)), float(beta_min),
                                      _ptr(_f32(z_bg)), _ptr(_f32(bg_out0)), _ptr(_f32(bg_rgb)), _ptr(_f32(d_rgb_values)),
                                      opt(d_weights), opt(d_depth_values), opt(d_depth_values_all), opt(bg_depth),
                                      _ptr(d_sdf), _ptr(d_rgb), _ptr(d_bo), _ptr(d_brgb),
                                      _ptr(d_beta_ray), _ptr(d_beta), _stream()), "svs_composite_bg_bwd")
    return d_sdf, d_rgb, d_bo, d_brgb, d_beta


def composite_bwd(z, sdf, rgb, depth_scale, beta_param, beta_min, d_rgb_values, d_weights=None, d_depth_values=None,
                  d_sdf_out=None, d_beta_out=None):
    """Reverse pass of `composite`: -> d_sdf (R*S,1), d_rgb (R*S,3), d_beta_param (1,).  d_sdf_out / d_beta_out: optional
    preallocated contiguous float32 tensors of those shapes to write into (the fused train step hands in views of its
    persistent buffers, which saves a fill and two copies per ray group)."""
    L = _lib.load()
    z = _f32(z)
    R, S = z.shape
    dev = z.device
    sdf, rgb, depth_scale = _f32(sdf), _f32(rgb), _f32(depth_scale)
    beta_param = _f32(beta_param).reshape(1)
    g_rgb = _f32(d_rgb_values)
    g_w = _f32(d_weights) if d_weights is not None else None
    g_d = _f32(d_depth_values).reshape(-1) if d_depth_values is not None else None
    for t, shape in ((d_sdf_out, (R * S, 1)), (d_beta_out, (1,))):
        if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError("composite_bwd: output tensor must be contiguous float32 of shape %s" % (shape,))
    d_sdf = d_sdf_out if d_sdf_out is not None else torch.empty(R * S, 1, device=dev)
    d_rgb = torch.empty(R * S, 3, device=dev)
    ws = torch.empty(R, device=dev)
    d_beta = d_beta_out if d_beta_out is not None else torch.empty(1, device=dev)
    _lib.check(L.svs_composite_bwd(R, S, _ptr(z), _ptr(sdf), _ptr(rgb), _ptr(depth_scale), _ptr(beta_param),
                                   float(beta_min), _ptr(g_rgb), _ptr(g_w), _ptr(g_d), _ptr(d_sdf), _ptr(d_rgb),
                                   _ptr(ws), _ptr(d_beta), _stream()), "svs_composite_bwd")
    return d_sdf, d_rgb, d_beta


_HOST_COPIES = {}


def _host_f32(t):
    """float32 host copy of a small camera matrix; copies of DEVICE tensors are cached per (address, version) so that a
    step launches without a device-to-host synchronisation (and can be captured in a hipGraph)."""
    if not (torch.is_tensor(t) and t.is_cuda):
        return torch.as_tensor(t, dtype=torch.float32)
    key = (t.data_ptr(), t._version, tuple(t.shape))
    hit = _HOST_COPIES.get(key)
    if hit is None:
        if len(_HOST_COPIES) > 256:
            _HOST_COPIES.clear()
        hit = _HOST_COPIES[key] = t.detach().to(dtype=torch.float32).cpu()
    return hit


_LOOKUP_CONSTS = {}


def clear_lookup_caches():
    """Drop the cached per-view constants of cost_lookup and the host copies of camera matrices -- with them the references
    that keep a stage's probability volumes and depth ranges on the GPU (called when a new MVS stage hands over its volumes:
    VolOpt.get_mvs_input, costvol.clear_caches)."""
    _LOOKUP_CONSTS.clear()
    _HOST_COPIES.clear()


def cost_lookup(views, same_view, img_res, *, xyz=None, cam=None, dirs=None, z=None, inverse_depth=False,
                same_view_dev=None):
    """VolOpt.cost_mapping (volsdf/vsdf.py:382-452).

    views: list of dict(K (4,4) host/any tensor, c2w (4,4), cost (D,H,W) device, z_mvs (D,H,W) device or
    (z_near, z_far) (H,W) device).  Points: xyz (R,S,3), or cam (3,), dirs (R,3), z (R,S).
    same_view_dev: optional device int32 tensor overriding same_view at run time (captured launch sequences).
    Returns pj (R,S), pi (R,S), valid (R,S) bool.
    """
    L = _lib.load()
    if xyz is not None:
        xyz = _f32(xyz)
        R, S = xyz.shape[:2]
        dev = xyz.device
    else:
        cam, dirs, z = _f32(cam).reshape(3), _f32(dirs), _f32(z)
        R, S = z.shape
        dev = z.device
    V = len(views)
    # the per-view constants (camera block, volume pointers and sizes) are the same every step of a stage: cached by the
    # identity / version / address of what they were built from (51 scalar reads of host tensors per call otherwise: 0.1 ms)
    ver = lambda t: (id(t), getattr(t, "_version", 0))
    key = tuple((ver(v["K"]), ver(v["c2w"]), v["cost"].data_ptr(), v["cost"].shape,
                 (v["z_mvs"].data_ptr(), v["z_mvs"].shape) if "z_mvs" in v else (v["z_near"].data_ptr(), v["z_far"].data_ptr()))
                for v in views)
    hit = _LOOKUP_CONSTS.get(key)
    if hit is None:
        vp = (ctypes.c_float * (17 * V))()
        dims = (ctypes.c_int * (3 * V))()
        keep, cost_p, near_p, far_p = [], [], [], []
        copied = False
        for j, v in enumerate(views):
            K, c2w = _host_f32(v["K"]), _host_f32(v["c2w"])
            vals = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1]] + [c2w[i, k] for i in range(3) for k in range(4)]
            for k, x in enumerate(vals):
                vp[17 * j + k] = float(x)
            cost = _f32(v["cost"]).reshape(v["cost"].shape[-3:])
            if "z_mvs" in v:
                zm = v["z_mvs"].reshape(v["z_mvs"].shape[-3:])
                zn, zf = _f32(zm[0]), _f32(zm[-1])
                copied = copied or zn.data_ptr() != zm[0].data_ptr() or zf.data_ptr() != zm[-1].data_ptr()
            else:
                zn, zf = _f32(v["z_near"]), _f32(v["z_far"])
                copied = copied or zn.data_ptr() != v["z_near"].data_ptr() or zf.data_ptr() != v["z_far"].data_ptr()
            # kept alive: the key holds the addresses of the ORIGINAL tensors (v["cost"], z range) -- not only those of the
            # float32 / contiguous copies _f32 may have made -- so neither can be recycled while the entry lives
            keep += [cost, zn, zf, v["K"], v["c2w"], v["cost"], v.get("z_mvs"), v.get("z_near"), v.get("z_far")]
            copied = copied or cost.data_ptr() != v["cost"].data_ptr()
            cost_p.append(cost); near_p.append(zn); far_p.append(zf)
            dims[3 * j], dims[3 * j + 1], dims[3 * j + 2] = cost.shape
        if len(_LOOKUP_CONSTS) >= 4:
            _LOOKUP_CONSTS.clear()
        hit = (vp, dims, _ptr_array(cost_p), _ptr_array(near_p), _ptr_array(far_p), keep)
        if not copied:
            # (a volume or a depth-range plane _f32 had to convert -- not float32 or not contiguous -- is looked up from a
            # private copy: the original may change without its address changing, so such a call is not cached)
            _LOOKUP_CONSTS[key] = hit
    vp, dims, cost_arr, near_arr, far_arr, _ = hit
    pj = torch.empty(R, S, device=dev)
    pi = torch.empty(R, S, device=dev)
    valid = torch.empty(R, S, dtype=torch.uint8, device=dev)
    _lib.check(L.svs_cost_lookup(_ptr(xyz), _ptr(cam), _ptr(dirs), _ptr(z), S, R * S, V, int(same_view),
                                 int(bool(inverse_depth)), float(img_res[1]), float(img_res[0]), vp,
                                 cost_arr, near_arr, far_arr, dims, _ptr(pj), _ptr(pi),
                                 _ptr(valid), _ptr(same_view_dev), _stream()), "svs_cost_lookup")
    return pj, pi, valid.view(torch.bool)              # the kernel writes 0 / 1: reinterpreted, not converted


def loss_fwd_bwd(rgb_values, rgb_target, weights, depth_values, grad_theta=None, pi=None, pj=None, *, rgb_weight=1.0,
                 eikonal_weight=0.1, mvs_weight=0.0, sparse_weight=0.0, gce=1.0, confi=0.0, annealed=False,
                 anneal_sparse=0.0, norm=None, anneal_dev=None, grad_theta_out=None):
    """VolSDFLoss.forward (volsdf/model/loss.py:80-114) and d(total)/d(model outputs) in one launch.
    norm = (R_total, n_eik_total): denominators of the means when the batch is processed in ray groups.
    anneal_dev: optional device float32[2] = {annealed, anneal_sparse} read at run time (captured launch sequences).
    Returns (losses[5] = rgb, eikonal, mvs, sparse, total; dict of gradients)."""
    L = _lib.load()
    rgb_values, rgb_target = _f32(rgb_values).reshape(-1, 3), _f32(rgb_target).reshape(-1, 3)
    weights, depth_values = _f32(weights), _f32(depth_values).reshape(-1)
    R, S = weights.shape
    dev = weights.device
    gt = _f32(grad_theta) if grad_theta is not None else None
    n_eik = gt.shape[0] if gt is not None else 0
    pi_, pj_ = (_f32(pi), _f32(pj)) if pi is not None else (None, None)
    losses = torch.empty(5, device=dev)
    d_rgb = torch.empty(R, 3, device=dev)
    # grad_theta_out: optional (n_eik,3) float32 destination of d loss / d grad_theta (the backward's own buffer: train.MlpBackward)
    d_gt = None
    if n_eik:
        ok = (grad_theta_out is not None and grad_theta_out.is_cuda and grad_theta_out.dtype == torch.float32
              and tuple(grad_theta_out.shape) == (n_eik, 3) and grad_theta_out.is_contiguous())
        d_gt = grad_theta_out if ok else torch.empty(n_eik, 3, device=dev)
    d_w = torch.empty(R, S, device=dev)
    d_dep = torch.empty(R, 1, device=dev)
    ws = torch.empty(L.svs_loss_workspace_bytes(R, n_eik) // 8, dtype=torch.float64, device=dev)
    _lib.check(L.svs_loss(R, S, n_eik, _ptr(rgb_values), _ptr(rgb_target), _ptr(gt), _ptr(weights), _ptr(pi_), _ptr(pj_),
                          _ptr(depth_values), float(rgb_weight), float(eikonal_weight), float(mvs_weight),
                          float(sparse_weight), float(gce), float(confi), int(bool(annealed)), float(anneal_sparse),
                          int(norm[0]) if norm else 0, int(norm[1]) if norm else 0, _ptr(losses), _ptr(d_rgb), _ptr(d_gt), _ptr(d_w), _ptr(d_dep), _ptr(ws),
                          _ptr(anneal_dev), _stream()), "svs_loss")
    return losses, dict(rgb_values=d_rgb, grad_theta=d_gt, weights=d_w, depth_values=d_dep)


class SamplerWorkspace:
    """Device buffers of the error-bounded sampler for R rays (allocated once, reused every step).
    group_rays: rays per convergence group (None: the whole batch, what one reference forward call does)."""

    def __init__(self, R, device, group_rays=None):
        L = _lib.load()
        self.R = R
        self.group_rays = int(group_rays) if group_rays else R
        self.cap, self.max_new = L.svs_sampler_cap(), L.svs_sampler_max_new()
        f = lambda *s: torch.empty(*s, device=device)
        self.z, self.sdf = f(R, self.cap), f(R, self.cap)
        self.beta, self.far = f(R), f(R)
        self.samples, self.samples_sdf = f(R, self.max_new), f(R, self.max_new)
        self.ctl = torch.zeros(L.svs_sampler_ctl_bytes(R, self.group_rays) // 4, dtype=torch.int32, device=device)
        self.ctl_stride = L.svs_sampler_ctl_stride()
        self.err = torch.zeros(1, dtype=torch.int32, device=device)


def sample_rays(packed, cam, dirs, beta_param, *, beta_min=1e-4, near, scene_bounding_sphere, sphere_scale, sdf_clamp_radius,
                N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10, max_total_iters=5,
                fast=-1, training=False, inverse_sphere_bg=False, add_tiny=0.0, inv_4log=None, rng=None,
                workspace=None, debug=None, sdf_override=None):
    """ErrorBoundSampler.get_z_vals (volsdf/model/ray_sampler.py:67-219) without host synchronisation.

    cam (3,) or (R,3), dirs (R,3); beta_param: the `density.beta` parameter (device scalar tensor or float);
    beta0 = |beta_param| + beta_min is formed on the device (density.py:28-30), so a step needs no host sync.
    rng: dict of device tensors for train mode: 'jitter' (R,N_eval), 'u' (R,N_samples), 'perm' (int32, >= N_extra),
         'eik_idx' (int32, R).   debug: optional dict that receives per-round index/cdf tensors.
    sdf_override: optional list of per-round (R, N_eval) tensors used instead of the MLP (parity tests).
    Returns z_vals (R, N_samples + N_samples_extra + 2) and z_samples_eik (R,1).
    """
    L = _lib.load()
    cam, dirs = _f32(cam), _f32(dirs)
    R = dirs.shape[0]
    dev = dirs.device
    ws = workspace or SamplerWorkspace(R, dev)
    assert ws.R == R
    rng = rng or {}
    if not torch.is_tensor(beta_param):
        beta_param = torch.tensor(float(beta_param), device=dev)
    beta_param = _f32(beta_param).reshape(1)
    max_iters = fast if fast >= 0 else max_total_iters
    if training and max_iters > 1 and "perm" in rng:
        raise NotImplementedError("train-mode extras for more than one round need the bin count on the host")
    if inv_4log is None:
        inv_4log = float(1.0 / (4.0 * torch.log(torch.tensor(eps + 1.0))))     # float32, ray_sampler.py:77
    cam_stride = 0 if cam.numel() == 3 else 3
    far = 2.0 * scene_bounding_sphere
    jitter = _f32(rng["jitter"]) if training and "jitter" in rng else None
    _lib.check(L.svs_sampler_init(_ptr(cam), cam_stride, _ptr(dirs), R, N_samples_eval, float(near), float(far),
                                  int(inverse_sphere_bg), float(scene_bounding_sphere), _ptr(jitter), float(inv_4log),
                                  max_iters, _ptr(ws.samples), _ptr(ws.beta), _ptr(ws.far), _ptr(ws.ctl), ws.group_rays,
                                  _ptr(ws.err), _stream()), "svs_sampler_init")
    n_out = (N_samples if max_iters > 0 else N_samples_eval) + N_samples_extra + 2
    z_final = torch.empty(R, n_out, device=dev)
    z_eik = torch.empty(R, 1, device=dev)
    u_final = _f32(rng["u"]) if training and "u" in rng else None
    extra_idx = rng["perm"].to(torch.int32).contiguous() if training and "perm" in rng else None
    eik_idx = rng["eik_idx"].to(torch.int32).contiguous() if training and "eik_idx" in rng else None

    def call(phase, i, dbg):
        _lib.check(L.svs_sampler_round(phase, R, i, max_iters, N_samples_eval, N_samples, N_samples_extra, _ptr(beta_param),
                                       float(beta_min), float(eps), beta_iters, float(add_tiny), float(near), _ptr(ws.far), _ptr(ws.z),
                                       _ptr(ws.sdf), _ptr(ws.beta), _ptr(ws.samples), _ptr(ws.samples_sdf), _ptr(ws.ctl),
                                       ws.group_rays, _ptr(u_final), _ptr(extra_idx), _ptr(eik_idx), _ptr(z_final), _ptr(z_eik),
                                       _ptr(dbg.get("samples_idx")), _ptr(dbg.get("inds")), _ptr(dbg.get("cdf")),
                                       _ptr(dbg.get("weights")), _stream()), "svs_sampler_round")

    if max_iters == 0:
        call(2, 0, {})
        return z_final, z_eik
    src = PointSource(cam=cam, dirs=dirs, z=ws.samples[:, :N_samples_eval]) if N_samples_eval == ws.max_new else None
    for i in range(max_iters):
        dbg = {}
        if debug is not None:
            dbg = dict(samples_idx=torch.full((R, ws.cap), -1, dtype=torch.int32, device=dev),
                       inds=torch.full((R, ws.max_new), -1, dtype=torch.int32, device=dev),
                       cdf=torch.zeros(R, ws.cap, device=dev), weights=torch.zeros(R, ws.cap, device=dev))
            debug.setdefault("rounds", []).append(dbg)
        if sdf_override is not None:
            if i < len(sdf_override):
                ws.samples_sdf[:, :N_samples_eval].copy_(sdf_override[i])
        else:
            if src is None:
                raise NotImplementedError("N_samples_eval must equal the kernel's row stride (128)")
            gate = ws.ctl.data_ptr() + 4 * (8 + i)          # Ctl.active[i] of group 0
            sdf_vals(packed, src, sdf_clamp_radius, sphere_scale, out=ws.samples_sdf, gate=gate,
                     gate_points=ws.group_rays * N_samples_eval, gate_stride=ws.ctl_stride)
        call(0, i, dbg)
        call(1, i, dbg)
        if debug is not None:
            dbg["z"] = ws.z.clone(); dbg["sdf"] = ws.sdf.clone(); dbg["beta"] = ws.beta.clone()
            dbg["samples"] = ws.samples.clone(); dbg["ctl"] = ws.ctl.clone()
    return z_final, z_eik
