"""torch-tensor wrappers for the CasMVSNet cost-volume kernels (csrc/svs_costvol.hip)."""
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib
from .ops import _f32, _ptr, _ptr_array, _stream


def relative_projection(src_proj, ref_proj):
    """models/CasMVSNet.py:622-625 and :290-292 on the host, in float64: proj (2,4,4) arrays / tensors ->
    12 floats: rows of (src @ inv(ref))[:3,:3] then [:3,3]."""
    def comb(P):
        P = np.asarray(P.detach().cpu() if torch.is_tensor(P) else P, np.float64)
        out = P[0].copy()
        out[:3, :4] = P[1][:3, :3] @ P[0][:3, :4]
        return out
    rel = comb(src_proj) @ np.linalg.inv(comb(ref_proj))
    return list(rel[:3, :3].reshape(-1)) + list(rel[:3, 3])


_HOST_CACHE = {}


def host_copy(t):
    """numpy copy of a small device tensor whose VALUES are launch arguments (projection matrices, the depth range):
    one device-to-host copy per distinct tensor, cached on storage + version -- the sample of a scan is the same
    tensor in every stage and every stage-loop iteration, so the stage loop runs without host synchronisation."""
    key = (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), str(t.device), t.dtype)
    hit = _HOST_CACHE.get(key)
    if hit is None:
        if len(_HOST_CACHE) >= 64:
            _HOST_CACHE.clear()
        # the entry keeps the tensor alive: its storage cannot be handed to another tensor while the key exists
        hit = _HOST_CACHE[key] = (t, t.detach().cpu().numpy())
    return hit[1]


_RT_CACHE = {}
_HWC_CACHE = {}


def _tensor_key(t):
    return (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), str(t.device), t.dtype)


def _rot_trans(proj_matrices):
    """(1,V,2,4,4) projection matrices -> ctypes float[12*(V-1)] for svs_warp_variance (host launch arguments).
    Cached per tensor like host_copy: a stage's matrices are the same tensor in every iteration of the stage loop."""
    key = _tensor_key(proj_matrices)
    hit = _RT_CACHE.get(key)
    if hit is not None:
        return hit[1]
    P = host_copy(proj_matrices)[0]
    n_src = P.shape[0] - 1
    rt = (ctypes.c_float * (12 * n_src))()
    for v in range(n_src):
        for k, x in enumerate(relative_projection(P[v + 1], P[0])):
            rt[12 * v + k] = float(x)
    if len(_RT_CACHE) >= 64:
        _RT_CACHE.clear()
    _RT_CACHE[key] = (proj_matrices, rt)
    return rt


def _hwc(f):
    """(C,H,W) source feature map -> its channel-last copy (H,W,C), cached per tensor (storage + version): a view's
    features are built once per scan and warped in every cost-volume build of the stage loop."""
    key = _tensor_key(f)
    hit = _HWC_CACHE.get(key)
    if hit is not None:
        return hit[1]
    L = _lib.load()
    C, H, W = f.shape
    o = torch.empty(H, W, C, device=f.device)
    _lib.check(L.svs_chw_to_hwc(_ptr(_f32(f)), _ptr(o), C, H, W, _stream()), "svs_chw_to_hwc")
    # bounded by bytes (a 32 x 1200 x 1600 float32 map and its copy are 490 MB), not by entries; `clear_caches()` drops
    # everything at the end of a scan / stage loop.  The key is (storage address, torch version counter, shape): a producer
    # that rewrites a cached feature map through this library's raw-pointer kernels does not bump the counter -- such
    # buffers must not be recycled while cached (the FeatureNet wrapper hands out fresh tensors).
    global _HWC_BYTES
    nbytes = 2 * f.numel() * 4
    if _HWC_BYTES + nbytes > _HWC_BUDGET:
        _HWC_CACHE.clear()
        _HWC_BYTES = 0
    _HWC_CACHE[key] = (f, o)          # keeps `f` alive: the key holds its address
    _HWC_BYTES += nbytes
    return o


_HWC_BYTES = 0
_HWC_BUDGET = int(os.environ.get("SVS_HWC_CACHE_BYTES", str(1 << 30)))


def clear_caches():
    """Drop the cached channel-last feature maps and projection constants, and the prior look-up's per-view constants (end of
    a scan / stage loop; StageLoop calls it per scan, VolOpt.get_mvs_input per stage)."""
    global _HWC_BYTES
    _HWC_CACHE.clear()
    _RT_CACHE.clear()
    _HWC_BYTES = 0
    from . import ops
    ops.clear_lookup_caches()


class SplitVolume:
    """A (C,D,H,W) volume in the form conv0 of the regularisation U-Net reads (include/svolsdf_hip.h,
    svs_split_volume_dims): fp16 hi / mid parts, channel-last 16-byte units, zero border.  The buffers are cached per
    shape: the border is zeroed once, producers rewrite the interior."""
    _cache = {}

    def __init__(self, C, D, H, W, device):
        L = _lib.load()
        self.C, self.D, self.H, self.W = C, D, H, W
        self.shape = (1, C, D, H, W)
        # (one buffer per shape AND stream: views processed on concurrent streams must not share it)
        key = (C, D, H, W, str(device), torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)
        buf = SplitVolume._cache.get(key)
        if buf is None:
            if len(SplitVolume._cache) >= 16:
                SplitVolume._cache.clear()
            nbytes = L.svs_split_volume_dims(C, D, H, W, None)
            buf = SplitVolume._cache[key] = torch.zeros(nbytes // 2, dtype=torch.float16, device=device)
        self.buf = buf
        self.device = buf.device
        # the buffer is shared by every SplitVolume of this shape and stream: a new one takes it over, and a holder of an
        # older object must not read it any more (`check_current`, called by the consumers)
        self._key = key
        self.generation = SplitVolume._generation[key] = SplitVolume._generation.get(key, 0) + 1

    _generation = {}

    def check_current(self):
        if SplitVolume._generation.get(self._key) != self.generation:
            raise RuntimeError("this SplitVolume's buffer has been reused by a later volume of the same shape on the same "
                               "stream (the buffers are cached per shape); consume a split volume before building the next")

    @staticmethod
    def pack(x):
        """float32 (C,D,H,W) -> SplitVolume"""
        L = _lib.load()
        x = _f32(x)
        C, D, H, W = x.shape
        sv = SplitVolume(C, D, H, W, x.device)
        _lib.check(L.svs_split_volume_pack(_ptr(x), _ptr(sv.buf), C, D, H, W, _stream()), "svs_split_volume_pack")
        return sv

    def float(self):
        """back to float32 (C,D,H,W): hi + mid (tests)"""
        self.check_current()
        L = _lib.load()
        dims = (ctypes.c_int * 2)()
        L.svs_split_volume_dims(self.C, self.D, self.H, self.W, dims)
        Hp, Wp = dims[0], dims[1]
        G = self.C // 8
        v = self.buf.view(self.D + 2, Hp, 2, G, Wp, 8)[1:self.D + 1, 1:self.H + 1, :, :, 1:self.W + 1].float()
        return (v[:, :, 0] + v[:, :, 1]).permute(2, 4, 0, 1, 3).reshape(self.C, self.D, self.H, self.W)


def pair_supported(C, Cout):
    return C in (8, 16, 32) and Cout <= 8 and not os.environ.get("SVS_CONV_PAIR_OFF")


def warp_variance(features, proj_matrices, depth_values, split=False):
    """DepthNet.forward step 2 (models/CasMVSNet.py:611-642).  features: list of (1,C,H,W) (reference first),
    proj_matrices: (1,V,2,4,4), depth_values (1,D,H,W) -> variance (1,C,D,H,W), or with split=True the same values
    as a SplitVolume (the producer side of the fused conv0, svs_conv3d_pair)."""
    L = _lib.load()
    ref = _f32(features[0][0])
    C, H, W = ref.shape
    D = depth_values.shape[1]
    dev = ref.device
    n_src = len(features) - 1
    hwc = [_hwc(f[0]) for f in features[1:]]
    rt = _rot_trans(proj_matrices)
    dv = _f32(depth_values[0])
    if split:
        sv = SplitVolume(C, D, H, W, dev)
        _lib.check(L.svs_warp_variance_split(_ptr(ref), _ptr_array(hwc), rt, n_src, C, D, H, W, _ptr(dv), _ptr(sv.buf),
                                             _stream()), "svs_warp_variance_split")
        return sv
    var = torch.empty(1, C, D, H, W, device=dev)
    _lib.check(L.svs_warp_variance(_ptr(ref), _ptr_array(hwc), rt, n_src, C, D, H, W, _ptr(dv), _ptr(var), 0, _stream()),
               "svs_warp_variance")
    return var


def homo_warp(src_fea, src_rel, depth_values):
    """homo_warping alone (models/CasMVSNet.py:280-315).  src_fea (C,H,W), src_rel: 12 floats
    (rows of (src_proj @ inv(ref_proj))[:3,:3], then [:3,3]), depth_values (D,H,W) -> (C,D,H,W)."""
    L = _lib.load()
    src = _f32(src_fea)
    C, H, W = src.shape
    dv = _f32(depth_values)
    D = dv.shape[0]
    hwc = torch.empty(H, W, C, device=src.device)
    _lib.check(L.svs_chw_to_hwc(_ptr(src), _ptr(hwc), C, H, W, _stream()), "svs_chw_to_hwc")
    rt = (ctypes.c_float * 12)(*[float(x) for x in src_rel])
    out = torch.empty(C, D, H, W, device=src.device)
    _lib.check(L.svs_warp_variance(_ptr(src), _ptr_array([hwc]), rt, 1, C, D, H, W, _ptr(dv), _ptr(out), 1, _stream()),
               "svs_warp_variance")
    return out


_WFRAG_CACHE = {}


def mfma_weight_fragments(weight):
    """[Cin][27][Cout] folded float32 weights -> the fp16 hi / mid A fragments of svs_conv3d_mfma
    ([k-step][piece][lane][8] fp16, see include/svolsdf_hip.h).  One-time repacking per layer (cached)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _WFRAG_CACHE.get(key)
    if hit is not None:
        return hit[0]
    Cin, _, Cout = weight.shape
    dev = weight.device
    KS = (27 * Cin + 31) // 32
    s = torch.arange(KS, device=dev).view(KS, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 64, 1)
    j = torch.arange(8, device=dev).view(1, 1, 8)
    kk = 32 * s + 8 * (lane >> 4) + j
    tap, ci, co = kk // Cin, kk % Cin, (lane & 15).expand(KS, 64, 8)
    ok = (tap < 27) & (co < Cout)
    w = weight[ci.clamp(max=Cin - 1), tap.clamp(max=26), co.clamp(max=Cout - 1)]
    w = torch.where(ok, w, torch.zeros_like(w)).float()
    hi = w.half()
    mid = (w - hi.float()).half()
    frag = torch.stack([hi, mid], 1).contiguous()          # (KS, 2, 64, 8) fp16
    _WFRAG_CACHE[key] = (frag, weight)                     # keep `weight` alive: the key holds its address
    return frag


def pair_weight_fragments(weight):
    """[Cin][27][Cout <= 8] folded float32 weights -> the fp16 hi / mid A fragments of svs_conv3d_pair
    ([k-step][piece][lane][8] fp16, include/svolsdf_hip.h): row m = (channel m & 7, x parity m >> 3),
    k = (((kd*3+kh)*4 + t)*G + g)*8 + c8 carries the weight of tap (kd, kh, kw = t - parity)."""
    key = ("pair", weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _WFRAG_CACHE.get(key)
    if hit is not None:
        return hit[0]
    Cin, _, Cout = weight.shape
    dev = weight.device
    G = Cin // 8
    KS = 9 * G
    s = torch.arange(KS, device=dev).view(KS, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 64, 1)
    j = torch.arange(8, device=dev).view(1, 1, 8)
    kk = 32 * s + 8 * (lane >> 4) + j
    c8, g, t, row9 = kk % 8, (kk // 8) % G, (kk // (8 * G)) % 4, kk // (32 * G)
    m = (lane & 15).expand(KS, 64, 8)
    co, kw = m & 7, t - (m >> 3)
    ok = (kw >= 0) & (kw <= 2) & (co < Cout)
    w = weight[8 * g + c8, (row9 * 3 + kw.clamp(0, 2)), co.clamp(max=Cout - 1)]
    w = torch.where(ok, w, torch.zeros_like(w)).float()
    hi = w.half()
    mid = (w - hi.float()).half()
    frag = torch.stack([hi, mid], 1).contiguous()          # (KS, 2, 64, 8) fp16
    _WFRAG_CACHE[key] = (frag, weight)
    return frag


_W2D_CACHE = {}


def conv2d_pack(weight):
    """(Cout,Cin,k,k) -> [ceil(Cout/8)][Cin][k][k][8] float32, the layout svs_conv2d reads through the scalar cache
    (include/svolsdf_hip.h).  Cached per weight tensor (address + version)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), weight.device)
    hit = _W2D_CACHE.get(key)
    if hit is not None:
        return hit[0]
    if len(_W2D_CACHE) > 64:
        _W2D_CACHE.clear()
    Cout, Cin, k, _ = weight.shape
    G = (Cout + 7) // 8
    w = torch.zeros(G * 8, Cin, k, k, device=weight.device, dtype=torch.float32)
    w[:Cout] = weight.detach().float()
    packed = w.view(G, 8, Cin, k, k).permute(0, 2, 3, 4, 1).contiguous()
    _W2D_CACHE[key] = (packed, weight)                     # keep `weight` alive: the key holds its address
    return packed


def conv2d(x, weight, bias=None, add=None, add_upsample2=False, stride=1, relu=False, out=None):
    """FeatureNet convolution (csrc/svs_conv2d.hip): x (Cin,H,W), weight (Cout,Cin,k,k) with BatchNorm folded, padding
    k // 2 -> (Cout,Ho,Wo) = [add +] relu?(conv + bias); add_upsample2: `add` is at half resolution (nearest x2)."""
    L = _lib.load()
    x = _f32(x)
    Cin, H, W = x.shape
    Cout, cin_w, k, k2 = weight.shape
    if cin_w != Cin or k != k2:
        raise ValueError("weight shape does not match the input")
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty(Cout, Ho, Wo, device=x.device)
    elif tuple(out.shape) != (Cout, Ho, Wo) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous float32 (Cout,Ho,Wo) tensor")
    bias = _f32(bias) if bias is not None else None
    add = _f32(add) if add is not None else None
    if add is not None and tuple(add.shape) != ((Cout, Ho // 2, Wo // 2) if add_upsample2 else (Cout, Ho, Wo)):
        raise ValueError("addend shape does not match the output")
    _lib.check(L.svs_conv2d(_ptr(x), _ptr(conv2d_pack(weight)), _ptr(bias), _ptr(add), int(bool(add_upsample2)), _ptr(out), Cin,
                            Cout, H, W, k, stride, int(bool(relu)), _stream()), "svs_conv2d")
    return out


_W2D_MFMA_CACHE = {}


def conv2d_mfma_supported(Cin, Cout, k, stride):
    return bool(_lib.load().svs_conv2d_mfma_supported(int(Cin), int(Cout), int(k), int(stride)))


def conv2d_mfma_frag(weight):
    """(Cout,Cin,k,k) float32 -> the fp16 hi / mid MFMA A fragments svs_conv2d_mfma reads (packed on the device by
    svs_conv2d_mfma_pack; cached per weight tensor: address + version)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), weight.device)
    hit = _W2D_MFMA_CACHE.get(key)
    if hit is not None:
        return hit[0]
    if len(_W2D_MFMA_CACHE) > 64:
        _W2D_MFMA_CACHE.clear()
    L = _lib.load()
    Cout, Cin, k, _ = weight.shape
    w = _f32(weight.detach())
    frag = torch.empty(L.svs_conv2d_mfma_wfrag_bytes(Cin, Cout, k) // 2, dtype=torch.float16, device=weight.device)
    _lib.check(L.svs_conv2d_mfma_pack(_ptr(w), Cin, Cout, k, _ptr(frag), _stream()), "svs_conv2d_mfma_pack")
    _W2D_MFMA_CACHE[key] = (frag, weight, w)               # keep `weight` alive: the key holds its address
    return frag


def conv2d_mfma(x, weight, bias=None, stride=1, relu=False):
    """A FeatureNet 3x3 (stride 1) / 5x5 (stride 2) convolution on the matrix cores (csrc/svs_conv2d_mfma.hip): x (Cin,H,W),
    weight (Cout,Cin,k,k), padding k // 2 -> (Cout,Ho,Wo) = relu?(conv + bias), float32 class (fp16x2)."""
    L = _lib.load()
    x = _f32(x)
    Cin, H, W = x.shape
    Cout, cin_w, k, k2 = weight.shape
    if cin_w != Cin or k != k2 or not conv2d_mfma_supported(Cin, Cout, k, stride):
        raise ValueError("shape not supported by svs_conv2d_mfma")
    pad = k // 2
    out = torch.empty(Cout, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, device=x.device)
    bias = _f32(bias) if bias is not None else None
    _lib.check(L.svs_conv2d_mfma(_ptr(x), _ptr(conv2d_mfma_frag(weight)), _ptr(bias), _ptr(out), Cin, Cout, H, W, k, stride,
                                 int(bool(relu)), _stream()), "svs_conv2d_mfma")
    return out


def conv2d_mfma_lateral(lat_in, lat_weight, lat_bias, lat_add, weight, bias=None, relu=False):
    """The FPN's lateral step fused into the 3x3 layer behind it (svs_conv2d_mfma_lateral): lat_in (8,H,W), lat_weight (32,8,1,1),
    lat_add (32,H/2,W/2), weight (Cout<=16,32,3,3) -> (Cout,H,W) = relu?(conv3x3(conv1x1(lat_in) + lat_bias + up2(lat_add)) + bias)."""
    L = _lib.load()
    lat_in, lat_add = _f32(lat_in), _f32(lat_add)
    _, H, W = lat_in.shape
    Cout = weight.shape[0]
    out = torch.empty(Cout, H, W, device=lat_in.device)
    lb = _f32(lat_bias) if lat_bias is not None else None
    bb = _f32(bias) if bias is not None else None
    _lib.check(L.svs_conv2d_mfma_lateral(_ptr(lat_in), _ptr(conv2d_pack(lat_weight)), _ptr(lb), _ptr(lat_add), _ptr(conv2d_mfma_frag(weight)),
                                         _ptr(bb), _ptr(out), Cout, H, W, int(bool(relu)), _stream()), "svs_conv2d_mfma_lateral")
    return out


_FPN_MFMA = os.environ.get("SVS_FPN_MFMA", "1") != "0"     # A/B: 0 = every layer of the pyramid on the float32 vector kernels
# (layer index -> stride) of svs_featurenet_fpn's 13 convolutions
_FPN_STRIDES = (1, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1)
_FPN_ADDEND = (9, 11)                                      # the lateral 1x1 convolutions take an addend: float32 kernels


class FeatureNetFpn:
    """The 13 convolutions of the 'fpn' FeatureNet enqueued by ONE library call (svs_featurenet_fpn): the per-launch host
    cost of going through Python 13 times was as large as the kernels' run time."""
    LAYERS = 13

    def __init__(self, base_channels):
        self.b = int(base_channels)
        self._ws, self._ws_key = None, None
        self._tables, self._tables_key = None, None

    def tables(self, layers):
        """layers: 13 (weight (Cout,Cin,k,k), bias or None) pairs in the library's order -> pointer tables (cached until
        a tensor changes)."""
        key = tuple((w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version)) for w, b in layers)
        if key != self._tables_key:
            packed = [conv2d_pack(w) for w, _ in layers]
            biases = [None if b is None else _f32(b) for _, b in layers]
            # the 3x3 / 5x5 layers with 8 / 16 / 32 input channels run on the matrix cores (svs_conv2d_mfma)
            # (not conv0.1, 8 -> 8 at full resolution: 15.5 us there against 13.4 on the vector kernel -- a quarter-full M tile)
            frags = [conv2d_mfma_frag(w) if (_FPN_MFMA and i not in _FPN_ADDEND and w.shape[2] > 1 and not (w.shape[1] <= 8 and w.shape[2] == 3)
                                             and conv2d_mfma_supported(w.shape[1], w.shape[0], w.shape[2], _FPN_STRIDES[i])) else None
                     for i, (w, _) in enumerate(layers)]
            self._tables = (_ptr_array(packed), _ptr_array(biases), packed, biases, layers, _ptr_array(frags), frags)
            self._tables_key = key
        return self._tables

    def __call__(self, image, layers):
        """image (3,H,W) -> stage1 (4b,H/4,W/4), stage2 (2b,H/2,W/2), stage3 (b,H,W)."""
        L = _lib.load()
        if len(layers) != self.LAYERS:
            raise ValueError("13 layers expected")
        image = _f32(image)
        _, H, W = image.shape
        if image.shape[0] != 3 or H % 4 or W % 4:
            raise ValueError("image must be (3,H,W) with H, W multiples of 4")
        dev, b = image.device, self.b
        if self._ws_key != (H, W, dev):
            self._ws = torch.empty(L.svs_featurenet_fpn_workspace_bytes(b, H, W) // 4, device=dev)
            self._ws_key = (H, W, dev)
        tb = self.tables(layers)
        wt, bt, ft = tb[0], tb[1], tb[5]
        s1 = torch.empty(4 * b, H // 4, W // 4, device=dev)
        s2 = torch.empty(2 * b, H // 2, W // 2, device=dev)
        s3 = torch.empty(b, H, W, device=dev)
        _lib.check(L.svs_featurenet_fpn2(_ptr(image), H, W, b, wt, bt, ft, _ptr(self._ws), _ptr(s1), _ptr(s2), _ptr(s3), _stream()),
                   "svs_featurenet_fpn2")
        return s1, s2, s3


_GEMM_ON = [True]


def gemm_enabled(on=None):
    """The MFMA implicit-GEMM path for the strided / transposed / coarse layers (svs_conv3d_gemm); off = the VALU direct
    convolutions (kept for A/B measurements and as the fall-back for channel counts the GEMM kernel does not cover)."""
    if on is not None:
        _GEMM_ON[0] = bool(on)
    return _GEMM_ON[0]


def rows_weight_fragments(weight):
    """[16][27][Cout <= 16] folded float32 weights -> the fp16 hi / mid A fragments of svs_conv3d_rows
    ([k-step][piece][lane][8], include/svolsdf_hip.h): k-step s, lane group kg = combination 4 s + kg = tap * G + g."""
    key = ("rows", weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _WFRAG_CACHE.get(key)
    if hit is not None:
        return hit[0]
    Cin, _, Cout = weight.shape
    dev = weight.device
    G = Cin // 8
    KS = (27 * G + 3) // 4
    s = torch.arange(KS, device=dev).view(KS, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 64, 1)
    j = torch.arange(8, device=dev).view(1, 1, 8)
    c = 4 * s + (lane >> 4)
    tap, g = (c // G).expand(KS, 64, 8), (c % G).expand(KS, 64, 8)
    co = (lane & 15).expand(KS, 64, 8)
    ok = (tap < 27) & (co < Cout)
    w = weight[8 * g + j, tap.clamp(max=26), co.clamp(max=Cout - 1)]
    w = torch.where(ok, w, torch.zeros_like(w)).float()
    hi = w.half()
    mid = (w - hi.float()).half()
    frag = torch.stack([hi, mid], 1).contiguous()
    _WFRAG_CACHE[key] = (frag, weight)
    return frag


def s2c8_weight_fragments(weight):
    """[8][27][Cout <= 16] folded float32 weights -> the fp16 hi / mid A fragments of svs_conv3d_s2c8
    ([9 k-steps][piece][lane][8], include/svolsdf_hip.h)."""
    key = ("s2c8", weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _WFRAG_CACHE.get(key)
    if hit is not None:
        return hit[0]
    Cin, _, Cout = weight.shape
    dev = weight.device
    s = torch.arange(9, device=dev).view(9, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 64, 1)
    j = torch.arange(8, device=dev).view(1, 1, 8).expand(9, 64, 8)
    kx, q = s // 3, s % 3
    row = (lane >> 4) + 4 * q
    co = (lane & 15).expand(9, 64, 8)
    ok = ((row < 9) & (co < Cout)).expand(9, 64, 8)
    tap = (row.clamp(max=8) * 3 + kx).expand(9, 64, 8)
    w = weight[j, tap, co.clamp(max=Cout - 1)]
    w = torch.where(ok, w, torch.zeros_like(w)).float()
    hi = w.half()
    mid = (w - hi.float()).half()
    frag = torch.stack([hi, mid], 1).contiguous()          # (9, 2, 64, 8) fp16
    _WFRAG_CACHE[key] = (frag, weight)
    return frag


def gemm_weight_fragments(weight, transposed):
    """[Cin][27][Cout] folded float32 weights -> the A fragments of svs_conv3d_gemm (packed on the device, cached)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), bool(transposed))
    hit = _WFRAG_CACHE.get(key)
    if hit is not None:
        return hit[0]
    L = _lib.load()
    Cin, _, Cout = weight.shape
    w = _f32(weight)
    frag = torch.empty(L.svs_conv3d_gemm_wfrag_bytes(Cin, Cout, int(transposed)), dtype=torch.uint8, device=weight.device)
    _lib.check(L.svs_conv3d_gemm_pack(_ptr(w), Cin, Cout, int(transposed), _ptr(frag), _stream()), "svs_conv3d_gemm_pack")
    _WFRAG_CACHE[key] = (frag, weight)
    return frag


def rows_supported(Cin, Cout):
    return Cin == 16 and Cout <= 16 and not os.environ.get("SVS_CONV_ROWS_OFF")


def conv3d(x, weight, bias=None, skip=None, stride=1, transposed=False, relu=True, split_out=False):
    """x (Cin,D,H,W); weight [Cin][27][Cout] folded; -> (Cout,Do,Ho,Wo).  split_out: return the result as a SplitVolume (the
    stride-2 convolution from 8 to 16 channels only: conv1 feeding conv2)."""
    L = _lib.load()
    if isinstance(x, SplitVolume):
        Cout = weight.shape[2]
        rows = x.C == 16 and Cout > 8                # conv2 (channel rows); otherwise conv0 (x-pair rows, Cout <= 8)
        fused = (not transposed and stride == 1 and skip is None
                 and (rows_supported(x.C, Cout) if rows else pair_supported(x.C, Cout)))
        if fused:
            x.check_current()
            out = torch.empty((Cout, x.D, x.H, x.W), device=x.device)
            if rows:
                _lib.check(L.svs_conv3d_rows(_ptr(x.buf), _ptr(rows_weight_fragments(weight)), _ptr(bias), _ptr(out), x.C, Cout,
                                             x.D, x.H, x.W, int(relu), _stream()), "svs_conv3d_rows")
            else:
                _lib.check(L.svs_conv3d_pair(_ptr(x.buf), _ptr(pair_weight_fragments(weight)), _ptr(bias), _ptr(out), x.C, Cout,
                                             x.D, x.H, x.W, int(relu), _stream()), "svs_conv3d_pair")
            return out
        x = x.float()                # no fused form for this layer (or switched off): back to the float32 volume
    x = _f32(x)
    Cin, D, H, W = x.shape
    Cout = weight.shape[2]
    if not transposed and stride == 1 and Cout == 1 and not os.environ.get("SVS_CONV_C1_OFF"):
        out = torch.empty((1, D, H, W), device=x.device)
        _lib.check(L.svs_conv3d_c1(_ptr(x), _ptr(_f32(weight)), _ptr(bias), _ptr(skip), _ptr(out), Cin, D, H, W, int(relu),
                                   _stream()), "svs_conv3d_c1")
        return out
    if not transposed and stride == 1 and Cin in (8, 16, 32) and Cout <= 16 and not os.environ.get("SVS_CONV_RING_OFF"):
        frag = mfma_weight_fragments(weight)
        out = torch.empty((Cout, D, H, W), device=x.device)
        _lib.check(L.svs_conv3d_mfma(_ptr(x), _ptr(frag), _ptr(bias), _ptr(skip), _ptr(out), Cin, Cout, D, H, W,
                                     int(relu), _stream()), "svs_conv3d_mfma")
        return out
    if transposed:
        shp = (Cout, 2 * D, 2 * H, 2 * W)
    else:
        shp = (Cout, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1)
    out = torch.empty(shp, device=x.device)
    if not transposed and stride == 2 and Cin == 8 and Cout <= 16 and W % 2 == 0 and not os.environ.get("SVS_CONV_S2C8_OFF"):
        frag = s2c8_weight_fragments(weight)
        if split_out and Cout == 16 and skip is None:
            sv = SplitVolume(Cout, *shp[1:], x.device)
            _lib.check(L.svs_conv3d_s2c8(_ptr(x), _ptr(frag), _ptr(bias), None, None, _ptr(sv.buf), Cout, D, H, W, int(relu),
                                         _stream()), "svs_conv3d_s2c8")
            return sv
        _lib.check(L.svs_conv3d_s2c8(_ptr(x), _ptr(frag), _ptr(bias), _ptr(skip), _ptr(out), None, Cout, D, H, W, int(relu),
                                     _stream()), "svs_conv3d_s2c8")
        return out
    if gemm_enabled() and L.svs_conv3d_gemm_supported(Cin, Cout):
        frag = gemm_weight_fragments(weight, transposed)
        _lib.check(L.svs_conv3d_gemm(_ptr(x), _ptr(frag), _ptr(bias), _ptr(skip), _ptr(out), Cin, Cout, D, H, W, stride,
                                     int(transposed), int(relu), _stream()), "svs_conv3d_gemm")
        return out
    _lib.check(L.svs_conv3d(_ptr(x), _ptr(weight), _ptr(bias), _ptr(skip), _ptr(out), Cin, Cout, D, H, W, stride,
                            int(transposed), int(relu), _stream()), "svs_conv3d")
    return out


def prob_depth_conf(reg, depth_values):
    """reg (D,H,W), depth_values (D,H,W) -> prob (D,H,W), depth (H,W), conf (H,W), index (H,W int32)."""
    L = _lib.load()
    reg, dv = _f32(reg), _f32(depth_values)
    D, H, W = reg.shape
    dev = reg.device
    prob = torch.empty(D, H, W, device=dev)
    depth = torch.empty(H, W, device=dev)
    conf = torch.empty(H, W, device=dev)
    idx = torch.empty(H, W, dtype=torch.int32, device=dev)
    _lib.check(L.svs_prob_depth_conf(_ptr(reg), _ptr(dv), D, H, W, _ptr(prob), _ptr(depth), _ptr(conf), _ptr(idx),
                                     _stream()), "svs_prob_depth_conf")
    return prob, depth, conf, idx


def depth_hypotheses(prev_depth, img_hw, ndepth, scale, dmin, dmax, pix_interval, inverse, device):
    """models/CasMVSNet.py:733-751 -> (D, H/scale, W/scale)."""
    L = _lib.load()
    H, W = img_hw
    out = torch.empty(ndepth, H // scale, W // scale, device=device)
    pd = _f32(prev_depth) if prev_depth is not None else None
    Hp, Wp = (pd.shape[-2], pd.shape[-1]) if pd is not None else (0, 0)
    _lib.check(L.svs_depth_hypotheses(_ptr(pd), Hp, Wp, H, W, ndepth, scale, float(dmin), float(dmax),
                                      float(pix_interval), int(bool(inverse)), _ptr(out), _stream()),
               "svs_depth_hypotheses")
    return out
