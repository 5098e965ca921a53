"""CPU oracle for the S-VolSDF volume-rendering hot path (numpy, float32).

TEST INFRASTRUCTURE ONLY.  This module is a restatement, written from scratch, of
the algorithm of the reference (cvlab-stonybrook/s-volsdf, file:line cited on every
function, paths relative to the reference root).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it; the product path
(`s-volsdf_amd/`) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  Every function here is checked in tests/test_oracle_golden.py
against golden arrays produced by importing the reference itself on CPU
(tests/golden/make_fixtures.py, run in the build container; the reference has no
tests or golden vectors of its own -- SURVEY.md section 4).

Numeric contract (what "bit-exact" means for the sampler), also in DESIGN.md section 2:
  * every elementary op (+ - * / sqrt, compare) is an IEEE float32 op in the
    reference's order, with no fused multiply-add except where the restated
    library routine itself uses one (written `_fma32`);
  * `torch.expm1(float32)` on CPU is Sleef's `Sleef_expm1f8_u10` (sleef 3.x as bundled
    with torch 2.10, `xexpm1f` / `expk2f` of sleefsimdsp.c, FMA build): restated here
    operation by operation as `sleef_expm1f`; bit-equal to the library routine
    exported by libtorch_cpu.so on 12.6 M inputs (tests/golden/check_primitives.py);
  * `torch.sum(float32 row, dim=-1)` on CPU is ATen's `cascade_sum`
    (aten/src/ATen/native/cpu/SumKernel.cpp: `vectorized_inner_sum` -> `row_sum` ->
    `multi_row_sum`, 8-lane vectors, 4 independent accumulators, cascade levels of
    16): restated as `aten_sum`; the AVX2 kernel also serves AVX-512 hosts (the stub
    is registered without an AVX-512 variant), so the order is the same on every
    x86-64 host; bit-equal to torch.sum for every row length 1..700 and longer rows;
  * `torch.exp(float32)` on CPU is NOT an open algorithm in an MKL build of torch: it
    dispatches to Intel MKL VML `vmsExp` (closed source), whose AVX-512 / AVX2 / SSE
    kernels return different float32 results on 1-4 % of all inputs (measured,
    check_primitives.py), so the reference's own exp depends on the host.  The
    contract pins the one OPEN implementation torch has: Sleef's `Sleef_expf8_u10`
    (what ATen's Vectorized<float>::exp() calls in a build without MKL), restated as
    `sleef_expf`.  The golden fixtures are generated with the reference's
    `torch.exp` bound to that very library routine (tests/golden/ref_shim.py::
    pin_open_exp), which makes them host-independent; `sampler64_hostexp_*.npz`
    holds the unpinned reference for the cross-check with the host's own exp;
  * `torch.sqrt(float32)` is likewise MKL VML in an MKL build (AVX-512 kernel: off by one
    ulp on 0.7 % of all inputs, AVX2 kernel: IEEE); pinned to the IEEE square root
    (ATen's Vectorized<float>::sqrt()), `ref_sqrt`;
  * cumsum accumulates in float64 and rounds each prefix to float32 (what torch-CPU
    cumsum does) in the CANONICAL BLOCKED ORDER of `canon_cumsum`
    (64 contiguous chunks, sequential inside a chunk, Kogge-Stone across chunks) --
    the order a 64-lane wavefront scan produces (float64 sums of <= 640 float32
    terms round to the same float32 in any order except at measure-zero ties).
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
F64 = np.float64


def _fma32(a, b, c):
    """float32 fused multiply-add with ONE rounding, emulated exactly: the product of two float32 is exact in
    float64; the float64 sum is rounded to odd (TwoSum error term), so the final rounding to float32 cannot
    double-round."""
    a = np.asarray(a, F32).astype(F64)
    b = np.asarray(b, F32).astype(F64)
    c = np.asarray(c, F32).astype(F64)
    with np.errstate(all="ignore"):
        p = a * b
        s = p + c
        bb = s - p
        e = (p - (s - bb)) + (c - bb)
        even = (np.atleast_1d(s).view(np.int64) & 1).reshape(np.shape(s)) == 0
        fix = (e != 0) & even & np.isfinite(s)
        s = np.where(fix, np.nextafter(s, np.where(e > 0, np.inf, -np.inf)), s)
        return s.astype(F32)


# ----------------------------------------------------------------------------------------
# exp / expm1: Sleef 3.x (bundled with torch), u10 single-precision routines, FMA build
# ----------------------------------------------------------------------------------------
_R_LN2F = F32(1.4426950216293335)          # 0x3fb8aa3b
_L2UF = F32(0.693145751953125)             # 0x3f317200
_L2LF = F32(1.428606765330187e-06)         # 0x35bfbe8e
_EXPF_C = tuple(F32(c) for c in (0.00019852761761285365, 0.0013930435525253415, 0.008333360776305199,
                                 0.041666485369205475, 0.1666666716337204, 0.5))
_EXPK2F_C = tuple(F32(c) for c in (0.00019809602235909551, 0.0013942564837634563, 0.008333456702530384,
                                   0.04166637361049652))


def _f(x):
    return np.asarray(x).astype(F32)


def _ldexp2f(u, q):
    """vldexp2_vf_vf_vi2: u * 2^(q>>1) * 2^(q - (q>>1)), two float32 multiplications"""
    h = q >> 1
    return _f(_f(u * np.ldexp(F32(1.0), h).astype(F32)) * np.ldexp(F32(1.0), q - h).astype(F32))


def sleef_expf(d) -> np.ndarray:
    """Sleef_expf8_u10 (xexpf, sleefsimdsp.c), float32 -> float32.  q = rint(d / ln2) by cvtps2dq (nearest even);
    s = fma(q, -L2U, d); s = fma(q, -L2L, s); degree-5 Horner in fma; u = fma(s*s, u, s) + 1; ldexp2;
    0 below -104, +inf above 100."""
    d = np.asarray(d, F32)
    with np.errstate(all="ignore"):
        dc = np.clip(np.nan_to_num(d, nan=0.0), -200.0, 200.0).astype(F32)
        q = np.rint(_f(dc * _R_LN2F)).astype(np.int32)
        qf = q.astype(F32)
        s = _fma32(qf, -_L2UF, dc)
        s = _fma32(qf, -_L2LF, s)
        u = np.full_like(s, _EXPF_C[0])
        for c in _EXPF_C[1:]:
            u = _fma32(u, s, c)
        u = _f(_fma32(_f(s * s), u, s) + F32(1.0))
        u = _ldexp2f(u, q)
    u = np.where(d < F32(-104.0), F32(0.0), u)
    u = np.where(d > F32(100.0), F32(np.inf), u)
    u = np.where(np.isnan(d), F32(np.nan), u)
    return u.astype(F32)


# double-float helpers of sleef (df.h, FMA variants); a double-float is a pair (x, y) of float32 arrays
def _dfadd2_f_f(x, y):
    s = _f(x + y)
    v = _f(s - x)
    return s, _f(_f(x - _f(s - v)) + _f(y - v))


def _dfadd2_f2_f(xx, xy, y):
    s, t = _dfadd2_f_f(xx, y)
    return s, _f(t + xy)


def _dfadd2_f2_f2(xx, xy, yx, yy):
    s, t = _dfadd2_f_f(xx, yx)
    return s, _f(t + _f(xy + yy))


def _dfmul_f2_f(xx, xy, y):
    s = _f(xx * y)
    return s, _fma32(xy, y, _fma32(xx, y, -s))


def _dfmul_f2_f2(xx, xy, yx, yy):
    s = _f(xx * yx)
    return s, _fma32(xx, yy, _fma32(xy, yx, _fma32(xx, yx, -s)))


def _dfsqu_f2(xx, xy):
    s = _f(xx * xx)
    return s, _fma32(_f(xx + xx), xy, _fma32(xx, xx, -s))


def sleef_expm1f(a) -> np.ndarray:
    """Sleef_expm1f8_u10 (xexpm1f = expk2f(a) + (-1) in double-float arithmetic, sleefsimdsp.c)."""
    a = np.asarray(a, F32)
    with np.errstate(all="ignore"):
        ac = np.clip(np.nan_to_num(a, nan=0.0), -200.0, 200.0).astype(F32)
        q = np.rint(_f(_f(ac + F32(0.0)) * _R_LN2F)).astype(np.int32)
        qf = q.astype(F32)
        sx, sy = _dfadd2_f2_f(ac, np.zeros_like(ac), _f(qf * -_L2UF))
        sx, sy = _dfadd2_f2_f(sx, sy, _f(qf * -_L2LF))
        u = np.full_like(sx, _EXPK2F_C[0])
        for c in _EXPK2F_C[1:]:
            u = _fma32(u, sx, c)
        tx, ty = _dfmul_f2_f(sx, sy, u)
        tx, ty = _dfadd2_f2_f(tx, ty, F32(0.1666666567325592))
        tx, ty = _dfmul_f2_f2(sx, sy, tx, ty)
        tx, ty = _dfadd2_f2_f(tx, ty, F32(0.5))
        qx, qy = _dfsqu_f2(sx, sy)
        mx, my = _dfmul_f2_f2(qx, qy, tx, ty)
        tx, ty = _dfadd2_f2_f2(sx, sy, mx, my)
        s = _f(F32(1.0) + tx)                                   # dfadd_vf2_vf_vf2(1, t)
        ty = _f(_f(_f(F32(1.0) - s) + tx) + ty)
        tx = s
        tx, ty = _ldexp2f(tx, q), _ldexp2f(ty, q)
        low = ac < F32(-104.0)
        tx, ty = np.where(low, F32(0.0), tx), np.where(low, F32(0.0), ty)
        dx, dy = _dfadd2_f2_f(tx, ty, F32(-1.0))
        x = _f(dx + dy)
    x = np.where(a > F32(88.72283172607422), F32(np.inf), x)
    x = np.where(a < F32(-16.63553237915039), F32(-1.0), x)
    x = np.where((a == 0) & np.signbit(a), F32(-0.0), x)
    x = np.where(np.isnan(a), F32(np.nan), x)
    return x.astype(F32)


# ----------------------------------------------------------------------------------------
# torch.sum(float32, dim=-1) on CPU: ATen cascade_sum (SumKernel.cpp), AVX2 kernel
# ----------------------------------------------------------------------------------------
_SUM_LANES = 8        # Vectorized<float>::size() of the kernel that is dispatched (AVX2, also on AVX-512 hosts)
_SUM_ILP = 4          # row_sum's ilp_factor
_SUM_LEVELS = 4       # multi_row_sum's num_levels


def _multi_row_sum(v):
    """multi_row_sum<acc_t, 4>: v (R, size, 4, L) -> (R, 4, L); level 0 takes `level_step` rows, then folds upwards."""
    R, size = v.shape[:2]
    clog = 0
    while (1 << clog) < size:
        clog += 1
    power = max(4, clog // _SUM_LEVELS)
    step, mask = 1 << power, (1 << power) - 1
    acc = np.zeros((_SUM_LEVELS,) + (R,) + v.shape[2:], F32)
    i = 0
    while i + step <= size:
        for _ in range(step):
            acc[0] = acc[0] + v[:, i]
            i += 1
        for j in range(1, _SUM_LEVELS):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = 0
            if (i & (mask << (j * power))) != 0:
                break
    while i < size:
        acc[0] = acc[0] + v[:, i]
        i += 1
    for j in range(1, _SUM_LEVELS):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def _row_sum(v):
    """row_sum<acc_t>: v (R, size, L) (L = 8 for vector loads, 1 for scalar loads) -> (R, L)"""
    R, size, L = v.shape
    n4 = size // _SUM_ILP
    part = _multi_row_sum(v[:, :n4 * _SUM_ILP].reshape(R, n4, _SUM_ILP, L))
    p0 = part[:, 0]
    for i in range(n4 * _SUM_ILP, size):
        p0 = p0 + v[:, i]
    for k in range(1, _SUM_ILP):
        p0 = p0 + part[:, k]
    return p0


def aten_sum(x) -> np.ndarray:
    """torch.sum(x, -1, keepdim=True) for a contiguous float32 x, in ATen's order (module docstring)."""
    x = np.asarray(x, F32)
    m = x.shape[-1]
    x2 = x.reshape(-1, m)
    R = x2.shape[0]
    with np.errstate(over="ignore", invalid="ignore"):
        if m < _SUM_LANES:                                   # scalar_inner_sum
            out = _row_sum(x2.reshape(R, m, 1))[:, 0]
        else:                                                # vectorized_inner_sum
            nv = m // _SUM_LANES
            vec = _row_sum(x2[:, :nv * _SUM_LANES].reshape(R, nv, _SUM_LANES))
            out = np.zeros(R, F32)
            for k in range(nv * _SUM_LANES, m):
                out = out + x2[:, k]
            for k in range(_SUM_LANES):
                out = out + vec[:, k]
    return out.reshape(x.shape[:-1] + (1,)).astype(F32)


# the three primitives as the restatement below calls them (module attributes, so that a test can bind the
# host's own exp for the cross-check against the unpinned reference)
ref_exp = sleef_expf
ref_expm1 = sleef_expm1f
ref_sum = aten_sum


def ref_sqrt(x):
    """torch.sqrt pinned to the IEEE square root (module docstring: in an MKL build the reference's `torch.sqrt` is
    MKL VML, whose AVX-512 kernel is off by one ulp on 0.7 % of all inputs)."""
    return np.sqrt(np.asarray(x, F32)).astype(F32)


# ----------------------------------------------------------------------------------------
# canonical scans
# ----------------------------------------------------------------------------------------
def canon_cumsum64(x) -> np.ndarray:
    """Inclusive cumsum along the last axis, float64 accumulation in the canonical blocked order.

    Lane i of 64 owns the contiguous chunk [i*c, (i+1)*c), c = ceil(m/64); prefixes are
    sequential inside a chunk; chunk totals are combined by a Kogge-Stone inclusive scan
    (d = 1,2,4,8,16,32: t[i] += t[i-d]); result[i*c+j] = offset_i + local_i[j].
    Returns float64 (callers round to float32).
    """
    x = np.asarray(x)
    m = x.shape[-1]
    c = max(1, -(-m // 64))
    pad = 64 * c - m
    xd = x.astype(F64)
    if pad:
        xd = np.concatenate([xd, np.zeros(x.shape[:-1] + (pad,), F64)], -1)
    xd = xd.reshape(x.shape[:-1] + (64, c))
    local = np.empty_like(xd)
    acc = np.zeros(xd.shape[:-1], F64)
    for j in range(c):
        acc = acc + xd[..., j]
        local[..., j] = acc
    tot = local[..., c - 1].copy()
    for d in (1, 2, 4, 8, 16, 32):
        nxt = tot.copy()
        nxt[..., d:] = tot[..., d:] + tot[..., :-d]
        tot = nxt
    off = np.concatenate([np.zeros(tot.shape[:-1] + (1,), F64), tot[..., :-1]], -1)
    out = off[..., None] + local
    return out.reshape(x.shape[:-1] + (64 * c,))[..., :m]


def canon_cumsum(x) -> np.ndarray:
    """torch.cumsum(float32) restatement: float64 accumulate, each prefix rounded to float32."""
    return canon_cumsum64(x).astype(F32)


# ----------------------------------------------------------------------------------------
# a1  rays      volsdf/utils/rend_util.py:60-95 (get_camera_params), :143-156 (lift)
# ----------------------------------------------------------------------------------------
def norm3(v):
    """torch `x.norm(2, dim=-1, keepdim=True)` of float32 3-vectors on CPU: the vectorised reduce kernel contracts
    acc + x*x into a fused multiply-add, i.e. sqrt(fma(x2, x2, fma(x1, x1, x0*x0))) (checked on 200 000 rows)."""
    v = np.asarray(v, F32)
    acc = _f(v[..., 0] * v[..., 0])
    for k in range(1, v.shape[-1]):
        acc = _fma32(v[..., k], v[..., k], acc)
    return np.sqrt(acc).astype(F32)[..., None]


def _normalize(v, eps=1e-12):
    """F.normalize: v / max(||v||_2, eps)"""
    return (v / np.maximum(norm3(v), F32(eps))).astype(F32)


def lift(x, y, z, K):
    """rend_util.py:143-156.  K (4,4); x,y,z (N,)."""
    fx, fy, cx, cy, sk = (F32(K[0, 0]), F32(K[1, 1]), F32(K[0, 2]), F32(K[1, 2]), F32(K[0, 1]))
    x_l = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_l = (y - cy) / fy * z
    return np.stack([x_l, y_l, z], -1).astype(F32)


def rays_from_uv(uv, pose, K):
    """rend_util.py:60-95 + volsdf/model/network.py:213-217.

    uv (R,2) pixel coords (x=col, y=row), pose (4,4) cam-to-world, K (4,4).
    Returns ray_dirs (R,3) unit, cam_loc (3,), depth_scale (R,1) = z of the unit ray in the camera frame.
    """
    uv = np.asarray(uv, F32)
    pose = np.asarray(pose, F32)
    K = np.asarray(K, F32)
    p_cam = lift(uv[:, 0], uv[:, 1], np.ones(uv.shape[0], F32), K)          # (R,3)
    world = (p_cam @ pose[:3, :3].T).astype(F32) + pose[:3, 3]
    dirs = _normalize(world - pose[:3, 3])
    depth_scale = _normalize(p_cam.copy())[:, 2:3]     # identity pose: dirs_cam = normalize(p_cam)
    return dirs, pose[:3, 3].copy(), depth_scale


def sphere_intersections(cam_loc, dirs, r):
    """rend_util.py:200-216.  cam_loc (R,3), dirs (R,3) -> (R,2) near/far, clamped >= 0."""
    # ray_cam_dot: torch.bmm of (1,3) x (3,1) -- ATen's small-matrix loop, plain multiply-add in index order;
    # cam_loc.norm(2, 1) ** 2: the SQUARE of the rounded norm (norm3), not the sum of squares
    dot = _f(_f(_f(dirs[:, 0] * cam_loc[:, 0]) + _f(dirs[:, 1] * cam_loc[:, 1])) + _f(dirs[:, 2] * cam_loc[:, 2]))[:, None]
    nrm = norm3(cam_loc)
    under = _f(_f(dot * dot) - _f(_f(nrm * nrm) - F32(r * r)))
    if (under <= 0).any():
        raise ValueError("BOUNDING SPHERE PROBLEM")       # the reference calls exit() here
    s = np.sqrt(under).astype(F32)
    return np.maximum(np.concatenate([-s - dot, s - dot], -1), F32(0)).astype(F32)


# ----------------------------------------------------------------------------------------
# a4/A2  positional encoding     volsdf/model/embedder.py:10-36
# ----------------------------------------------------------------------------------------
def posenc(x, L):
    """[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]; x (P,d) -> (P, d*(1+2L))."""
    x = np.asarray(x, F32)
    out = [x]
    for k in range(L):
        f = F32(2.0 ** k)
        out.append(np.sin(x * f, dtype=F32))
        out.append(np.cos(x * f, dtype=F32))
    return np.concatenate(out, -1).astype(F32)


# ----------------------------------------------------------------------------------------
# a5  SDF MLP      volsdf/model/network.py:71-131
# ----------------------------------------------------------------------------------------
def weightnorm(g, v):
    """nn.utils.weight_norm(dim=0): w = g * v / ||v||_row   (network.py:64-65)."""
    v = np.asarray(v, F32)
    nrm = np.sqrt((v.astype(F64) ** 2).sum(1, keepdims=True)).astype(F32)
    return (np.asarray(g, F32).reshape(-1, 1) * (v / nrm)).astype(F32)


def effective_weights(params, prefix, n_layers):
    """params: dict name -> array with torch state_dict names. Returns [(W,b)] effective per layer."""
    out = []
    for l in range(n_layers):
        b = np.asarray(params[f"{prefix}.lin{l}.bias"], F32)
        if f"{prefix}.lin{l}.weight_g" in params:
            W = weightnorm(params[f"{prefix}.lin{l}.weight_g"], params[f"{prefix}.lin{l}.weight_v"])
        else:
            W = np.asarray(params[f"{prefix}.lin{l}.weight"], F32)
        out.append((W, b))
    return out


def softplus100(a):
    """nn.Softplus(beta=100), threshold 20 (network.py:69): a if 100a > 20 else log1p(exp(100a))/100."""
    a = np.asarray(a, F32)
    t = a * F32(100.0)
    with np.errstate(over="ignore"):
        soft = (np.log1p(np.exp(np.minimum(t, F32(30.0)), dtype=F32), dtype=F32) / F32(100.0)).astype(F32)
    return np.where(t > F32(20.0), a, soft).astype(F32)


def sigmoid100(a):
    """d softplus100 / da = sigmoid(100 a) (1 above the threshold)."""
    t = np.asarray(a, F32).astype(F64) * 100.0
    s = 1.0 / (1.0 + np.exp(-np.clip(t, -700, 700)))
    return np.where(t > 20.0, 1.0, s).astype(F32)


INV_SQRT2 = F32(1.0 / np.sqrt(2.0))


def sdf_mlp_forward(layers, x, multires=6, skip_in=(4,), return_cache=False):
    """ImplicitNetwork.forward, network.py:71-88.  layers = [(W,b)]*9; x (P,d_in) -> (P, 1+feat)."""
    x = np.asarray(x, F32)
    inp = posenc(x, multires) if multires > 0 else x
    h = inp
    cache = []
    n = len(layers)
    for l, (W, b) in enumerate(layers):
        if l in skip_in:
            h = (np.concatenate([h, inp], 1) / F32(np.sqrt(2.0))).astype(F32)
        a = (h @ W.T + b).astype(F32)
        cache.append((h, a))
        h = softplus100(a) if l < n - 1 else a
    if return_cache:
        return h, inp, cache
    return h


def sdf_clamp(sdf, x, sdf_bounding_sphere, sphere_scale):
    """network.py:110-112 / :128-130: min(sdf, scale*(R - |x|)) when sdf_bounding_sphere > 0."""
    if sdf_bounding_sphere > 0.0:
        nrm = np.sqrt((x * x).sum(1, keepdims=True, dtype=F32)).astype(F32)
        sphere = (F32(sphere_scale) * (F32(sdf_bounding_sphere) - nrm)).astype(F32)
        return np.minimum(sdf, sphere), sphere
    return sdf, None


def sdf_vals(layers, x, sdf_bounding_sphere=3.0, sphere_scale=20.0, multires=6):
    """ImplicitNetwork.get_sdf_vals, network.py:125-131 -> (P,1)."""
    out = sdf_mlp_forward(layers, x, multires)
    return sdf_clamp(out[:, :1], np.asarray(x, F32), sdf_bounding_sphere, sphere_scale)[0]


def posenc_vjp(x, g_pe, L):
    """d/dx of posenc contracted with g_pe (P, d*(1+2L)) -> (P,d)."""
    d = x.shape[1]
    gx = g_pe[:, :d].astype(F64).copy()
    xd = x.astype(F64)
    for k in range(L):
        f = 2.0 ** k
        gs = g_pe[:, d * (1 + 2 * k): d * (2 + 2 * k)].astype(F64)
        gc = g_pe[:, d * (2 + 2 * k): d * (3 + 2 * k)].astype(F64)
        gx += f * (np.cos(xd * f) * gs - np.sin(xd * f) * gc)
    return gx.astype(F32)


def sdf_outputs(layers, x, sdf_bounding_sphere=3.0, sphere_scale=20.0, multires=6, skip_in=(4,),
                clamp=True):
    """ImplicitNetwork.get_outputs, network.py:105-123 -> sdf (P,1), feature (P,256), d sdf/dx (P,3).

    The gradient is the reverse-mode derivative of (clamped) sdf w.r.t. x, written out by hand
    (the reference gets it from torch.autograd.grad, :115-121).  `clamp=False` gives
    ImplicitNetwork.gradient (:90-103), which differentiates the raw output column 0.
    """
    x = np.asarray(x, F32)
    out, inp, cache = sdf_mlp_forward(layers, x, multires, skip_in, return_cache=True)
    n = len(layers)
    d_pe = inp.shape[1]
    g = np.zeros_like(out)
    g[:, 0] = 1.0                                  # d sdf / d a_8
    g_inp = np.zeros_like(inp)
    for l in range(n - 1, -1, -1):
        W, _ = layers[l]
        h, a = cache[l]
        if l < n - 1:
            g = (g * sigmoid100(a)).astype(F32)
        g = (g @ W).astype(F32)                    # grad wrt this layer's input
        if l in skip_in:
            g = (g / F32(np.sqrt(2.0))).astype(F32)
            g_inp += g[:, -d_pe:]
            g = g[:, :-d_pe]
    g_inp += g
    grad = posenc_vjp(x, g_inp, multires) if multires > 0 else g_inp
    sdf = out[:, :1]
    if clamp and sdf_bounding_sphere > 0.0:
        sdf_c, sphere = sdf_clamp(sdf, x, sdf_bounding_sphere, sphere_scale)
        nrm = np.sqrt((x * x).sum(1, keepdims=True, dtype=F32)).astype(F32)
        use_sphere = sphere < sdf                  # torch.minimum: ties keep grad of both halves /2; measure-zero
        grad = np.where(use_sphere, (-F32(sphere_scale) * x / nrm).astype(F32), grad)
        sdf = sdf_c
    return sdf.astype(F32), out[:, 1:].astype(F32), grad.astype(F32)


# ----------------------------------------------------------------------------------------
# a6  radiance MLP     volsdf/model/network.py:170-190
# ----------------------------------------------------------------------------------------
def rgb_mlp_forward(layers, points, normals, view_dirs, feat, mode="idr", multires_view=1):
    vd = posenc(view_dirs, multires_view) if multires_view > 0 else np.asarray(view_dirs, F32)
    if mode == "idr":
        h = np.concatenate([points, vd, normals, feat], -1).astype(F32)
    else:
        h = np.concatenate([vd, feat], -1).astype(F32)
    n = len(layers)
    for l, (W, b) in enumerate(layers):
        h = (h @ W.T + b).astype(F32)
        if l < n - 1:
            h = np.maximum(h, F32(0))
    return (1.0 / (1.0 + np.exp(-h.astype(F64)))).astype(F32)


# ----------------------------------------------------------------------------------------
# a7  Laplace density     volsdf/model/density.py:21-30
# ----------------------------------------------------------------------------------------
def get_beta(beta_param, beta_min=1e-4):
    return F32(abs(F32(beta_param)) + F32(beta_min))


def laplace_density(sdf, beta):
    """alpha*(0.5 + 0.5*sign(s)*expm1(-|s|/beta)), alpha = 1/beta; beta scalar or broadcastable."""
    sdf = np.asarray(sdf, F32)
    beta = np.asarray(beta, F32)
    alpha = (F32(1.0) / beta).astype(F32)
    e = ref_expm1((-np.abs(sdf) / beta).astype(F32))
    return (alpha * (F32(0.5) + F32(0.5) * np.sign(sdf).astype(F32) * e)).astype(F32)


# ----------------------------------------------------------------------------------------
# a2  uniform sampler     volsdf/model/ray_sampler.py:22-43
# ----------------------------------------------------------------------------------------
def linspace32(start, end, n):
    """torch.linspace(start,end,n) in float32: step=(end-start)/(n-1) (float32);
    fma(step, i, start) for i < n/2, fma(-step, n-1-i, end) otherwise (ATen RangeFactories)."""
    step = (F32(end) - F32(start)) / F32(n - 1)
    i = np.arange(n)
    lo = _fma32(step, i.astype(F32), F32(start))
    hi = _fma32(-step, (n - 1 - i).astype(F32), F32(end))
    return np.where(i < n // 2, lo, hi).astype(F32)


def linspace01(n):
    return linspace32(0.0, 1.0, n)


def uniform_z(near, far, n_samples, t_rand=None):
    """near (R,1)/scalar, far (R,1)/scalar; t_rand (R,n) in train mode (the torch.rand draw of :39)."""
    t = linspace01(n_samples)[None, :]
    near = np.asarray(near, F32).reshape(-1, 1)
    far = np.asarray(far, F32).reshape(-1, 1)
    z = (near * (F32(1.0) - t) + far * t).astype(F32)
    if t_rand is not None:
        if z.shape[0] == 1:
            z = np.repeat(z, t_rand.shape[0], 0)
        mids = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
        upper = np.concatenate([mids, z[:, -1:]], -1)
        lower = np.concatenate([z[:, :1], mids], -1)
        z = (lower + (upper - lower) * np.asarray(t_rand, F32)).astype(F32)
    return z


# ----------------------------------------------------------------------------------------
# a3/a4  error-bounded sampler     volsdf/model/ray_sampler.py:67-229
# ----------------------------------------------------------------------------------------
def d_star_bound(z, sdf):
    """ray_sampler.py:97-111.  z, sdf (R,n) -> dists (R,n-1), d_star (R,n-1)."""
    dists = (z[:, 1:] - z[:, :-1]).astype(F32)
    a, b, c = dists, np.abs(sdf[:, :-1]), np.abs(sdf[:, 1:])
    a2, b2, c2 = (a * a).astype(F32), (b * b).astype(F32), (c * c).astype(F32)
    first = (a2 + b2).astype(F32) <= c2
    second = (a2 + c2).astype(F32) <= b2
    d_star = np.zeros_like(dists)
    d_star[first] = b[first]
    d_star[second] = c[second]
    s = ((a + b + c).astype(F32) / F32(2.0)).astype(F32)
    area = (((s * (s - a)).astype(F32) * (s - b)).astype(F32) * (s - c)).astype(F32)
    mask = ~first & ~second & ((b + c).astype(F32) - a > 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        tri = ((F32(2.0) * ref_sqrt(area)).astype(F32) / a).astype(F32)
    d_star[mask] = tri[mask]
    same_sign = (np.sign(sdf[:, 1:]) * np.sign(sdf[:, :-1])) == 1
    return dists, (same_sign.astype(F32) * d_star).astype(F32)


def error_bound(beta, sdf, dists, d_star):
    """ErrorBoundSampler.get_error_bound, ray_sampler.py:221-229.  beta scalar or (R,1). -> (R,)"""
    beta = np.asarray(beta, F32)
    density = laplace_density(sdf, beta)
    sfe = np.concatenate([np.zeros((dists.shape[0], 1), F32), (dists * density[:, :-1]).astype(F32)], -1)
    integral = canon_cumsum(sfe)
    with np.errstate(over="ignore", invalid="ignore"):
        eps_sec = ((ref_exp((-d_star / beta).astype(F32)) * (dists * dists).astype(F32)).astype(F32)
                   / (F32(4.0) * (beta * beta).astype(F32)).astype(F32)).astype(F32)
        err_int = canon_cumsum(eps_sec)
        bound = ((np.minimum(ref_exp(err_int), F32(1.0e6)) - F32(1.0)).astype(F32)
                 * ref_exp(-integral[:, :-1])).astype(F32)
    return bound.max(-1)


def ray_weights(z, sdf, beta, last_dist=1e10):
    """ray_sampler.py:126-132 / network.py:281-295: density -> free energy -> alpha, T, weights."""
    density = laplace_density(sdf, beta)
    dists = np.concatenate([(z[:, 1:] - z[:, :-1]).astype(F32),
                            np.full((z.shape[0], 1), last_dist, F32)], -1)
    fe = (dists * density).astype(F32)
    sfe = np.concatenate([np.zeros((z.shape[0], 1), F32), fe[:, :-1]], -1)
    alpha = (F32(1.0) - ref_exp(-fe)).astype(F32)
    trans = ref_exp(-canon_cumsum(sfe))
    return (alpha * trans).astype(F32), trans, dists


def searchsorted_right(cdf, u):
    """torch.searchsorted(cdf, u, right=True) rowwise: number of cdf entries <= u."""
    return (cdf[:, None, :] <= u[:, :, None]).sum(-1).astype(np.int64)


def inverse_cdf(cdf, bins, u):
    """ray_sampler.py:173-185 -> samples (R,N), inds (R,N) int64."""
    n = cdf.shape[-1]
    inds = searchsorted_right(cdf, u)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, n - 1)
    cdf_b = np.take_along_axis(cdf, below, 1)
    cdf_a = np.take_along_axis(cdf, above, 1)
    bin_b = np.take_along_axis(bins, below, 1)
    bin_a = np.take_along_axis(bins, above, 1)
    denom = (cdf_a - cdf_b).astype(F32)
    denom = np.where(denom < F32(1e-5), F32(1.0), denom)
    t = ((u - cdf_b).astype(F32) / denom).astype(F32)
    return (bin_b + (t * (bin_a - bin_b).astype(F32)).astype(F32)).astype(F32), inds


def stable_merge(z, samples):
    """torch.sort(cat[z, samples]) values + indices with ties resolved to the lower index."""
    cat = np.concatenate([z, samples], -1)
    idx = np.argsort(cat, axis=-1, kind="stable")
    return np.take_along_axis(cat, idx, -1), idx.astype(np.int64)


def extras_index_eval(n, n_extra):
    """torch.linspace(0, n-1, n_extra).long()  (ray_sampler.py:203), float32 linspace then truncation."""
    return linspace32(0.0, float(n - 1), n_extra).astype(np.int64)


def sampler_round(z, sdf, beta_in, beta0, *, upsample_allowed, training=False, u_final=None,
                  N_samples=64, N_samples_eval=128, eps=0.1, beta_iters=10, add_tiny=0.0):
    """One pass of the `while` body of ErrorBoundSampler.get_z_vals after the sdf merge
    (ray_sampler.py:96-190): d*, beta line search, weights, convergence test, pdf/cdf, inverse CDF and,
    when up-sampling, the merge-sort.

    z, sdf (R,n): current bins and their sdf; beta_in (R,): beta carried over from the previous round
    (the Lemma-2 bound on the first); upsample_allowed = (total_iters + 1 < max_total_iters).
    Returns a dict (see keys below); 'z_next'/'samples_idx' only when up-sampling.
    """
    R = z.shape[0]
    beta0 = F32(beta0)
    dists, d_star = d_star_bound(z, sdf)

    curr = error_bound(beta0, sdf, dists, d_star)
    beta = np.asarray(beta_in, F32).copy()
    beta[curr <= F32(eps)] = beta0
    beta_min, beta_max = np.full(R, beta0, F32), beta
    for _ in range(beta_iters):
        mid = ((beta_min + beta_max) / F32(2.0)).astype(F32)
        curr = error_bound(mid[:, None], sdf, dists, d_star)
        beta_max = np.where(curr <= F32(eps), mid, beta_max).astype(F32)
        beta_min = np.where(curr > F32(eps), mid, beta_min).astype(F32)
    beta = beta_max

    weights, trans, _ = ray_weights(z, sdf, beta[:, None])

    not_converge = bool(beta.max() > beta0)          # batch-global, ray_sampler.py:136
    upsample = not_converge and upsample_allowed
    if upsample:
        N = N_samples_eval
        with np.errstate(over="ignore", invalid="ignore"):
            b = beta[:, None]
            eps_sec = ((ref_exp((-d_star / b).astype(F32)) * (dists * dists).astype(F32)).astype(F32)
                       / (F32(4.0) * (b * b).astype(F32)).astype(F32)).astype(F32)
            err_int = canon_cumsum(eps_sec)
            bound_op = ((np.minimum(ref_exp(err_int), F32(1.0e6)) - F32(1.0)).astype(F32)
                        * trans[:, :-1]).astype(F32)
        pdf = (bound_op + F32(add_tiny)).astype(F32)
    else:
        N = N_samples
        pdf = (weights[:, :-1] + F32(1e-5)).astype(F32)
    pdf = (pdf / ref_sum(pdf)).astype(F32)
    cdf = np.concatenate([np.zeros((R, 1), F32), canon_cumsum(pdf)], -1)

    if upsample or not training:
        u = np.repeat(linspace01(N)[None], R, 0)
    else:
        u = np.asarray(u_final, F32)
    samples, inds = inverse_cdf(cdf, z, u)

    rec = dict(n=z.shape[1], z=z, sdf=sdf, d_star=d_star, beta=beta, weights=weights, pdf=pdf, cdf=cdf, u=u,
               inds=inds, samples=samples, upsample=upsample, not_converge=not_converge)
    if upsample:
        rec["z_next"], rec["samples_idx"] = stable_merge(z, samples)
    return rec


def error_bound_sampler(sdf_fn, ray_dirs, cam_loc, beta0, *, near=1e-4, scene_bounding_sphere=3.0,
                        N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                        max_total_iters=5, fast=-1, training=False, inverse_sphere_bg=False,
                        N_samples_inverse_sphere=0, add_tiny=0.0, rng=None, inv_4log=None, trace=None,
                        sdf_override=None):
    """ErrorBoundSampler.get_z_vals, ray_sampler.py:67-219.

    sdf_fn(points (P,3)) -> (P,1) is `model.implicit_network.get_sdf_vals`.
    rng: dict of the train-mode draws (SURVEY note R): 'jitter' (R,128), 'u' (R,64),
         'perm' (n,) randperm of the final bin count, 'eik_idx' (R,), ['jitter_bg' (R,32)].
    inv_4log: float32 1/(4*log(1+eps)) as computed by the host (torch.log in float32, :77).
    sdf_override: optional list of per-round (R*128,1) sdf arrays used INSTEAD of sdf_fn, so the
         sampler can be replayed on fixed sdf inputs (bit-exact index tests).
    trace: optional list that receives one dict per round (see sampler_round).
    Returns z_vals (R, N_samples+N_samples_extra+2) [BG: (z_vals, z_bg)], z_samples_eik (R,1).
    """
    ray_dirs = np.asarray(ray_dirs, F32)
    cam_loc = np.asarray(cam_loc, F32)
    if cam_loc.ndim == 1:
        cam_loc = np.repeat(cam_loc[None], ray_dirs.shape[0], 0)
    R = ray_dirs.shape[0]
    rng = rng or {}
    beta0 = F32(beta0)
    far_default = F32(2.0 * scene_bounding_sphere)
    max_iters = fast if fast >= 0 else max_total_iters

    if inverse_sphere_bg:
        far_u = sphere_intersections(cam_loc, ray_dirs, scene_bounding_sphere)[:, 1:]
    else:
        far_u = far_default
    z = uniform_z(F32(near), far_u, N_samples_eval, rng.get("jitter") if training else None)
    if z.shape[0] == 1:
        z = np.repeat(z, R, 0)
    samples, samples_idx = z, None

    dists = (z[:, 1:] - z[:, :-1]).astype(F32)
    if inv_4log is None:
        inv_4log = F32(1.0) / (F32(4.0) * F32(math.log(F32(eps) + F32(1.0))))
    bound = (F32(inv_4log) * ref_sum((dists * dists).astype(F32))[:, 0]).astype(F32)
    beta = ref_sqrt(bound)

    total_iters, not_converge = 0, True
    sdf = None
    while not_converge and total_iters < max_iters:
        points = (cam_loc[:, None, :] + samples[:, :, None] * ray_dirs[:, None, :]).astype(F32)
        if sdf_override is not None:
            samples_sdf = np.asarray(sdf_override[total_iters], F32).reshape(-1, 1)
        else:
            samples_sdf = np.asarray(sdf_fn(points.reshape(-1, 3)), F32)
        if samples_idx is not None:
            sdf_merge = np.concatenate([sdf.reshape(R, z.shape[1] - samples.shape[1]),
                                        samples_sdf.reshape(R, samples.shape[1])], -1)
            sdf = np.take_along_axis(sdf_merge, samples_idx, 1)
        else:
            sdf = samples_sdf.reshape(R, -1)

        total_iters += 1
        rec = sampler_round(z, sdf, beta, beta0, upsample_allowed=total_iters < max_iters, training=training,
                            u_final=rng.get("u"), N_samples=N_samples, N_samples_eval=N_samples_eval, eps=eps,
                            beta_iters=beta_iters, add_tiny=add_tiny)
        rec["samples_sdf"] = samples_sdf.reshape(R, -1)      # sdf of this round's new samples (sdf_fn output)
        rec["beta_in"] = beta
        beta, samples, not_converge = rec["beta"], rec["samples"], rec["not_converge"]
        if rec["upsample"]:
            z, samples_idx = rec["z_next"], rec["samples_idx"]
        if trace is not None:
            trace.append(rec)

    z_final, z_eik = sampler_finalize(z, samples, near=near, far=far_default, N_samples_extra=N_samples_extra,
                                      training=training, rng=rng,
                                      far_rays=(sphere_intersections(cam_loc, ray_dirs, scene_bounding_sphere)[:, 1:]
                                                if inverse_sphere_bg else None))
    if inverse_sphere_bg:
        z_bg = uniform_z(F32(0.0), F32(1.0), N_samples_inverse_sphere,
                         rng.get("jitter_bg") if training else None)
        if z_bg.shape[0] == 1:
            z_bg = np.repeat(z_bg, R, 0)
        z_bg = (z_bg * F32(1.0 / scene_bounding_sphere)).astype(F32)
        return (z_final, z_bg), z_eik
    return z_final, z_eik


def sampler_finalize(z, z_samples, *, near, far, N_samples_extra=32, training=False, rng=None, far_rays=None):
    """ray_sampler.py:192-212: extras (near, far, N_samples_extra bins), final sort, eikonal sample pick."""
    R = z.shape[0]
    rng = rng or {}
    near_c = np.full((R, 1), near, F32)
    far_c = np.full((R, 1), far, F32) if far_rays is None else np.asarray(far_rays, F32)
    if N_samples_extra > 0:
        if training:
            sampling_idx = np.asarray(rng["perm"])[:N_samples_extra]
        else:
            sampling_idx = extras_index_eval(z.shape[1], N_samples_extra)
        z_extra = np.concatenate([near_c, far_c, z[:, sampling_idx]], -1)
    else:
        z_extra = np.concatenate([near_c, far_c], -1)
    z_final = np.sort(np.concatenate([z_samples, z_extra], -1), -1).astype(F32)
    if training and "eik_idx" in rng:
        idx = np.asarray(rng["eik_idx"]).reshape(-1, 1)
    else:
        idx = np.zeros((R, 1), np.int64)      # the eval-mode draw is unused downstream (network.py:258)
    return z_final, np.take_along_axis(z_final, idx, 1)


# ----------------------------------------------------------------------------------------
# a8  compositing    volsdf/model/network.py:281-295 and the reductions :237-256,270-276
# ----------------------------------------------------------------------------------------
def composite(z, sdf, rgb, beta, depth_scale, normals=None):
    """z (R,S), sdf (R,S), rgb (R,S,3), beta scalar, depth_scale (R,1)."""
    weights, _, _ = ray_weights(z, sdf, F32(beta))
    out = {"weights": weights}
    out["rgb_values"] = (weights[:, :, None] * rgb).sum(1, dtype=F32)
    wsum = weights.sum(1, keepdims=True, dtype=F32)
    out["depth_values"] = (depth_scale * ((weights * z).sum(1, keepdims=True, dtype=F32)
                                          / (wsum + F32(1e-8)))).astype(F32)
    out["depth_vals"] = (z * depth_scale).astype(F32)
    if normals is not None:
        nrm = np.sqrt((normals * normals).sum(-1, keepdims=True, dtype=F32)).astype(F32)
        out["normal_map"] = (weights[:, :, None] * (normals / nrm)).sum(1, dtype=F32)
    return out


# ----------------------------------------------------------------------------------------
# a9  VolSDFNetwork.forward    volsdf/model/network.py:206-279
# ----------------------------------------------------------------------------------------
def render_forward(params, uv, pose, K, *, beta_param, fast=-1, training=False, rng=None,
                   sampler_conf=None, scene_bounding_sphere=3.0, sphere_scale=20.0, trace=None):
    """Full forward of the fg-only DTU model.  params: state-dict-named arrays.  Returns dict like the reference."""
    sampler_conf = dict(sampler_conf or {})
    sdf_layers = effective_weights(params, "implicit_network", 9)
    rgb_layers = effective_weights(params, "rendering_network", 5)
    beta = get_beta(beta_param)
    dirs, cam, depth_scale = rays_from_uv(uv, pose, K)
    R = dirs.shape[0]
    cam_r = np.repeat(cam[None], R, 0)
    sdf_fn = lambda p: sdf_vals(sdf_layers, p, scene_bounding_sphere, sphere_scale)
    z, z_eik = error_bound_sampler(sdf_fn, dirs, cam_r, beta, fast=fast, training=training, rng=rng,
                                   scene_bounding_sphere=scene_bounding_sphere, trace=trace, **sampler_conf)
    S = z.shape[1]
    points = (cam_r[:, None, :] + z[:, :, None] * dirs[:, None, :]).astype(F32)
    pf = points.reshape(-1, 3)
    sdf, feat, grad = sdf_outputs(sdf_layers, pf, scene_bounding_sphere, sphere_scale)
    dirs_flat = np.repeat(dirs[:, None, :], S, 1).reshape(-1, 3)
    rgb = rgb_mlp_forward(rgb_layers, pf, grad, dirs_flat, feat).reshape(R, S, 3)
    out = composite(z, sdf.reshape(R, S), rgb, beta, depth_scale,
                    normals=None if training else grad.reshape(R, S, 3))
    out["xyz"] = points
    out["z_vals"] = z
    out["sdf"] = sdf.reshape(R, S)
    out["gradients"] = grad.reshape(R, S, 3)
    if training:
        eik_uniform = np.asarray(rng["eik_points"], F32)
        eik_near = (cam_r[:, None, :] + z_eik[:, :, None] * dirs[:, None, :]).reshape(-1, 3).astype(F32)
        eik = np.concatenate([eik_uniform, eik_near], 0)
        out["grad_theta"] = sdf_outputs(sdf_layers, eik, scene_bounding_sphere, sphere_scale, clamp=False)[2]
    return out


# ----------------------------------------------------------------------------------------
# a9 (BG)  VolSDFNetworkBG.forward    volsdf/model/network_bg.py:37-214
# ----------------------------------------------------------------------------------------
def depth2pts_outside(ray_o, ray_d, depth, r):
    """network_bg.py:182-214 (NeRF++ inverted sphere): ray_o, ray_d (..., 3), depth (...) = 1 / distance to the
    origin in [0, 1/r]  ->  pts (..., 4) = (unit direction of the point, depth), depth_real (...)."""
    ray_o, ray_d, depth = np.asarray(ray_o, F32), np.asarray(ray_d, F32), np.asarray(depth, F32)
    o_dot_d = (ray_d * ray_o).sum(-1, dtype=F32)
    under_sqrt = (o_dot_d ** 2 - ((ray_o ** 2).sum(-1, dtype=F32) - F32(r) ** 2)).astype(F32)
    d_sphere = (np.sqrt(under_sqrt) - o_dot_d).astype(F32)
    p_sphere = (ray_o + d_sphere[..., None] * ray_d).astype(F32)
    p_mid = (ray_o - o_dot_d[..., None] * ray_d).astype(F32)
    p_mid_norm = norm3(p_mid)[..., 0]
    rot_axis = np.cross(ray_o, p_sphere).astype(F32)
    rot_axis = (rot_axis / norm3(rot_axis)).astype(F32)
    phi = np.arcsin(p_mid_norm / F32(r)).astype(F32)
    theta = np.arcsin(p_mid_norm * depth).astype(F32)
    rot_angle = (phi - theta)[..., None].astype(F32)
    c, sn = np.cos(rot_angle).astype(F32), np.sin(rot_angle).astype(F32)
    p_new = (p_sphere * c + np.cross(rot_axis, p_sphere).astype(F32) * sn
             + rot_axis * (rot_axis * p_sphere).sum(-1, keepdims=True, dtype=F32) * (F32(1.0) - c)).astype(F32)
    p_new = (p_new / norm3(p_new)).astype(F32)
    pts = np.concatenate([p_new, depth[..., None]], -1).astype(F32)
    d1 = (-o_dot_d / (ray_d * ray_d).sum(-1, dtype=F32)).astype(F32)
    ray_d_cos = (F32(1.0) / norm3(ray_d)[..., 0]).astype(F32)
    depth_real = (F32(1.0) / (depth + F32(1e-6)) * np.cos(theta).astype(F32) * ray_d_cos + d1).astype(F32)
    return pts, depth_real


def fg_weights_bg_model(z, z_max, sdf, beta):
    """VolSDFNetworkBG.volume_rendering, network_bg.py:147-164: the last interval ends at the sphere exit z_max;
    returns weights (R,S), bg_transmittance (R,) = transmittance behind the last sample."""
    sigma = laplace_density(sdf, F32(beta))
    dists = np.concatenate([z[:, 1:] - z[:, :-1], z_max[:, None] - z[:, -1:]], -1).astype(F32)
    free = (dists * sigma).astype(F32)
    shifted = np.concatenate([np.zeros((z.shape[0], 1), F32), free], -1)
    alpha = (F32(1.0) - ref_exp(-free)).astype(F32)
    trans = ref_exp(-canon_cumsum(shifted))
    return (alpha * trans[:, :-1]).astype(F32), trans[:, -1].astype(F32), dists


def bg_weights(z_bg, bg_sigma):
    """bg_volume_rendering, network_bg.py:166-180; z_bg (R,N) descending inverse depths, bg_sigma (R,N) = |sdf|."""
    dists = np.concatenate([z_bg[:, :-1] - z_bg[:, 1:], np.full((z_bg.shape[0], 1), 1e10, F32)], -1).astype(F32)
    free = (dists * bg_sigma).astype(F32)
    shifted = np.concatenate([np.zeros((z_bg.shape[0], 1), F32), free[:, :-1]], -1)
    alpha = (F32(1.0) - ref_exp(-free)).astype(F32)
    trans = ref_exp(-canon_cumsum(shifted))
    return (alpha * trans).astype(F32)


def render_forward_bg(params, uv, pose, K, *, beta_param, fast=-1, training=False, rng=None, near_pose=None,
                      scene_bounding_sphere=3.0, trace=None):
    """VolSDFNetworkBG.forward (network_bg.py:37-145) with the bmvs.yaml configuration."""
    r = scene_bounding_sphere
    sdf_layers = effective_weights(params, "implicit_network", 9)
    rgb_layers = effective_weights(params, "rendering_network", 5)
    bg_sdf_layers = effective_weights(params, "bg_implicit_network", 9)
    bg_rgb_layers = effective_weights(params, "bg_rendering_network", 2)
    beta = get_beta(beta_param)
    dirs, cam, depth_scale = rays_from_uv(uv, pose, K)
    R = dirs.shape[0]
    cam_r = np.repeat(cam[None], R, 0)
    sdf_fn = lambda p: sdf_vals(sdf_layers, p, 0.0, 1.0)
    (z_all, z_bg), z_eik = error_bound_sampler(sdf_fn, dirs, cam_r, beta, near=0.0, scene_bounding_sphere=r, fast=fast,
                                               training=training, rng=rng, inverse_sphere_bg=True,
                                               N_samples_inverse_sphere=32, add_tiny=1e-6, trace=trace)
    z_max, z = z_all[:, -1], z_all[:, :-1]
    S = z.shape[1]
    points = (cam_r[:, None, :] + z[:, :, None] * dirs[:, None, :]).astype(F32)
    pf = points.reshape(-1, 3)
    sdf, feat, grad = sdf_outputs(sdf_layers, pf, 0.0, 1.0)
    view = dirs
    if not training:
        view = rays_from_uv(uv, near_pose, K)[0]
    rgb = rgb_mlp_forward(rgb_layers, pf, grad, np.repeat(view[:, None, :], S, 1).reshape(-1, 3), feat).reshape(R, S, 3)
    weights, bg_trans, _ = fg_weights_bg_model(z, z_max, sdf.reshape(R, S), beta)
    fg_rgb = (weights[:, :, None] * rgb).sum(1, dtype=F32)
    # background
    z_bg = z_bg[:, ::-1].copy()                      # 1 -> 0
    Nb = z_bg.shape[1]
    bg_pts, bg_depth = depth2pts_outside(np.repeat(cam_r[:, None, :], Nb, 1), np.repeat(dirs[:, None, :], Nb, 1), z_bg, r)
    bg_out = sdf_mlp_forward(bg_sdf_layers, bg_pts.reshape(-1, 4), multires=10)
    bg_sigma = np.abs(bg_out[:, :1]).astype(F32)
    bg_rgb = rgb_mlp_forward(bg_rgb_layers, None, None, np.repeat(view[:, None, :], Nb, 1).reshape(-1, 3), bg_out[:, 1:],
                             mode="nerf", multires_view=4).reshape(R, Nb, 3)
    bw = bg_weights(z_bg, bg_sigma.reshape(R, Nb))
    bg_rgb_values = (bw[:, :, None] * bg_rgb).sum(1, dtype=F32)
    weights_all = np.concatenate([weights, bg_trans[:, None] * bw], 1).astype(F32)
    depth_vals_all = (depth_scale * np.concatenate([z, bg_depth], 1)).astype(F32)
    out = {
        "rgb_values": (fg_rgb + bg_trans[:, None] * bg_rgb_values).astype(F32),
        "depth_values_all": ((weights_all * depth_vals_all).sum(1, keepdims=True, dtype=F32)
                             / (weights_all.sum(1, keepdims=True, dtype=F32) + F32(1e-8))).astype(F32),
        "depth_vals": (z * depth_scale).astype(F32),
        "weights": weights,
        "xyz": points,
    }
    out["depth_values"] = ((weights * out["depth_vals"]).sum(1, keepdims=True, dtype=F32)
                           / (weights.sum(1, keepdims=True, dtype=F32) + F32(1e-8))).astype(F32)
    out.update(z_vals=z, z_max=z_max, z_bg=z_bg, bg_points=bg_pts, bg_depth=bg_depth, bg_sigma=bg_sigma.reshape(R, Nb),
               bg_rgb=bg_rgb, bg_weights=bw, bg_transmittance=bg_trans, sdf=sdf.reshape(R, S))
    if training:
        eik_near = (cam_r[:, None, :] + z_eik[:, :, None] * dirs[:, None, :]).reshape(-1, 3).astype(F32)
        eik = np.concatenate([np.asarray(rng["eik_points"], F32), eik_near], 0)
        out["grad_theta"] = sdf_outputs(sdf_layers, eik, 0.0, 1.0, clamp=False)[2]
    else:
        nrm = np.sqrt((grad * grad).sum(-1, keepdims=True, dtype=F32)).astype(F32)
        out["normal_map"] = (weights[:, :, None] * (grad / nrm).reshape(R, S, 3)).sum(1, dtype=F32)
    return out


# ----------------------------------------------------------------------------------------
# a10  MVS prior lookup     volsdf/vsdf.py:382-452 (VolOpt.cost_mapping)
# ----------------------------------------------------------------------------------------
def _grid_sample_zeros(vol, coords):
    """F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) for one channel.
    vol: (H,W) or (D,H,W); coords: (...,2) as (x,y) or (...,3) as (x,y,z), normalised to [-1,1]."""
    nd = vol.ndim
    sizes = vol.shape[::-1]                       # (W,H[,D]) matches coord order x,y[,z]
    pix = [((coords[..., a] + F32(1.0)) / F32(2.0) * F32(sizes[a] - 1)).astype(F32) for a in range(nd)]
    base = [np.floor(p) for p in pix]
    frac = [(p - b).astype(F32) for p, b in zip(pix, base)]
    out = np.zeros(coords.shape[:-1], F32)
    for corner in range(1 << nd):
        w = np.ones(coords.shape[:-1], F32)
        idx = []
        ok = np.ones(coords.shape[:-1], bool)
        for a in range(nd):
            hi = (corner >> a) & 1
            ia = base[a] + hi
            w = (w * (frac[a] if hi else (F32(1.0) - frac[a]))).astype(F32)
            ok &= (ia >= 0) & (ia <= sizes[a] - 1)
            idx.append(np.clip(np.nan_to_num(ia, nan=0.0, posinf=0.0, neginf=0.0), 0, sizes[a] - 1).astype(np.int64))
        vals = vol[tuple(idx[::-1])]
        out += np.where(ok, w * vals, F32(0.0)).astype(F32)
    return out


def cost_mapping(xyz, view_index, views, img_res, inverse_depth=False):
    """VolOpt.cost_mapping, vsdf.py:382-452.

    xyz (R,S,3) world points; view_index: which entry of `views` the batch was rendered from;
    views: list of dicts(K (4,4), c2w (4,4), cost (D,Hc,Wc) probability volume, z_mvs (D,Hc,Wc) depth hypotheses);
    img_res (H,W) of the SceneDataset.  Returns pj (R,S), pi (R,S), valid (R,S) bool.
    """
    R, S, _ = xyz.shape
    _h, _w = img_res
    pj = np.zeros((R, S), F32)
    pi = np.zeros((R, S), F32)
    valid = np.zeros((R, S), bool)
    with np.errstate(all="ignore"):
        for i, v in enumerate(views):
            K, c2w = np.asarray(v["K"], F32), np.asarray(v["c2w"], F32)[:3]
            fx, fy, cx, cy, sk = K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1]
            p = (xyz - c2w[:, 3].reshape(1, 1, 3)).astype(F32)
            p = (p @ c2w[:, :3]).astype(F32)
            z = p[..., 2]
            x = (p[..., 0] / z).astype(F32)
            y = (p[..., 1] / z).astype(F32)
            y = (y * fy + cy).astype(F32)
            x = (x * fx + cx + (y - cy) * sk / fy).astype(F32)
            x = (x / F32((_w - 1) / 2) - F32(1.0)).astype(F32)
            y = (y / F32((_h - 1) / 2) - F32(1.0)).astype(F32)
            inval = (z < 1e-5) | (x > 1.001) | (x < -1.001) | (y > 1.001) | (y < -1.001)
            x = np.where(inval, F32(-99.0), x)
            y = np.where(inval, F32(-99.0), y)
            z = np.where(inval, F32(-99.0), z)
            xy = np.stack([x, y], -1)
            near = _grid_sample_zeros(np.asarray(v["z_mvs"][0], F32), xy)
            far = _grid_sample_zeros(np.asarray(v["z_mvs"][-1], F32), xy)
            if inverse_depth:
                far = np.where(inval, F32(1e-8), far)
                zn = (F32(2.0) * (F32(1.0) - near / z) / (F32(1.0) - near / far) - F32(1.0)).astype(F32)
            else:
                zn = (F32(2.0) * (z - near) / (far - near) - F32(1.0)).astype(F32)
            inval = (near < 1e-5) | (far < 1e-5) | (zn > 1.01) | (zn < -1.01) | inval
            x = np.where(inval, F32(-99.0), x)
            y = np.where(inval, F32(-99.0), y)
            zn = np.where(inval, F32(-99.0), zn)
            cost = _grid_sample_zeros(np.asarray(v["cost"], F32), np.stack([x, y, zn], -1))
            if i == view_index:
                pi = cost
            else:
                pj = (pj + cost).astype(F32)
                valid |= ~inval
    pi = np.where(valid, pi, F32(0.0)).astype(F32)
    return pj, pi, valid


# ----------------------------------------------------------------------------------------
# a11  loss     volsdf/model/loss.py:80-114 (VolSDFLoss.forward)
# ----------------------------------------------------------------------------------------
def volsdf_loss(out, rgb_gt, rgb_smooth, iter_step, *, eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0,
                sparse_weight=1.0, anneal_rgb=200, gce=0.5, confi=1e-3):
    """Returns dict(rgb_loss, eikonal_loss, mvs_loss, sparse_loss, loss) as float32 scalars.
    out: dict with rgb_values (R,3), grad_theta (2R,3) [optional], weights, pi, pj (R,S) [optional], depth_values (R,1)."""
    res = {}
    rgb_gt = np.asarray(rgb_gt, F32).reshape(-1, 3)
    res["rgb_loss"] = np.abs(out["rgb_values"] - rgb_gt).mean(dtype=F64).astype(F32)
    if "grad_theta" in out:
        nrm = np.sqrt((out["grad_theta"].astype(F64) ** 2).sum(1))
        res["eikonal_loss"] = (((nrm - 1.0) ** 2).mean()).astype(F32)
    else:
        res["eikonal_loss"] = F32(0.0)
    has_mvs = "pi" in out
    if has_mvs and mvs_weight > 0:
        pw = (out["pi"] * out["pj"]).astype(F64)
        w = out["weights"].astype(F64)
        if gce == 1:
            l = -pw * w
        elif gce == 0:
            l = -pw * np.log(w + 1e-8)
        else:
            l = -pw * w ** gce * np.log(w + 1e-8)
        l = l.sum(1) * (pw.sum(1) > confi)
        res["mvs_loss"] = l.mean().astype(F32)
    else:
        res["mvs_loss"] = F32(0.0)
    anneal_on = sparse_weight > 0 and anneal_rgb > 0 and iter_step < anneal_rgb
    if has_mvs and anneal_on:
        conf = (out["pi"] * out["pj"]).astype(F64).sum(-1)
        dep = out.get("depth_values_all", out["depth_values"]).astype(F64).reshape(-1)
        res["sparse_loss"] = ((1.0 / (dep + 1e-3)) * (conf < confi)).mean().astype(F32)
    else:
        res["sparse_loss"] = F32(0.0)
    anneal_sparse = 0.0
    if anneal_on:
        t = iter_step / anneal_rgb
        anneal_sparse = 0.0 if t >= 1 else (1.0 if t <= 0 else 1.0 + (0.0 - 1.0) * min(t, 1.0))
        conf = (out["pi"] * out["pj"]).astype(F64).sum(-1)
        l = np.abs(out["rgb_values"] - np.asarray(rgb_smooth, F32).reshape(-1, 3)).astype(F64).mean(-1)
        res["rgb_loss"] = (l * (conf < 1e-8)).mean().astype(F32)
    res["loss"] = F32(rgb_weight * res["rgb_loss"] + eikonal_weight * res["eikonal_loss"] +
                      mvs_weight * res["mvs_loss"] + sparse_weight * anneal_sparse * res["sparse_loss"])
    return res
