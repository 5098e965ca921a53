"""Orchestration of the training backward through the fused MLPs (csrc/svs_mlp_bwd.hip, svs_wgrad.hip).

Host code only sequences kernel launches on the current stream and owns the scratch buffers; every
arithmetic step is a HIP kernel.  The reference gets these gradients from torch.autograd
(volsdf/vsdf.py:215), including the double backward through network.py:115-121.
"""
import ctypes

import torch

from . import lib as _lib
from .ops import _f32, _ptr, _ptr_array, _stream

KBLOCK = 128 * 64           # floats per wave-tile activation block
RBUF = 4 * KBLOCK + 1024    # radiance forward activations per tile


def _off(t, n_floats):
    """device pointer `n_floats` floats into tensor t"""
    return ctypes.c_void_p(t.data_ptr() + 4 * n_floats)


class MlpBackward:
    """Scratch + packed training streams for one (ImplicitNetwork, RenderingNetwork) pair."""

    def __init__(self, device):
        L = _lib.load()
        self.dev = device
        self.sdf_stream = torch.empty(L.svs_stream_bytes(2) // 4, device=device)
        self.rgb_stream = torch.empty(L.svs_stream_bytes(4) // 4, device=device)
        self.ws = torch.empty(L.svs_pack_workspace_bytes() // 4, device=device)
        self._n = None

    def _alloc(self, n_total, n_main):
        if self._n == (n_total, n_main):
            return
        L = _lib.load()
        z = lambda nbytes: torch.zeros(nbytes // 4, device=self.dev)
        self.zbuf = z(L.svs_rgb_zbuf_bytes(n_main))           # zero-initialised once: zbar_4 only writes its first tile
        self.feat_bar = z(L.svs_block_bytes(n_main, 1))
        self.ubuf = z(L.svs_sdf_ubuf_bytes(n_total))
        self.a2buf = z(L.svs_block_bytes(n_total, 8))
        self.abuf = z(L.svs_block_bytes(n_total, 8))
        self.pebuf = z(L.svs_block_bytes(n_total, 1))
        self.sbar = z(L.svs_block_bytes(n_total, 1) // (128 * 2))  # 32 floats per tile
        self.dWk = torch.empty(14, 256, 288, device=self.dev)
        self.dbk = torch.empty(14, 256, device=self.dev)
        self.row0 = torch.empty(257, device=self.dev)
        self._n = (n_total, n_main)

    def run(self, sdf_params, rgb_params, keep, d_rgb, d_sdf, d_grad_extra, out=None):
        """sdf_params / rgb_params: (weight_v list, weight_g list or None, bias list) of the two MLPs.
        keep: dict filled by ops.sdf_outputs / ops.rgb_eval (hbuf, gbuf, clamp_mask, src, rbuf, feat_tiles, rgb).
        d_rgb (n_main,3): dL/d rgb of the ray samples; d_sdf (n_main,1) or None; d_grad_extra (n_extra,3) or None:
        dL/d(d sdf/dx) of the extra (eikonal) points that follow the ray samples in the launch.
        out: optional (sdf_out, rgb_out) lists of (grad_v, grad_g, grad_b) tensors to write into (e.g. views of a flat
        gradient buffer) instead of allocating.
        Returns (sdf_grads, rgb_grads): lists of (grad_v, grad_g, grad_b) per layer."""
        L = _lib.load()
        src = keep["src"]
        n_total, n_main = src.n, keep["rgb"].shape[0]
        if n_main % 32:
            raise NotImplementedError("rays*samples must be a multiple of 32 (1024 x 98 is)")
        self._alloc(n_total, n_main)
        dev = self.dev
        sv, sg, sb = [[_f32(t) for t in x] if x is not None else None for x in sdf_params]
        rv, rg, rb = [[_f32(t) for t in x] if x is not None else None for x in rgb_params]
        st = _stream()
        _lib.check(L.svs_pack_stream(2, _ptr_array(sv), _ptr_array(sg) if sg else None, _ptr_array(sb), _ptr(self.ws),
                                     _ptr(self.sdf_stream), st), "svs_pack_stream(sdf train)")
        sdf_scale_ws = self.ws       # (pack workspace is reused; scales are recomputed inside unpack)
        _lib.check(L.svs_pack_stream(4, _ptr_array(rv), _ptr_array(rg) if rg else None, _ptr_array(rb), _ptr(self.ws),
                                     _ptr(self.rgb_stream), st), "svs_pack_stream(rgb bwd)")
        del sdf_scale_ws
        # ---- radiance MLP: input gradients
        d_rgb = _f32(d_rgb)
        d_normals = torch.empty(n_main, 3, device=dev)
        _lib.check(L.svs_rgb_bwd(n_main, _ptr(d_rgb), _ptr(keep["rgb"]), _ptr(keep["rbuf"]), _ptr(self.rgb_stream),
                                 _ptr(self.zbuf), _ptr(self.feat_bar), _ptr(d_normals), st), "svs_rgb_bwd")
        # ---- SDF MLP: pass A (needs nbar), pass B (needs sbar, fbar)
        d_grad = d_normals if d_grad_extra is None else torch.cat([d_normals, _f32(d_grad_extra)], 0)
        if d_grad.shape[0] != n_total:
            raise ValueError("d_grad_extra must cover the points that follow the ray samples")
        d_sdf_full = torch.zeros(n_total, device=dev)
        if d_sdf is not None:
            d_sdf_full[:n_main] = _f32(d_sdf).reshape(-1)
        hbuf, gbuf, mask = keep["hbuf"], keep["gbuf"], keep["clamp_mask"]
        # ---- weight gradients (kernel order), one GEMM over the points per layer.  The radiance GEMMs only need
        # rgb_bwd's outputs: they run on a side stream and fill the CUs the SDF sweeps leave idle in their tail round.
        H8, U9, A8 = 8 * KBLOCK, 9 * KBLOCK, 8 * KBLOCK
        rbuf, feat = keep["rbuf"], keep["feat_tiles"]

        def wgrad(slot, n_pts, a0, sa0, b0, sb0, a1=None, a1h=None, sa1=0, sh1=0, b1=None, sb1=0, extra=None, sx=0):
            _lib.check(L.svs_wgrad(a0, None, b0, sa0, 0, sb0, a1, a1h, b1, sa1, sh1, sb1, extra, sx, n_pts,
                                   _off(self.dWk, slot * 256 * 288), 288, _off(self.dbk, slot * 256), _stream()),
                       "svs_wgrad")

        def unpack(slot, mp, rows, cols, row_off, v, g, row0=None, dst=None):
            if dst is not None:
                gv, gg, gb = dst
            else:
                gv = torch.empty(rows, cols, device=dev)
                gg = torch.empty(rows, 1, device=dev) if g is not None else None
                gb = torch.empty(rows, device=dev)
            _lib.check(L.svs_unpack_wgrad(_off(self.dWk, slot * 256 * 288), _off(self.dbk, slot * 256), 288, mp, rows,
                                          cols, row_off, _ptr(v), _ptr(g), row0, _ptr(gv), _ptr(gg), _ptr(gb), _stream()),
                       "svs_unpack_wgrad")
            return gv, gg, gb

        return_rgb = []

        def radiance_side():
            wgrad(9, n_main, _off(self.zbuf, 0), 5 * KBLOCK, _ptr(feat), KBLOCK, extra=_off(rbuf, 4 * KBLOCK), sx=RBUF)
            for l in range(1, 5):
                wgrad(9 + l, n_main, _off(self.zbuf, l * KBLOCK), 5 * KBLOCK, _off(rbuf, (l - 1) * KBLOCK), RBUF)
            for l in range(5):
                rows, cols = rv[l].shape
                return_rgb.append(unpack(9 + l, 2 if l == 0 else 0, rows, cols, 0, rv[l], rg[l] if rg else None,
                                         dst=out[1][l] if out else None))

        return self._finish(L, keep, src, n_total, n_main, d_grad, d_sdf_full, hbuf, gbuf, mask, wgrad, unpack,
                            radiance_side, return_rgb, sv, sg, out, H8, U9, A8)

    def _finish(self, L, keep, src, n_total, n_main, d_grad, d_sdf_full, hbuf, gbuf, mask, wgrad, unpack, radiance_side,
                return_rgb, sv, sg, out, H8, U9, A8):
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
        self.dWk.zero_(); self.dbk.zero_(); self.row0.zero_()
        fork = torch.cuda.Event(); fork.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(fork)
            radiance_side()
            join = torch.cuda.Event(); join.record(self._side)
        st = _stream()
        _lib.check(L.svs_sdf_bwd_a(*src.args(), _ptr(d_grad), _ptr(mask), _ptr(hbuf), _ptr(gbuf), _ptr(self.sdf_stream),
                                   _ptr(self.ubuf), _ptr(self.a2buf), _ptr(self.pebuf), st), "svs_sdf_bwd_a")
        _lib.check(L.svs_sdf_bwd_b(n_total, _ptr(d_sdf_full), _ptr(mask), _ptr(self.feat_bar), n_main, _ptr(hbuf),
                                   _ptr(gbuf), _ptr(self.a2buf), _ptr(self.sdf_stream), _ptr(self.abuf), _ptr(self.sbar),
                                   st), "svs_sdf_bwd_b")
        _lib.check(L.svs_lin8_row0_grad(_ptr(hbuf), _ptr(self.ubuf), _ptr(self.sbar), n_total, _ptr(self.row0), st),
                   "svs_lin8_row0_grad")
        ev = self.timer_events = ([torch.cuda.Event(enable_timing=True) for _ in range(2)]
                                  if getattr(self, "time_wgrad", False) else None)
        if ev:
            ev[0].record()
        wgrad(0, n_total, _off(self.abuf, 0), A8, _ptr(self.pebuf), KBLOCK,
              _off(gbuf, 0), _off(hbuf, 0), H8, H8, _off(self.ubuf, 0), U9)
        for l in range(1, 8):
            wgrad(l, n_total, _off(self.abuf, l * KBLOCK), A8, _off(hbuf, (l - 1) * KBLOCK), H8,
                  _off(gbuf, l * KBLOCK), _off(hbuf, l * KBLOCK), H8, H8, _off(self.ubuf, l * KBLOCK), U9)
        wgrad(8, n_main, _ptr(self.feat_bar), KBLOCK, _off(hbuf, 7 * KBLOCK), H8)
        if ev:
            ev[1].record()
        sdf_grads = []
        for l in range(9):
            rows, cols = sv[l].shape
            sdf_grads.append(unpack(l, 1 if l == 4 else 0, rows, cols, 1 if l == 8 else 0, sv[l], sg[l] if sg else None,
                                    _ptr(self.row0) if l == 8 else None, dst=out[0][l] if out else None))
        main.wait_event(join)
        return sdf_grads, return_rgb
