"""Orchestration of the training backward through the fused MLPs (csrc/svs_mlp_bwd.hip, svs_wgrad.hip).

Host code only sequences kernel launches on HIP streams and owns the scratch buffers; every arithmetic step is
a HIP kernel.  The reference gets these gradients from torch.autograd (volsdf/vsdf.py:215), including the
double backward through network.py:115-121.

Structure (so that a batch can be processed in several ray groups on concurrent streams, see trainer.TrainStep):
  TrainStreams   packed training weight streams, rebuilt once per optimiser step, shared by all groups
  WGradAccum     kernel-order weight-gradient accumulators (float atomics), zeroed once per step, shared
  MlpBackward    per-group scratch + the launches: radiance backward, SDF pass A / pass B, weight-gradient GEMMs
  finalize()     kernel order -> parameter gradients (weight-norm backward), once per step
"""
import ctypes
import os

import torch

from . import lib as _lib
from .ops import _f32, _ptr, _ptr_array, _stream, default_precision, is_h2, F16X2, F16X2_HALF

KBLOCK = 128 * 64           # floats per wave-tile activation block
KREC = 64                   # floats per record of a scaled block (csrc/svs_blocks_h2.h)


def block_stride(n_points):
    """Floats between consecutive blocks of one wave tile in the multi-block activation buffers (hbuf, gbuf, ubuf, a2buf,
    abuf), which are laid out [block][wave tile] with the tile count padded to whole workgroups (svs_mlp_dev.h)."""
    return ((n_points + 127) // 128) * 4 * KBLOCK
LDW = 288


def n_tiles_padded(n_points):
    """wave tiles of a launch over n_points, padded to whole workgroups"""
    return ((n_points + 127) // 128) * 4


def record_off(n_points, n_blocks, block):
    """Float offset of the records of block `block` in a [block][tile] buffer of n_blocks scaled blocks per tile: the
    records follow the slots, in slot order (include/svolsdf_hip.h, "Records")."""
    T = n_tiles_padded(n_points)
    return n_blocks * T * KBLOCK + block * T * KREC


# Experiment, OFF by default: in captured sequences the prior look-up as a branch beside the fused SDF / radiance launches and
# lin8's first-row gradient beside the SDF weight-gradient launch (51 us of small launches off a 256-ray step's critical
# chain).  Measured with launch plans, A/B twice on one box: 1.315 / 1.311 against 1.319 / 1.311 ms (DTU model, 256 rays), 1.55
# against 1.495 with the background model -- the branches' stream crossings and the fifth busy stream cost what they save.
_PLAN_BRANCHES = os.environ.get("SVS_PLAN_BRANCHES", "0") == "1"
_RGB_MERGE = os.environ.get("SVS_RGB_WGRAD_MERGE", "0")       # 0 (default) | 1 | auto (MlpBackward.accumulate: an experiment)
_WGRAD_SPLIT = os.environ.get("SVS_WGRAD_SPLIT", "0")         # 0 (default) | 1 | auto (MlpBackward.accumulate: an experiment)


def _off(t, n_floats):
    """device pointer `n_floats` floats into tensor t"""
    return t.data_ptr() + 4 * n_floats


class TrainStreams:
    def __init__(self, device, precision=None):
        L = _lib.load()
        self.precision = default_precision() if precision is None else int(precision)
        self.sdf = torch.empty(L.svs_stream_bytes(2) // 4, device=device)
        self.rgb = torch.empty(L.svs_stream_bytes(4) // 4, device=device)
        self.ws = torch.empty(L.svs_pack_workspace_bytes() // 4, device=device)

    def pack(self, sdf_params, rgb_params):
        L = _lib.load()
        sv, sg, sb = [[_f32(t) for t in x] if x is not None else None for x in sdf_params]
        rv, rg, rb = [[_f32(t) for t in x] if x is not None else None for x in rgb_params]
        self._keep = (sv, sg, sb, rv, rg, rb)
        st = _stream()
        _lib.check(L.svs_pack_stream(2, self.precision, _ptr_array(sv), _ptr_array(sg) if sg else None, _ptr_array(sb), _ptr(self.ws),
                                     _ptr(self.sdf), st), "svs_pack_stream(sdf train)")
        _lib.check(L.svs_pack_stream(4, self.precision, _ptr_array(rv), _ptr_array(rg) if rg else None, _ptr_array(rb), _ptr(self.ws),
                                     _ptr(self.rgb), st), "svs_pack_stream(rgb bwd)")


class WGradAccum:
    def __init__(self, device):
        # one allocation, one memset per step
        n = [14 * 256 * LDW, 14 * 256, 260, 4]
        self.flat = torch.zeros(sum(n), device=device)
        parts = torch.split(self.flat, n)
        self.dWk = parts[0].view(14, 256, LDW)
        self.dbk = parts[1].view(14, 256)
        self.row0 = parts[2][:257]
        # fp16x2: maxima of the gradient-like GEMM operands, published by the sweeps: [0] SDF abar / u,
        # [1] radiance zbar, [2] feature-vector gradient
        self.absmax = parts[3]

    def zero(self):
        self.flat.zero_()


def _unpack_all(jobs):
    """[(dWk, dbk, ldw, map, rows, cols, row_off, v, g, row0, gv, gg, gb)] (c_void_p / None / int) -> one launch."""
    L = _lib.load()
    val = lambda x: x.value if isinstance(x, ctypes.c_void_p) else x
    arr = (_lib.UnpackJob * len(jobs))(*[_lib.UnpackJob(*[val(x) for x in j]) for j in jobs])
    _lib.check(L.svs_unpack_wgrad_multi(ctypes.cast(arr, ctypes.c_void_p), len(jobs), _stream()), "svs_unpack_wgrad_multi")


def finalize(accum, sdf_params, rgb_params, out=None, nets=(0, 1)):
    """kernel-order accumulators -> (sdf_grads, rgb_grads): lists of (grad_v, grad_g, grad_b) per layer; `out`
    optionally names the destination tensors (views of a flat gradient buffer).  One launch for all 14 layers; `nets`
    restricts it to the SDF network (0) or the radiance network (1): a data-parallel step unpacks the radiance gradients as
    soon as their GEMM launch has retired, so that their bucket can be all-reduced beside the SDF backward (trainer.py)."""
    dev = accum.dWk.device
    res, jobs, keep = [], [], []
    for gi, (params, base, n) in enumerate(((sdf_params, 0, 9), (rgb_params, 9, 5))):
        v, g, _ = params
        group = []
        if gi not in nets:
            res.append(group)
            continue
        for l in range(n):
            rows, cols = v[l].shape
            if out is not None:
                gv, gg, gb = out[gi][l]
            else:
                gv = torch.empty(rows, cols, device=dev)
                gg = torch.empty(rows, 1, device=dev) if g is not None else None
                gb = torch.empty(rows, device=dev)
            is_sdf = gi == 0
            mp = 1 if (is_sdf and l == 4) else (2 if (not is_sdf and l == 0) else 0)
            row_off = 1 if (is_sdf and l == 8) else 0
            row0 = _ptr(accum.row0) if (is_sdf and l == 8) else None
            vl, gl = _f32(v[l]), (_f32(g[l]) if g is not None else None)
            keep += [vl, gl]
            jobs.append((_off(accum.dWk, (base + l) * 256 * LDW), _off(accum.dbk, (base + l) * 256), LDW, mp, rows, cols,
                         row_off, _ptr(vl), _ptr(gl), row0, _ptr(gv), _ptr(gg), _ptr(gb)))
            group.append((gv, gg, gb))
        res.append(group)
    _unpack_all(jobs)
    return res[0], res[1]


class MlpBackward:
    """Per-group scratch and launches of the MLP backward."""

    def __init__(self, device, streams=None, accum=None):
        self.dev = device
        self.streams = streams or TrainStreams(device)
        self.accum = accum or WGradAccum(device)
        self._n = None
        self._side = None
        self._job_cache = {}
        self.time_wgrad = False
        self.timer_events = None

    def _alloc(self, n_total, n_main):
        if self._n == (n_total, n_main):
            return
        L = _lib.load()
        z = lambda nbytes: torch.zeros(nbytes // 4, device=self.dev)
        self.zbuf = z(L.svs_rgb_zbuf_bytes(n_main))           # zero-initialised once: zbar_4 only writes its first tile
        self.feat_bar = z(L.svs_block_bytes(n_main, 1))
        self.ubuf = z(L.svs_sdf_ubuf_bytes(n_total))
        # float32 kernels only: the fp16x2 pass B re-forms a2 from ubuf and gbuf (8 KB per point less traffic and memory)
        self.a2buf = z(L.svs_block_bytes(n_total, 8)) if not is_h2(self.streams.precision) else None
        self.abuf = z(L.svs_block_bytes(n_total, 8))
        self.pebuf = z(L.svs_block_bytes(n_total, 1))
        self.sbar = z(L.svs_block_bytes(n_total, 1) // (128 * 2))  # 32 floats per tile
        # d loss / d sdf of all points of the launch: the ray samples' part is rewritten every step (sdf_grad_out()), the
        # tail -- the eikonal points, which have no such term -- stays zero
        self.d_sdf_full = torch.zeros(n_total, device=self.dev)
        # d loss / d (d sdf / dx) of all points of the launch, pass A's input: rows [0, n_main) = d loss / d normals, written by
        # the radiance backward; the tail = the eikonal points' gradients, written by the loss kernel (grad_extra_out()) --
        # no concatenation launch between the radiance backward and pass A
        self.d_grad_full = torch.zeros(n_total, 3, device=self.dev)
        self._n = (n_total, n_main)

    def _cache_jobs(self, key, jobs):
        if len(self._job_cache) >= 64:
            self._job_cache.clear()
        arr = (_lib.WGradJob * len(jobs))(*jobs)
        hit = self._job_cache[key] = (ctypes.cast(arr, ctypes.c_void_p), len(jobs), arr)    # (arr kept alive beside its address)
        return hit[:2]

    def sdf_grad_out(self, n_total, n_main):
        """(n_main,1) view of the persistent d_sdf buffer: compositing's backward writes into it directly."""
        self._alloc(n_total, n_main)
        return self.d_sdf_full[:n_main].view(n_main, 1)

    def grad_extra_out(self, n_total, n_main):
        """(n_total - n_main, 3) view of the persistent d loss / d (d sdf / dx) buffer: the loss kernel writes the eikonal
        points' gradients into it directly."""
        self._alloc(n_total, n_main)
        return self.d_grad_full[n_main:]

    def accumulate(self, keep, d_rgb, d_sdf, d_grad_extra, wait=True, side=True, defer_wgrad=False, extra=None):
        """Launches the backward of one ray group on the current stream (+ a side stream for the radiance weight
        gradients) and adds its weight gradients into self.accum.  Returns the event that marks the end of the side
        stream's work; wait=True also makes the current stream wait for it.  (A caller that runs groups on forked streams
        passes wait=False and lets its ORIGIN stream wait for the event: see trainer.TrainStep._device_step.)
        keep: dict filled by ops.sdf_outputs / ops.rgb_eval (hbuf, gbuf, clamp_mask, src, rbuf, feat_tiles, rgb).
        d_rgb (n_main,3); d_sdf (n_main,1) or None; d_grad_extra (n_extra,3) or None: dL/d(d sdf/dx) of the extra
        (eikonal) points that follow the ray samples in the launch.
        defer_wgrad: only the sweeps run (radiance backward, pass A, pass B, lin8's first row); the group's weight-gradient
        JOBS are returned instead of launched -- dict(rgb=[jobs], sdf=[jobs], ev_rgb, ev_all, key) -- for another group's
        call to take along as `extra`: that call's two weight-gradient launches then cover both groups (a small ray group's
        own launches cost a ring fill and one workgroup per CU each for a twentieth of the points)."""
        L = _lib.load()
        src = keep["src"]
        n_total, n_main = src.n, keep["rgb"].shape[0]
        if n_main % 32:
            raise NotImplementedError("rays*samples of a group must be a multiple of 32")
        self._alloc(n_total, n_main)
        dev, acc, S = self.dev, self.accum, self.streams
        LS = block_stride(n_total)
        hbuf, gbuf, mask = keep["hbuf"], keep["gbuf"], keep["clamp_mask"]
        rbuf, feat = keep["rbuf"], keep["feat_tiles"]

        prec = S.precision
        h2 = is_h2(prec)

        def addr(x):
            return x.value if isinstance(x, ctypes.c_void_p) else x

        def job(slot, n_pts, amax, a0, sa0, b0, sb0, a1=None, sa1=0, b1=None, sb1=0, extra=None, sx=0, rec0=None, rec1=None,
                bias=True):
            return _lib.WGradJob(addr(a0), addr(b0), sa0, sb0, addr(a1), addr(b1), sa1, sb1,
                                 addr(extra), sx, n_pts, LDW, addr(_off(acc.dWk, slot * 256 * LDW)),
                                 addr(_off(acc.dbk, slot * 256)) if bias else None, addr(_off(acc.absmax, amax)) if h2 else None,
                                 addr(rec0) if h2 else None, addr(rec1) if h2 else None)

        def wgrad_multi(cached):
            arr, n = cached
            _lib.check(L.svs_wgrad_multi(arr, n, prec, _stream()), "svs_wgrad_multi")

        def job_list(key, build):
            """the group's own jobs (ctypes structures), built once per configuration"""
            hit = self._job_cache.get(key)
            if hit is None:
                if len(self._job_cache) >= 64:
                    self._job_cache.clear()
                hit = self._job_cache[key] = build()
            return hit

        def job_array(key, own, more):
            """own jobs (+ those of a folded-in group) as the array one launch takes"""
            key = ("arr",) + key + ((more["key"],) if more else ())
            hit = self._job_cache.get(key)
            if hit is None:
                return self._cache_jobs(key, list(own) + (list(more["jobs"]) if more else []))
            return hit[:2]

        # ---- radiance MLP: input gradients
        d_rgb = _f32(d_rgb)
        d_grad = self.d_grad_full
        d_normals = d_grad[:n_main]
        n_extra = 0 if d_grad_extra is None else d_grad_extra.shape[0]
        if n_main + n_extra != n_total:
            raise ValueError("d_grad_extra must cover the points that follow the ray samples")
        if n_extra and d_grad_extra.data_ptr() != d_grad[n_main:].data_ptr():
            # (a caller that did not write into grad_extra_out(); by a kernel, not copy_: a captured device copy is a node a
            # launch plan cannot replay, csrc/svs_plan.hip)
            torch.mul(_f32(d_grad_extra), 1.0, out=d_grad[n_main:])
        _lib.check(L.svs_rgb_bwd(n_main, _ptr(d_rgb), _ptr(keep["rgb"]), _ptr(rbuf), _ptr(S.rgb), prec, _ptr(self.zbuf),
                                 _ptr(self.feat_bar), _ptr(d_normals), _ptr(acc.absmax) if h2 else None, _stream()),
                   "svs_rgb_bwd")
        if d_sdf is not None and d_sdf.data_ptr() == self.d_sdf_full.data_ptr() and d_sdf.numel() == n_main:
            d_sdf_full = self.d_sdf_full              # written in place by the caller (sdf_grad_out())
        else:
            d_sdf_full = torch.zeros(n_total, device=dev)
            if d_sdf is not None:
                d_sdf_full[:n_main] = _f32(d_sdf).reshape(-1)
        # ---- the radiance weight-gradient GEMMs only need rgb_bwd's outputs: they run on a side stream and fill the
        # CUs the SDF sweeps leave idle in their tail round
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        if os.environ.get("SVS_RGB_WGRAD_SIDE", "1") == "0":   # A/B switch: the radiance weight gradients in line, in front of pass A
            side = False
        side_stream = self._side if side else main          # side=False: everything on the current stream
        # the job lists are a function of the buffers' addresses and the point counts: built once per configuration (a
        # step's scratch comes back at the same addresses from torch's caching allocator), ~40 ctypes structures a step
        rkey = ("rgb", n_main, prec, self.zbuf.data_ptr(), feat.data_ptr(), rbuf.data_ptr(), acc.dWk.data_ptr())

        def rgb_jobs():
            LSm = block_stride(n_main)          # rbuf = [4 blocks][tile] + extras [tile][1024], zbuf = [5 blocks][tile]
            zrec = lambda l: _off(self.zbuf, record_off(n_main, 5, l))
            jobs = [job(9, n_main, 1, _off(self.zbuf, 0), KBLOCK, _ptr(feat), KBLOCK, extra=_off(rbuf, 4 * LSm), sx=1024,
                        rec0=zrec(0))]
            for l in range(1, 5):
                jobs.append(job(9 + l, n_main, 1, _off(self.zbuf, l * LSm), KBLOCK, _off(rbuf, (l - 1) * LSm), KBLOCK,
                                rec0=zrec(l)))
            return jobs

        # Experiment (SVS_WGRAD_SPLIT = 1 | auto: up to 40 960 points; OFF by default).  Small batches (config 4's 256 rays per
        # GPU): the step is the SUM of its kernels' latencies, the sweeps' workgroups do not fill the chip and HBM idles under
        # them.  The second-order half of the SDF weight gradients -- ghat_l x u_l^T, complete once pass A is -- rides with
        # the radiance weight gradients in ONE launch beside pass B, and the launch that ends the step is the first-order
        # half (abar_l x h_l^T, bias gradients) only.  Same sums, same accumulators (gradient tests pass).  Round 4 had the
        # second-order launch queued BEHIND the radiance launch (it started when pass B was half done): 1.46 against 1.42 ms.
        # Round 5, merged, A/B three times on one box: 1.186 against 1.194 ms at 256 rays (DTU), 1.358 against 1.324 with the
        # background model, 1.94 / 1.94 at 512 rays -- the launch that ends the step is not HBM-bound at this size (26 items
        # per workgroup: ring fill, atomic flush and the spread of the workgroups' finishing times), halving its bytes takes
        # 15 % off it, and pass B runs slower beside 0.7 GB of streaming.  Off.
        split = h2 and side and not defer_wgrad and extra is None and \
            (_WGRAD_SPLIT == "1" or (_WGRAD_SPLIT == "auto" and n_total <= 40960))
        # Experiment (SVS_RGB_WGRAD_MERGE = 1 | auto: up to 40 960 points; OFF by default): the radiance network's weight
        # gradients ride in the SDF network's launch at the end of the step instead of running beside pass A on a stream of
        # their own.  One workgroup of the GEMM takes a whole CU (8 waves, 136 KB of LDS), and so does one of a sweep (one
        # 512-register wave per SIMD): beside pass A the radiance launch's 256 workgroups take turns with the sweep's ~200 -- at
        # 256 rays pass A lasts 122 us (195 with the background networks' launches beside it as well), pass B 226.  Merged, the
        # sweeps run alone and the last launch grows -- A/B three times on one box: 1.215 against 1.208 ms (DTU model), 1.39
        # against 1.363 with the background model, 2.02 / 2.01 at 512 rays: what the concurrent launch costs the sweeps is less
        # than its own duration.  Off.
        merge_rgb = h2 and side and not defer_wgrad and extra is None and not split and \
            (_RGB_MERGE == "1" or (_RGB_MERGE == "auto" and n_total <= 40960))
        deferred = None
        if defer_wgrad:
            ev_rgb = torch.cuda.Event(); ev_rgb.record(main)
            deferred = dict(rgb=dict(jobs=job_list(rkey, rgb_jobs), key=rkey), ev_rgb=ev_rgb)
            join = None
        elif split:
            join = None
        elif merge_rgb:
            join = None
        else:
            fork = torch.cuda.Event(); fork.record(main)
            with torch.cuda.stream(side_stream):
                if side:
                    side_stream.wait_event(fork)
                if extra is not None:
                    side_stream.wait_event(extra["ev_rgb"])
                wgrad_multi(job_array(rkey, job_list(rkey, rgb_jobs), extra["rgb"] if extra else None))
                join = torch.cuda.Event(); join.record(side_stream)
        # ---- SDF MLP: pass A (needs nbar), pass B (needs sbar, fbar), then its weight gradients
        st = _stream()
        _lib.check(L.svs_sdf_bwd_a(*src.args(), _ptr(d_grad), _ptr(mask), _ptr(hbuf), _ptr(gbuf), _ptr(S.sdf), prec,
                                   _ptr(self.ubuf), _ptr(self.a2buf), _ptr(self.pebuf),
                                   _ptr(acc.absmax) if h2 else None, st), "svs_sdf_bwd_a")
        join2 = None
        if split:
            k2 = ("sdf2", n_total, prec, self.ubuf.data_ptr(), gbuf.data_ptr(), acc.dWk.data_ptr())
            urec = lambda l: _off(self.ubuf, record_off(n_total, 9, l))
            second = lambda: [job(l, n_total, 0, _off(gbuf, l * LS), KBLOCK, _off(self.ubuf, l * LS), KBLOCK,
                                  rec0=urec(l), bias=False) for l in range(8)]
            after_a = torch.cuda.Event(); after_a.record(main)
            with torch.cuda.stream(side_stream):
                side_stream.wait_event(after_a)
                # one launch: the radiance network's five layers + the eight second-order products of the SDF network
                wgrad_multi(job_array(rkey + k2, job_list(rkey, rgb_jobs), dict(jobs=job_list(k2, second), key=k2)))
                join2 = torch.cuda.Event(); join2.record(side_stream)
        _lib.check(L.svs_sdf_bwd_b(n_total, _ptr(d_sdf_full), _ptr(mask), _ptr(self.feat_bar), n_main, _ptr(hbuf),
                                   _ptr(gbuf), _ptr(self.a2buf), _ptr(self.ubuf), _ptr(S.sdf), prec, _ptr(self.abuf),
                                   _ptr(self.sbar), _ptr(acc.absmax) if h2 else None, st),
                   "svs_sdf_bwd_b")
        # the first row of lin8's weight gradient: a 257-vector reduction over two blocks (0.04 ms alone).  On the side stream,
        # beside the weight-gradient launch, it was starved to the length of that launch (one workgroup of the GEMM per CU
        # leaves it a quarter of the register file): it runs in front of it on this stream
        # (experiment, SVS_PLAN_BRANCHES=1: in a captured sequence on the radiance weight gradients' stream after all)
        row0_aside = side and not defer_wgrad and _PLAN_BRANCHES and torch.cuda.is_current_stream_capturing()
        if row0_aside:
            after_b = torch.cuda.Event(); after_b.record(main)
            with torch.cuda.stream(side_stream):
                side_stream.wait_event(after_b)
                _lib.check(L.svs_lin8_row0_grad(_ptr(hbuf), _ptr(self.ubuf), _ptr(self.sbar), n_total, prec, _ptr(acc.row0),
                                                _stream()), "svs_lin8_row0_grad")
                join = torch.cuda.Event(); join.record(side_stream)
        else:
            _lib.check(L.svs_lin8_row0_grad(_ptr(hbuf), _ptr(self.ubuf), _ptr(self.sbar), n_total, prec, _ptr(acc.row0),
                                            _stream()), "svs_lin8_row0_grad")
        ev = self.timer_events = ([torch.cuda.Event(enable_timing=True) for _ in range(2)] if self.time_wgrad else None)
        if ev:
            ev[0].record()
        skey = ("sdf1" if split else "sdf", n_total, n_main, prec, self.abuf.data_ptr(), self.ubuf.data_ptr(), self.pebuf.data_ptr(),
                hbuf.data_ptr(), gbuf.data_ptr(), self.feat_bar.data_ptr(), acc.dWk.data_ptr())

        def sdf_jobs():
            arec = lambda l: _off(self.abuf, record_off(n_total, 8, l))
            urec = lambda l: _off(self.ubuf, record_off(n_total, 9, l))
            second = (lambda l: dict(a1=_off(gbuf, l * LS), sa1=KBLOCK, b1=_off(self.ubuf, l * LS), sb1=KBLOCK, rec1=urec(l))) \
                if not split else (lambda l: {})
            jobs = [job(0, n_total, 0, _off(self.abuf, 0), KBLOCK, _ptr(self.pebuf), KBLOCK, rec0=arec(0), **second(0))]
            for l in range(1, 8):
                jobs.append(job(l, n_total, 0, _off(self.abuf, l * LS), KBLOCK, _off(hbuf, (l - 1) * LS), KBLOCK, rec0=arec(l),
                                **second(l)))
            jobs.append(job(8, n_main, 2, _ptr(self.feat_bar), KBLOCK, _off(hbuf, 7 * LS), KBLOCK,
                            rec0=_off(self.feat_bar, record_off(n_main, 1, 0))))
            return jobs

        if defer_wgrad:
            deferred["sdf"] = dict(jobs=job_list(skey, sdf_jobs), key=skey)
            deferred["ev_all"] = torch.cuda.Event(); deferred["ev_all"].record(main)
            deferred["hold"] = self._hold = (d_grad, d_sdf_full, d_normals, d_rgb)
            return deferred
        if extra is not None:
            main.wait_event(extra["ev_all"])
        if merge_rgb:
            wgrad_multi(job_array(skey + rkey, job_list(skey, sdf_jobs), dict(jobs=job_list(rkey, rgb_jobs), key=rkey)))
            join = torch.cuda.Event(); join.record(main)
        else:
            wgrad_multi(job_array(skey, job_list(skey, sdf_jobs), extra["sdf"] if extra else None))
        if ev:
            ev[1].record()
        if join2 is not None and not row0_aside:
            join = join2                      # (recorded on the same side stream, after the radiance launch's event)
        if wait and side:
            main.wait_event(join)
        self._hold = (d_grad, d_sdf_full, d_normals, d_rgb)      # keep inputs alive until the streams are joined
        return join

    def run(self, sdf_params, rgb_params, keep, d_rgb, d_sdf, d_grad_extra, out=None):
        """Whole backward of a single group: pack, zero, accumulate, finalize.  Returns (sdf_grads, rgb_grads)."""
        self.streams.pack(sdf_params, rgb_params)
        self.accum.zero()
        self.accumulate(keep, d_rgb, d_sdf, d_grad_extra)
        return finalize(self.accum, sdf_params, rgb_params, out)


class BgBackward:
    """Training backward of the background networks of VolSDFNetworkBG (bg_implicit_network: ordinary backprop, no
    second-order sweep because its input gradient is never used; bg_rendering_network): the kernels of csrc/svs_bg_h2.hip
    (fp16x2) or csrc/svs_bg_f32.hip (float32 MFMA) + the shared pass-B sweep and weight-gradient kernels."""
    BGRBUF = KBLOCK + 1024

    def __init__(self, device, precision=None):
        L = _lib.load()
        self.dev = device
        self.precision = default_precision() if precision is None else int(precision)
        self.sdf_stream = torch.empty(L.svs_stream_bytes(6) // 4, device=device)
        self.rgb_stream = torch.empty(L.svs_stream_bytes(8) // 4, device=device)
        self.ws = torch.empty(L.svs_pack_workspace_bytes() // 4, device=device)
        # one allocation, one memset per step (as WGradAccum)
        n = [11 * 256 * LDW, 11 * 256, 260, 4]
        self.flat = torch.zeros(sum(n), device=device)
        parts = torch.split(self.flat, n)
        self.dWk = parts[0].view(11, 256, LDW)                   # 0..8 implicit layers, 9..10 radiance layers
        self.dbk = parts[1].view(11, 256)
        self.row0 = parts[2][:257]
        self.absmax = parts[3]
        self._scratch = {}          # per concurrent ray group: (n, zbuf, feat_bar, abuf, sbar); the accumulators are shared

    def pack(self, sdf_wb, rgb_wb):
        L = _lib.load()
        self._keep = []
        for which, (w, b), stream in ((6, sdf_wb, self.sdf_stream), (8, rgb_wb, self.rgb_stream)):
            w, b = [_f32(t) for t in w], [_f32(t) for t in b]
            self._keep += [w, b]
            _lib.check(L.svs_pack_stream(which, self.precision, _ptr_array(w), None, _ptr_array(b), _ptr(self.ws), _ptr(stream),
                                         _stream()), "svs_pack_stream(bg backward)")

    def zero(self):
        self.flat.zero_()

    def _alloc(self, n, slot):
        cur = self._scratch.get(slot)
        if cur is not None and cur[0] == n:
            return cur[1:]
        L = _lib.load()
        z = lambda nbytes: torch.zeros(nbytes // 4, device=self.dev)
        cur = (n, z(L.svs_block_bytes(n, 2)),          # zbuf: zero-initialised once, only 4 + 1 tiles are ever written
               z(L.svs_block_bytes(n, 1)), z(L.svs_block_bytes(n, 8)), z(L.svs_block_bytes(n, 1) // (128 * 2)))
        self._scratch[slot] = cur
        return cur[1:]

    def accumulate(self, keep, d_bg_rgb, d_bg_out0, slot=0):
        """keep: what ops.bg_sdf_eval / ops.bg_rgb_eval stored; d_bg_rgb (P,3), d_bg_out0 (P,1); slot: scratch set (one per
        concurrent ray group)."""
        L = _lib.load()
        P = keep["bg_rgb"].shape[0]
        zbuf, feat_bar, abuf, sbar = self._alloc(P, slot)
        st = _stream()
        hbuf, ghat7, pebuf, rbuf, feat = keep["bg_hbuf"], keep["bg_ghat7"], keep["bg_pebuf"], keep["bg_rbuf"], keep["bg_feat"]
        d_bg_rgb, d_bg_out0 = _f32(d_bg_rgb), _f32(d_bg_out0)
        prec = self.precision
        _lib.check(L.svs_bg_rgb_bwd(P, _ptr(d_bg_rgb), _ptr(keep["bg_rgb"]), _ptr(rbuf), _ptr(self.rgb_stream), prec, _ptr(zbuf),
                                    _ptr(feat_bar), _ptr(self.absmax), st), "svs_bg_rgb_bwd")
        _lib.check(L.svs_bg_sdf_bwd(P, _ptr(d_bg_out0), _ptr(feat_bar), _ptr(hbuf), _ptr(ghat7), _ptr(self.sdf_stream), prec,
                                    _ptr(abuf), _ptr(sbar), _ptr(self.absmax), st), "svs_bg_sdf_bwd")
        _lib.check(L.svs_lin8_row0_grad(_ptr(hbuf), None, _ptr(sbar), P, prec, _ptr(self.row0), st), "svs_lin8_row0_grad")
        LS, Z2 = block_stride(P), 2 * KBLOCK
        T = n_tiles_padded(P)

        def addr(x):
            return x.value if isinstance(x, ctypes.c_void_p) else x

        def job(slot, amax, a0, sa0, b0, sb0, rec0, extra=None, sx=0):
            return _lib.WGradJob(addr(a0), addr(b0), sa0, sb0, None, None, 0, 0, addr(extra), sx, P, LDW,
                                 addr(_off(self.dWk, slot * 256 * LDW)), addr(_off(self.dbk, slot * 256)),
                                 addr(_off(self.absmax, amax)), addr(rec0), None)

        arec = lambda l: _off(abuf, record_off(P, 8, l))
        jobs = [job(0, 0, _off(abuf, 0), KBLOCK, _ptr(pebuf), KBLOCK, arec(0))]
        for l in range(1, 8):
            jobs.append(job(l, 0, _off(abuf, l * LS), KBLOCK, _off(hbuf, (l - 1) * LS), KBLOCK, arec(l)))
        jobs.append(job(8, 2, _ptr(feat_bar), KBLOCK, _off(hbuf, 7 * LS), KBLOCK, _off(feat_bar, record_off(P, 1, 0))))
        # bg zbuf slots are [tile][2 blocks]; its records, behind the slots, are [block][tile][64] like everywhere else
        zrec = lambda b: _off(zbuf, 2 * T * KBLOCK + b * T * KREC)
        jobs.append(job(9, 1, _off(zbuf, 0), Z2, _ptr(feat), KBLOCK, zrec(0), extra=_off(rbuf, KBLOCK), sx=self.BGRBUF))
        jobs.append(job(10, 1, _off(zbuf, KBLOCK), Z2, _ptr(rbuf), self.BGRBUF, zrec(1)))
        arr = (_lib.WGradJob * len(jobs))(*jobs)
        _lib.check(L.svs_wgrad_multi(ctypes.cast(arr, ctypes.c_void_p), len(jobs), prec, st), "svs_wgrad_multi(bg)")
        self._hold = getattr(self, "_hold", {})
        self._hold[slot] = (d_bg_rgb, d_bg_out0)

    def finalize(self, sdf_wb, rgb_wb, out=None):
        """kernel-order accumulators -> ([(grad_w, grad_b)] * 9, [(grad_w, grad_b)] * 2); one launch"""
        res, jobs, keep = [], [], []
        for gi, ((w, b), base) in enumerate(((sdf_wb, 0), (rgb_wb, 9))):
            group = []
            for l in range(len(w)):
                rows, cols = w[l].shape
                gw, gb = out[gi][l] if out is not None else (torch.empty(rows, cols, device=self.dev), torch.empty(rows, device=self.dev))
                is_sdf = gi == 0
                mp = 3 if (is_sdf and l == 4) else (4 if (not is_sdf and l == 0) else 0)
                row_off = 1 if (is_sdf and l == 8) else 0
                row0 = _ptr(self.row0) if (is_sdf and l == 8) else None
                wl = _f32(w[l])
                keep.append(wl)
                jobs.append((_off(self.dWk, (base + l) * 256 * LDW), _off(self.dbk, (base + l) * 256), LDW, mp, rows, cols, row_off,
                             _ptr(wl), None, row0, _ptr(gw), None, _ptr(gb)))
                group.append((gw, gb))
            res.append(group)
        _unpack_all(jobs)
        return res[0], res[1]


def algorithmic_bytes_per_point(precision=None):
    """HBM bytes per point that the HBM-bound launches of the training backward read or write ONCE (the figures
    bench.py's roofline prices).  One 256-feature block slot = 1024 bytes per point in float32
    (precision f32); on the fp16x2 path (csrc/svs_blocks_h2.h) a PAIR block is 1024 bytes per point, its hi plane alone
    512, a HALF block 512.
      svs_sdf_bwd_a   float32: reads h_1..h_8, ghat_0..ghat_7; writes u_0..u_8, a2_0..a2_7, the PE block
                      fp16x2 (round 5): reads h_1..h_8; writes u_0..u_8, the PE block (a2 is re-formed by pass B)
      svs_sdf_bwd_b   float32: reads h_1..h_8, a2_0..a2_7, ghat_7, fbar; writes abar_0..abar_7
                      fp16x2: reads h_1..h_8 (hi planes in the one-piece mode), u_1..u_8, ghat_0..ghat_7, fbar; writes abar_0..abar_7
      wgrad_sdf       per layer abar_l, h_l (hi plane), ghat_l, u_l (l = 0..7) + fbar, h_8 (hi plane) for lin8
      wgrad_radiance  zbar_0..zbar_4, r_0..r_3 (hi planes), the feature block (hi plane), the 16 extra input rows
      svs_sdf_outputs (training launch; MFMA-bound, listed for its SECOND roofline) writes h_1..h_8, ghat_0..ghat_7, the
                      feature block; its reverse sweep reads h_1..h_8 back"""
    precision = default_precision() if precision is None else precision
    if is_h2(precision):
        # F16X2: every block with both pieces (1024 B per point and block); F16X2_HALF: scaled blocks and the planes the
        # sweeps / the weight gradient read of a pair block are 512 B
        half = 512 if precision == F16X2_HALF else 1024
        pair = 1024
        return {"svs_sdf_bwd_a": 8 * half + 9 * half + pair,
                "svs_sdf_bwd_b": 8 * half + 8 * half + 8 * half + half + 8 * half,
                "wgrad_sdf": 8 * (half + half + half + half) + half + half,
                "wgrad_radiance": 5 * half + 4 * half + half + 128,
                "svs_lin8_row0_grad": pair + half,
                "svs_sdf_outputs": 8 * pair + 8 * pair + 8 * half + pair}
    blk = 1024
    return {"svs_sdf_bwd_a": (8 + 8 + 9 + 8 + 1) * blk, "svs_sdf_bwd_b": (8 + 8 + 1 + 1 + 8) * blk,
            "wgrad_sdf": (8 * 4 + 2) * blk, "wgrad_radiance": (5 + 4 + 1) * blk + 128, "svs_lin8_row0_grad": 2 * blk,
            "svs_sdf_outputs": (8 + 8 + 8 + 1) * blk}
