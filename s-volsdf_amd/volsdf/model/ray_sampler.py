"""Ray samplers with the reference's class names and call surface (volsdf/model/ray_sampler.py).

`ErrorBoundSampler.get_z_vals` enqueues the HIP sampler (svs_sampler.hip) with the gated fused SDF-MLP
launches between rounds; no host synchronisation happens inside a call.  All randomness is drawn from
torch's CPU generator in the reference's order (ray_sampler.py:39,170,201,211) and handed to the kernels.
"""
import abc

import torch

from svs_hip import ops


class RaySampler(metaclass=abc.ABCMeta):
    def __init__(self, near, far):
        self.near, self.far = near, far

    @abc.abstractmethod
    def get_z_vals(self, ray_dirs, cam_loc, model):
        pass


class UniformSampler(RaySampler):
    """ray_sampler.py:15-43.  Kept as host glue (tiny); the error-bounded sampler's own uniform
    initialisation runs inside its init kernel."""

    def __init__(self, scene_bounding_sphere, near, N_samples, take_sphere_intersection=False, far=-1):
        super().__init__(near, 2.0 * scene_bounding_sphere if far == -1 else far)
        self.N_samples = N_samples
        self.scene_bounding_sphere = scene_bounding_sphere
        self.take_sphere_intersection = take_sphere_intersection

    def get_z_vals(self, ray_dirs, cam_loc, model, iter_step=None):
        from volsdf.utils import rend_util
        dev = ray_dirs.device
        n = ray_dirs.shape[0]
        near = self.near * torch.ones(n, 1, device=dev)
        if not self.take_sphere_intersection:
            far = self.far * torch.ones(n, 1, device=dev)
        else:
            far = rend_util.get_sphere_intersections(cam_loc, ray_dirs, r=self.scene_bounding_sphere)[:, 1:]
        t_vals = torch.linspace(0., 1., steps=self.N_samples).to(dev)
        z_vals = near * (1. - t_vals) + far * t_vals
        if model.training:
            mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
            upper = torch.cat([mids, z_vals[..., -1:]], -1)
            lower = torch.cat([z_vals[..., :1], mids], -1)
            t_rand = torch.rand(z_vals.shape).to(dev)
            z_vals = lower + (upper - lower) * t_rand
        return z_vals


class ErrorBoundSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, N_samples_eval, N_samples_extra, eps, beta_iters,
                 max_total_iters, inverse_sphere_bg=False, N_samples_inverse_sphere=0, add_tiny=0.0):
        super().__init__(near, 2.0 * scene_bounding_sphere)
        self.N_samples, self.N_samples_eval, self.N_samples_extra = N_samples, N_samples_eval, N_samples_extra
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_samples_eval,
                                              take_sphere_intersection=inverse_sphere_bg)
        self.eps, self.beta_iters, self.max_total_iters = eps, beta_iters, max_total_iters
        self.scene_bounding_sphere = scene_bounding_sphere
        self.add_tiny = add_tiny
        self.inverse_sphere_bg = inverse_sphere_bg
        if inverse_sphere_bg:
            self.inverse_sphere_sampler = UniformSampler(1.0, 0.0, N_samples_inverse_sphere, False, far=1.0)
        # float32 constant of ray_sampler.py:77, computed the way the reference computes it
        self._inv_4log = float(1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0))))
        self._ws = None

    def draw_train_rng(self, R, dev, extra=None, out=None, stream=None):
        """The sampler's train-mode draws for R rays: same calls, same order as the reference
        (ray_sampler.py:39,170,201,211; CPU generator), then uploaded to the device.

        The draws are generated straight into a small ring of persistent pinned host buffers and uploaded with
        non-blocking copies: a pageable `.to(device)` blocks the host until all earlier work of the stream has
        finished, which kept the host from running ahead of the GPU (0.55 ms of GPU idle time at the start of every
        step).  extra: optional callable(slot_dict) drawing further tensors AFTER the sampler's (the model's
        eikonal points, network.py:261), so the generator order of the reference is preserved.
        out: optional dict of persistent device tensors to upload into (the static inputs of a captured step) instead of
        fresh ones; missing entries are created in it.
        stream: optional side stream THE CALLER ALREADY OWNS to run the uploads on (into device buffers that belong to the
        ring slot), so that they execute while the previous step is still computing instead of in front of this step's
        first kernel (45 us of idle GPU per step in the kernel trace; the host runs ahead of the GPU).  The current stream
        waits for the copies; the copies wait until the kernels that last read the slot's device buffers (4 draws ago) are
        done (`consumed`, recorded on the consumer's stream when the next draw is made).  A stream of its own for this was
        measured 20 % SLOWER: one more stream changes how the runtime maps streams onto its 4 hardware queues."""
        n_out = self.N_samples + self.N_samples_extra + 2
        if dev.type != "cuda":
            host = dict(jitter=torch.rand(R, self.N_samples_eval), u=torch.rand(R, self.N_samples),
                        perm=torch.randperm(self.N_samples_eval)[:self.N_samples_extra].to(torch.int32),
                        eik_idx=torch.randint(n_out, (R,)).to(torch.int32))
            if self.inverse_sphere_bg:
                host["jitter_bg"] = torch.rand(R, self.inverse_sphere_sampler.N_samples)
            if extra is not None:
                extra(host, None)
            return host
        ring = getattr(self, "_rng_ring", None)
        if ring is None or ring["key"] != (R, str(dev)):
            # One pinned buffer per slot, the draws are views of it: a step's draws travel in ONE host-to-device copy
            # (five separate copies were 0.4 ms of host time per step).  Layout in 4-byte words; the int32 arrays are views
            # of the same storage.  `eik_points` (R,3) is the model's draw (extra()), reserved here so that it travels along.
            nbg = self.inverse_sphere_sampler.N_samples if self.inverse_sphere_bg else 0
            sizes = dict(jitter=R * self.N_samples_eval, u=R * self.N_samples, perm=self.N_samples_extra, eik_idx=R,
                         jitter_bg=R * nbg, eik_points=3 * R)
            shapes = dict(jitter=(R, self.N_samples_eval), u=(R, self.N_samples), perm=(self.N_samples_extra,), eik_idx=(R,),
                          jitter_bg=(R, nbg), eik_points=(R, 3))

            def make_slot():
                total = sum(sizes.values())
                pin = torch.empty(total, dtype=torch.float32).pin_memory()
                slot, off = dict(_pin=pin, _off={}, event=None), 0
                for k, n in sizes.items():
                    if n == 0:
                        continue
                    v = pin[off:off + n]
                    slot[k] = (v.view(torch.int32) if k in ("perm", "eik_idx") else v).view(shapes[k])
                    slot["_off"][k] = (off, n)
                    off += n
                slot["perm64"] = torch.empty(self.N_samples_eval, dtype=torch.int64)
                slot["eik64"] = torch.empty(R, dtype=torch.int64)
                return slot
            ring = self._rng_ring = dict(key=(R, str(dev)), i=0, slots=[make_slot() for _ in range(4)])
        slot = ring["slots"][ring["i"] % len(ring["slots"])]
        ring["i"] += 1
        if slot["event"] is not None:
            slot["event"].synchronize()          # the uploads that last read this slot (4 steps ago) are done
        torch.rand(R, self.N_samples_eval, out=slot["jitter"])
        torch.rand(R, self.N_samples, out=slot["u"])
        torch.randperm(self.N_samples_eval, out=slot["perm64"])
        slot["perm"].copy_(slot["perm64"][:self.N_samples_extra])
        torch.randint(n_out, (R,), out=slot["eik64"])
        slot["eik_idx"].copy_(slot["eik64"])
        names = ["jitter", "u", "perm", "eik_idx"]
        if self.inverse_sphere_bg:
            # the inverse-sphere sampler jitters too, after the eikonal pick (ray_sampler.py:215 -> :39)
            torch.rand(R, self.inverse_sphere_sampler.N_samples, out=slot["jitter_bg"])
            names.append("jitter_bg")
        if extra is not None:
            names += extra(slot, R)

        def upload(dbuf):
            """slot -> device: one copy of the packed buffer, then per-name views of it (names that do not live in the
            packed buffer -- a caller's own extra draws -- are copied one by one)."""
            packed = [k for k in names if k in slot["_off"] and slot[k].data_ptr() == slot["_pin"].data_ptr() + 4 * slot["_off"][k][0]]
            if "_all" not in dbuf:
                dbuf["_all"] = torch.empty(slot["_pin"].shape, dtype=torch.float32, device=dev)
            ops.stage_in(dbuf["_all"], slot["_pin"])          # (a kernel that reads the pinned buffer: svs_stage_in)
            for k in names:
                if k in packed:
                    off, n = slot["_off"][k]
                    v = dbuf["_all"][off:off + n]
                    dbuf[k] = (v.view(torch.int32) if slot[k].dtype == torch.int32 else v).view(slot[k].shape)
                else:
                    if k not in dbuf or dbuf[k].shape != slot[k].shape:
                        dbuf[k] = torch.empty(slot[k].shape, dtype=slot[k].dtype, device=dev)
                    dbuf[k].copy_(slot[k], non_blocking=True)
            return {k: dbuf[k] for k in names}

        if stream is not None and out is None:
            cur = torch.cuda.current_stream()
            last = ring.get("last")
            if last is not None:
                last["consumed"] = torch.cuda.Event()
                last["consumed"].record(cur)
            ring["last"] = slot
            dbuf = slot.setdefault("dev", {})
            if slot.get("consumed") is not None:
                stream.wait_event(slot["consumed"])
            with torch.cuda.stream(stream):
                res = upload(dbuf)
                slot["event"] = torch.cuda.Event()
                slot["event"].record(stream)
            cur.wait_event(slot["event"])
            return res
        if out is None:
            out = {}
        res = upload(out)
        slot["event"] = torch.cuda.Event()
        slot["event"].record()
        return res

    def get_z_vals(self, ray_dirs, cam_loc, model, fast=-1, iter_step=None, rng=None):
        """ray_dirs (R,3), cam_loc (R,3) or (3,) -> z_vals (R, N_samples+N_samples_extra+2), z_samples_eik (R,1).
        rng: optional pre-drawn train-mode draws for exactly these rays (a slice of draw_train_rng's output)."""
        dev = ray_dirs.device
        R = ray_dirs.shape[0]
        max_iters = fast if fast >= 0 else self.max_total_iters
        if model.training:
            if max_iters != 1:
                raise NotImplementedError("train-mode sampling is implemented for fast=1 (what VolOpt.train_step uses)")
            if rng is None:
                rng = self.draw_train_rng(R, dev)
        else:
            rng = None
        if not hasattr(self, "_ws_by_R"):
            self._ws_by_R = {}
        # group_rays: rays per convergence group (None: the batch of this call, as in the reference; the image
        # renderer sets it to split_n_pixels so that one large launch reproduces the reference's per-chunk decisions)
        group = getattr(self, "group_rays", None)
        key = (R, str(dev), torch.cuda.current_stream().cuda_stream, group)
        if key not in self._ws_by_R:
            self._ws_by_R[key] = ops.SamplerWorkspace(R, dev, group_rays=group)
        self._ws = self._ws_by_R[key]
        net = model.implicit_network
        z, z_eik = ops.sample_rays(model.packed_mlp(), cam_loc, ray_dirs, model.density.beta, beta_min=model.density.beta_min_value, near=self.near,
                                   scene_bounding_sphere=self.scene_bounding_sphere, sphere_scale=net.sphere_scale,
                                   sdf_clamp_radius=net.sdf_bounding_sphere, N_samples=self.N_samples,
                                   N_samples_eval=self.N_samples_eval, N_samples_extra=self.N_samples_extra, eps=self.eps,
                                   beta_iters=self.beta_iters, max_total_iters=self.max_total_iters, fast=fast,
                                   training=model.training, inverse_sphere_bg=self.inverse_sphere_bg,
                                   add_tiny=self.add_tiny, inv_4log=self._inv_4log, rng=rng, workspace=self._ws)
        if self.inverse_sphere_bg:
            # inverse-sphere samples + their 4-D points in one small kernel; the model picks the points up from
            # _bg_last (z_bg descending, points, conventional depths); the reference returns z_bg ascending
            jit = rng.get("jitter_bg") if (model.training and rng) else None
            self._bg_last = ops.bg_points(cam_loc, ray_dirs, self.inverse_sphere_sampler.N_samples,
                                          self.scene_bounding_sphere, jitter=jit)
            # (the reference returns z_bg ascending; the drop-in model reads _bg_last and asks for no flip)
            z = (z, torch.flip(self._bg_last[0], dims=[-1]) if getattr(self, "return_bg_ascending", True) else None)
        return z, z_eik

    def get_error_bound(self, beta, model, sdf, z_vals, dists, d_star):
        """ray_sampler.py:221-229 on tensors (host glue for analysis; the sampler kernels evaluate it in LDS)."""
        density = model.density(sdf.reshape(z_vals.shape), beta=beta)
        sfe = torch.cat([torch.zeros(dists.shape[0], 1, device=dists.device), dists * density[:, :-1]], dim=-1)
        integral = torch.cumsum(sfe, dim=-1)
        err_int = torch.cumsum(torch.exp(-d_star / beta) * (dists ** 2.) / (4 * beta ** 2), dim=-1)
        bound = (torch.clamp(torch.exp(err_int), max=1.e6) - 1.0) * torch.exp(-integral[:, :-1])
        return bound.max(-1)[0]
