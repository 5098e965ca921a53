"""VolSDF networks with the reference's class names, constructor arguments, state-dict keys and
forward() contract (volsdf/model/network.py), evaluated by the fused HIP kernels of svs_mlp.hip.

`train.model_class: volsdf.model.network.VolSDFNetwork` (config/vol/dtu.yaml:4) resolves to this module
when s-volsdf_amd/ precedes the reference on sys.path; checkpoints interchange with the reference
(`implicit_network.lin{0..8}.{weight_g,weight_v,bias}`, `rendering_network.lin{0..4}.*`, `density.beta`).
"""
import numpy as np
import torch
import torch.nn as nn

from svs_hip import ops
from volsdf.model.density import LaplaceDensity
from volsdf.model.embedder import get_embedder
from volsdf.model.ray_sampler import ErrorBoundSampler
from volsdf.utils import rend_util


def _dev(module):
    return next(module.parameters()).device


class ImplicitNetwork(nn.Module):
    """SDF MLP, network.py:10-131.  Parameters are created exactly as the reference creates them (nn.Linear +
    geometric initialisation + weight_norm), so seeds and checkpoints carry over; evaluation is the fused
    kernel, which is specialised to the 8 x 256 / skip-4 / PE-6 / 256-feature architecture of every config."""

    def __init__(self, feature_vector_size, sdf_bounding_sphere, d_in, d_out, dims, geometric_init=True, bias=1.0,
                 skip_in=(), weight_norm=True, multires=0, sphere_scale=1.0):
        super().__init__()
        self.sdf_bounding_sphere, self.sphere_scale = sdf_bounding_sphere, sphere_scale
        dims = [d_in] + list(dims) + [d_out + feature_vector_size]
        self.embed_fn = None
        if multires > 0:
            self.embed_fn, dims[0] = get_embedder(multires, input_dims=d_in)
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.weight_norm = weight_norm
        self._supported = (d_in == 3 and d_out == 1 and feature_vector_size == 256 and multires == 6 and
                           list(dims[1:-1]) == [256] * 8 and self.skip_in == (4,))
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if l + 1 in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                if l == self.num_layers - 2:
                    torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    torch.nn.init.constant_(lin.bias, -bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self._packed, self._packed_key = None, None

    # ---- packed weights ----------------------------------------------------------------------------
    def _layer_tensors(self):
        n = self.num_layers - 1
        lins = [getattr(self, f"lin{l}") for l in range(n)]
        if self.weight_norm:
            return [m.weight_v for m in lins], [m.weight_g for m in lins], [m.bias for m in lins]
        return [m.weight for m in lins], None, [m.bias for m in lins]

    def packed(self, owner=None):
        if not self._supported:
            raise NotImplementedError("the fused SDF kernel is built for d_in=3, 8x256, skip_in=[4], multires=6, 256 features")
        v, g, b = self._layer_tensors()
        key = tuple((t.data_ptr(), t._version) for t in v + (g or []) + b)
        pk = owner if owner is not None else self._packed
        if pk is None:
            pk = self._packed = ops.PackedMlp(v[0].device)
        if self._packed_key != key or getattr(pk, "_sdf_key", None) != key:
            pk.pack_sdf(v, g, b)
            pk._sdf_key = self._packed_key = key
        return pk

    # ---- reference call surface (inference; gradients w.r.t. parameters arrive with the backward kernels) ----
    def forward(self, input):
        """(P,3) -> (P,257) = [sdf (unclamped), feature]."""
        sdf, _, _, _, rows = ops.sdf_outputs(self.packed(), ops.PointSource(points=input), 0.0, self.sphere_scale,
                                             want_feature_rows=True)
        return torch.cat([sdf, rows], 1)

    def gradient(self, x, with_sdf=False):
        sdf, grad, _, _, _ = ops.sdf_outputs(self.packed(), ops.PointSource(points=x), 0.0, self.sphere_scale)
        return (grad, sdf) if with_sdf else grad

    def get_outputs(self, x):
        sdf, grad, _, _, rows = ops.sdf_outputs(self.packed(), ops.PointSource(points=x), self.sdf_bounding_sphere,
                                                self.sphere_scale, want_feature_rows=True)
        return sdf, rows, grad

    def get_sdf_vals(self, x):
        return ops.sdf_vals(self.packed(), ops.PointSource(points=x), self.sdf_bounding_sphere, self.sphere_scale)


class RenderingNetwork(nn.Module):
    """Radiance MLP, network.py:134-190 (mode 'idr', 4 x 256, PE-1 view directions)."""

    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0):
        super().__init__()
        self.mode = mode
        dims = [d_in + feature_vector_size] + list(dims) + [d_out]
        self.embedview_fn = None
        if multires_view > 0:
            self.embedview_fn, input_ch = get_embedder(multires_view)
            dims[0] += input_ch - 3
        self.num_layers = len(dims)
        self.weight_norm = weight_norm
        self._supported = (mode == "idr" and dims == [271, 256, 256, 256, 256, 3])
        for l in range(self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self._packed_key = None

    def pack_into(self, pk):
        if not self._supported:
            raise NotImplementedError("the fused radiance kernel is built for mode='idr', 271->4x256->3")
        lins = [getattr(self, f"lin{l}") for l in range(self.num_layers - 1)]
        if self.weight_norm:
            v, g, b = [m.weight_v for m in lins], [m.weight_g for m in lins], [m.bias for m in lins]
        else:
            v, g, b = [m.weight for m in lins], None, [m.bias for m in lins]
        key = tuple((t.data_ptr(), t._version) for t in v + (g or []) + b)
        if getattr(pk, "_rgb_key", None) != key:
            pk.pack_rgb(v, g, b)
            pk._rgb_key = key
        return pk


class VolSDFNetwork(nn.Module):
    """network.py:192-295.  forward(input, fast) -> dict with the reference's keys."""

    def __init__(self, conf):
        super().__init__()
        self.feature_vector_size = conf.get_int('feature_vector_size')
        self.scene_bounding_sphere = conf.get_float('scene_bounding_sphere', default=1.0)
        self.white_bkgd = conf.get_bool('white_bkgd', default=False)
        self.register_buffer("bg_color", torch.tensor(conf.get_list("bg_color", default=[1.0, 1.0, 1.0])).float(),
                             persistent=False)
        self.implicit_network = ImplicitNetwork(self.feature_vector_size,
                                                0.0 if self.white_bkgd else self.scene_bounding_sphere,
                                                **conf.get_config('implicit_network'))
        self.rendering_network = RenderingNetwork(self.feature_vector_size, **conf.get_config('rendering_network'))
        self.density = LaplaceDensity(**conf.get_config('density'))
        self.ray_sampler = ErrorBoundSampler(self.scene_bounding_sphere, **conf.get_config('ray_sampler'))
        self._pk = None

    def packed_mlp(self, rgb=True):
        """Packed weight streams (re-packed only when a parameter changed).  rgb=False: the SDF streams only (the fused
        train step packs the radiance stream on a side stream, trainer.TrainStep)."""
        if self._pk is None or self._pk.device != _dev(self):
            self._pk = ops.PackedMlp(_dev(self))
        self.implicit_network.packed(owner=self._pk)
        if rgb:
            self.rendering_network.pack_into(self._pk)
        return self._pk

    # ---- parameters in a fixed order (shared by the autograd bridge and the fused trainer) -----------------
    def mlp_params(self):
        """((sdf weight_v, weight_g, bias), (rgb weight_v, weight_g, bias)) as lists per layer."""
        def grab(net, n):
            lins = [getattr(net, f"lin{l}") for l in range(n)]
            if net.weight_norm:
                return [m.weight_v for m in lins], [m.weight_g for m in lins], [m.bias for m in lins]
            return [m.weight for m in lins], None, [m.bias for m in lins]
        return grab(self.implicit_network, 9), grab(self.rendering_network, 5)

    def _flat_param_list(self):
        (sv, sg, sb), (rv, rg, rb) = self.mlp_params()
        out = []
        for v, g, b in ((sv, sg, sb), (rv, rg, rb)):
            for l in range(len(v)):
                out += [v[l]] + ([g[l]] if g is not None else []) + [b[l]]
        return out + [self.density.beta]

    def invalidate_packed(self):
        """Call after parameters were changed outside torch's version counter (fused optimiser kernel)."""
        self.implicit_network._packed_key = None
        if self._pk is not None:
            self._pk._sdf_key = self._pk._rgb_key = None

    def forward(self, input, fast=-1):
        if self.training and torch.is_grad_enabled():
            params = self._flat_param_list()
            input, n_valid, n_pad = pad_rays(input, self.ray_sampler.N_samples + self.ray_sampler.N_samples_extra + 2)
            res = _RenderFunction.apply(self, input, fast, *params)
            rgb_values, depth_values, weights, grad_theta, depth_vals, xyz = res
            out = {'rgb_values': rgb_values, 'depth_values': depth_values, 'depth_vals': depth_vals, 'weights': weights,
                   'xyz': xyz, 'grad_theta': grad_theta}
            return cut_rays(out, n_valid, n_pad)
        return self._forward_impl(input, fast, None)

    def draw_train_rng(self, R, dev, out=None, stream=None):
        """All train-mode random draws of one forward for R rays, in the reference's order (sampler draws, then the
        uniform eikonal points of network.py:261).  Slices of it can be handed to _forward_impl per ray group."""
        rb = self.scene_bounding_sphere

        def eik(slot, n):
            if n is None:       # CPU tensors (no device)
                slot["eik_points"] = torch.empty(R, 3).uniform_(-rb, rb)
                return ["eik_points"]
            if "eik_points" not in slot:                 # (the sampler's ring reserves it inside its packed pinned buffer)
                slot["eik_points"] = torch.empty(R, 3).pin_memory()
            slot["eik_points"].uniform_(-rb, rb)
            return ["eik_points"]

        return self.ray_sampler.draw_train_rng(R, dev, extra=eik, out=out, stream=stream)

    @staticmethod
    def slice_rng(rng, lo, hi):
        return {k: (v if k == "perm" else v[lo:hi].contiguous()) for k, v in rng.items() if not k.startswith("_")}

    def _forward_impl(self, input, fast, keep, rng=None):
        """network.py:206-279 on the HIP kernels.  keep: dict that receives what the backward kernels need.
        rng: optional pre-drawn random draws for exactly these rays (train mode)."""
        intrinsics, uv, pose = input["intrinsics"], input["uv"], input["pose"]
        if uv.shape[0] != 1:
            raise NotImplementedError("batch_size 1 only (runner.py:166)")
        net = self.implicit_network
        pk = self.packed_mlp()
        ray_dirs, cam_loc, depth_scale = ops.rays_from_uv(uv[0], pose[0], intrinsics[0])       # network.py:213-217
        num_pixels = ray_dirs.shape[0]

        if self.training and rng is None:
            n_valid = input.get("_valid_rays", num_pixels)
            rng = self.draw_train_rng(n_valid, ray_dirs.device)
            if n_valid < num_pixels:
                rng = pad_rng(rng, num_pixels)
        z_vals, z_samples_eik = self.ray_sampler.get_z_vals(ray_dirs, cam_loc, self, fast=fast,
                                                            iter_step=input.get("iter_step", 1), rng=rng)
        N_samples = z_vals.shape[1]
        n_main = num_pixels * N_samples
        hook = input.get("_after_sampling")          # (trainer: work that depends on the sample depths only)
        if hook is not None:
            hook(cam_loc, ray_dirs, z_vals)
        eikonal_points = None
        if self.training:
            # eikonal samples (network.py:258-266) ride in the same launch as the ray samples; they
            # differentiate the raw network output (ImplicitNetwork.gradient), the ray samples the clamped sdf
            # [uniform points in the bounding sphere ; one point per ray at the sampler's extra depth]: one launch straight
            # into the launch's point list
            eikonal_points = ops.eikonal_points(rng["eik_points"], cam_loc, z_samples_eik, ray_dirs)
        src = ops.PointSource(points=eikonal_points, cam=cam_loc, dirs=ray_dirs, z=z_vals)
        sdf, gradients, feat_tiles, _, _ = ops.sdf_outputs(pk, src, net.sdf_bounding_sphere, net.sphere_scale,
                                                           clamp_n=n_main, keep=keep)
        grad_theta = gradients[n_main:]
        sdf, gradients = sdf[:n_main], gradients[:n_main]
        src_main = ops.PointSource(cam=cam_loc, dirs=ray_dirs, z=z_vals)
        hook = input.get("_before_rgb")              # (trainer: the radiance weight stream is packed on another stream)
        if hook is not None:
            hook()
        rgb_flat = ops.rgb_eval(pk, src_main, gradients, ray_dirs, feat_tiles, keep=keep)
        comp = ops.composite(z_vals, sdf, rgb_flat, depth_scale, self.density.beta, self.density.beta_min_value,
                             normals=None if self.training else gradients)
        if keep is not None:
            keep.update(z_vals=z_vals, sdf=sdf, rgb_flat=rgb_flat, depth_scale=depth_scale, cam_loc=cam_loc,
                        ray_dirs=ray_dirs)
        rgb_values = comp["rgb_values"]
        if self.white_bkgd:
            acc_map = torch.sum(comp["weights"], -1)
            rgb_values = rgb_values + (1. - acc_map[..., None]) * self.bg_color.unsqueeze(0)
        output = {'rgb_values': rgb_values, 'depth_values': comp["depth_values"], 'depth_vals': comp["depth_vals"],
                  'weights': comp["weights"]}
        if not input.get("_skip_xyz"):
            # (the reference hands the sample positions to cost_mapping; the fused train step looks the prior up from
            # (cam, dirs, z) directly and skips these two launches: svs_hip/trainer.py)
            output['xyz'] = cam_loc.view(1, 1, 3) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)
        if self.training:
            output['grad_theta'] = grad_theta
        else:
            output['normal_map'] = comp["normal_map"]
        return output

    def backward_from_output_grads(self, keep, g_rgb_values, g_weights=None, g_depth_values=None, g_grad_theta=None,
                                   out=None):
        """d loss / d parameters from d loss / d (rgb_values, weights, depth_values, grad_theta): compositing
        backward, then the fused MLP backward.  Returns (sdf_grads, rgb_grads, d_beta)."""
        from svs_hip.train import MlpBackward
        dev = keep["z_vals"].device
        if getattr(self, "_mlp_bwd", None) is None or self._mlp_bwd.dev != dev:
            self._mlp_bwd = MlpBackward(dev)
        R = keep["z_vals"].shape[0]
        if g_rgb_values is None:
            g_rgb_values = torch.zeros(R, 3, device=dev)
        g_weights = self.white_bkgd_weight_grad(g_rgb_values, g_weights, keep["z_vals"].shape[1])
        d_sdf, d_rgb, d_beta = ops.composite_bwd(keep["z_vals"], keep["sdf"], keep["rgb_flat"], keep["depth_scale"],
                                                 self.density.beta, self.density.beta_min_value, g_rgb_values,
                                                 g_weights, g_depth_values)
        n_extra = keep["src"].n - keep["rgb"].shape[0]
        if g_grad_theta is None and n_extra:
            g_grad_theta = torch.zeros(n_extra, 3, device=dev)
        sdf_p, rgb_p = self.mlp_params()
        sdf_g, rgb_g = self._mlp_bwd.run(sdf_p, rgb_p, keep, d_rgb, d_sdf, g_grad_theta, out=out)
        return sdf_g, rgb_g, d_beta

    def white_bkgd_weight_grad(self, g_rgb_values, g_weights, S):
        """white_bkgd (network.py:246-248): rgb_values += (1 - sum(weights)) * bg_color, so every weight of a ray receives
        -(d loss / d rgb_values) . bg_color on top of its direct gradient.  No-op for the default black background."""
        if not self.white_bkgd:
            return g_weights
        extra = -(g_rgb_values @ self.bg_color.to(g_rgb_values.device)).unsqueeze(1).expand(-1, S)
        return extra.contiguous() if g_weights is None else g_weights + extra

    def volume_rendering(self, z_vals, sdf):
        """network.py:281-295 -> (weights, dists)."""
        R = z_vals.shape[0]
        dev = z_vals.device
        comp = ops.composite(z_vals, sdf, torch.zeros(z_vals.numel(), 3, device=dev), torch.ones(R, 1, device=dev),
                             self.density.beta, self.density.beta_min_value)
        dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full((R, 1), 1e10, device=dev)], -1)
        return comp["weights"], dists


def pad_rng(rng, n_pad):
    """Train-mode draws made for the caller's rays, extended to a padded batch by repeating the last ray's rows: the random
    stream is consumed exactly as for the unpadded batch and the caller's rays see the same draws."""
    out = {}
    for k, v in rng.items():
        if k.startswith("_"):                  # (the packed upload buffer of ray_sampler.draw_train_rng)
            continue
        if k == "perm" or not torch.is_tensor(v) or v.shape[0] >= n_pad:
            out[k] = v
        else:
            out[k] = torch.cat([v, v[-1:].expand(n_pad - v.shape[0], *v.shape[1:])], 0).contiguous()
    return out


def pad_rays(input, samples_per_ray):
    """The backward kernels want rays x samples to be a multiple of 32 (svs_hip/train.py): a batch that is not gets its last
    ray repeated.  -> (input, rays of the caller, rays after padding)."""
    import math
    R = input["uv"].shape[1]
    pad = (-R) % (32 // math.gcd(samples_per_ray, 32))
    if pad == 0:
        return input, R, R
    uv = input["uv"]
    inp = dict(input)
    inp["uv"] = torch.cat([uv, uv[:, -1:].expand(uv.shape[0], pad, uv.shape[2])], 1)
    inp["_valid_rays"] = R                  # _forward_impl draws for R rays and repeats the last ray's draws (pad_rng)
    return inp, R, R + pad


def cut_rays(out, n_valid, n_pad):
    """Model outputs of a padded batch without the padding rays.  Plain slicing: autograd hands the padding zero gradients."""
    if n_valid == n_pad:
        return out
    res = {}
    for k, t in out.items():
        if not torch.is_tensor(t) or t.dim() == 0:
            res[k] = t
        elif k == "grad_theta" and t.shape[0] == 2 * n_pad:
            res[k] = torch.cat([t[:n_valid], t[n_pad:n_pad + n_valid]], 0)
        elif t.shape[0] == n_pad:
            res[k] = t[:n_valid]
        elif t.shape[0] % n_pad == 0:                      # flattened (rays x samples, ...) tensors
            res[k] = t.reshape(n_pad, -1, *t.shape[1:])[:n_valid].reshape(-1, *t.shape[1:])
        else:
            res[k] = t
    return res


class _RenderFunction(torch.autograd.Function):
    """Autograd bridge: lets `loss.backward()` of the reference trainer (volsdf/vsdf.py:215) drive the hand-written
    backward kernels.  Inputs after `fast` are the parameters in VolSDFNetwork._flat_param_list() order."""

    @staticmethod
    def forward(ctx, model, input, fast, *params):
        keep = {}
        out = model._forward_impl(input, fast, keep)
        ctx.model, ctx.keep = model, keep
        ctx.mark_non_differentiable(out['depth_vals'], out['xyz'])
        return (out['rgb_values'], out['depth_values'], out['weights'], out['grad_theta'], out['depth_vals'], out['xyz'])

    @staticmethod
    def backward(ctx, g_rgb_values, g_depth_values, g_weights, g_grad_theta, _g1, _g2):
        model = ctx.model
        sdf_g, rgb_g, d_beta = model.backward_from_output_grads(ctx.keep, g_rgb_values, g_weights, g_depth_values,
                                                                g_grad_theta)
        grads = []
        for group, wn in ((sdf_g, model.implicit_network.weight_norm), (rgb_g, model.rendering_network.weight_norm)):
            for gv, gg, gb in group:
                grads += [gv] + ([gg] if wn else []) + [gb]
        grads.append(d_beta.reshape(model.density.beta.shape))
        return (None, None, None, *grads)
