"""VolSDFNetworkBG with the reference's class name, constructor, state-dict keys and forward() contract
(volsdf/model/network_bg.py): the foreground VolSDF model plus the NeRF++ inverted-sphere background, evaluated by
the fused HIP kernels (svs_mlp_h2.hip for the foreground networks, svs_bg_h2.hip for the background networks,
svs_render.hip for the inverse-sphere points and the fg/bg compositing).

`train.model_class: volsdf.model.network_bg.VolSDFNetworkBG` (config/vol/bmvs.yaml:4).  Checkpoint keys:
`implicit_network.*`, `rendering_network.*`, `density.beta`, `bg_implicit_network.lin{0..8}.{weight,bias}`,
`bg_rendering_network.lin{0,1}.{weight,bias}`.
"""
import os

import torch
import torch.nn as nn

from svs_hip import ops
from volsdf.model.density import AbsDensity, LaplaceDensity
from volsdf.model.network import ImplicitNetwork, RenderingNetwork, _dev
from volsdf.model.ray_sampler import ErrorBoundSampler


class VolSDFNetworkBG(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.feature_vector_size = conf.get_int('feature_vector_size')
        self.scene_bounding_sphere = conf.get_float('scene_bounding_sphere', default=1.0)
        # foreground object's networks (no sphere clamp: network_bg.py:25)
        self.implicit_network = ImplicitNetwork(self.feature_vector_size, 0.0, **conf.get_config('implicit_network'))
        self.rendering_network = RenderingNetwork(self.feature_vector_size, **conf.get_config('rendering_network'))
        self.density = LaplaceDensity(**conf.get_config('density'))
        self.ray_sampler = ErrorBoundSampler(self.scene_bounding_sphere, inverse_sphere_bg=True,
                                             **conf.get_config('ray_sampler'))
        # background's networks
        bg_feature_vector_size = conf.get_int('bg_network.feature_vector_size')
        self.bg_implicit_network = ImplicitNetwork(bg_feature_vector_size, 0.0, **conf.get_config('bg_network.implicit_network'))
        self.bg_rendering_network = RenderingNetwork(bg_feature_vector_size, **conf.get_config('bg_network.rendering_network'))
        self.bg_density = AbsDensity(**conf.get_config('bg_network.density', default={}))
        bi, br = self.bg_implicit_network, self.bg_rendering_network
        if not (bi.num_layers == 10 and not bi.weight_norm and bi.lin0.weight.shape == (256, 84) and
                bi.lin3.weight.shape == (172, 256) and bi.lin8.weight.shape == (257, 256) and br.mode == "nerf" and
                not br.weight_norm and br.num_layers == 3 and br.lin0.weight.shape == (128, 283)):
            raise NotImplementedError("the background kernels are built for bmvs.yaml's bg_network (4-D PE-10 8x256 "
                                      "skip-4 implicit network, 283->128->3 'nerf' radiance network, no weight-norm)")
        self._pk = self._pk_bg = None
        self._bg_key = None
        self._bg_streams = {}              # per calling stream: the stream the background networks' forward runs on

    # ---- packed weights -------------------------------------------------------------------------------------
    def packed_mlp(self, rgb=True):
        if self._pk is None or self._pk.device != _dev(self):
            self._pk = ops.PackedMlp(_dev(self))
        self.implicit_network.packed(owner=self._pk)
        if rgb:
            self.rendering_network.pack_into(self._pk)
        return self._pk

    def bg_params(self):
        bi, br = self.bg_implicit_network, self.bg_rendering_network
        sw = [getattr(bi, f"lin{l}").weight for l in range(9)], [getattr(bi, f"lin{l}").bias for l in range(9)]
        rw = [getattr(br, f"lin{l}").weight for l in range(2)], [getattr(br, f"lin{l}").bias for l in range(2)]
        return sw, rw

    def packed_bg(self):
        if self._pk_bg is None or self._pk_bg.device != _dev(self):
            self._pk_bg, self._bg_key = ops.PackedBg(_dev(self)), None
        sw, rw = self.bg_params()
        key = tuple((t.data_ptr(), t._version) for t in sw[0] + sw[1] + rw[0] + rw[1])
        if key != self._bg_key:
            self._pk_bg.pack(sw, rw)
            self._bg_key = key
        return self._pk_bg

    def mlp_params(self):
        """((sdf weight_v, weight_g, bias), (rgb weight_v, weight_g, bias)) of the foreground networks"""
        def grab(net, n):
            lins = [getattr(net, f"lin{l}") for l in range(n)]
            return [m.weight_v for m in lins], [m.weight_g for m in lins], [m.bias for m in lins]
        return grab(self.implicit_network, 9), grab(self.rendering_network, 5)

    def _flat_param_list(self):
        (sv, sg, sb), (rv, rg, rb) = self.mlp_params()
        out = []
        for v, g, b in ((sv, sg, sb), (rv, rg, rb)):
            for l in range(len(v)):
                out += [v[l], g[l], b[l]]
        out.append(self.density.beta)
        for w, b in self.bg_params():
            for l in range(len(w)):
                out += [w[l], b[l]]
        return out

    def invalidate_packed(self):
        self.implicit_network._packed_key = None
        if self._pk is not None:
            self._pk._sdf_key = self._pk._rgb_key = None
        self._bg_key = None

    # ---- forward --------------------------------------------------------------------------------------------
    def draw_train_rng(self, R, dev, out=None, stream=None):
        rb = self.scene_bounding_sphere

        def eik(slot, n):
            if n is None:
                slot["eik_points"] = torch.empty(R, 3).uniform_(-rb, rb)
                return ["eik_points"]
            if "eik_points" not in slot:
                slot["eik_points"] = torch.empty(R, 3).pin_memory()
            slot["eik_points"].uniform_(-rb, rb)
            return ["eik_points"]

        return self.ray_sampler.draw_train_rng(R, dev, extra=eik, out=out, stream=stream)

    @staticmethod
    def slice_rng(rng, lo, hi):
        return {k: (v if k == "perm" else v[lo:hi].contiguous()) for k, v in rng.items() if not k.startswith("_")}

    def forward(self, input, fast=-1):
        if self.training and torch.is_grad_enabled():
            # autograd bridge: the reference's own train_step (loss.backward(), clip_grad_norm_, torch Adam,
            # volsdf/vsdf.py:214-219) drives the hand-written backward kernels, as with the DTU model
            from .network import cut_rays, pad_rays
            rs = self.ray_sampler
            input, n_valid, n_pad = pad_rays(input, rs.N_samples + rs.N_samples_extra + 1)
            res = _RenderFunctionBG.apply(self, input, fast, *self._flat_param_list())
            rgb_values, depth_values_all, depth_values, weights, grad_theta, depth_vals, xyz = res
            return cut_rays({'rgb_values': rgb_values, 'depth_values_all': depth_values_all, 'depth_values': depth_values,
                             'depth_vals': depth_vals, 'weights': weights, 'xyz': xyz, 'grad_theta': grad_theta}, n_valid, n_pad)
        return self._forward_impl(input, fast, None)

    def backward_from_output_grads(self, keep, g_rgb_values, g_weights=None, g_depth_values=None, g_depth_values_all=None,
                                   g_grad_theta=None):
        """d loss / d parameters (in _flat_param_list() order) from d loss / d (rgb_values, weights, depth_values,
        depth_values_all, grad_theta): fg / bg compositing backward, then the fused MLP backwards of all four networks."""
        from svs_hip.train import BgBackward, MlpBackward, finalize
        dev = keep["z_vals"].device
        if getattr(self, "_mlp_bwd", None) is None or self._mlp_bwd.dev != dev:
            self._mlp_bwd, self._bg_bwd = MlpBackward(dev), BgBackward(dev)
        R = keep["z_vals"].shape[0]
        if g_rgb_values is None:
            g_rgb_values = torch.zeros(R, 3, device=dev)
        d_sdf, d_rgb, d_bo, d_brgb, d_beta = ops.composite_bg_bwd(
            keep["z_vals"], keep["z_max"], keep["sdf"], keep["rgb_flat"], keep["depth_scale"], self.density.beta,
            self.density.beta_min_value, keep["z_bg"], keep["bg_out0"], keep["bg_rgb"], g_rgb_values, g_weights,
            g_depth_values, d_depth_values_all=g_depth_values_all, bg_depth=keep["bg_depth"])
        n_extra = keep["src"].n - keep["rgb"].shape[0]
        if g_grad_theta is None and n_extra:
            g_grad_theta = torch.zeros(n_extra, 3, device=dev)
        sdf_p, rgb_p = self.mlp_params()
        bg_sdf_wb, bg_rgb_wb = self.bg_params()
        bw, bgb = self._mlp_bwd, self._bg_bwd
        bw.streams.pack(sdf_p, rgb_p)
        bgb.pack(bg_sdf_wb, bg_rgb_wb)
        bw.accum.zero(); bgb.zero()
        bgb.accumulate(keep, d_brgb, d_bo)
        bw.accumulate(keep, d_rgb, d_sdf, g_grad_theta)
        sdf_g, rgb_g = finalize(bw.accum, sdf_p, rgb_p)
        bg_sdf_g, bg_rgb_g = bgb.finalize(bg_sdf_wb, bg_rgb_wb)
        grads = []
        for group in (sdf_g, rgb_g):
            for gv, gg, gb in group:
                grads += [gv, gg, gb]
        grads.append(d_beta.reshape(self.density.beta.shape))
        for group in (bg_sdf_g, bg_rgb_g):
            for gw, gb in group:
                grads += [gw, gb]
        return grads

    def _forward_impl(self, input, fast, keep, rng=None):
        """network_bg.py:37-145 on the HIP kernels."""
        intrinsics, uv, pose = input["intrinsics"], input["uv"], input["pose"]
        if uv.shape[0] != 1:
            raise NotImplementedError("batch_size 1 only (runner.py:166)")
        net = self.implicit_network
        pk, pkb = self.packed_mlp(), self.packed_bg()
        ray_dirs, cam_loc, depth_scale = ops.rays_from_uv(uv[0], pose[0], intrinsics[0])
        R = ray_dirs.shape[0]
        if self.training and rng is None:
            n_valid = input.get("_valid_rays", R)
            rng = self.draw_train_rng(n_valid, ray_dirs.device)
            if n_valid < R:
                from .network import pad_rng
                rng = pad_rng(rng, R)
        self.ray_sampler.return_bg_ascending = False          # (this model takes the background samples from _bg_last)
        try:
            (z_all, _), z_samples_eik = self.ray_sampler.get_z_vals(ray_dirs, cam_loc, self, fast=fast,
                                                                    iter_step=input.get("iter_step", 1), rng=rng)
        finally:
            self.ray_sampler.return_bg_ascending = True
        z_bg, bg_pts, bg_depth = self.ray_sampler._bg_last
        z_vals, z_max = ops.split_last(z_all)
        S, Nb = z_vals.shape[1], z_bg.shape[1]
        hook = input.get("_after_sampling")          # (trainer: work that depends on the sample depths only)
        if hook is not None:
            hook(cam_loc, ray_dirs, z_vals)
        n_main = R * S
        eikonal_points = None
        if self.training:
            eikonal_points = ops.eikonal_points(rng["eik_points"], cam_loc, z_samples_eik, ray_dirs)
        view_dirs = ray_dirs
        if not self.training:
            # nearest training view's directions (network_bg.py:69-74)
            view_dirs, _, _ = ops.rays_from_uv(uv[0], input["near_pose"][0].to(uv.device), intrinsics[0])

        def background():
            # network_bg.py:78-100: the background networks on the inverse-sphere samples
            hook = input.get("_before_bg")           # (trainer: their weight streams are packed on another stream)
            if hook is not None:
                hook()
            out0, feat = ops.bg_sdf_eval(pkb, bg_pts, keep=keep)
            return out0, ops.bg_rgb_eval(pkb, view_dirs, Nb, feat, R * Nb, keep=keep)

        # The background networks need the sampler's inverse-sphere points and the view directions only: in a train step
        # they run on a stream of their own BESIDE the fg networks (R x 32 points against R x 100: at 256 rays per GPU the
        # two launches together still fit the chip once) and compositing waits for them.  Not while a launch sequence is
        # being captured (a fork of a forked stream that joins its parent: the runtime defect of trainer._device_step's note).
        bg_join = None
        cur = torch.cuda.current_stream()
        if (self.training and keep is not None and os.environ.get("SVS_BG_SIDE", "1") != "0"
                and (not torch.cuda.is_current_stream_capturing() or getattr(self, "_side_ok_in_capture", False))):
            key = cur.cuda_stream
            bs = self._bg_streams.get(key)
            if bs is None:
                bs = self._bg_streams[key] = torch.cuda.Stream(device=ray_dirs.device)
            fork = torch.cuda.Event(); fork.record(cur)
            with torch.cuda.stream(bs):
                bs.wait_event(fork)
                bg_out0, bg_rgb = background()
                bg_join = torch.cuda.Event(); bg_join.record(bs)
        src = ops.PointSource(points=eikonal_points, cam=cam_loc, dirs=ray_dirs, z=z_vals)
        sdf, gradients, feat_tiles, _, _ = ops.sdf_outputs(pk, src, 0.0, net.sphere_scale, clamp_n=n_main, keep=keep)
        grad_theta = gradients[n_main:]
        sdf, gradients = sdf[:n_main], gradients[:n_main]
        src_main = ops.PointSource(cam=cam_loc, dirs=ray_dirs, z=z_vals)
        hook = input.get("_before_rgb")              # (trainer: the radiance weight stream is packed on another stream)
        if hook is not None:
            hook()
        rgb_flat = ops.rgb_eval(pk, src_main, gradients, view_dirs, feat_tiles, keep=keep)
        if bg_join is not None:
            torch.cuda.current_stream().wait_event(bg_join)
        else:
            bg_out0, bg_rgb = background()
        comp = ops.composite_bg(z_vals, z_max, sdf, rgb_flat, depth_scale, self.density.beta, self.density.beta_min_value,
                                z_bg, bg_out0, bg_rgb, bg_depth, normals=None if self.training else gradients)
        if keep is not None:
            keep.update(z_vals=z_vals, z_max=z_max, sdf=sdf, rgb_flat=rgb_flat, depth_scale=depth_scale, cam_loc=cam_loc,
                        ray_dirs=ray_dirs, z_bg=z_bg, bg_out0=bg_out0, bg_depth=bg_depth, comp=comp)
        points = None if input.get("_skip_xyz") else cam_loc.view(1, 1, 3) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)
        output = {'rgb_values': comp["rgb_values"], 'depth_values_all': comp["depth_values_all"],
                  'depth_values': comp["depth_values"], 'depth_vals': comp["depth_vals"], 'weights': comp["weights"],
                  'xyz': points}
        if self.training:
            output['grad_theta'] = grad_theta
        else:
            output['normal_map'] = comp["normal_map"]
        return output


class _RenderFunctionBG(torch.autograd.Function):
    """Autograd bridge of the fg + background model (see volsdf.model.network._RenderFunction).  Inputs after `fast` are
    the parameters in VolSDFNetworkBG._flat_param_list() order."""

    @staticmethod
    def forward(ctx, model, input, fast, *params):
        keep = {}
        out = model._forward_impl(input, fast, keep)
        ctx.model, ctx.keep = model, keep
        ctx.mark_non_differentiable(out['depth_vals'], out['xyz'])
        return (out['rgb_values'], out['depth_values_all'], out['depth_values'], out['weights'], out['grad_theta'],
                out['depth_vals'], out['xyz'])

    @staticmethod
    def backward(ctx, g_rgb_values, g_depth_values_all, g_depth_values, g_weights, g_grad_theta, _g1, _g2):
        grads = ctx.model.backward_from_output_grads(ctx.keep, g_rgb_values, g_weights, g_depth_values, g_depth_values_all,
                                                     g_grad_theta)
        return (None, None, None, *grads)
