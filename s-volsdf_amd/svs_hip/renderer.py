"""Whole-image rendering: the chunk pipeline of VolOpt.render_step (volsdf/vsdf.py:237-287) and render_mvs.

The reference renders an image as ceil(total_pixels / split_n_pixels) forward calls of split_n_pixels = 500 rays and
moves six tensors per chunk to the host (885 chunks x 6 D2H copies for a 768x576 image, volsdf/utils/general.py:24-58).
Here one forward call covers `rays_per_launch` rays (thousands: the MLP kernels only fill the GPU from ~32k points
upwards and the activations of 16k rays fit easily in 288 GB), its outputs are written straight into full-image
DEVICE tensors, and nothing is copied to the host unless the caller asks for it.

The chunk size is part of the reference's result: the sampler decides "not converged -> one more up-sampling round"
per forward call, i.e. per chunk of split_n_pixels rays (ray_sampler.py:136).  That decision is reproduced exactly:
the sampler's device-side control block has one record per group of split_n_pixels rays
(ErrorBoundSampler.group_rays), so a large launch makes the same per-chunk decisions as 500-ray calls.
"""
import torch


@torch.no_grad()
def render_image(model, model_input, total_pixels, split_n_pixels=500, rays_per_launch=8000, fast=-1,
                 keys=("rgb_values", "normal_map", "depth_values", "depth_vals", "weights", "xyz")):
    """model: VolSDFNetwork in eval mode; model_input: dict(intrinsics (1,4,4), uv (1,total_pixels,2), pose (1,4,4)).
    Returns the merged outputs of VolOpt.render_step (the arrays utils.merge_output builds) as device tensors:
    rgb_values (N,3), normal_map (N,3), depth_values (N,1), depth_vals (N,S), weights (N,S), xyz (N,S,3)."""
    if model.training:
        raise ValueError("render_image renders in eval mode (VolOpt.render_step calls model.eval())")
    uv = model_input["uv"]
    if uv.shape[0] != 1 or uv.shape[1] != total_pixels:
        raise ValueError("uv must be (1, total_pixels, 2)")
    per_launch = max(split_n_pixels, (rays_per_launch // split_n_pixels) * split_n_pixels)
    sampler = model.ray_sampler
    prev_group = getattr(sampler, "group_rays", None)
    sampler.group_rays = split_n_pixels
    out = {}
    try:
        for lo in range(0, total_pixels, per_launch):
            hi = min(lo + per_launch, total_pixels)
            chunk = dict(model_input)
            chunk["uv"] = uv[:, lo:hi].contiguous()
            res = model(chunk, fast=fast)
            for k in keys:
                v = res[k]
                if k not in out:
                    out[k] = torch.empty((total_pixels,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                out[k][lo:hi].copy_(v)
    finally:
        sampler.group_rays = prev_group
    return out


def depth_image(outputs, img_res, scale_factor=1.0, min_acc=0.2):
    """The depth map VolOpt.render_step hands to the MVS stage (vsdf.py:259-263): depth_values as an (H, W) image times
    the dataset's scale factor; pixels whose accumulated weight is below 0.2 get the maximum depth."""
    H, W = img_res
    depth = outputs["depth_values"].reshape(H, W) * scale_factor
    acc = outputs["weights"].sum(1).reshape(H, W)
    return torch.where(acc < min_acc, depth.max(), depth)
