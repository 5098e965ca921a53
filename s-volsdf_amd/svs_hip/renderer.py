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


def shard_pixels(total_pixels, split_n_pixels, rank, world):
    """Contiguous pixel range [lo, hi) of `rank`: whole chunks of split_n_pixels rays, so that every rank makes the
    same per-chunk sampler decisions as a single-GPU render (the chunks are independent; no exchange is needed until
    the image is assembled).  Chunks are dealt out as evenly as possible (the first ranks take one more)."""
    n_chunks = (total_pixels + split_n_pixels - 1) // split_n_pixels
    base, extra = divmod(n_chunks, world)
    c0 = rank * base + min(rank, extra)
    c1 = c0 + base + (1 if rank < extra else 0)
    return min(c0 * split_n_pixels, total_pixels), min(c1 * split_n_pixels, total_pixels)


def gather_image(part, lo, hi, total_pixels, world):
    """Assembles the full-image tensors from every rank's part (dict of (hi - lo, ...) tensors) with one all-gather per
    key (RCCL on the GPU box, gloo in the CPU test); ranks pad to the largest part."""
    import torch.distributed as dist
    dev = next(iter(part.values())).device        # (RCCL gathers device tensors; gloo in the CPU test takes host tensors)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo], dtype=torch.int64, device=dev))
    sizes = [int(s) for s in sizes]
    m = max(sizes)
    out = {}
    for k, v in part.items():
        pad = torch.zeros((m,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[:v.shape[0]] = v
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out[k] = torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)
        assert out[k].shape[0] == total_pixels
    return out


@torch.no_grad()
def render_image(model, model_input, total_pixels, split_n_pixels=500, rays_per_launch=8000, fast=-1,
                 keys=("rgb_values", "normal_map", "depth_values", "depth_vals", "weights", "xyz"), rank=0, world=1):
    """model: VolSDFNetwork in eval mode; model_input: dict(intrinsics (1,4,4), uv (1,total_pixels,2), pose (1,4,4)).
    Returns the merged outputs of VolOpt.render_step (the arrays utils.merge_output builds) as device tensors:
    rgb_values (N,3), normal_map (N,3), depth_values (N,1), depth_vals (N,S), weights (N,S), xyz (N,S,3).
    world > 1: every rank renders its shard_pixels() range of whole chunks and the image is assembled with
    gather_image (one all-gather per output)."""
    if model.training:
        raise ValueError("render_image renders in eval mode (VolOpt.render_step calls model.eval())")
    uv = model_input["uv"]
    if uv.shape[0] != 1 or uv.shape[1] != total_pixels:
        raise ValueError("uv must be (1, total_pixels, 2)")
    per_launch = max(split_n_pixels, (rays_per_launch // split_n_pixels) * split_n_pixels)
    sampler = model.ray_sampler
    prev_group = getattr(sampler, "group_rays", None)
    sampler.group_rays = split_n_pixels
    p_lo, p_hi = shard_pixels(total_pixels, split_n_pixels, rank, world) if world > 1 else (0, total_pixels)
    out = {}
    try:
        for lo in range(p_lo, p_hi, per_launch):
            hi = min(lo + per_launch, p_hi)
            chunk = dict(model_input)
            chunk["uv"] = uv[:, lo:hi].contiguous()
            res = model(chunk, fast=fast)
            for k in keys:
                v = res[k]
                if k not in out:
                    out[k] = torch.empty((p_hi - p_lo,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                out[k][lo - p_lo:hi - p_lo].copy_(v)
    finally:
        sampler.group_rays = prev_group
    if world > 1:
        return gather_image(out, p_lo, p_hi, total_pixels, world)
    return out


def depth_image(outputs, img_res, scale_factor=1.0, min_acc=0.2):
    """The depth map VolOpt.render_step hands to the MVS stage (vsdf.py:259-263): depth_values as an (H, W) image times
    the dataset's scale factor; pixels whose accumulated weight is below 0.2 get the maximum depth."""
    H, W = img_res
    depth = outputs["depth_values"].reshape(H, W) * scale_factor
    acc = outputs["weights"].sum(1).reshape(H, W)
    return torch.where(acc < min_acc, depth.max(), depth)
