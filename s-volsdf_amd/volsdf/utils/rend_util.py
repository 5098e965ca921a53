"""Camera / ray utilities on the hot path, reference names (volsdf/utils/rend_util.py:14-22,60-95,143-156,200-216).

`get_camera_params` runs the HIP ray kernel for device tensors.  The image-IO helpers of the reference
module (load_rgb, load_K_Rt_from_P, ...) are outside the hot path and are not provided here.
"""
import torch

from svs_hip import ops


def get_psnr(img1, img2, normalize_rgb=False):
    if normalize_rgb:
        img1, img2 = (img1 + 1.) / 2., (img2 + 1.) / 2.
    mse = torch.mean((img1 - img2) ** 2)
    return -10. * torch.log(mse) / torch.log(torch.tensor([10.], device=mse.device))


def get_camera_params(uv, pose, intrinsics):
    """uv (B,N,2), pose (B,4,4), intrinsics (B,4,4) -> ray_dirs (B,N,3), cam_loc (B,3).  B must be 1
    (the trainer's batch size, runner.py:164-169); quaternion poses are not on the hot path."""
    if pose.shape[1] == 7:
        raise NotImplementedError("quaternion poses are not used by the S-VolSDF trainer")
    if uv.shape[0] != 1:
        raise NotImplementedError("batch_size 1 only (runner.py:166)")
    dirs, cam, _ = ops.rays_from_uv(uv[0], pose[0], intrinsics[0])
    return dirs[None], cam[None]


def get_sphere_intersections(cam_loc, ray_directions, r=1.0):
    """(n,3),(n,3) -> (n,2) near/far clamped at 0; raises instead of the reference's exit() on a miss."""
    dot = (ray_directions * cam_loc).sum(-1, keepdim=True)
    under = dot ** 2 - (cam_loc.norm(2, 1, keepdim=True) ** 2 - r ** 2)
    if (under <= 0).any():
        raise RuntimeError("BOUNDING SPHERE PROBLEM!")
    s = torch.sqrt(under)
    return torch.cat([-s - dot, s - dot], -1).clamp_min(0.0)
