"""Host glue with the reference's names (volsdf/utils/general.py:6-58): class lookup and pixel chunking."""
import os

import torch


def mkdir_ifnotexists(directory):
    os.makedirs(directory, exist_ok=True)


def get_class(kls):
    parts = kls.split(".")
    m = __import__(".".join(parts[:-1]))
    for comp in parts[1:]:
        m = getattr(m, comp)
    return m


def split_input(model_input, total_pixels, n_pixels=10000):
    """Chunks of n_pixels rays (the chunk size is part of the result: the sampler's convergence test is
    chunk-global, ray_sampler.py:136)."""
    split = []
    dev = model_input["uv"].device
    for indx in torch.split(torch.arange(total_pixels, device=dev), n_pixels, dim=0):
        data = dict(model_input)
        data["uv"] = torch.index_select(model_input["uv"], 1, indx)
        for key in ("object_mask", "rgb"):
            if key in data:
                data[key] = torch.index_select(model_input[key], 1, indx)
        split.append(data)
    return split


def merge_output(res, total_pixels, batch_size):
    out = {}
    for entry in res[0]:
        if res[0][entry] is None:
            continue
        nd = res[0][entry].dim()
        if nd == 1:
            out[entry] = torch.cat([r[entry].reshape(batch_size, -1, 1) for r in res], 1).reshape(batch_size * total_pixels)
        elif nd == 2:
            out[entry] = torch.cat([r[entry].reshape(batch_size, -1, r[entry].shape[-1]) for r in res], 1).reshape(
                batch_size * total_pixels, -1)
        elif nd == 3:
            out[entry] = torch.cat([r[entry].reshape(batch_size, -1, r[entry].shape[-2], r[entry].shape[-1]) for r in res],
                                   1).reshape(batch_size * total_pixels, -1, res[0][entry].shape[-1])
        else:
            raise NotImplementedError
    return out
