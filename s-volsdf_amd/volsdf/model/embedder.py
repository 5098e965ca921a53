"""NeRF positional encoding, same call surface as the reference's volsdf/model/embedder.py:38-50.

The fused HIP MLP kernels compute the encoding in registers; this host version exists for callers that
want the encoding as a tensor (meshing / analysis code) and for shape bookkeeping (`out_dim`).
"""
import torch


class Embedder:
    def __init__(self, input_dims, num_freqs, include_input=True):
        self.input_dims, self.num_freqs, self.include_input = input_dims, num_freqs, include_input
        self.out_dim = input_dims * ((1 if include_input else 0) + 2 * num_freqs)

    def embed(self, x):
        parts = [x] if self.include_input else []
        for k in range(self.num_freqs):
            f = float(2 ** k)
            parts += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(parts, -1)


def get_embedder(multires, input_dims=3):
    e = Embedder(input_dims, multires)
    return e.embed, e.out_dim
