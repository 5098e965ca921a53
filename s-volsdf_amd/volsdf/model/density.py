"""Density classes with the reference's names and parameters (volsdf/model/density.py:5-47).

`LaplaceDensity.beta` is the learnable parameter `density.beta` of the checkpoint; the density itself is
evaluated inside the HIP sampler / compositing kernels (svs::laplace_density).  The tensor-level
`density_func` below is host glue for callers outside the hot path (it is what the kernels compute).
"""
import torch
import torch.nn as nn


class Density(nn.Module):
    def __init__(self, params_init={}):
        super().__init__()
        for p in params_init:
            setattr(self, p, nn.Parameter(torch.tensor(params_init[p])))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.register_buffer("beta_min", torch.tensor(beta_min), persistent=False)
        self.beta_min_value = float(beta_min)     # host copy: kernels take it by value, no device read-back per step

    def density_func(self, sdf, beta=None):
        if beta is None:
            beta = self.get_beta()
        alpha = 1 / beta
        return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))

    def get_beta(self):
        return self.beta.abs() + self.beta_min


class AbsDensity(Density):
    def density_func(self, sdf, beta=None):
        return torch.abs(sdf)
