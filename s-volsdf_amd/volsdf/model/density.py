"""Density classes with the reference's names and parameters (volsdf/model/density.py:5-47).

`LaplaceDensity.beta` is the learnable parameter `density.beta` of the checkpoint; the density itself is
evaluated inside the HIP sampler / compositing kernels (svs::laplace_density).  The tensor-level
`density_func` below is host glue for callers outside the hot path (it is what the kernels compute).
"""
import torch
from torch import nn


class Density(nn.Module):
    """Base: one learnable scalar per entry of `params_init` (the checkpoint key is the entry's name)."""

    def __init__(self, params_init=None):
        super().__init__()
        for name, value in (params_init or {}).items():
            self.register_parameter(name, nn.Parameter(torch.as_tensor(value, dtype=torch.get_default_dtype())))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)

    def density_func(self, sdf, beta=None):
        raise NotImplementedError


class LaplaceDensity(Density):
    """sigma(s) = (1/beta) * Psi_beta(-s), Psi the CDF of a zero-mean Laplace distribution of scale beta (:21-30)."""

    def __init__(self, params_init=None, beta_min=0.0001):
        super().__init__(params_init)
        # host copy: the kernels take beta_min by value, no device read-back per step
        self.beta_min_value = float(beta_min)
        self.register_buffer("beta_min", torch.tensor(self.beta_min_value), persistent=False)

    def get_beta(self):
        return self.beta_min + torch.abs(self.beta)

    def density_func(self, sdf, beta=None):
        b = self.get_beta() if beta is None else beta
        half = 0.5 * torch.sign(sdf) * torch.expm1(-torch.abs(sdf) / b)
        return (1.0 / b) * (0.5 + half)


class AbsDensity(Density):
    def density_func(self, sdf, beta=None):
        return sdf.abs()
