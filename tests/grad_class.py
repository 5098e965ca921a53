"""What "parameter gradients in the float32 accuracy class" means in the GPU parity tests (test helper)."""


def per_tensor_errors(got, ref64, ref32=None):
    """{name: (error of `got`, largest error of the float32-autograd evaluations ref32)}: max |x - ref64| over the tensor's
    largest |ref64|"""
    out = {}
    for n in got:
        r = ref64[n]
        den = float(r.abs().max()) + 1e-30
        e32 = max((float((g[n].double() - r).abs().max()) / den for g in (ref32 or [])), default=0.0)
        out[n] = (float((got[n].double() - r).abs().max()) / den, e32)
    return out


def f32_yardstick(autograd, p0, trials=3, seed=0):
    """What arithmetic of the float32 class does to this gradient -- a list of {name: gradient} whose largest per-tensor
    deviation from the float64 gradient is the yardstick:
      * `trials` float32-autograd evaluations (torch: the reference's own arithmetic, volsdf/vsdf.py:214-219), the first at
        the parameters p0, the others at p0 (1 + 6e-8 N(0,1)) -- half a float32 ulp of noise on every parameter;
      * four FLOAT64 evaluations at p0 (1 + 1e-6 N(0,1)): the accuracy class of the fp16x2 forward (sdf and hidden activations
        to 1.5e-6, north_star's bound being 1e-4).
    A step's gradient is only piecewise smooth (ReLU masks of the radiance network, the sphere clamp, L1 signs): one unit of
    one well-weighted point switching side moves a tensor's gradient by 1e-4 of its largest entry (tools/dev/
    grad_sensitivity.py: the float64 gradient of the 256-ray bmvs step moves by 2.4e-4 on rendering_network.lin3 under 1e-6
    noise, by 2e-6 on lin4, which sits behind no mask), and which units sit within rounding distance of their kink differs
    from one evaluation to the next.  autograd(dtype, params) -> {name: gradient}."""
    import torch
    g = torch.Generator().manual_seed(seed)
    noisy = lambda eps: {k: v.double() * (1 + eps * torch.randn(v.shape, generator=g, dtype=torch.float64).to(v.device))
                         for k, v in p0.items()}
    out = [autograd(torch.float32, p0 if t == 0 else noisy(6e-8)) for t in range(trials)]
    out += [autograd(torch.float64, noisy(1e-6)) for _ in range(4)]
    return out


def assert_f32_class(errs, what, floor=3e-5, factor=4.0, floors=None):
    """The float32 accuracy class, per tensor: within `floor` of float64 autograd (relative to the tensor's largest entry),
    or -- where arithmetic of that class itself does not hold that (f32_yardstick) -- within `factor` times the largest deviation
    of the yardstick evaluations.  floors: per-tensor overrides of `floor` (density.beta: its gradient amplifies the error of
    the forward's sdf values by 1 / beta, and the fp16x2 forward holds sdf to 1.5e-6, north_star's bound being 1e-4)."""
    floors = floors or {}
    worst = max(errs, key=lambda n: errs[n][0])
    print(f"{what}: worst per-tensor gradient error vs float64 autograd {errs[worst][0]:.2e} ({worst}; float32 autograd "
          f"there: {errs[worst][1]:.2e}); tensors above {floor:g}: "
          + (", ".join(f"{n} {e:.1e} (f32 {f:.1e})" for n, (e, f) in errs.items() if e >= floor) or "none"))
    bad = {n: v for n, v in errs.items() if v[0] >= max(floors.get(n, floor), factor * v[1])}
    assert not bad, bad
