"""What "parameter gradients in the float32 accuracy class" means in the GPU parity tests (test helper)."""


class Err(tuple):
    """(error of the kernels, largest error of all yardstick evaluations) as before, plus .f32 = the error of PLAIN float32
    autograd at the same parameters (the reference's own arithmetic) and .noise64 = the largest deviation of the float64
    evaluations at 1e-6 parameter noise (how much the gradient itself moves inside the accuracy class of the forward)."""
    def __new__(cls, e, yard, f32=0.0, noise64=0.0):
        o = super().__new__(cls, (e, yard))
        o.f32, o.noise64 = f32, noise64
        return o


def per_tensor_errors(got, ref64, ref32=None, n_f32=3):
    """{name: Err}: max |x - ref64| over the tensor's largest |ref64|.  ref32 = f32_yardstick(...): its first `n_f32` entries
    are float32-autograd evaluations (the first one at the unperturbed parameters), the rest float64 evaluations under noise."""
    out = {}
    for n in got:
        r = ref64[n]
        den = float(r.abs().max()) + 1e-30
        es = [float((g[n].double() - r).abs().max()) / den for g in (ref32 or [])]
        out[n] = Err(float((got[n].double() - r).abs().max()) / den, max(es, default=0.0), es[0] if es else 0.0,
                     max(es[n_f32:], default=0.0))
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


# Tensors whose gradient sits behind hard kinks -- the ReLU masks of the radiance networks (network.py:170-190,
# network_bg.py:25-35): one unit of one well-weighted point on the other side of its kink moves such a tensor's gradient by
# ~1e-4 of its largest entry, and which units sit within rounding distance differs between any two evaluations whose hidden
# activations differ in their last bits (tools/dev/grad_sensitivity.py).  For these the ratio to plain
# float32 autograd is REPORTED, and the bound is the measured movement of the float64 gradient under 1e-6 parameter noise.
KINK_TENSORS = ("rendering_network.", "bg_rendering_network.")
RATIO_MAX = 8.0          # kernels vs plain float32 autograd, per tensor, wherever the kernels are above the absolute floor
F32_FLOOR = 3e-6         # a float32-autograd error below this is noise of its own (observed 2e-6 ... 2.6e-5)


def assert_f32_class(errs, what, floor=3e-5, factor=4.0, floors=None, ratio_max=RATIO_MAX):
    """The float32 accuracy class, per tensor: within `floor` of float64 autograd (relative to the tensor's largest entry),
    or -- where arithmetic of that class itself does not hold that (f32_yardstick) -- within `factor` times the largest deviation
    of the yardstick evaluations.  floors: per-tensor overrides of `floor` (density.beta: its gradient amplifies the error of
    the forward's sdf values by 1 / beta, and the fp16x2 forward holds sdf to 1.5e-6, north_star's bound being 1e-4).

    Since round 4 the RATIO to the reference's own arithmetic is printed and bounded as well, so that "float32 class" cannot
    drift: a tensor above `floor` must be within `ratio_max` x the error of plain float32 autograd at the same parameters --
    except the KINK_TENSORS, whose bound is the float64 gradient's own movement under 1e-6 noise (`factor` x)."""
    floors = floors or {}
    worst = max(errs, key=lambda n: errs[n][0])
    ratio = {n: v[0] / max(getattr(v, "f32", 0.0), F32_FLOOR) for n, v in errs.items()}
    above = {n: v for n, v in errs.items() if v[0] >= floor}
    print(f"{what}: worst per-tensor gradient error vs float64 autograd {errs[worst][0]:.2e} ({worst}; float32 autograd "
          f"there: {errs[worst][1]:.2e}); tensors above {floor:g}: "
          + (", ".join(f"{n} {v[0]:.1e} (f32 autograd {getattr(v, 'f32', 0.0):.1e}: ratio {ratio[n]:.1f}; f64 under 1e-6 noise "
                       f"{getattr(v, 'noise64', 0.0):.1e})" for n, v in above.items()) or "none")
          + f"; largest ratio to float32 autograd over all tensors: {max(ratio.values()):.1f}")
    bad = {n: v for n, v in errs.items() if v[0] >= max(floors.get(n, floor), factor * v[1])}
    assert not bad, bad
    if any(hasattr(v, "f32") and v.f32 > 0 for v in errs.values()):
        drift = {}
        for n, v in above.items():
            if v[0] < floors.get(n, floor):
                continue
            if n.startswith(KINK_TENSORS):
                if v[0] > factor * v.noise64:
                    drift[n] = (v[0], "kink tensor: float64 gradient under 1e-6 noise moves by", v.noise64)
            elif ratio[n] > ratio_max:
                drift[n] = (v[0], "ratio to float32 autograd", ratio[n])
        assert not drift, drift
