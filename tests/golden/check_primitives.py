"""What torch-CPU really evaluates for the primitives the sampler's bit-exactness hangs on -- the measurements behind
the numeric contract of DESIGN.md section 2 / oracle/svs_oracle.py.  Build container only (needs gcc and this torch build).

    python tests/golden/check_primitives.py

Prints, for float32 inputs:
  * torch.exp  vs  MKL VML's three dispatch kernels (AVX-512 / AVX2 / SSE2, called directly through their exported
    symbols), Sleef_expf8_u10 and the correctly rounded result: which one torch.exp IS on this host, and how often the
    kernels disagree with each other (the reference's exp is host-dependent and closed source);
  * torch.sqrt likewise (MKL's AVX-512 kernel is not correctly rounded);
  * torch.expm1 vs Sleef_expm1f8_u10 / Sleef_expm1f16_u10 (identical: torch.expm1 is Sleef on every host);
  * torch.sum(dim=-1) vs the restated cascade order for every row length 1..700 and a few long rows, under
    ATEN_CPU_CAPABILITY = default and avx2 (identical: the AVX2 kernel serves AVX-512 hosts as well);
  * the numpy restatements of oracle/svs_oracle.py vs the library routines.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_shim          # noqa: E402
import svs_oracle as orc  # noqa: E402
import torch             # noqa: E402

F32 = np.float32
LIB = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))


def mkl_kernel(name, x):
    """mkl_vml_kernel_s<Op>_<arch>HAynn(int n, const float* a, float* r): the per-architecture VML kernels are exported."""
    try:
        f = getattr(LIB, name)
    except AttributeError:
        return None
    f.restype = None
    out = np.zeros_like(x)
    f(ctypes.c_int(x.size), x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def ne(a, b):
    return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())


def main():
    rng = np.random.default_rng(1)
    n = 1 << 21
    x = np.concatenate([-rng.random(n) * 20, rng.random(n) * 14, -np.exp(rng.random(n) * 20 - 15),
                        rng.standard_normal(n) * 1e-3, -rng.random(n) * 103]).astype(F32)
    x = x[: x.size // 16 * 16].copy()
    print(f"torch {torch.__version__}, capability {torch.backends.cpu.get_cpu_capability()}, {x.size} inputs")
    t_exp = torch.exp(torch.from_numpy(x)).numpy()
    cr = np.exp(x.astype(np.float64)).astype(F32)
    sleef = ref_shim.torch_vec8("Sleef_expf8_u10")(x)
    print("exp:   torch.exp != correctly rounded:", ne(t_exp, cr), "  != Sleef_expf8_u10:", ne(t_exp, sleef))
    for k in ("Z0", "L9", "E2"):
        o = mkl_kernel(f"mkl_vml_kernel_sExp_{k}HAynn", x)
        if o is not None:
            print(f"       MKL sExp {k}HA: != torch.exp {ne(o, t_exp)},  != correctly rounded {ne(o, cr)}")
    print("       oracle sleef_expf != Sleef_expf8_u10:", ne(orc.sleef_expf(x), sleef))

    xs = np.abs(np.exp(rng.random(n * 4) * 40 - 30)).astype(F32)
    t_sqrt = torch.sqrt(torch.from_numpy(xs)).numpy()
    print("sqrt:  torch.sqrt != IEEE:", ne(t_sqrt, np.sqrt(xs)))
    for k in ("Z0", "L9", "E2"):
        o = mkl_kernel(f"mkl_vml_kernel_sSqrt_{k}HAynn", xs)
        if o is not None:
            print(f"       MKL sSqrt {k}HA: != torch.sqrt {ne(o, t_sqrt)},  != IEEE {ne(o, np.sqrt(xs))}")

    t_m1 = torch.expm1(torch.from_numpy(x)).numpy()
    s_m1 = ref_shim.torch_vec8("Sleef_expm1f8_u10")(x)
    print("expm1: torch.expm1 != Sleef_expm1f8_u10:", ne(t_m1, s_m1), "  oracle sleef_expm1f != library:",
          ne(orc.sleef_expm1f(x), s_m1))

    code = ("import numpy as np, torch, sys; sys.path.insert(0, %r); import svs_oracle as orc\n"
            "rng = np.random.default_rng(0); bad = 0\n"
            "for m in list(range(1, 701)) + [1024, 2047, 4096, 5000, 70000]:\n"
            "    x = (rng.random((16, m)) * np.exp(rng.random((16, 1)) * 12 - 8)).astype(np.float32)\n"
            "    bad += int((orc.aten_sum(x)[:, 0] != torch.sum(torch.from_numpy(x), -1).numpy()).sum())\n"
            "print('sum:   capability', torch.backends.cpu.get_cpu_capability(), ': restated cascade order != torch.sum on', bad, 'rows')\n"
            % os.path.join(HERE, "..", "..", "oracle"))
    for cap in (None, "avx2"):
        env = dict(os.environ)
        if cap:
            env["ATEN_CPU_CAPABILITY"] = cap
        print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())


if __name__ == "__main__":
    main()
