"""Builds libsvolsdf_hip.so (hand-written HIP kernels + the C-ABI of include/svolsdf_hip.h) for gfx950.

    python s-volsdf_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The library is written in-tree (s-volsdf_amd/lib/), which is
git-ignored but travels with the repo snapshot to the GPU box.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsvolsdf_hip.so")
ARCH = "gfx950"

# translation unit -> extra flags.  The sampler/compositing units implement the bit-exact numeric contract
# (IEEE float32 ops in the reference's order): no fma contraction there.
UNITS = {
    "svs_common.cpp": [],
    "svs_pack.hip": [],
    "svs_mlp.hip": [],
    "svs_mlp_h2.hip": [],
    "svs_bg_h2.hip": [],
    "svs_bg_f32.hip": [],
    "svs_sampler.hip": ["-ffp-contract=off"],
    "svs_render.hip": ["-ffp-contract=off"],
    "svs_costvol.hip": ["-ffp-contract=off"],
    "svs_conv_mfma.hip": [],
    "svs_conv_pair.hip": [],
    "svs_conv_gemm.hip": [],
    "svs_conv2d.hip": [],
    "svs_conv2d_mfma.hip": [],
    "svs_wgrad.hip": [],
    "svs_mlp_bwd.hip": [],
    "svs_mlp_bwd_h2.hip": [],
    "svs_optim.hip": ["-ffp-contract=off"],
    "svs_fusion.hip": ["-ffp-contract=off"],
    "svs_cloud.hip": ["-ffp-contract=off"],
    "svs_plan.hip": [],
}
# The two-waves-per-SIMD experiments of round 3 (svs_sdf_vals16 / svs_sdf_vals_pair, measured no faster
# than the default kernel) are built only on request: SVS_BUILD_EXPERIMENTS=1.
EXPERIMENTS = os.environ.get("SVS_BUILD_EXPERIMENTS", "0") == "1"
if EXPERIMENTS:
    UNITS.update({"svs_mlp_w16.hip": [], "svs_mlp_h2p.hip": []})
BASE_FLAGS = (["-DSVS_EXPERIMENTAL_KERNELS"] if EXPERIMENTS else []) + ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
              "-x", "hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode()); h.update(f.read())
    h.update(repr(sorted(UNITS.items())).encode())
    h.update(repr(BASE_FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = os.path.join(LIBDIR, "build.stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for unit, extra in UNITS.items():
        obj = os.path.join(LIBDIR, unit.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + BASE_FLAGS + extra + ["-I", CSRC, "-c", os.path.join(CSRC, unit), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((unit, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for unit, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {unit}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
