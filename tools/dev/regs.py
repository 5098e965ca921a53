"""Register / scratch usage of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/dev/regs.py svs_mlp_h2 [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "s-volsdf_amd", "csrc")


def main():
    unit, extra = sys.argv[1], sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-I", CSRC] + extra + [
        "-c", os.path.join(CSRC, unit + ".hip"), "-o", "/tmp/regs_" + unit + ".o", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    cur, d = None, {}
    keys = {"VGPRs": "V", "AGPRs": "A", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ"}
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur, d = m.group(1), {}
        for k, short in keys.items():
            m = re.search(" " + re.escape(k) + r": (\d+)", line)
            if m:
                d[short] = m.group(1)
        if "LDS Size" in line and cur:
            name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()[:90]
            print(f"{name:90s} " + " ".join(f"{s}={d.get(s)}" for s in keys.values()))
        if "error" in line:
            print(line)
    return out.returncode


if __name__ == "__main__":
    sys.exit(main())
