"""Register / scratch / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/dev/kres.py svs_mlp_bwd_h2.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "s-volsdf_amd", "csrc")
unit, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-x", "hip",
       "-I", CSRC] + extra + ["-c", os.path.join(CSRC, unit), "-o", "/tmp/kres_%d.o" % os.getpid(),
                              "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}


def flush():
    if not cur:
        return
    name = subprocess.run(["c++filt", cur.get("Name", "?")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("svs::mlp::", "").replace("void ", "")
    print("%-60s vgpr %s agpr %s scratch %s occ %s lds %s" % (name, cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"),
                                                             cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))


for line in out.splitlines():
    m = re.search(r"remark:\s+(?:Function )?([A-Za-z \[\]/]+): (\S+)", line)
    if not m:
        if "error" in line or "warning:" in line:
            print(line)
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Name":
        flush()
        cur = {}
    cur[k] = v
flush()
try:
    os.remove("/tmp/kres_%d.o" % os.getpid())
except OSError:
    pass
