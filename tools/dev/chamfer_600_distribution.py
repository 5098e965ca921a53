import sys, os, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import chamfer_parity
# dev aid: the distribution of the 600-step Chamfer distance over six seeds per path (what tests/test_gpu_chamfer_parity.py is built on)
res = chamfer_parity.measure(steps=600, seeds=(0, 1, 2, 3, 4, 5), paths=("hip", "torch_f32"), rays=512, timeout=1500, prior=True)
for p in ("hip", "torch_f32"):
    print(p, [round(r.get("overall_mm", -1), 3) for r in res[p]["runs"]], [ (round(r.get("accuracy_mm",-1),2), round(r.get("completeness_mm",-1),2)) for r in res[p]["runs"]])
