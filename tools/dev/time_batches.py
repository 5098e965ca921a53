"""dev aid: host and GPU time of one device-side train batch (svs_hip/batches.py)"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/s-volsdf_amd", ROOT + "/tests", ROOT + "/tests/golden"]
from synthetic_scene import SyntheticSceneDataset
from svs_hip.batches import DeviceBatches
ds = SyntheticSceneDataset(img_res=(576, 768))
db = DeviceBatches(ds, 1024, "cuda:0")
for _ in range(5): db.batch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(50): db.batch()
e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"batch(): host {1e3 * (t1 - t0) / 50:.3f} ms, GPU {e0.elapsed_time(e1) / 50:.3f} ms")
