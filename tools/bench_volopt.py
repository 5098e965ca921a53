"""End-to-end steps per second of `VolOpt.run` (the reference's training loop, volsdf/vsdf.py:322-367) on the synthetic
in-memory dataset at 576 x 768, 1024 pixels per step, with (a) the reference's batch source -- DataLoader over a
SceneDataset-style dataset: full pixel grid per item, torch.randperm on the CPU, one torch thread -- and (b) the opt-in
device batch source (svs_hip/batches.py).  The step itself is the same; bench.py times it alone."""
import copy, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import torch
import test_gpu_volopt as tv


def main():
    os.chdir(tempfile.mkdtemp())
    res = {}
    for name, dev_batches in (("dataloader", False), ("device_batches", True)):
        args = tv.make_args()
        args["vol"]["dataset"]["img_res"] = [576, 768]
        args["vol"]["train"]["num_pixels"] = 1024
        args["vol"]["train"]["render_freq"] = 10 ** 9
        args["vol"]["train"]["checkpoint_freq"] = 10 ** 9
        args["max_h"], args["max_w"] = 576, 768
        v = tv.build(args, device_batches=dev_batches)
        v._preview = lambda *a, **k: None
        v.save_checkpoints = lambda *a, **k: None
        v.run(opt_stepN=60)                      # warm-up (kernel attribute set-up, schedule measurement)
        torch.cuda.synchronize()
        n0, t0 = v.total_step, time.perf_counter()
        v.run(opt_stepN=100 if not dev_batches else 400)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = dict(steps=v.total_step - n0, ms_per_step=round(1e3 * dt / (v.total_step - n0), 3),
                         rays_per_s=round(1024 * (v.total_step - n0) / dt))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
