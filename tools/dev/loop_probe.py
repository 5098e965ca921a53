"""Where the time of `VolOpt.run`'s loop goes on the host (dataset vs enqueue, alone and overlapped)."""
import json, os, sys, tempfile, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import torch
import test_gpu_volopt as tv


def build(**kw):
    args = tv.make_args()
    args["vol"]["dataset"]["img_res"] = [576, 768]
    args["vol"]["train"]["num_pixels"] = 1024
    args["vol"]["train"]["render_freq"] = 10 ** 9
    args["vol"]["train"]["checkpoint_freq"] = 10 ** 9
    args["max_h"], args["max_w"] = 576, 768
    v = tv.build(args, **kw)
    v._preview = lambda *a, **k: None
    v.save_checkpoints = lambda *a, **k: None
    return v


def timed_run(v, n):
    torch.cuda.synchronize()
    n0, t0 = v.total_step, time.perf_counter()
    v.run(opt_stepN=n)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / (v.total_step - n0)


def main():
    os.chdir(tempfile.mkdtemp())
    res = {}
    v = build()
    v.run(opt_stepN=60)
    res["default_ms"] = timed_run(v, 150)
    # dataset alone
    ds, dl = v.train_dataset, v.train_dataloader
    t = []
    for _ in range(10):
        t0 = time.perf_counter(); ds.change_sampling_idx(v.num_pixels); t1 = time.perf_counter()
        b = next(iter(dl)); t2 = time.perf_counter()
        t.append((t1 - t0, t2 - t1))
    res["dataset_randperm_ms"] = 1e3 * sorted(x[0] for x in t)[5]
    res["dataset_getitem_ms"] = 1e3 * sorted(x[1] for x in t)[5]
    # enqueue alone: the same batch over and over, host time per train_step while the GPU queue is never empty
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        v.train_step(b, False, _resample=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res["enqueue_only_host_ms"] = 1e3 * (t1 - t0) / 100
    res["enqueue_only_wall_ms"] = 1e3 * (t2 - t0) / 100
    v2 = build(overlap_loader=True)
    v2.run(opt_stepN=60)
    res["overlap_ms"] = timed_run(v2, 150)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
