"""dev aid: the three reference views of a scan through one cost-volume stage -- one after the other vs on three streams"""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden")]
import synth
from models.CasMVSNet import CascadeMVSNet
dev = torch.device("cuda:0")
H, W = 512, 640
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
m = CascadeMVSNet(refine=False, ndepths=[192, 32, 8], depth_interals_ratio=[1.0, 0.5, 0.5], share_cr=False, cr_base_chs=[8, 8, 8],
                  grad_method="detach")
for st, cin in enumerate((32, 16, 8)):
    m.cost_regularization[st].load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_costreg_params(100 + st, cin).items()})
m.to(dev).eval()
views = []
for v in range(3):
    feats, proj, depth_values = synth.make_mvs_sample(3 + v, img_hw=(H, W))
    sample = dict(imgs=torch.zeros(1, 3, 3, H, W, device=dev), depth_values=G(depth_values)[None],
                  proj_matrices={k: G(x)[None] for k, x in proj.items()})
    views.append((sample, [{k: G(x)[None] for k, x in f.items()} for f in feats]))
streams = [torch.cuda.Stream() for _ in range(3)]

def run(concurrent):
    outs = [None] * 3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    main = torch.cuda.current_stream()
    ev[0].record()
    for st in range(3):
        if concurrent:
            fork = torch.cuda.Event(); fork.record(main)
            for i, (sample, feats) in enumerate(views):
                streams[i].wait_event(fork)
                with torch.cuda.stream(streams[i]):
                    outs[i], _ = m(st, sample, features=feats, extra=None, outputs=outs[i], int_r=m.depth_interals_ratio[st])
            for s in streams:
                main.wait_stream(s)
        else:
            for i, (sample, feats) in enumerate(views):
                outs[i], _ = m(st, sample, features=feats, extra=None, outputs=outs[i], int_r=m.depth_interals_ratio[st])
        ev[st + 1].record()
    torch.cuda.synchronize()
    return [ev[k].elapsed_time(ev[k + 1]) for k in range(3)], outs
for mode in (False, True, False, True):
    for _ in range(2): run(mode)
    ts = [run(mode)[0] for _ in range(4)]
    best = [min(t[k] for t in ts) for k in range(3)]
    print("concurrent" if mode else "sequential", ["%.3f" % b for b in best], "sum %.3f ms for 3 views" % sum(best))
a = run(False)[1]; b = run(True)[1]
print("max |depth difference|", max(float((a[i]["depth"] - b[i]["depth"]).abs().max()) for i in range(3)))
