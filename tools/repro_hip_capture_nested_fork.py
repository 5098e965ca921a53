"""Reproducer of a HIP runtime defect met while capturing the train step ("hipGraph capture"): during
stream capture, a stream forked from an already forked stream that joins back into its PARENT FORK (not into the origin
stream) makes hipStreamEndCapture crash.  Each variant runs in a child process (the crash is a segfault) on one GPU:
    python tools/repro_hip_capture_nested_fork.py
svs_hip/trainer.py avoids the pattern: join events of nested forks are waited on by the origin stream."""
import os, sys, subprocess
CHILD = r'''
import os, torch
v = os.environ["VARIANT"]
dev = torch.device("cuda:0")
a = torch.randn(1 << 20, device=dev); b = torch.randn(1 << 20, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def E(stream):
    e = torch.cuda.Event(); e.record(stream); return e
def run():
    main = torch.cuda.current_stream()
    keep = [a * 2]
    if v == "siblings_cross":
        f = E(main)
        with torch.cuda.stream(s1):
            s1.wait_event(f); keep.append(a + 1); e1 = E(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(f); keep.append(b + 1); s2.wait_event(e1); keep.append(b + 2); e2 = E(s2)
        main.wait_event(e1); main.wait_event(e2)
    elif v == "refork":
        for _ in range(2):
            f = E(main)
            with torch.cuda.stream(s1):
                s1.wait_event(f); keep.append(a + 1); e1 = E(s1)
            main.wait_event(e1); keep.append(a * 3)
    elif v == "second_wait":
        f = E(main)
        with torch.cuda.stream(s1):
            s1.wait_event(f); keep.append(a + 1)
        keep.append(a * 3)
        p = E(main)
        with torch.cuda.stream(s1):
            s1.wait_event(p); keep.append(a + 2); e1 = E(s1)
        keep.append(a * 4)
        main.wait_event(e1)
    elif v == "selfwait":
        p = E(main); main.wait_event(p); keep.append(a + 1)
    elif v == "wait_stream":
        s1.wait_stream(main)
        with torch.cuda.stream(s1):
            keep.append(a + 1); e1 = E(s1)
        main.wait_event(e1)
    elif v == "nested":
        f = E(main)
        with torch.cuda.stream(s1):
            s1.wait_event(f); keep.append(a + 1)
            f2 = E(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(f2); keep.append(b + 1); e2 = E(s2)
            s1.wait_event(e2); e1 = E(s1)
        main.wait_event(e1)
    elif v == "nested_join_main":
        f = E(main)
        with torch.cuda.stream(s1):
            s1.wait_event(f); keep.append(a + 1)
            f2 = E(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(f2); keep.append(b + 1); e2 = E(s2)
            keep.append(a + 5); e1 = E(s1)
        main.wait_event(e1); main.wait_event(e2)
    return keep
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    keep = run()
g.replay(); g.replay()
torch.cuda.synchronize()
print("OK", v)
'''
for v in ("siblings_cross", "refork", "second_wait", "selfwait", "wait_stream", "nested", "nested_join_main"):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, VARIANT=v), capture_output=True, text=True)
    print(v, "rc", r.returncode, r.stdout.strip()[-40:], (r.stderr.strip().splitlines() or [""])[-1][:200] if r.returncode else "", flush=True)
