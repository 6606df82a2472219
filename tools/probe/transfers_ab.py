#!/usr/bin/env python3
"""Where does the serving step lose time when every batch comes from pinned host memory?  Same pipelined loop as
bench.py's `incl_transfers` leg, one ingredient at a time:
  a  resident input (the headline loop)
  b  two alternating resident buffers, no copy
  c  b + H2D on a copy stream each step
  d  c + holding every step's token tensor and the D2H at the end  (= bench.py's leg)
  e  c with the H2D issued one step EARLIER (copy for step i+1 enqueued before forward i)
  p  a with the engine's per-launch HIP-event profiling on (what bench.py's headline leg runs with)
  c1 the copy alone beside the loop (nothing waits for it)      c2  c1 + the forward waits for its copy
usage: python tools/probe/transfers_ab.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doc2tex_amd import Model, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = synth.make_config("C2", device="cuda")
m = Model(cfg)
m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False)
m.eval().to("cuda")
m.pipelined = True      # bench.py's serving configuration
m.decode_chains = 3
m.decode_group = 6
dev = torch.device("cuda:0")
host = synth.synth_images(64, 128, 512, seed=1).pin_memory()
img = host.to(dev)
text = torch.full((64, 1), 1, dtype=torch.long, device=dev)


def fwd(x):
    with torch.no_grad():
        return m(x, text, is_train=False, is_test=False)


def prime():
    for n in range(6, 0, -1):
        for _ in range(6):
            for _ in range(n):
                fwd(img)
            m.synchronize()


def run(mode):
    for _ in range(5):
        fwd(img)
    m.synchronize()
    torch.cuda.synchronize()
    bufs = [img.clone(), img.clone()]
    cs = torch.cuda.Stream(device=dev)
    free = [torch.cuda.Event(), torch.cuda.Event()]
    for e in free:
        e.record()
    toks = []
    main = torch.cuda.current_stream(dev)

    def h2d(k):
        with torch.cuda.stream(cs):
            cs.wait_event(free[k])
            bufs[k].copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return ev
    torch.cuda.synchronize()
    m.engine().profile(mode == "p")
    t0 = time.perf_counter()
    nxt = h2d(0) if mode == "e" else None
    for i in range(steps):
        k = i & 1
        if mode in ("a", "p"):
            out = fwd(img)
        elif mode == "b":
            out = fwd(bufs[k])
        elif mode == "c1":  # the copy alone beside the loop: nothing waits for it, the forward reads the resident batch
            with torch.cuda.stream(cs):
                bufs[k].copy_(host, non_blocking=True)
            out = fwd(img)
        elif mode == "c2":  # + the forward waits for its copy (no buffer hand-back)
            with torch.cuda.stream(cs):
                bufs[k].copy_(host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            main.wait_event(ev)
            out = fwd(bufs[k])
        elif mode in ("c", "d"):
            ev = h2d(k)
            main.wait_event(ev)
            out = fwd(bufs[k])
            free[k].record()
            if mode == "d":
                toks.append(out[0])
        else:
            ev = nxt
            main.wait_event(ev)
            out = fwd(bufs[k])
            free[k].record()
            if i + 1 < steps:
                nxt = h2d(k ^ 1)
    m.synchronize()
    if toks:
        torch.stack(toks).to("cpu", non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m.engine().profile(False)
    m.engine().profile_read()
    print(f"{mode}: {dt * 1e3:7.2f} ms/step  {64 / dt:7.1f} formulas/s", flush=True)


# the copy alone
torch.cuda.synchronize()
b = torch.empty_like(img)
for _ in range(3):
    b.copy_(host, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    b.copy_(host, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"H2D alone: {dt * 1e3:.3f} ms per {host.numel() * 4 / 1e6:.1f} MB = {host.numel() * 4 / dt / 1e9:.1f} GB/s", flush=True)
prime()
for mode in ("a", "c1", "c2", "c", "a", "c1", "c2", "c"):
    run(mode)
