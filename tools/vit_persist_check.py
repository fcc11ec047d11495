"""TransHE as one persistent launch (th_set_vit_mode 2) against one launch per layer (mode 1): values + time.
    python tools/vit_persist_check.py [N_c ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from transhuman_amd import hip
dev = torch.device("cuda:0")
net = bench.build_net(dev)


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for nc in [int(a) for a in sys.argv[1:]] or [500, 300, 37, 1, 800, 1100]:
    for V in (3, 1):
        g = torch.randn(V, nc, 192, device=dev)
        pe = torch.rand(V, nc, 3, device=dev) * 2 - 1
        hip.set_vit_mode(1, dev)
        ref = net.ViT(g, pe, mask=None).clone()
        t2 = timed(lambda: net.ViT(g, pe, mask=None))
        hip.set_vit_mode(2, dev)
        outs = [net.ViT(g, pe, mask=None).clone() for _ in range(5)]
        t1 = timed(lambda: net.ViT(g, pe, mask=None))
        same = all(torch.equal(o, outs[0]) for o in outs)
        hip.set_vit_mode(1, dev)
        print(f"N_c {nc:5d} V {V}: per-layer {t2:.3f} ms  persistent {t1:.3f} ms  repeatable {same}  max diff {max(float((o - ref).abs().max()) for o in outs):.3e}  finite {bool(torch.isfinite(outs[0]).all())}", flush=True)
