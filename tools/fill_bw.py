#!/usr/bin/env python
"""HBM write / copy stream rates on the GPU box (reference points for the write-bound kernels K8 / K5 / K4)."""
import time, torch
dev = torch.device("cuda:0")
n = 3 * 512 * 512 * 260
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def tm(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
t = tm(lambda: a.fill_(1.0)); print(f"fill {n*4/1e9:.2f} GB: {t*1e6:.0f} us = {n*4/t/1e12:.2f} TB/s")
t = tm(lambda: a.zero_()); print(f"zero: {t*1e6:.0f} us = {n*4/t/1e12:.2f} TB/s")
t = tm(lambda: b.copy_(a)); print(f"copy: {t*1e6:.0f} us = {2*n*4/t/1e12:.2f} TB/s (read+write)")
