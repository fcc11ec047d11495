#!/usr/bin/env python
"""K12 vs torch/MIOpen on the ResNet-stem convolution shapes (N = 3 views): python tools/conv_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transhuman_amd import hip
dev = torch.device("cuda:0")
hip.load_library()
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (ci, co, ks, st, H) in ((3, 64, 7, 2, 512), (64, 64, 3, 1, 128), (64, 128, 3, 2, 128), (128, 128, 3, 1, 64), (64, 128, 1, 2, 128)):
    conv = torch.nn.Conv2d(ci, co, ks, st, ks // 2, bias=False).to(dev)
    x = torch.randn(3, ci, H, H, device=dev)
    a = tm(lambda: hip.conv2d(x, conv)); b = tm(lambda: conv(x))
    err = float((hip.conv2d(x, conv) - conv(x)).abs().max())
    print(f"conv {ci}->{co} k{ks} s{st} @{H}: K12 {a:.1f} us   torch {b:.1f} us   max|diff| {err:.2e}")
x = torch.randn(3, 64, 256, 256, device=dev)
print(f"maxpool: K12 {tm(lambda: hip.maxpool3x3s2(x)):.1f} us   torch {tm(lambda: torch.nn.functional.max_pool2d(x, 3, 2, 1)):.1f} us")
