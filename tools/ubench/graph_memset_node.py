#!/usr/bin/env python
"""Stand-alone probe of the ROCm 7.2 problem behind profiles/r05_l_vit_graph_memset_node.txt: a hipGraph captured from
   hipMemsetAsync(B, 0) -> kernel(out = f(B)) [-> more kernels], replayed many times with B poisoned (NaN) by an ordinary kernel in
   front of every replay.  If the memset node of a replay is dropped, reordered behind its consumer or applied to the wrong range,
   `out` holds NaN.  Prints the number of bad replays per variant (memset node / fill kernel in its place; one or several streams).

       python tools/ubench/graph_memset_node.py [replays]
"""
import ctypes
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def raw_stream():
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def build(n, use_memset, chain, lead=0):
    B = torch.empty(n, device="cuda")
    src = src0 = torch.randint(-1000, 1000, (n,), device="cuda").float()   # (integers: every sum below is exact)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(lead):                           # kernel nodes IN FRONT of the memset (TransHE: x + pe, then the clear)
            src = src + 0.0
        if use_memset:
            rc = hip.hipMemsetAsync(ctypes.c_void_p(B.data_ptr()), 0, B.numel() * 4, raw_stream())
            assert rc == 0, rc
        else:
            B.zero_()                                   # (an elementwise fill kernel)
        B[: n // 2].copy_(src[: n // 2])                # a producer writes part of the buffer (the real keys) ...
        out = B * 2.0                                   # ... the consumer reads all of it (the padding must be zero)
        for _ in range(chain):                          # a tail of small dependent kernels, like the rest of a forward
            out = out + 1.0
    return g, B, src0, out


def run(name, use_memset, chain, replays, streams, lead=0, n=1 << 18):
    g, B, src, out = build(n, use_memset, chain, lead)
    ss = [torch.cuda.Stream() for _ in range(streams)]
    want = torch.cat([src[: n // 2] * 2.0, torch.zeros(n - n // 2, device="cuda")]) + float(chain)
    bad = 0
    for i in range(replays):
        s = ss[i % streams]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            B.fill_(float("nan"))
            g.replay()
            ok = torch.equal(out, want)
        torch.cuda.current_stream().wait_stream(s)
        bad += 0 if bool(ok) else 1
    print(f"{name:40s} bad replays: {bad} / {replays}", flush=True)


if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for chain in (0, 60):
        for streams in (1, 3):
            run(f"memset node, chain {chain}, {streams} stream(s)", True, chain, R, streams)
            run(f"fill kernel, chain {chain}, {streams} stream(s)", False, chain, R, streams)
    # the shape of the TransHE forward: one kernel in front, 1 474 560 bytes cleared, 60 kernels behind
    for streams in (1, 3):
        run(f"memset 2nd node, 368640 floats, {streams} stream(s)", True, 60, R, streams, lead=1, n=368640)
        run(f"fill   2nd node, 368640 floats, {streams} stream(s)", False, 60, R, streams, lead=1, n=368640)
