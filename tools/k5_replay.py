#!/usr/bin/env python
"""CPU replay of K5's texel-row reuse on the headline frame: how many DISTINCT corner texel rows (1 KiB each) a batch of B
consecutive entries of the valid-sample list touches per reference view, for the ray-major list (round 3) and for the
depth-major-within-64-ray-groups list, B = 16 / 32 / 64.  Distinct / (4 B) is the share of today's texture-path bytes a
kernel that loads every distinct row once (into LDS) would still move.
    python tools/k5_replay.py [--res 512]            (CPU only; ~2 min, scipy cKDTree for the hull test)"""
import argparse
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transhuman_amd import synth  # noqa: E402
from transhuman_amd.dist import shard_ray_indices  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--samples", type=int, default=64)
ap.add_argument("--world", type=int, default=1)
args = ap.parse_args()
H = W = args.res
S = args.samples
b = synth.make_batch(H, W, 3, seed=0, all_rays=True)
idx = shard_ray_indices(H, W, args.world, 0, tile=8, tile_major=True).numpy()      # bench.py's ray order (8x8 tiles)
ro, rd = b["ray_o"][0].numpy()[idx], b["ray_d"][0].numpy()[idx]
near, far = b["near"][0].numpy()[idx], b["far"][0].numpy()[idx]
t = np.linspace(0.0, 1.0, S, dtype=np.float32)
z = near[:, None] * (1 - t) + far[:, None] * t
R = ro.shape[0]
verts = b["tar_smpl_vertice"][0].numpy()
tree = cKDTree(verts)
mask = np.zeros((R, S), bool)
for s0 in range(0, R, 8192):
    p = ro[s0:s0 + 8192, None] + rd[s0:s0 + 8192, None] * z[s0:s0 + 8192, :, None]
    d, _ = tree.query(p.reshape(-1, 3), k=1, distance_upper_bound=0.1)
    mask[s0:s0 + 8192] = np.isfinite(d).reshape(-1, S)
print("valid samples", int(mask.sum()), "hit rays", int(mask.any(1).sum()))
rr, ss = np.nonzero(mask)                       # ray-major ascending
pts = ro[rr] + rd[rr] * z[rr, ss][:, None]
Rm, Tm, Km = (b[k][0][0].numpy().astype(np.float64) for k in ("input_R", "input_T", "input_K"))
sc = 2.0 / (W - 1) * (W - 1) / 2.0        # (pixel coords -> texel coords is the identity at full resolution, see th_bilinear_setup)
corner = []
for v in range(3):
    cam = pts.astype(np.float64) @ Rm[v].T + Tm[v].reshape(1, 3)
    uvw = cam @ Km[v].T
    ix = np.clip(uvw[:, 0] / uvw[:, 2], 0, W - 1)
    iy = np.clip(uvw[:, 1] / uvw[:, 2], 0, H - 1)
    x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    corner.append(np.stack([y0 * W + x0, y0 * W + x1, y1 * W + x0, y1 * W + x1], 1))
orders = {"ray-major": np.arange(len(rr)),
          "depth-major in 64-ray groups": np.lexsort((rr % 64, ss, rr // 64)),
          "depth-major in 32-ray groups": np.lexsort((rr % 32, ss, rr // 32)),
          "depth-major in 16-ray groups": np.lexsort((rr % 16, ss, rr // 16))}
for name, o in orders.items():
    for B in (16, 32, 64):
        n = len(o) // B * B
        tot, big = 0.0, {48: 0, 64: 0}
        per_view = []
        for v in range(3):
            c = corner[v][o[:n]].reshape(-1, B * 4)
            cs = np.sort(c, axis=1)
            u = 1 + (np.diff(cs, axis=1) != 0).sum(1)
            per_view.append(u.mean() / (4 * B))
            tot += u.mean()
            for k in big:
                big[k] += int((u > k).sum())
        print(f"{name:30s} B={B:3d}: distinct rows / corner reads = {tot / (12 * B):.3f}  per view " +
              " ".join(f"{x:.3f}" for x in per_view) + f"  batches with U > 48: {big[48] / (3 * n / B):.4f}, > 64: {big[64] / (3 * n / B):.4f}")

# ---- pair sharing inside K5's waves (depth-major in 16-ray groups): rows 2k / 2k+1 of the list are the two half-waves of a
# wave pass; how often can the second row take corner texels from the first row's registers?
o = orders["depth-major in 16-ray groups"]
n2 = len(o) // 2 * 2
for v in range(3):
    c = corner[v][o[:n2]].reshape(-1, 2, 4)            # [pair][A|B][i00 i01 i10 i11]
    A, B = c[:, 0], c[:, 1]
    same = (A == B).all(1)
    plus = (B[:, 0] == A[:, 1]) & (B[:, 2] == A[:, 3]) & ~same
    minus = (B[:, 1] == A[:, 0]) & (B[:, 3] == A[:, 2]) & ~same
    down = (B[:, 0] == A[:, 2]) & (B[:, 1] == A[:, 3]) & ~same & ~plus & ~minus
    up = (B[:, 2] == A[:, 0]) & (B[:, 3] == A[:, 1]) & ~same & ~plus & ~minus
    saved = same.mean() * 0.5 + (plus.mean() + minus.mean() + down.mean() + up.mean()) * 0.25
    print(f"view {v}: pairs with the same 4 texels {same.mean():.3f}, shifted by +1 / -1 in x {plus.mean():.3f} / {minus.mean():.3f}, "
          f"by +1 / -1 in y {down.mean():.3f} / {up.mean():.3f} -> corner-load bytes saved {saved:.3f}")
