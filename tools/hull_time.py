import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from transhuman_amd import synth, hip
from transhuman_amd.dist import shard_ray_indices
dev = torch.device("cuda:0")
b = synth.batch_to(synth.make_batch(512, 512, 3, seed=0, all_rays=True), dev)
def tm(idx, tag):
    P = hip.Points(b["ray_o"][0][idx].contiguous(), b["ray_d"][0][idx].contiguous(), b["near"][0][idx].contiguous(), b["far"][0][idx].contiguous(), 64)
    for _ in range(3): hip.hull_mask(P, b["tar_smpl_vertice"][0])
    torch.cuda.synchronize(); ts=[]
    for _ in range(10):
        t0=time.perf_counter(); m,h = hip.hull_mask(P, b["tar_smpl_vertice"][0]); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print(tag, "rays", P.R, "hits", int(h.sum()), "valid", int(m.sum()), "ms", round(np.median(ts)*1e3,3))
allr = torch.arange(512*512, device=dev)
for seq in (True, False):          # TH_HULL_SEQ is read per launch: the one-lane-per-sample form, then the wave-cooperative default
    if seq: os.environ["TH_HULL_SEQ"] = "1"
    else: os.environ.pop("TH_HULL_SEQ", None)
    tm(allr, "full row-major" + (" (TH_HULL_SEQ=1)" if seq else ""))
    tm(shard_ray_indices(512,512,1,0,tile=8,tile_major=True).to(dev), "full tile-major" + (" (TH_HULL_SEQ=1)" if seq else ""))
for r in (0,3):
    tm(shard_ray_indices(512,512,8,r,tile=8,tile_major=True).to(dev), f"shard {r}/8")
