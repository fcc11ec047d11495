"""Off-line parity tail of whole frames: EVERY ray of a 512 x 512 x 64 frame, three ways.

    python tools/dense_tail.py [--frame dense|real] [--cpu-rays 4096] [--out gpurun_out/r05_tail.json]

  gpu      the HIP path (fused fp16 hi/lo x3 kernel, and the per-layer fp32 MFMA path)
  o32      oracle/th_oracle.py in fp32 -- the reference's arithmetic
  t64      the same graph in float64 on the same fp32 inputs (oracle.widen): the exact result both approximate

The oracle is plain torch, so it runs on the HIP device as well as on the host: t64 and an fp32 evaluation (torch-ROCm's
kernels: what the reference itself would compute on this GPU) over all 262 144 rays take a minute on the device instead of
an hour on 32 host threads.  The device evaluations are tied to the host oracle on `--cpu-rays` randomly picked rays
(fp32 and float64 on the host cores), which also gives the CPU-fp32 distances on that subset.
Writes histograms (count of rays per decade of max |rgb, acc| difference), maxima, 99.99th percentiles and the worst rays.
Test infrastructure only (imports oracle/).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import th_oracle as O                       # noqa: E402
from transhuman_amd import synth, hip                   # noqa: E402
from transhuman_amd.config import get_cfg               # noqa: E402
from util import make_sd, make_net, synth_assign, real_assign, csr, can_centres64, can64      # noqa: E402

EDGES = [0.0, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 2e-5, 3e-5, 5e-5, 7e-5, 1e-4, 1.5e-4, 2e-4, 3e-4, 1e-3, 1.0]


def to_dev(obj, dev):
    if torch.is_tensor(obj):
        return obj.to(dev)
    if isinstance(obj, dict):
        return {k: to_dev(v, dev) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_dev(v, dev) for v in obj)
    return obj


def oracle_frame(bc, sd, assign, dev, dtype, block=8192, pick=None, samples=64):
    """[R, 4] (rgb | acc) of the oracle on `dev` in `dtype` over the rays `pick` (default: all), in blocks of rays"""
    off, mem = csr(assign)
    b, s = to_dev(bc, dev), to_dev(sd, dev)
    if dtype != torch.float32:
        b, s = O.widen(b, dtype), O.widen(s, dtype)
    cc = can_centres64(assign).to(dev)
    R = bc["ray_o"].shape[1]
    idx = np.arange(R) if pick is None else np.asarray(pick)
    out = torch.zeros((len(idx), 4), dtype=torch.float64)
    with torch.no_grad():
        hol, pix = O.encoder_forward(s, b["input_imgs"][0][0])
        for a in range(0, len(idx), block):
            sub = dict(b)
            sel = torch.as_tensor(idx[a:a + block], device=dev)
            for k in ("ray_o", "ray_d", "near", "far"):
                sub[k] = b[k][:, sel]
            o, _ = O.render_fast(s, sub, hol, pix, off, mem, cc, n_samples=samples, small_frame_rays=-1)
            out[a:a + block, :3] = o["rgb_map"][0].double().cpu()
            out[a:a + block, 3] = o["acc_map"][0].double().cpu()
    return out


def oracle_sigma_points(bc, sd, assign, pts, dev, dtype, block=65536):
    """sigma_raw of the oracle on `dev` in `dtype` at the points `pts` [N, 3] (all taken as inside the hull): the per-point part of
    if_mesh_renderer.Renderer.render :46-100 (oracle.render_sigma_grid without its host-side mask / zero view directions)"""
    off, mem = csr(assign)
    b, s = to_dev(bc, dev), to_dev(sd, dev)
    if dtype != torch.float32:
        b, s = O.widen(b, dtype), O.widen(s, dtype)
    cc = can_centres64(assign).to(dev)
    pts = pts.to(dev).to(dtype)
    out = torch.zeros(pts.shape[0], dtype=torch.float64)
    with torch.no_grad():
        hol, pix = O.encoder_forward(s, b["input_imgs"][0][0])
        fc = O.frame_constants(s, b, hol, off, mem, cc, 12)
        ps = O.world2smpl(pts, b["Rh"][0], b["Th"][0])
        for a in range(0, pts.shape[0], block):
            x = pts[a:a + block]
            pf = O.pixel_aligned(pix, x, b)
            vd = torch.zeros((x.shape[0], 27), device=dev, dtype=dtype)
            m = torch.ones(x.shape[0], dtype=torch.bool, device=dev)
            raw = O.network_forward(s, pf, vd, ps[a:a + block], fc["centres"], fc["blend"], fc["tokens"], m)
            out[a:a + block] = raw[:, 3].double().cpu()
    return out


def tail_report(r, bc, sd, assign, dev, host_rays=768, host64_every=4, seed=5, hits_only=True):
    """One frame three ways (HIP path through Renderer.render_fast, oracle fp32 and float64 on the device, tied to the host oracle on a
    subset): the distances over ALL rays.  Returns a dict of tensors / numbers; the caller asserts."""
    o = r.render_fast(synth.batch_to(bc, dev), is_train=False)
    img = torch.cat([o["rgb_map"][0], o["acc_map"][0][:, None]], dim=1).double().cpu()
    stats = dict(r.last_stats)
    hip.drop_workspaces(dev)
    torch.cuda.empty_cache()
    t64 = oracle_frame(bc, sd, assign, dev, torch.float64)
    o32 = oracle_frame(bc, sd, assign, dev, torch.float32)
    rs = np.random.RandomState(seed)
    hits = torch.nonzero(img[:, 3] > 0).reshape(-1).numpy()
    pool = hits if hits_only else np.arange(img.shape[0])
    pick = np.sort(rs.choice(pool, min(host_rays, len(pool)), replace=False))
    c32 = oracle_frame(bc, sd, assign, torch.device("cpu"), torch.float32, pick=pick)
    c64 = oracle_frame(bc, sd, assign, torch.device("cpu"), torch.float64, pick=pick[::host64_every])
    d32, d64, n64 = dist(img, o32), dist(img, t64), dist(o32, t64)
    k = int(round(d32.numel() * 0.9999))
    return {"img": img, "o32": o32, "t64": t64, "stats": stats, "pick": pick,
            "tie64": float(dist(t64[pick[::host64_every]], c64).max()), "g_host": dist(img[pick], c32),
            "d32": d32, "d64": d64, "n64": n64, "p9999": {"d32": float(d32.kthvalue(k)[0]), "d64": float(d64.kthvalue(k)[0]),
                                                          "n64": float(n64.kthvalue(k)[0])}}


def flips(bc, sd, assign, dev, rays, gpu, o32, t64, samples=64):
    off, mem = csr(assign)
    out = []
    if len(rays) == 0:
        return out
    cc = can_centres64(assign).to(dev)
    raws = {}
    for dt in (torch.float32, torch.float64):
        b, s = to_dev(bc, dev), to_dev(sd, dev)
        if dt != torch.float32:
            b, s = O.widen(b, dt), O.widen(s, dt)
        sub = dict(b)
        sel = torch.as_tensor(np.asarray(rays), device=dev)
        for k in ("ray_o", "ray_d", "near", "far"):
            sub[k] = b[k][:, sel]
        with torch.no_grad():
            hol, pix = O.encoder_forward(s, b["input_imgs"][0][0])
            _, aux = O.render_fast(s, sub, hol, pix, off, mem, cc, n_samples=samples, small_frame_rays=-1)
        raw = torch.zeros((len(rays), samples, 4), dtype=torch.float64)
        msk = torch.zeros((len(rays), samples), dtype=torch.bool)
        raw[aux["hit"].cpu()] = aux["raw"].double().cpu()
        msk[aux["hit"].cpu()] = aux["mask"].cpu()
        raws[dt] = (raw, msk)
        if dt == torch.float64:
            # near-ties of the discrete decisions, in float64: the gap between the 7th and the 8th nearest token centre of
            # every valid sample (the 7-NN SET changes when it closes) and the distance of the hull test from its threshold
            pts, _ = O.sampling_points(sub["ray_o"][0], sub["ray_d"][0], sub["near"][0], sub["far"][0], samples)
            ps = O.world2smpl(pts, b["Rh"][0], b["Th"][0]).reshape(-1, 3)
            dc = torch.cdist(ps, aux["centres"].to(ps)).sort(dim=1)[0]
            gap78 = (dc[:, 7] - dc[:, 6]).view(len(rays), samples).cpu()
            dv = torch.cdist(pts.reshape(-1, 3), b["tar_smpl_vertice"][0]).min(dim=1)[0]
            hull_margin = (dv - 0.1).abs().view(len(rays), samples).cpu()
    r32, m = raws[torch.float32]
    r64, _ = raws[torch.float64]
    for i, ray in enumerate(rays):
        sg32, sg64 = r32[i, :, 3], r64[i, :, 3]
        fl = torch.nonzero(m[i] & ((sg32 > 0) != (sg64 > 0))).reshape(-1)
        near0 = torch.nonzero(m[i] & (sg64.abs() < 1e-4)).reshape(-1)
        out.append({"ray": int(ray), "gpu_vs_o32": float(dist(gpu[ray:ray + 1], o32[ray:ray + 1])), "gpu_vs_t64": float(dist(gpu[ray:ray + 1], t64[ray:ray + 1])),
                    "o32_vs_t64": float(dist(o32[ray:ray + 1], t64[ray:ray + 1])),
                    "min_gap_7th_8th_neighbour_over_valid_samples": float(gap78[i][m[i]].min()) if bool(m[i].any()) else None,
                    "min_hull_margin_over_samples": float(hull_margin[i].min()),
                    "samples_sign_flip_o32_t64": [int(j) for j in fl], "sigma_raw_t64_at_flips": [float(sg64[j]) for j in fl],
                    "samples_with_abs_sigma_raw_below_1e-4": [int(j) for j in near0], "their_sigma_raw_t64": [float(sg64[j]) for j in near0]})
    return out


def dist(a, b):
    """per-ray max |rgb, acc| difference"""
    return (a.double() - b.double()).abs().max(dim=1)[0]


def summary(d):
    d = d.cpu().double()
    n = d.numel()
    h = np.histogram(d.numpy(), bins=EDGES)[0]
    k = max(1, int(round(n * 0.9999)))
    order = torch.argsort(d, descending=True)[:5]
    return {"rays": int(n), "max": float(d.max()), "p9999": float(d.kthvalue(k)[0]), "mean": float(d.mean()),
            "over_1e-4": int((d > 1e-4).sum()), "over_5e-5": int((d > 5e-5).sum()),
            "hist_edges": EDGES, "hist_counts": [int(x) for x in h],
            "worst_rays": [int(i) for i in order], "worst_vals": [float(d[i]) for i in order]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frame", default="dense", choices=["dense", "real"])
    ap.add_argument("--cpu-rays", type=int, default=4096)
    ap.add_argument("--cpu64-rays", type=int, default=1024)
    ap.add_argument("--nc", type=int, default=500)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_tail.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    hip.load_library()
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = 64, args.nc
    from transhuman_amd.networks.renderer import if_clight_renderer
    net = make_net(12).to(dev)
    # (N_c = 1500: the reference's own ragged kmeans_dict_1500, tests/golden/kmeans_pc2voxel.npz)
    assign = real_assign(args.nc) if args.nc == 1500 else synth_assign(args.nc)
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=assign)
    if args.frame == "dense":
        bc = synth.make_batch(512, 512, 3, seed=0, all_rays=True, dense=True, focal=6000.0, dilate=64)
    else:
        bc = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
    b = synth.batch_to(bc, dev)
    img = {}
    for mode, name in ((1, "fused"), (0, "fp32_mfma")):
        hip.set_mlp_mode(mode)
        o = r.render_fast(b, is_train=False)
        img[name] = torch.cat([o["rgb_map"][0], o["acc_map"][0][:, None]], dim=1).double().cpu()
        st = dict(r.last_stats)
    hip.set_mlp_mode(1)
    hip.drop_workspaces(dev)
    torch.cuda.empty_cache()
    sd = make_sd()
    res = {"frame": args.frame, "stats": {k: int(v) for k, v in st.items()}, "guard": hip.guard_state(dev)}
    t0 = time.time()
    t64 = oracle_frame(bc, sd, assign, dev, torch.float64)
    res["t64_device_s"] = time.time() - t0
    t0 = time.time()
    o32 = oracle_frame(bc, sd, assign, dev, torch.float32)
    res["o32_device_s"] = time.time() - t0
    # the device evaluations tied to the host oracle
    rs = np.random.RandomState(11)
    R = bc["ray_o"].shape[1]
    if args.frame == "dense":
        pick = np.sort(rs.choice(R, args.cpu_rays, replace=False))
    else:
        hits = torch.nonzero(img["fused"][:, 3] > 0).reshape(-1).numpy()
        pick = np.sort(np.concatenate([rs.choice(hits, min(len(hits), args.cpu_rays * 7 // 8), replace=False),
                                       rs.choice(R, args.cpu_rays // 8, replace=False)]))
        pick = np.unique(pick)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    t0 = time.time()
    c32 = oracle_frame(bc, sd, assign, torch.device("cpu"), torch.float32, pick=pick)
    res["o32_host_s"] = time.time() - t0
    p64 = pick[:: max(1, len(pick) // args.cpu64_rays)]
    t0 = time.time()
    c64 = oracle_frame(bc, sd, assign, torch.device("cpu"), torch.float64, pick=p64)
    res["t64_host_s"] = time.time() - t0
    pos64 = np.searchsorted(pick, p64)
    res["tie"] = {
        "rays_host_fp32": int(len(pick)), "rays_host_float64": int(len(p64)),
        "t64_device_vs_t64_host": float(dist(t64[p64], c64).max()),
        "o32_device_vs_o32_host": summary(dist(o32[pick], c32)),
    }
    res["all_rays"] = {
        "fused_vs_o32_device": summary(dist(img["fused"], o32)),
        "fused_vs_t64": summary(dist(img["fused"], t64)),
        "fp32mfma_vs_o32_device": summary(dist(img["fp32_mfma"], o32)),
        "fp32mfma_vs_t64": summary(dist(img["fp32_mfma"], t64)),
        "o32_device_vs_t64": summary(dist(o32, t64)),
        "fused_vs_fp32mfma": summary(dist(img["fused"], img["fp32_mfma"])),
    }
    res["host_subset"] = {
        "fused_vs_o32_host": summary(dist(img["fused"][pick], c32)),
        "fp32mfma_vs_o32_host": summary(dist(img["fp32_mfma"][pick], c32)),
        "o32_host_vs_t64": summary(dist(c32, t64[pick])),
        "fused_vs_t64": summary(dist(img["fused"][pick], t64[pick])),
    }
    # excess of the HIP path over the reference's own fp32 noise, ray by ray: g64 - o64 (negative: closer to the truth)
    ex = dist(img["fused"], t64) - dist(o32, t64)
    res["all_rays"]["fused_excess_over_o32_noise"] = {"max": float(ex.max()), "p9999": float(ex.kthvalue(int(ex.numel() * 0.9999))[0]),
                                                      "rays_gpu_closer": int((ex < 0).sum())}
    # rays further than 1e-4 from either oracle: are they discrete flips?  sigma_raw of a sample within rounding of 0 changes
    # sign between two evaluations; relu(sigma) * delta with delta = 1e10 on a ray's last sample (nerf_net_utils.py:33-35) then
    # turns a 1e-7 difference into alpha = 0 vs alpha = 1, and the progressive RGB pass (cross_transformer.py:298) adds or
    # drops the sample's colour.  For every such ray: the samples whose sigma_raw differs in sign between the fp32 and the
    # float64 oracle, and |sigma_raw| there.
    bad = torch.nonzero((dist(img["fused"], o32) > 1e-4) | (dist(img["fused"], t64) > 1e-4) | (dist(o32, t64) > 1e-4)).reshape(-1).numpy()
    res["rays_over_1e-4_any_pair"] = flips(bc, sd, assign, dev, bad[:64], img["fused"], o32, t64)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    brief = {k: (v["max"], v["p9999"], v["over_1e-4"]) for k, v in res["all_rays"].items() if "max" in v and "p9999" in v and "over_1e-4" in v}
    print(json.dumps({"tie": res["tie"]["t64_device_vs_t64_host"], "o32_tie_max": res["tie"]["o32_device_vs_o32_host"]["max"],
                      "all_rays(max,p9999,>1e-4)": brief,
                      "host_subset": {k: (v["max"], v["over_1e-4"]) for k, v in res["host_subset"].items()},
                      "times": {k: round(v, 1) for k, v in res.items() if k.endswith("_s")}}, indent=1))


if __name__ == "__main__":
    main()
