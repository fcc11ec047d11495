#!/usr/bin/env python
"""bench.py -- rays/s of the TransHuman rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload real|dense]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic 512x512 frame with 64
samples/ray, V=3 reference views, N_c=500 tokens (BASELINE.json configs[1]):
encoder (hand-written HIP too: K12 convolutions, K11 BatchNorm, K8 upsample/concat) -> paint/group -> TransHE ->
DPaRF tables -> [sample placement, hull mask, compaction, DPaRF, pixel gather,
per-point MLP, compositing] -> image.  All inputs are resident in HBM before
the timed region.  With N>1 GPUs the frame's rays are dealt to ranks in
diagonal 8x8-pixel tiles, the encoder's constants are recomputed on every rank
(cheaper than shipping the 0.82 GB feature map), TransHE of frame j runs on rank
j mod N and its tokens are broadcast, and the image is assembled with one RCCL
all_gather -- "strong" scaling: total work fixed, value = rays of the frame /
max-over-ranks time.  The timed loop runs Renderer.render_sequence (the frame
constants of frame i+1 on a second stream under the shading of frame i); the
reference's own call pattern, render_fast per frame, is timed after it and
reported as render_fast_ms_per_step, followed by the `extra` block (the other
configurations of BASELINE.json, each with an oracle spot check).

Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel (the fused per-point MLP: fp32-class arithmetic
as three fp16 MFMA products per MAC), measured live with HIP events on the launch stream;
`cpu_baseline` is the CPU oracle (oracle/th_oracle.py, a port of the reference
arithmetic) timed on this host on a bounded sample of the same frame.
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from transhuman_amd import synth                                    # noqa: E402
from transhuman_amd.config import get_cfg                           # noqa: E402
from transhuman_amd.dist import shard_ray_indices, gather_image, ImageGatherer, DeferredSum, TokenExchange, StemExchange   # noqa: E402

MFMA_F32_PEAK = 157.3e12        # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
SIGMA_BIAS = -1.7


def algorithmic_mlp_flops(V, n_valid, n_pos):
    """Exact MAC counts from the layer shapes (cross_transformer.py:96-126); V=3 gives SURVEY 8a-8's
    1 543 040 (sigma) + 764 416 (rgb) MAC per sample."""
    mac_sigma = V * 491264 + V * V * 384 + 65792
    mac_rgb = V * 249216 + 16768
    return 2.0 * (n_valid * mac_sigma + n_pos * mac_rgb)


def executed_mlp_flops(V, n_valid, n_pos, map_fold=False):
    """Dense MACs the fused kernel actually issues (per operand product triple counted once): the fc_0 token
    part and fc_1 are folded away, the f-consuming layers have K = 272 (260 real), and view_fc is folded over
    feature_fc / rgb_res_0 (128-wide products on inter, f and the 32-wide view-direction rows).  map_fold (TH_ROWS_TEX): the
    three layers that read f (alpha_res_0', [Wa R0' ; rgb_res_1']) are applied to the map's texels once per frame
    (map_fold_kernel: 512 x 272 MACs per texel of the cropped map -- not counted here, < 1 % of the frame) and the fused kernel
    blends their outputs instead of multiplying."""
    f_sigma = 0 if map_fold else 256 * 272
    f_rgb = 0 if map_fold else 256 * 272
    row_sigma = 384 * 256 + f_sigma + 384 * 256 + 256 * 256             # kv1, alpha_res_0', kv0, fc_2
    row_rgb = 128 * 256 + 128 * 32 + f_rgb                              # (Wa F), Wd, stacked [Wa R0' ; rgb_res_1']
    mac_sigma = V * row_sigma + 256 * 64 + 256 * 256                      # + fc_0 PE part, fc_3 (per sample)
    mac_rgb = V * row_rgb + 128 * 128                                     # + fc_4
    return 2.0 * (n_valid * mac_sigma + n_pos * mac_rgb)


MFMA_F16_PEAK = 2500e12         # dense fp16/bf16 MFMA peak (AMD's 5 PF figure is 2:1 sparse)


def hbm_traffic():
    """HBM bytes per full launch of the three per-sample kernels from the committed PMC passes: profiles/hbm_traffic.json
    is written by `tools/pmc_hbm_summary.py DIR --json` from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_hbm.sh (2 *
    FETCH_SIZE + WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md) and names the summary it came from."""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(p) as f:
            text = f.read()
        return json.loads(text[text.index("{"):])           # (tolerates a commented table in front of the object)
    except (OSError, ValueError):
        return None


def roofline_block(mlp_mode, achieved, flops_step, stage_ms, launches, executed_step=None, n_valid=0, folded_flops=0.0, fold_ms=0.0):
    """Dominant kernel = the per-point MLP.  `achieved` counts ALGORITHMIC fp32 FLOPs (the reference's
    layer shapes).  mode 1 (default): the fused kernel evaluates every fp32 MAC as three fp16 MFMA MACs
    (hi*hi + hi*lo + lo*hi, fp32 accumulate), so the pipe it is bound by is the fp16 MFMA pipe at one third
    of its 2.5 PFLOP/s dense peak; mode 0: fp32 MFMA GEMM launches, peak 157.3 TFLOP/s."""
    if mlp_mode == 1:
        peak = MFMA_F16_PEAK / 3.0
        from transhuman_amd import hip as _hip
        k8 = _hip.fused_waves() == 8
        kernel = (("mlp_fused8_kernel<3> (8 waves per workgroup, two per SIMD; fp16 hi/lo split x3 on v_mfma_f32_16x16x32_f16" if k8 else
                   "mlp_fused_kernel<3,1,true> (fp16 hi/lo split x3 on v_mfma_f32_32x32x16_f16") +
                  "; TH_ROWS_TEX: alpha_res_0 / rgb_res_0 / "
                  "rgb_res_1 are applied to the map's texels once per frame (map_fold_kernel) and the kernel blends texel rows of "
                  "the folded maps: the ALGORITHMIC FLOPs still count those layers per sample), rank 0" if _hip.tex_rows_enabled() else
                  "mlp_fused_kernel<3,1> (fp16 hi/lo split x3 on v_mfma_f32_32x32x16_f16), rank 0")
    else:
        peak = MFMA_F32_PEAK
        kernel = "per-point MLP stage (gemm_f32_mfma_kernel x14 + glue kernels), rank 0"
    t = hbm_traffic() if mlp_mode == 1 else None
    # the committed PMC pass measured launches of t["launch_samples"] samples: bytes per sample x this run's samples per launch
    per_launch = (float(n_valid) / max(launches, 1.0) / t["launch_samples"]) if t else 0.0
    traffic = t["mlp_fused_bytes_per_launch"] * per_launch if t else None
    # `frac`: the reference's layer shapes (ALGORITHMIC FLOPs) over the time of EVERY kernel that evaluates them -- the fused
    # kernel's launches plus map_fold_kernel's (the three folded layers are its work; its HIP-event span on the side stream is
    # stretched by the overlap with the shading, which errs on the low side).  `frac_executed` / `mfma_pipe_util` say how busy
    # the matrix pipe really is; the round-4 line divided by the fused kernel's time alone (kept as `frac_fused_kernel_time_only`).
    total_s = max((stage_ms + fold_ms) * 1e-3, 1e-12)
    frac = flops_step / total_s / peak
    return {"bound": "mfma", "kernel": kernel, "achieved": flops_step / total_s / 1e12, "peak": peak / 1e12,
            "unit": "TFLOP/s", "frac": frac, "traffic": traffic,
            "frac_note": "algorithmic FLOPs / (mlp_fused_kernel + map_fold_kernel time) / (fp16 MFMA peak / 3); read with "
                         "frac_executed (MFMA work the pipe actually did / time / the same peak) and mfma_pipe_util",
            "frac_fused_kernel_time_only": achieved / peak,
            "fold_kernel_ms_per_step": fold_ms,
            "traffic_note": (f"HBM bytes per launch ({n_valid / max(launches, 1.0):.0f} samples), rocprofv3 PMC (" + t["source"] +
                             f", measured on a launch of {t['launch_samples']} samples); algorithmic " +
                             f"{t['mlp_fused_algorithmic_bytes_per_launch'] * per_launch:.3g} (" + t["algorithmic_note"] + ")") if t else
                            "no committed PMC pass (profiles/hbm_traffic.json absent)",
            # TH_ROWS_TEX: alpha_res_0 / rgb_res_0 / rgb_res_1 (cross_transformer.py:316, :334, :346; 98 304 + 147 456 MAC per (sample,
            # view) in the reference) are evaluated per TEXEL by map_fold_kernel, not by this kernel: the same figure with their
            # algorithmic FLOPs taken out of the numerator -- what the kernel achieves on the layers it still multiplies
            "frac_without_folded_layers": ((flops_step - folded_flops) / max(stage_ms * 1e-3, 1e-12) / peak) if folded_flops else None,
            "folded_flop_per_step": folded_flops or None,
            "frac_of_fp32_mfma_peak": achieved / MFMA_F32_PEAK,
            # the same ALGORITHMIC number against the raw dense fp16 MFMA peak (what a reader who does not accept the /3 sees)
            "frac_of_fp16_mfma_peak": achieved / MFMA_F16_PEAK,
            # executed MFMA work (3 fp16 products per executed fp32 MAC) / time / raw fp16 peak: how busy the matrix pipe is
            "mfma_pipe_util": (3.0 * executed_step / max(stage_ms * 1e-3, 1e-12) / MFMA_F16_PEAK) if (executed_step and mlp_mode == 1) else None,
            "traffic_per_frame": ({"measured_K4_K5_K6_bytes": (t["mlp_fused_bytes_per_launch"] + t["pixgather_bytes_per_launch"] +
                                                               t["dparf_bytes_per_launch"]) * float(n_valid) / t["launch_samples"],
                                   "survey_8d_algorithmic_bytes": 1.22e9,
                                   "gather_kernel": t.get("gather_kernel", "pixgather_s256_kernel"),
                                   "note": "PMC bytes per sample x this frame's valid samples against SURVEY 8d's "
                                           "unique-footprint figure.  With K5's rows (TH_ROWS_TEX=0) the gap was the K5 -> HBM -> "
                                           "K6 row round trip (rows written once, read twice: 26 GB per frame); with the texel "
                                           "hand-over what is left above the footprint are texel rows that miss L2 when a tile "
                                           "of the fused kernel copies them (twice per tile) and the token-branch records"} if t else None),
            "algorithmic_flop_per_step": flops_step, "kernel_ms_per_step": stage_ms,
            "launches_per_step": launches,
            # what the matrix pipes really did (after the algebraic folds): executed FLOPs / time / peak
            "executed_flop_per_step": executed_step,
            "frac_executed": (executed_step / max(stage_ms * 1e-3, 1e-12) / peak) if (executed_step and mlp_mode == 1) else None}


def texel_handover_block(V, n_valid, gather_ms):
    """TH_ROWS_TEX (default on the fused path, split map): the producer of the pixel branch is K5t (k_pixtex.hip) -- it writes, per
    32-sample tile, the list of distinct corner texels and per (sample, view) a 32-byte record + 16 bytes of colour; the rows
    themselves are formed inside the fused kernel (fill_tex) from texels of the map and never reach HBM."""
    tiles = (n_valid + 31) // 32
    t = hbm_traffic()
    hbm = (t["pixgather_bytes_per_launch"] * n_valid / t["launch_samples"]) if t and "pixtex" in t.get("gather_kernel", "") else None
    return {"kernel": f"pixtex_kernel<{V}> (K5t: per-tile texel lists + per-row records; the rows are blended inside the fused kernel)",
            "ms_per_step": gather_ms,
            "bytes_written_per_step": float(n_valid) * V * 32 + tiles * 512.0,
            "bytes_per_sample": V * 32 + 16,
            "hbm_bytes_per_step_pmc": hbm,
            "row_handover": "TH_ROWS_TEX",
            "replaces": "pixgather_s256_kernel (K5): 3 x 1088 B per sample written to HBM and read back twice; TH_ROWS_TEX=0 "
                        "switches back (then this block reports K5 against its texture-path ceiling)",
            "note": "runs beside K4 (neighbour records) on a second stream"}


def gather_block(V, n_valid, gather_ms, n_cu=256, clock_hz=2.4e9):
    """K5 (pixgather_s256_kernel) against its own ceiling: bytes through the texture path per (sample, view) row = 4 corner
    texels of 1040 B (1 KiB of latents + 16 B colour) + one 1088-byte split row written; tools/ubench/load_rate.hip
    reaches ~50 B/clk/CU for 1 KiB row gathers out of L2."""
    rows = float(n_valid) * V
    byts = rows * (4 * 1040 + 1088)
    sec = max(gather_ms * 1e-3, 1e-12)
    t = hbm_traffic()
    hbm = (t["pixgather_bytes_per_launch"] * n_valid / t["launch_samples"]) if t else None
    return {"kernel": "pixgather_s256_kernel", "texture_path_bytes_per_step": byts, "ms_per_step": gather_ms,
            # bytes through the TEXTURE path (L1 / L2 hits included) -- NOT HBM bytes: this figure can exceed the HBM peak
            "texture_path_TB_per_s": byts / sec / 1e12,
            "hbm_TB_per_s": (hbm / sec / 1e12) if hbm else None,
            "hbm_note": ("HBM bytes from the committed PMC pass (" + t["source"] + "), scaled to this frame's samples") if t else None,
            "B_per_clk_per_CU": byts / sec / clock_hz / n_cu, "ubench_ceiling_B_per_clk_per_CU": 50.0,
            "frac_of_ubench_ceiling": byts / sec / clock_hz / n_cu / 50.0,
            # the same launch with every corner load hitting L1 (tools/k5_locality.py `one`, profiles/r04_e_k5_bound.txt): what the
            # kernel's own instruction stream + its 6.8 GB of stores cost; the rest is the latency of the frame's L1 misses
            "all_hit_floor_ms_standalone": 1.44,
            "note": f"at the nominal {clock_hz / 1e9:.1f} GHz, {n_cu} CUs; runs beside K4 (neighbour records) on a second stream"}


def load_assign(k, body):
    p = os.path.join(ROOT, "tests", "golden", "synth_assign.npz")
    if os.path.exists(p):
        d = np.load(p)
        if f"assign_{k}" in d.files:
            return d[f"assign_{k}"].astype(np.int64)
    return synth.kmeans_assign(body, k).astype(np.int64)


def build_net(device):
    from transhuman_amd.networks.cross_transformer import Network
    torch.manual_seed(0)
    net = Network()
    net.load_state_dict(synth.det_state_dict(net.state_dict(), seed=0, sigma_bias=SIGMA_BIAS))
    net.train()                                                     # run.py:29
    return net.to(device)


def cpu_baseline(batch, assign, n_samples, stride=64, gpu_img=None):
    """Oracle on the host cores, bounded sample: every `stride`-th ray of the same frame
    (frame constants computed once, per-ray part extrapolated linearly).  With `gpu_img` ([R,5] rgb|acc|depth of
    the timed GPU frame) the same rays double as the in-job parity check: max |rgb,acc| difference and PSNR
    (-10 log10 mse, if_nerf.py:34-37) of the GPU image against the oracle."""
    from oracle import th_oracle as O
    from transhuman_amd.networks.cross_transformer import Network
    # torch's intra-op pool stops scaling (and thrashes) far below the 256 hardware threads of the
    # GPU box on these small per-chunk ops: use at most 32 and report that number
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = Network()
    sd = synth.det_state_dict(net.state_dict(), seed=0, sigma_bias=SIGMA_BIAS)
    off, mem = synth.csr_from_assign(assign)
    body = batch["tar_smpl_vertice_smplcoord"][0].numpy()
    can = torch.from_numpy(body.astype(np.float64) * 1.02 + 0.001)
    can_c = torch.stack([can[torch.as_tensor(mem[off[i]:off[i + 1]], dtype=torch.long)].mean(0)
                         for i in range(len(off) - 1)])
    R = batch["ray_o"].shape[1]
    sub = dict(batch)
    for k in ("ray_o", "ray_d", "near", "far"):
        sub[k] = batch[k][:, ::stride]
    Rs = sub["ray_o"].shape[1]
    with torch.no_grad():
        t0 = time.perf_counter()
        hol, pix = O.encoder_forward(sd, batch["input_imgs"][0][0])
        t1 = time.perf_counter()
        o_out, _ = O.render_fast(sd, sub, hol, pix, off, mem, can_c, n_samples=n_samples, vit_depth=12,
                                 small_frame_rays=-1)
        t2 = time.perf_counter()
        fc_t0 = time.perf_counter()
        O.frame_constants(sd, batch, hol, off, mem, can_c, 12)
        t_fc = time.perf_counter() - fc_t0
    t_rays = max((t2 - t1) - t_fc, 1e-9)
    est_frame = (t1 - t0) + t_fc + t_rays * (R / Rs)
    res = {"value": R / est_frame, "unit": "rays/s", "cores": cores, "kind": "port",
           "sample": f"oracle/th_oracle.py (torch CPU fp32, {cores} threads): every {stride}th ray of the same "
                     f"512x512x{n_samples} frame ({Rs} rays, brute-force K=1 hull test), per-ray time scaled to "
                     f"{R} rays + per-frame constants once; measured {t2 - t0:.1f} s"}
    if gpu_img is not None:
        g = gpu_img[::stride].detach().cpu().double()
        ref = torch.cat([o_out["rgb_map"][0], o_out["acc_map"][0][:, None]], dim=1).double()
        diff = g[:, :4] - ref
        hit = int((o_out["acc_map"][0] > 0).sum())
        res["gpu_vs_oracle"] = {"rays": Rs, "rays_hit": hit, "max_abs_rgb_acc": float(diff.abs().max()),
                                "psnr_rgb_db": float(-10.0 * torch.log10(torch.clamp((diff[:, :3] ** 2).mean(), min=1e-30))),
                                "bar": "1e-4 on rgb/acc (BASELINE.json north_star)"}
    return res


def oracle_rays_check(batch_cpu, assign, n_samples, ray_idx, gpu_img, sd=None, truth=False):
    """max |rgb, acc| of the GPU image against the CPU oracle on the rays `ray_idx` of the frame (masked branch).
    truth=True adds the float64 evaluation of the same graph on the same fp32 inputs (oracle.widen): the distance of the
    GPU image AND of the fp32 oracle from it -- whether the HIP path is further from the exact result than the
    reference's own fp32 arithmetic."""
    from oracle import th_oracle as O
    from transhuman_amd.networks.cross_transformer import Network
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    if sd is None:
        torch.manual_seed(0)
        sd = synth.det_state_dict(Network().state_dict(), seed=0, sigma_bias=SIGMA_BIAS)
    off, mem = synth.csr_from_assign(assign)
    body = batch_cpu["tar_smpl_vertice_smplcoord"][0].numpy()
    can = torch.from_numpy(body.astype(np.float64) * 1.02 + 0.001)
    can_c = torch.stack([can[torch.as_tensor(mem[off[i]:off[i + 1]], dtype=torch.long)].mean(0)
                         for i in range(len(off) - 1)])
    sub = dict(batch_cpu)
    for k in ("ray_o", "ray_d", "near", "far"):
        sub[k] = batch_cpu[k][:, ray_idx]
    with torch.no_grad():
        hol, pix = O.encoder_forward(sd, batch_cpu["input_imgs"][0][0])
        o_out, _ = O.render_fast(sd, sub, hol, pix, off, mem, can_c, n_samples=n_samples, vit_depth=12, small_frame_rays=-1)
    ref = torch.cat([o_out["rgb_map"][0], o_out["acc_map"][0][:, None]], dim=1).double()
    g = gpu_img[ray_idx].detach().cpu().double()[:, :4]
    d = g - ref
    res = {"rays": int(len(ray_idx)), "rays_hit": int((o_out["acc_map"][0] > 0).sum()),
           "max_abs_rgb_acc": float(d.abs().max()),
           "psnr_rgb_db": float(-10.0 * torch.log10(torch.clamp((d[:, :3] ** 2).mean(), min=1e-30)))}
    if truth:
        sub64, sd64 = O.widen(sub), O.widen(sd)
        with torch.no_grad():
            hol, pix = O.encoder_forward(sd64, sub64["input_imgs"][0][0])
            t_out, _ = O.render_fast(sd64, sub64, hol, pix, off, mem, can_c, n_samples=n_samples, vit_depth=12, small_frame_rays=-1)
        t = torch.cat([t_out["rgb_map"][0], t_out["acc_map"][0][:, None]], dim=1)
        res["vs_float64_oracle"] = {"gpu": float((g - t).abs().max()), "fp32_oracle": float((ref - t).abs().max()),
                                    "note": "max |rgb, acc| against the float64 evaluation of the same graph on the same fp32 "
                                            "inputs: the fp32 oracle's own distance from it is the reference's rounding noise"}
    return res


def fused_vs_fp32(renderer, b, hip):
    """The whole frame `b` rendered by the fused fp16 hi/lo x3 kernel (mode 1) and by the per-layer fp32 MFMA path (mode 0,
    itself golden-checked): max and 99.99th percentile of |d rgb|, |d acc| over ALL rays of the frame."""
    try:
        hip.set_mlp_mode(1)
        o1 = renderer.render_fast(b)
        st = dict(renderer.last_stats)
        hip.set_mlp_mode(0)
        o0 = renderer.render_fast(b)
    finally:
        hip.set_mlp_mode(1)
    d_rgb = (o1["rgb_map"][0].double() - o0["rgb_map"][0].double()).abs().max(dim=-1)[0]
    d_acc = (o1["acc_map"][0].double() - o0["acc_map"][0].double()).abs()
    n = d_rgb.numel()
    k = max(1, int(round(n * 0.9999)))
    return {"rays": int(n), "valid_samples": int(st["valid_samples"]),
            "max_abs_rgb": float(d_rgb.max()), "p9999_abs_rgb": float(d_rgb.kthvalue(k)[0]),
            "max_abs_acc": float(d_acc.max()), "p9999_abs_acc": float(d_acc.kthvalue(k)[0]),
            "worst_ray": int(torch.argmax(torch.maximum(d_rgb, d_acc))),
            "vs": "per-layer fp32 MFMA path (th_set_mlp_mode 0), same kernels otherwise; bar 5e-5 (half the 1e-4 budget)"}


def time_steps(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def run_extras(dev, net, args, H, W, V):
    """The other configurations of BASELINE.json / SURVEY 8d, measured AFTER (outside) the timed region of the
    headline run on the same device, each with its own oracle spot check: S-dense (every sample of a hit ray inside
    the hull), C4 (N_c = 1500), C3 (orbit along the reference's gen_path_virt, rays generated on device) and C5
    (sigma on a grid^3 voxel grid + marching cubes).  Short runs (3 frames each): secondary numbers."""
    from oracle import th_oracle as O
    from transhuman_amd import hip
    from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
    from transhuman_amd.networks.renderer.if_mesh_renderer import Renderer as MeshRenderer
    cfg = get_cfg()
    extra = {}
    rs = np.random.RandomState(3)

    def frame_case(name, nc, dense, n_hit_rays=192, modes=False):
        cfg.num_class = nc
        bc = synth.make_batch(H, W, V, seed=0, all_rays=True, dense=dense)
        body = bc["tar_smpl_vertice_smplcoord"][0].numpy()
        assign = load_assign(nc, body)
        r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=assign)
        b = synth.batch_to(bc, dev)
        seq = r.render_sequence(itertools.repeat(b))
        ms, out = time_steps(lambda: next(seq), 8, warmup=3)        # (8 frames: the first few of a sequence fill its pipeline)
        img = torch.cat([out["rgb_map"][0], out["acc_map"][0][:, None], out["depth_map"][0][:, None]], dim=1)
        hit = torch.nonzero(out["acc_map"][0] > 0).reshape(-1).cpu().numpy()
        idx = np.sort(np.concatenate([rs.choice(hit, n_hit_rays, replace=False), rs.choice(H * W, 64, replace=False)]))
        chk = oracle_rays_check(bc, assign, args.samples, idx, img)
        extra[name] = {"ms_per_frame": ms, "rays_per_s": H * W / ms * 1e3, "valid_samples": int(r.last_stats["valid_samples"]),
                       "n_clusters": nc, "gpu_vs_oracle": chk}
        seq.close()
        if modes:
            extra[name]["fused_vs_fp32_full_frame"] = fused_vs_fp32(r, b, hip)

    frame_case("S_dense", args.nc, True)
    frame_case("C4_nc1500", 1500, False, n_hit_rays=448, modes=True)
    cfg.num_class = args.nc

    # SURVEY 8d's S-dense regime proper: a long lens on the torso + a per-ray slab hugging the surface -> (nearly) every one
    # of the R x S = 16.8 M samples is valid (32 full passes of the per-sample stage, ~77 TFLOP of algorithmic MLP work):
    # launch tails, the hull stage and the frame constants are negligible, the fused kernel's number stands alone
    bc = synth.make_batch(H, W, V, seed=0, all_rays=True, dense=True, focal=6000.0 * W / 512, dilate=64)
    body = bc["tar_smpl_vertice_smplcoord"][0].numpy()
    assign = load_assign(args.nc, body)
    r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=assign)
    b = synth.batch_to(bc, dev)
    seq = r.render_sequence(itertools.repeat(b))
    next(seq); next(seq)
    hip.profile_enable(True)
    hip.profile_read()
    ms, out = time_steps(lambda: next(seq), 3, warmup=0)
    prof = hip.profile_read()
    hip.profile_enable(False)
    n_valid = int(r.last_stats["valid_samples"])
    img = torch.cat([out["rgb_map"][0], out["acc_map"][0][:, None], out["depth_map"][0][:, None]], dim=1)
    pts = hip.Points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], args.samples)
    n_pos = count_sigma_positive(hip, net, r.last_frame, pts)
    flops = algorithmic_mlp_flops(V, n_valid, n_pos)
    mlp_ms = prof["mlp"][0] / 3.0
    idx = np.sort(rs.choice(H * W, args.dense_oracle_rays, replace=False))
    seq.close()
    dense_modes = fused_vs_fp32(r, b, hip)
    extra["S_dense_full"] = {"ms_per_frame": ms, "rays_per_s": H * W / ms * 1e3, "valid_samples": n_valid,
                             "sigma_pos_samples": n_pos, "algorithmic_mlp_flop": flops, "mlp_ms_per_frame": mlp_ms,
                             "mlp_TFLOP_per_s": flops / max(mlp_ms * 1e-3, 1e-12) / 1e12,
                             "roofline_frac": flops / max(mlp_ms * 1e-3, 1e-12) / (MFMA_F16_PEAK / 3.0),
                             "gpu_vs_oracle": oracle_rays_check(bc, assign, args.samples, idx, img, truth=True),
                             "fused_vs_fp32_full_frame": dense_modes}
    hip.drop_workspaces(dev)
    torch.cuda.empty_cache()

    # C3: orbit (the reference's virtual camera path, rays on device)
    bc = synth.make_batch(H, W, V, seed=0, all_rays=True)
    body = bc["tar_smpl_vertice_smplcoord"][0].numpy()
    assign = load_assign(args.nc, body)
    r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=assign)
    b = synth.batch_to(bc, dev)
    verts = bc["tar_smpl_vertice"][0].numpy()
    bounds = np.stack([verts.min(0), verts.max(0)]).astype(np.float32)
    bounds[0, 2] -= 0.05; bounds[1, 2] += 0.05
    K = np.array([[600.0 * W / 512, 0, W / 2], [0, 600.0 * W / 512, H / 2], [0, 0, 1]], np.float32)
    from transhuman_amd.camera_path import gen_path_virt, synthetic_rig
    centre = 0.5 * (bounds[0] + bounds[1]).astype(np.float64)
    w2c = gen_path_virt(synthetic_rig(centre=tuple(centre.tolist())), render_views=60)

    px = shard_ray_indices(H, W, 1, 0, tile=8, tile_major=True).to(dev)      # 8 x 8 pixel tiles (the sample list's unit)

    def frames():
        i = 0
        while True:
            RT = w2c[i % 60]
            rays = hip.gen_rays(K, RT[:3, :3].astype(np.float32), RT[:3, 3:].astype(np.float32), bounds, H, W, device=dev,
                                compact=False)
            sh = dict(b)
            for k in ("ray_o", "ray_d", "near", "far"):
                sh[k] = rays[k][px][None]
            yield sh
            i += 1
    seq = r.render_sequence(frames())
    ms, out = time_steps(lambda: next(seq), 6, warmup=2)
    lb = r.last_batch
    img = torch.cat([out["rgb_map"][0], out["acc_map"][0][:, None], out["depth_map"][0][:, None]], dim=1)
    hit = torch.nonzero(out["acc_map"][0] > 0).reshape(-1).cpu().numpy()
    idx = np.sort(np.concatenate([rs.choice(hit, min(192, len(hit)), replace=False), rs.choice(H * W, 64, replace=False)]))
    bc2 = dict(bc)
    for k in ("ray_o", "ray_d", "near", "far"):
        bc2[k] = lb[k].cpu()
    extra["C3_orbit"] = {"ms_per_frame": ms, "rays_per_s": H * W / ms * 1e3, "camera_path": "gen_path_virt(21-camera rig, 60 views)",
                         "hit_rays": int(r.last_stats["hit_rays"]), "gpu_vs_oracle": oracle_rays_check(bc2, assign, args.samples, idx, img)}
    seq.close()

    # C5: sigma grid + marching cubes (its own 16.7 M-point workspace: give the frame workspaces back first)
    hip.drop_workspaces(dev)
    torch.cuda.empty_cache()
    g = args.grid
    mr = MeshRenderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=assign)
    mb = dict(bc)
    mb["pts"] = synth.make_grid_pts(bc, g)
    mbd = synth.batch_to(mb, dev)
    old_th = cfg.mesh_th
    cfg.mesh_th = 0.5                      # sigma_raw of the synthetic weights is O(1) (the reference's 20 fits trained weights)
    try:
        ms, out = time_steps(lambda: mr.render(mbd), 4, warmup=2)
        flat = mbd["pts"].reshape(-1, 3)
        sig_ms, _ = time_steps(lambda: hip.eval_sigma_grid(net, mr.prepare_frame(mbd), flat), 4, warmup=1)
        cube = out["cube"][10:-10, 10:-10, 10:-10]
        # oracle on a sample of voxels near the surface + a few anywhere
        nz = np.flatnonzero(cube.reshape(-1) != 0)
        pick = np.sort(np.concatenate([rs.choice(nz, 1500, replace=False), rs.choice(g * g * g, 500, replace=False)]))
        torch.manual_seed(0)
        from transhuman_amd.networks.cross_transformer import Network
        sd = synth.det_state_dict(Network().state_dict(), seed=0, sigma_bias=SIGMA_BIAS)
        off, mem = synth.csr_from_assign(assign)
        can = torch.from_numpy(body.astype(np.float64) * 1.02 + 0.001)
        can_c = torch.stack([can[torch.as_tensor(mem[off[i]:off[i + 1]], dtype=torch.long)].mean(0) for i in range(len(off) - 1)])
        with torch.no_grad():
            hol, pix = O.encoder_forward(sd, bc["input_imgs"][0][0])
            ref = O.render_sigma_grid(sd, mb, mb["pts"].reshape(-1, 3)[pick].reshape(1, -1, 1, 1, 3), hol, pix, off, mem, can_c)
        d = np.abs(cube.reshape(-1)[pick] - ref.reshape(-1).numpy())
        mesh = out["mesh"]
        extra["C5_mesh"] = {"grid": g, "ms_per_frame": ms, "sigma_grid_ms": sig_ms, "voxels_per_s": g ** 3 / ms * 1e3,
                            "valid_voxels": int(mr.last_stats["valid_samples"]), "mesh_vertices": int(mesh.vertices.shape[0]),
                            "mesh_triangles": int(mesh.faces.shape[0]), "mesh_th": 0.5,
                            "gpu_vs_oracle": {"voxels": int(len(pick)), "max_abs_sigma": float(d.max()),
                                              "max_abs_sigma_oracle": float(np.abs(ref.reshape(-1).numpy()).max()),
                                              "note": "sigma is the RAW density (tens to hundreds here), not alpha: the 1e-4 bar is on "
                                                      "rgb / alpha; every valid voxel of the grid against the device oracle: "
                                                      "tests/test_gpu_round6.py"}}
    finally:
        cfg.mesh_th = old_th
    return extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)          # (1.5 s of timed frames at N = 1)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="real", choices=["real", "dense", "orbit", "mesh"],
                    help="real/dense: the headline frame (SURVEY 8d C2); orbit: C3, a new target camera every step, rays "
                         "made on device (th_gen_rays); mesh: C5, sigma on a --grid^3 voxel grid (voxels/s)")
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--nc", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short post-run measurements of the other configurations")
    ap.add_argument("--cpu-stride", type=int, default=128)
    ap.add_argument("--dense-oracle-rays", type=int, default=1024,
                    help="extras: rays of the all-valid S_dense_full frame checked against the CPU oracle")
    ap.add_argument("--mlp-mode", type=int, default=1, help="1 fused fp16-split MFMA kernel, 0 per-layer fp32 MFMA")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="render every frame with Renderer.render_fast (constants -> shading back to back on one stream) "
                         "instead of Renderer.render_sequence (constants of frame i+1 on a second stream under the "
                         "shading of frame i)")
    ap.add_argument("--emulate-rank", type=int, default=0, help="with --emulate-world: which rank's shard")
    ap.add_argument("--tile", type=int, default=8, help="edge of the pixel tiles dealt round-robin to the ranks")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="developer aid: on ONE GPU render rank 0's shard of an N-rank job (no collectives); the JSON "
                         "line is then per-rank time, not a bench result")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # TH_ONE_GPU=1 (developer aid, with TH_DIST_BACKEND=gloo): every rank of a torchrun job uses cuda:0 -- the multi-rank
    # control flow (ray shards, token exchange, deferred count, image gather) end to end on a one-GPU box; RCCL refuses
    # two ranks on one device, gloo moves CUDA tensors through the host
    if os.environ.get("TH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TH_FORCE_DIST=1 runs the RCCL code path (process group, all_reduce, all_gather) even with one rank
    dist_on = world > 1 or os.environ.get("TH_FORCE_DIST") == "1"
    if dist_on and "RANK" not in os.environ:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TH_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    if os.environ.get("TH_MAIN_PRIORITY"):       # A/B switch: the shading stream as a high-priority stream (-1)
        torch.cuda.set_stream(torch.cuda.Stream(dev, priority=int(os.environ["TH_MAIN_PRIORITY"])))
    from transhuman_amd import hip
    from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
    hip.set_mlp_mode(args.mlp_mode)
    cfg = get_cfg()
    cfg.N_samples = args.samples
    cfg.num_class = args.nc
    H = W = args.res
    V = 3
    batch_cpu = synth.make_batch(H, W, V, seed=0, all_rays=True, dense=(args.workload == "dense"))
    body = batch_cpu["tar_smpl_vertice_smplcoord"][0].numpy()
    assign = load_assign(args.nc, body)
    net = build_net(dev)
    can = body.astype(np.float64) * 1.02 + 0.001
    renderer = Renderer(net, vertex_can=can, pc2voxel_ind=assign)
    batch = synth.batch_to(batch_cpu, dev)
    R = batch["ray_o"].shape[1]
    tile_major = os.environ.get("TH_RAY_ORDER", "tile") == "tile"
    emu = args.emulate_world if world == 1 and args.emulate_world > 1 else 0
    my_idx = shard_ray_indices(H, W, emu or world, args.emulate_rank if emu else rank, tile=args.tile,
                               tile_major=tile_major).to(dev)
    shard = dict(batch)
    for k in ("ray_o", "ray_d", "near", "far"):
        shard[k] = batch[k][:, my_idx].contiguous()
    # The timed loop alternates TWO synthetic frames (seed 0 / seed 1: another pose, other reference images) so that nothing a
    # frame leaves behind -- caches, graph-ring instances, demand lists -- can be credited to identical inputs.
    # TH_BENCH_ONE_FRAME=1: the single repeated frame of rounds 1-5.
    frames = [shard]
    batches = [batch]
    if args.workload in ("real", "dense") and os.environ.get("TH_BENCH_ONE_FRAME") != "1":
        batch_b = synth.batch_to(synth.make_batch(H, W, V, seed=1, all_rays=True, dense=(args.workload == "dense")), dev)
        shard_b = dict(batch_b)
        for k in ("ray_o", "ray_d", "near", "far"):
            shard_b[k] = batch_b[k][:, my_idx].contiguous()
        frames.append(shard_b)
        batches.append(batch_b)

    if args.workload in ("orbit", "mesh"):
        return run_secondary(args, world, rank, dev, dist_on, renderer, net, batch, batch_cpu, H, W, V)

    gatherer = ImageGatherer(my_idx, R, world) if dist_on else None      # shard layout exchanged once

    # The frames of the job arrive as a stream (the reference renders one dataset item after the other, run.py:96-118):
    # Renderer.render_sequence computes the per-frame constants (encoder, paint, TransHE) of step i+1 on a second HIP
    # stream while step i shades; every step still does one full frame of work (the look-ahead of the last timed
    # step replaces the constants the first timed step received from the warm-up).
    sharded = dist_on or emu
    # TransHE is not replicated: frame j's tokens are computed by rank j % world and broadcast (1.15 MB) from the
    # side stream, two collectives ahead of their use (transhuman_amd/dist.py TokenExchange)
    tokens_x = TokenExchange() if dist_on else (TokenExchange(emulate=(emu, args.emulate_rank)) if emu else None)
    # ... and from 4 ranks on the ResNet stem too: its low-resolution latents (69 MB) are broadcast, the 0.82 GB map is
    # still built locally (dist.StemExchange; TH_STEM_EXCHANGE=0|1 overrides)
    stem_x = None
    if (dist_on or emu) and StemExchange.wanted(emu or world):
        stem_x = StemExchange() if dist_on else StemExchange(emulate=(emu, args.emulate_rank))
    seq = None if args.no_pipeline else renderer.render_sequence(itertools.cycle(frames),
                                                                  small_frame_rays=-1 if sharded else 2400,
                                                                  token_exchange=tokens_x, stem_exchange=stem_x)

    # The reference's R' <= 2400 switch (if_clight_renderer.py:551) looks at the WHOLE frame.  Shards are rendered in the
    # (overwhelmingly common) masked mode; the per-rank hit-ray counts th_render_rays reports anyway are summed with an
    # 8-byte all-reduce and only if the frame total is <= 2400 the shard is rendered again in the reference's un-masked
    # mode.  The all-reduce has its own communicator and stream and is checked AFTER the frame's work is queued, so the
    # host never waits for the shading before it can queue the next frame.
    whole_frame_hits = DeferredSum(dev) if dist_on else None
    dist_wait = [0.0]

    step_no = [0]
    stats_of = {}

    def step():
        # per frame: ray-only stage (hull mask, compaction) -> [per-frame constants] -> shading + compositing.
        shard = frames[step_no[0] % len(frames)]
        step_no[0] += 1
        if seq is not None:
            out = next(seq)
        else:
            out = renderer.render_fast(shard, small_frame_rays=-1 if sharded else 2400)
        stats_of[(step_no[0] - 1) % len(frames)] = dict(renderer.last_stats)
        if dist_on:
            whole_frame_hits.start(renderer.last_stats["hit_rays"])
        local = torch.cat([out["rgb_map"][0], out["acc_map"][0][:, None], out["depth_map"][0][:, None]], dim=1)
        if dist_on:
            img = gatherer(local)
            tw = time.perf_counter()
            total_hits = whole_frame_hits.result()            # (host wait for the 8-byte all-reduce: not the host's own cost)
            dist_wait[0] += time.perf_counter() - tw
            if total_hits <= 2400:
                fr = renderer.last_frame if seq is not None else None
                out = renderer.render_fast(shard, frame=fr, small_frame_rays=1 << 30)
                local = torch.cat([out["rgb_map"][0], out["acc_map"][0][:, None], out["depth_map"][0][:, None]], dim=1)
                img = gatherer(local)
        else:
            img = torch.zeros((R, 5), dtype=local.dtype, device=dev)
            img[my_idx] = local
        return img, dict(renderer.last_stats)

    for _ in range(args.warmup):
        step()
    hip.profile_enable(os.environ.get("TH_NO_PROF") != "1")
    hip.profile_read()
    # shader-clock probes: one 15 us wave per timed step (up to 64) on a stream of its own -- it starts as soon as a CU
    # has room, i.e. beside whatever the device is running when the host queues it (the fused MLP, 85 % of the time): the
    # clock the DVFS governor holds UNDER the load, not the one it jumps to when the queue drains
    n_probe = min(args.steps, 64) if os.environ.get("TH_NO_PROF") != "1" else 0
    clk = torch.zeros((max(n_probe, 1), 4), dtype=torch.int64, device=dev)
    probe_stream = torch.cuda.Stream(dev)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    hip.host_wait_read(dev)
    dist_wait[0] = 0.0
    # per-step device time: one event behind every step on the shading stream (the frame pipeline hands out frame i when frame
    # i + 1 is queued: consecutive events are one steady-state frame apart)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        img, stats = step()
        marks[i + 1].record()
        if i < n_probe:
            with torch.cuda.stream(probe_stream):
                hip.clock_probe(clk[i])
    host_dt = time.perf_counter() - t0             # the host is done queueing here (the device may still be working)
    # ... of which it spent this long blocked on counts / guard snapshots / the whole-frame hit count of a multi-rank job
    host_wait_ms = hip.host_wait_read(dev) + dist_wait[0] * 1e3
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0 and os.environ.get("TH_SAVE_IMAGE"):     # developer aid: the last timed frame [R, 5] (rgb, acc, depth)
        import numpy as _np
        _np.save(os.environ["TH_SAVE_IMAGE"], img.detach().cpu().numpy())
    prof = hip.profile_read()
    hip.profile_enable(False)
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]) if args.steps > 0 else np.zeros(1)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    # algorithmic work of the dominant stage on this rank (counts from the rendered frames themselves: the mean over the
    # alternating frames)
    per_frame = []
    for k, sh in enumerate(frames):
        st_k = stats_of.get(k, stats)
        pts = hip.Points(sh["ray_o"][0], sh["ray_d"][0], sh["near"][0], sh["far"][0], args.samples)
        frame = renderer.prepare_frame(batches[k])
        nv_k = st_k["valid_samples"]
        per_frame.append({"hit_rays": int(st_k["hit_rays"]), "valid_samples": int(nv_k),
                          "sigma_pos_samples": int(count_sigma_positive(hip, net, frame, pts) if st_k["unmasked"] == 0 else nv_k)})
    n_valid = int(round(sum(f["valid_samples"] for f in per_frame) / len(per_frame)))
    n_pos = int(round(sum(f["sigma_pos_samples"] for f in per_frame) / len(per_frame)))
    # the fused kernel from inside (th_fused_cycles): cycles per phase and per tile, and the shader clock UNDER the kernel, over
    # one un-pipelined frame of each kind after the timed region
    fused_inside = None
    if args.mlp_mode == 1 and world == 1 and os.environ.get("TH_NO_PROF") != "1":
        cnt = torch.zeros(64, dtype=torch.int64, device=dev)
        hip.fused_cycles(cnt)
        try:
            for sh in frames:
                renderer.render_fast(sh, small_frame_rays=-1 if sharded else 2400)
            torch.cuda.synchronize()
        finally:
            hip.fused_cycles(None)
        cc = cnt.cpu().numpy().astype(np.float64)
        if cc[0] > 0 and cc[63] > 0:
            fused_inside = {"sampled_tiles": int(cc[0]), "cycles_per_tile": cc[62] / cc[0], "us_per_tile": cc[63] / cc[0] * 0.01,
                            "clock_GHz_inside_the_launch": cc[62] / (cc[63] * 10.0),
                            "cycles_between_barriers": [round(x / cc[0]) for x in cc[1:62] if x > 0],
                            "mfma_cycles_per_tile_and_simd": 49600.0 if hip.fused_waves(dev) == 8 else 47800.0,
                            "waves_per_workgroup": hip.fused_waves(dev),
                            "note": "thread 0 of every 16th tile, one frame of each kind through render_fast after the timed region; "
                                    "mfma cycles = the tile's MFMA instructions x their issue interval on one SIMD (2 988 x 16.6 / 1 494 x 32)"}
    mlp_ms, mlp_launches = prof["mlp"]
    flops_step = algorithmic_mlp_flops(V, n_valid, n_pos)
    mlp_s_per_step = mlp_ms * 1e-3 / max(args.steps, 1)
    achieved = flops_step / max(mlp_s_per_step, 1e-12)

    if rank == 0:
        res = {
            "metric": "rays/sec (512x512, 64 samples/ray)",
            "value": R * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # spread over the timed steps (HIP events on the shading stream, one per step; rank 0)
            "ms_per_step_min": float(step_ms.min()), "ms_per_step_median": float(np.median(step_ms)),
            "ms_per_step_max": float(step_ms.max()),
            # rays/s depends on the synthetic frame's hit rate (19.5 % of the rays hit the hull here); shaded samples/s does not --
            # the all-valid S_dense_full frame of `extra` runs at the same samples/s and 1/8 of the rays/s
            "valid_samples_per_s_rank0": float(n_valid) * args.steps / dt,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (fp16 hi/lo x3 MFMA emulation)" if args.mlp_mode == 1 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"S-{args.workload}: synthetic capsule body (6890 verts), {H}x{W} rays, {args.samples} "
                            f"samples/ray, V={V} reference views, kmeans N_c={args.nc}, ViT depth 12 "
                            f"(BASELINE.json configs[1] analogue); step = encoder + paint/group + TransHE + "
                            f"hull mask + DPaRF + pixel gather + MLP + compositing",
                "frames_alternating": per_frame,
                "rays": R, "hit_rays_rank0": stats["hit_rays"], "valid_samples_rank0": n_valid,
                "sigma_pos_samples_rank0": n_pos, "parallelism": f"ray-tile x{world}" if world > 1 else "single",
                "frame_pipeline": "off" if seq is None else "constants(i+1) on a 2nd HIP stream under shading(i)",
                "stem_exchange": stem_x is not None,
                "map_crop": os.environ.get("TH_MAP_CROP") != "0",      # pixel map written inside the hull's texel box only
            },
            "roofline": roofline_block(args.mlp_mode, achieved, flops_step, mlp_ms / max(args.steps, 1),
                                       mlp_launches / max(args.steps, 1), executed_mlp_flops(V, n_valid, n_pos, map_fold=args.mlp_mode == 1 and hip.tex_rows_enabled(dev)), n_valid=n_valid,
                                       folded_flops=(2.0 * V * (98304.0 * n_valid + 147456.0 * n_pos)) if (args.mlp_mode == 1 and hip.tex_rows_enabled(dev)) else 0.0,
                                       fold_ms=prof["fold"][0] / max(args.steps, 1)),
            "gather": (texel_handover_block if hip.tex_rows_enabled(dev) else gather_block)(V, n_valid, prof["gather"][0] / max(args.steps, 1)),
            # what the range guard of the fp16 hi/lo split has switched on this device (every entry false = the fast paths
            # ran; a tripped MLP guard means per-layer fp32 launches, ~7x slower frames) + the last table read (fp16 bit
            # patterns of max |x| per split tensor: f, s, p, n, inter, fc4_in; fp32 bits: conv_in; fp16: vit_in)
            "range_guard": dict(hip.guard_state(dev), fallback=bool(hip.guard_state(dev)["mlp_fp32_fallback"]),
                                slots=[int(x) for x in (hip.last_range or [])]),
            "host_queue_ms_per_step": host_dt / max(args.steps, 1) * 1e3,      # wall time of the queueing loop (incl. its blocking waits)
            # the host's OWN cost per frame: the loop's wall time minus the time it sat in blocking waits (sample counts,
            # range-guard snapshots: back-pressure from the device, th_host_wait_read) -- Python + ctypes + launch calls.
            # A frame shorter than this is host-bound whatever the kernels do.
            "host_pure_ms_per_step": (host_dt * 1e3 - host_wait_ms) / max(args.steps, 1),
            "host_wait_ms_per_step": host_wait_ms / max(args.steps, 1),
            "shader_clock_GHz": shader_clock(clk, n_probe),
            "stage_ms_per_step": {k: v[0] / max(args.steps, 1) for k, v in prof.items()},
            "stage_note": "HIP-event spans per stage; with the frame pipeline hull / fold run on the side stream under the "
                          "other stages (their spans are stretched by the overlap and do not add to the frame time); vit = 0: "
                          "TransHE is a replayed hipGraph in the frame paths, its launches carry no stage events",
        }
        if fused_inside is not None:
            # the roofline priced at the clock the chip holds under the kernel (the peaks above assume 2.4 GHz)
            g = fused_inside["clock_GHz_inside_the_launch"]
            rf = res["roofline"]
            rf["clock_GHz_inside_the_launch"] = g
            rf["peak_at_launch_clock"] = rf["peak"] * g / 2.4
            rf["frac_at_launch_clock"] = rf["frac"] * 2.4 / g
            if rf.get("frac_executed") is not None:
                rf["frac_executed_at_launch_clock"] = rf["frac_executed"] * 2.4 / g
            fused_inside["mfma_busy_by_cycles"] = fused_inside["mfma_cycles_per_tile_and_simd"] / fused_inside["cycles_per_tile"]
            res["fused_kernel_inside"] = fused_inside
        res["peak_device_GiB"] = {"headline_allocated": torch.cuda.max_memory_allocated(dev) / 2**30,
                                  "headline_reserved": torch.cuda.max_memory_reserved(dev) / 2**30}
        if emu:
            res["config"]["emulated_rank0_of"] = emu
        if world == 1 and not emu:
            # the reference's own call pattern (run.py:96-118 calls renderer.render_fast per item, no look-ahead):
            # measured after the timed region on the same device
            ms_rf, _ = time_steps(lambda: renderer.render_fast(shard), max(3, args.steps // 2), warmup=1)
            res["render_fast_ms_per_step"] = ms_rf
            res["render_fast_rays_per_s"] = R / ms_rf * 1e3
            # the number a caller of the UNCHANGED run.py gets (run.py:52,109 call render_fast per frame; `value` above is
            # Renderer.render_sequence, the frame pipeline a video / evaluation loop can opt into)
            res["dropin_ms_per_step"] = ms_rf
            res["dropin_rays_per_s"] = R / ms_rf * 1e3
            if args.mlp_mode == 1 and not args.no_extras:
                res["fused_vs_fp32_full_frame"] = fused_vs_fp32(renderer, batch, hip)
        if world == 1 and not args.no_cpu_baseline and not emu:
            res["cpu_baseline"] = cpu_baseline(batch_cpu, assign, args.samples, stride=args.cpu_stride, gpu_img=img)
        if world == 1 and not args.no_extras and not args.no_cpu_baseline and not emu and args.workload == "real":
            try:        # secondary numbers must never cost the headline line
                res["extra"] = run_extras(dev, net, args, H, W, V)
            except Exception as e:        # noqa: BLE001
                res["extra"] = {"error": f"{type(e).__name__}: {e}"}
            cfg.num_class = args.nc
            res["peak_device_GiB"].update(with_extras_allocated=torch.cuda.max_memory_allocated(dev) / 2**30,
                                          with_extras_reserved=torch.cuda.max_memory_reserved(dev) / 2**30)
    else:
        res = None
    finish(dist_on, res)


def shader_clock(clk, n):
    """median / min / max GHz of the probes queued behind the timed steps (th_clock_probe), None without probes"""
    if n <= 0:
        return None
    c = clk[:n].cpu().double()
    ok = c[:, 1] > 0
    if not bool(ok.any()):
        return None
    g = (c[ok, 0] / (10.0 * c[ok, 1])).numpy()
    return {"median": float(np.median(g)), "min": float(g.min()), "max": float(g.max()), "probes": int(ok.sum()),
            "note": "s_memtime ticks / s_memrealtime (100 MHz) of a one-wave probe per timed step on its own stream (runs beside "
                    "the frame's kernels: the clock under load)"}


def finish(dist_on, res):
    """Tear the process group down first, flush whatever the C libraries (RCCL's version banner) still hold in
    their stdio buffers, and only then print the ONE JSON line, so that it is the last line on stdout."""
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if res is not None and os.environ.get("TH_BENCH_MEM") == "1":      # developer aid: peak device memory of the job
        print(f"[bench] peak device memory: allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, "
              f"reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB", file=sys.stderr)
    if res is not None:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


def run_secondary(args, world, rank, dev, dist_on, renderer, net, batch, batch_cpu, H, W, V):
    """The two non-headline workloads of SURVEY 8d (their own metrics; no roofline / cpu_baseline blocks)."""
    import math
    from transhuman_amd import hip
    if dist_on:
        import torch.distributed as dist
    verts = batch_cpu["tar_smpl_vertice"][0].numpy()
    if args.workload == "orbit":
        # C3: 60-view orbit around the body (run.py --type visualize): per step a new target camera -> rays on
        # device (th_gen_rays, the reference's get_rays + get_near_far) -> pixel tiles dealt to the ranks -> render ->
        # one all_gather.
        bounds = np.stack([verts.min(0), verts.max(0)]).astype(np.float32)
        bounds[0, 2] -= 0.05; bounds[1, 2] += 0.05                                   # can_smpl.py:228-230
        centre = 0.5 * (bounds[0] + bounds[1]).astype(np.float64)
        K = np.array([[600.0 * W / 512, 0, W / 2], [0, 600.0 * W / 512, H / 2], [0, 0, 1]], np.float32)
        n_views = 60
        # the reference's virtual camera path (render_utils.py:318-364 gen_path_virt, as can_smpl_perform.py:40-42
        # builds it once per sequence from the capture rig; frame i uses render_w2c[i % n_views], :68-70) over the
        # synthetic 21-camera rig
        from transhuman_amd.camera_path import gen_path_virt, synthetic_rig
        render_w2c = gen_path_virt(synthetic_rig(centre=tuple(centre.tolist())), render_views=n_views)

        def camera(i):
            RT = render_w2c[i % n_views]
            return K, RT[:3, :3].astype(np.float32), RT[:3, 3:].astype(np.float32)

        # every pixel is a ray (rays that miss the body box get near = far = 0 from K9: no sample can pass the hull test,
        # they come out as background exactly like the pixels the reference never renders): no per-frame compaction of
        # the ray list, hence no host round trip and no index arithmetic per frame; pixel tiles are dealt to the ranks
        # like in the headline workload, layout exchanged once
        # (also on one GPU: 8 x 8 pixel tiles make 16 consecutive rays an 8 x 2 block -- the unit of the sample list)
        my_px = shard_ray_indices(H, W, world, rank, tile=8, tile_major=True).to(dev)
        gatherer = ImageGatherer(my_px, H * W, world) if dist_on and world > 1 else None

        def frames():
            """the video: one batch per target camera (made under render_sequence's side stream: ray generation
            overlaps the shading of the previous frame)"""
            i = 0
            while True:
                rays = hip.gen_rays(*camera(i), bounds, H, W, device=dev, compact=False)
                sh = dict(batch)
                for k in ("ray_o", "ray_d", "near", "far"):
                    sh[k] = rays[k][my_px][None]
                yield sh
                i += 1

        seq = renderer.render_sequence(frames(), small_frame_rays=-1 if world > 1 else 2400)

        def step(i):
            out = next(seq)
            img = torch.cat([out["rgb_map"][0], out["acc_map"][0][:, None], out["depth_map"][0][:, None]], dim=1)
            if gatherer is not None:
                img = gatherer(img)
            else:
                full = torch.empty((H * W, 5), dtype=img.dtype, device=dev)
                full[my_px] = img
                img = full
            return img, H * W
        units, unit_name = H * W, "rays/sec (512x512 orbit, 64 samples/ray, rays generated on device)"
    else:
        # C5: mesh extraction grid (if_mesh_renderer.py:46-100): sigma on grid^3 voxel centres over the body box,
        # voxels dealt to the ranks in contiguous 4096-voxel runs
        from transhuman_amd.networks.renderer.if_mesh_renderer import Renderer as MeshRenderer
        mr = MeshRenderer(net, vertex_can=renderer.vertex_can.numpy(), pc2voxel_ind=renderer.pc2voxel_ind.cpu().numpy())
        g = args.grid
        mb = dict(batch_cpu)
        mb["pts"] = synth.make_grid_pts(batch_cpu, g)
        mb = synth.batch_to(mb, dev)
        nvox = g * g * g
        mine = torch.arange(nvox, device=dev)
        if world > 1:
            # run r of 4096 voxels; the runs of one x-slice (g*g/4096 of them) are shifted from slice to slice so that
            # a rank does not own the same y-band of every slice (see dist.shard_ray_indices)
            from transhuman_amd.dist import _tile_skew
            run = mine // 4096
            per_slice = max(1, (g * g) // 4096)
            mine = mine[((run % per_slice + _tile_skew(world) * (run // per_slice)) % world) == rank]

        def step(i):
            out = mr.render(mb, pts_slice=mine)
            sig = out["sigma"]
            if dist_on:
                from transhuman_amd.dist import gather_image
                sig = gather_image(sig[:, None], mine, nvox, world)[:, 0]
            return sig, nvox
        units, unit_name = nvox, f"voxels/sec ({g}^3 sigma grid)"

    for i in range(args.warmup):
        step(i)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res, n_last = step(args.warmup + i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0 and os.environ.get("TH_SAVE_IMAGE"):     # developer aid: the last frame [R, 5] / sigma grid [grid^3]
        np.save(os.environ["TH_SAVE_IMAGE"], res.detach().cpu().numpy())
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    res = None
    if rank == 0:
        res = {
            "metric": unit_name, "value": units * args.steps / dt, "unit": unit_name.split(" ")[0].replace("sec", "s"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"S-{args.workload} (SURVEY 8d {'C3' if args.workload == 'orbit' else 'C5'}), V={V}, "
                                   f"N_c={args.nc}", "units_last_step": int(n_last),
                       "stats": {k: int(v) for k, v in (renderer.last_stats if args.workload == 'orbit' else mr.last_stats).items()},
                       "parallelism": f"x{world}" if world > 1 else "single"}}
    finish(dist_on, res)


def count_sigma_positive(hip, net, frame, pts):
    """# of valid samples with sigma_raw > 0 (the RGB branch's algorithmic work): read it off the
    mesh-style sigma evaluation of the same samples."""
    o, d = pts.ray_o, pts.ray_d
    z = pts.near[:, None] * pts.omt[None, :] + pts.far[:, None] * pts.t[None, :]
    p = o[:, None, :] + d[:, None, :] * z[..., None]
    n = 0
    flat = p.reshape(-1, 3)
    step = 1 << 22
    for s in range(0, flat.shape[0], step):
        sig, _ = hip.eval_sigma_grid(net, frame, flat[s:s + step])
        n += int((sig > 0).sum())
    return n


if __name__ == "__main__":
    main()
