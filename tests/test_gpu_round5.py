"""GPU tests added in round 5.

* Every ray of the headline frame against the oracle evaluated ON THE DEVICE in fp32 (torch-ROCm's kernels: what the reference
  itself computes on this GPU) and in float64 (the exact value of the same graph), tied to the host oracle on a subset: the
  tail of the error distribution over all 262 144 rays, not a 1024-ray sample (profiles/r05_a_tail_*.json hold the off-line
  run of tools/dense_tail.py, S-dense included).
* BASELINE configs[0] as written: V = 1, 128 x 128 rays, 32 samples per ray, the reference's own kmeans_dict_300.
Everything goes through the C ABI (transhuman_amd.hip)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import th_oracle as O
from transhuman_amd import synth
from util import make_sd, make_net, synth_assign, real_assign, csr, can_centres64, can64, maxdiff

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu
BAR = 1e-4


@pytest.fixture(scope="module")
def hip(gpu):
    from transhuman_amd import hip as H
    H.load_library()
    return H


def _renderer(net, nc, samples, assign):
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_clight_renderer
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = samples, nc
    return if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=assign)


def test_headline_frame_every_ray_against_the_oracle(hip, gpu):
    """|gpu - oracle| over ALL rays of the 512 x 512 x 64 S-real frame.  The bar is 1e-4 on rgb / alpha; a ray may exceed it
    only where the reference's own algorithm is discontinuous and two fp32 evaluations land on different sides: the 7th and
    8th nearest token centre of a sample closer than fp32 resolves (the 7-NN SET changes), or sigma_raw within rounding of
    zero on a ray's last sample (delta = 1e10 turns it into alpha = 0 or 1).  Such rays are counted, bounded and shown to be
    exactly that."""
    import dense_tail as D
    net = make_net(12).to(gpu)
    assign = synth_assign(500)
    r = _renderer(net, 500, 64, assign)
    bc = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
    o = r.render_fast(synth.batch_to(bc, gpu), is_train=False)
    img = torch.cat([o["rgb_map"][0], o["acc_map"][0][:, None]], dim=1).double().cpu()
    assert r.last_stats["valid_samples"] > 1500000
    hip.drop_workspaces(gpu)
    sd = make_sd()
    t64 = D.oracle_frame(bc, sd, assign, gpu, torch.float64)
    o32 = D.oracle_frame(bc, sd, assign, gpu, torch.float32)
    # the device evaluations are the host oracle's (fp32: to its own rounding noise; float64: to 1e-12)
    rs = np.random.RandomState(5)
    hits = torch.nonzero(img[:, 3] > 0).reshape(-1).numpy()
    pick = np.sort(rs.choice(hits, 768, replace=False))
    c32 = D.oracle_frame(bc, sd, assign, torch.device("cpu"), torch.float32, pick=pick)
    c64 = D.oracle_frame(bc, sd, assign, torch.device("cpu"), torch.float64, pick=pick[::4])
    assert float(D.dist(t64[pick[::4]], c64).max()) < 1e-9
    g_host = D.dist(img[pick], c32)
    assert float(g_host.max()) < BAR, float(g_host.max())
    d32, d64, n64 = D.dist(img, o32), D.dist(img, t64), D.dist(o32, t64)
    k = int(round(d32.numel() * 0.9999))
    print(f"all rays: |gpu - o32| max {float(d32.max()):.3e} p99.99 {float(d32.kthvalue(k)[0]):.3e} over 1e-4: {int((d32 > BAR).sum())};"
          f" |gpu - t64| p99.99 {float(d64.kthvalue(k)[0]):.3e} over: {int((d64 > BAR).sum())};"
          f" |o32 - t64| p99.99 {float(n64.kthvalue(k)[0]):.3e} over: {int((n64 > BAR).sum())}; host subset max {float(g_host.max()):.3e}")
    assert float(d32.kthvalue(k)[0]) < 2e-5 and float(d64.kthvalue(k)[0]) < 2e-5
    # the HIP path is not further from the exact result than the reference's fp32 arithmetic is (99.99th percentiles)
    assert float(d64.kthvalue(k)[0]) <= float(n64.kthvalue(k)[0]) + 1e-5
    bad = torch.nonzero((d32 > BAR) | (d64 > BAR)).reshape(-1).numpy()
    assert len(bad) <= 8, len(bad)            # (observed: 3 - 5 rays of 262 144, rounds 5 and 6; the reference's own fp32 has 7 against the exact value)
    for rec in D.flips(bc, sd, assign, gpu, bad, img, o32, t64):
        tie = rec["min_gap_7th_8th_neighbour_over_valid_samples"] is not None and rec["min_gap_7th_8th_neighbour_over_valid_samples"] < 1e-6
        flip = len(rec["samples_sign_flip_o32_t64"]) > 0 or len(rec["samples_with_abs_sigma_raw_below_1e-4"]) > 0
        edge = rec["min_hull_margin_over_samples"] < 1e-6
        assert tie or flip or edge, rec


@pytest.mark.parametrize("focal,expect_unmasked", [(150.0, 0), (40.0, 1)])
def test_config0_as_written(hip, gpu, focal, expect_unmasked):
    """BASELINE.json configs[0]: ONE reference view, 128 x 128 rays, 32 samples per ray, the reference's kmeans_dict_300
    (tests/golden/kmeans_pc2voxel.npz) -- both sides of the R' <= 2400 switch (if_clight_renderer.py:551): the body filling the
    frame (masked branch) and a small body (un-masked branch: every sample of the hit rays shaded)."""
    from transhuman_amd.config import get_cfg
    net = make_net(12).to(gpu)
    assign = real_assign(300)
    r = _renderer(net, 300, 32, assign)
    bc = synth.make_batch(128, 128, 1, seed=0, all_rays=True, focal=focal)
    out = r.render_fast(synth.batch_to(bc, gpu), is_train=False)
    st = dict(r.last_stats)
    print("configs[0]:", st)
    assert st["unmasked"] == expect_unmasked and st["hit_rays"] > 100
    assert (st["hit_rays"] <= 2400) == bool(expect_unmasked)
    off, mem = csr(assign)
    sd = make_sd()
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    with torch.no_grad():
        hol, pix = O.encoder_forward(sd, bc["input_imgs"][0][0])
        ref, _ = O.render_fast(sd, bc, hol, pix, off, mem, can_centres64(assign), n_samples=32)
    assert int((ref["acc_map"][0] > 0).sum()) > 100
    d = max(maxdiff(out["rgb_map"].cpu(), ref["rgb_map"]), maxdiff(out["acc_map"].cpu(), ref["acc_map"]))
    dd = maxdiff(out["depth_map"].cpu(), ref["depth_map"])
    print(f"configs[0] focal {focal}: max |rgb, acc| {d:.3e}, depth {dd:.3e}")
    assert d < BAR and dd < 1e-3
    get_cfg().N_samples, get_cfg().num_class = 64, 500
    hip.drop_workspaces(gpu)


def _img(o):
    return torch.cat([o["rgb_map"][0], o["acc_map"][0][:, None], o["depth_map"][0][:, None]], dim=1)


def test_demand_driven_map_equals_the_cropped_map(hip, gpu, monkeypatch):
    """The map written only where the frame's valid samples / painted vertices read it (th_render_predemand, round 5) against the
    map written over the row spans of the hull's box (round 3 / 4): same image bit for bit, through render_fast and through the
    frame pipeline, for the whole frame and for a rank's shard (8 x 8 tiles, the demand of a rank of 8), both branches of the
    R' <= 2400 rule; and a frame built for one sample list, then used with other rays, completes its map first."""
    from transhuman_amd.dist import shard_ray_indices
    net = make_net(12).to(gpu)
    r = _renderer(net, 500, 64, synth_assign(500))
    for (res, focal) in ((256, 300.0), (128, 40.0)):          # masked branch / un-masked branch (<= 2400 hit rays)
        bc = synth.make_batch(res, res, 3, seed=0, all_rays=True, focal=focal)
        b = synth.batch_to(bc, gpu)
        my = shard_ray_indices(res, res, 8, 3, tile=8).to(gpu)
        sh = dict(b)
        for k in ("ray_o", "ray_d", "near", "far"):
            sh[k] = b[k][:, my].contiguous()
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("TH_MAP_DEMAND", mode)
            o_full = _img(r.render_fast(b, is_train=False))
            st = dict(r.last_stats)
            o_sh = _img(r.render_fast(sh, is_train=False, small_frame_rays=-1))
            seq = [_img(o) for o in r.render_sequence([b, sh, b], small_frame_rays=2400)]
            out[mode] = (o_full, o_sh, seq, st)
        assert out["1"][3]["hit_rays"] > 100 and (out["1"][3]["unmasked"] == 1) == (focal == 40.0)
        assert torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1])
        for a, c in zip(out["0"][2], out["1"][2]):
            assert torch.equal(a, c)
        assert float(out["1"][0][:, 3].max()) > 0.05
    # a demand-built frame handed to OTHER rays: the C side completes the map (th_map_source.demand) -- same pixels as a plain frame
    monkeypatch.setenv("TH_MAP_DEMAND", "1")
    bc = synth.make_batch(256, 256, 3, seed=0, all_rays=True, focal=300.0)
    b = synth.batch_to(bc, gpu)
    my = shard_ray_indices(256, 256, 8, 5, tile=8).to(gpu)
    sh = dict(b)
    for k in ("ray_o", "ray_d", "near", "far"):
        sh[k] = b[k][:, my].contiguous()
    pts = hip.Points(sh["ray_o"][0], sh["ray_d"][0], sh["near"][0], sh["far"][0], n_samples=64)
    hip.render_prepass(pts, b["tar_smpl_vertice"][0], 3, 0.1, -1, n_clusters=500)
    dm = r.predemand(b, pts)
    assert dm is not None
    frame = r.prepare_frame(b, demand=dm)
    assert frame.map.demand is not None
    o_other = _img(r.render_fast(b, is_train=False, frame=frame))            # all rays, not the shard the demand was made for
    monkeypatch.setenv("TH_MAP_DEMAND", "0")
    o_ref = _img(r.render_fast(b, is_train=False))
    assert torch.equal(o_other, o_ref)
    hip.drop_workspaces(gpu)


@pytest.mark.parametrize("tag", ["small", "large"])
def test_render_fast_with_depth_jitter_and_density_noise(hip, gpu, tag):
    """The reference's two sampling randomisations through the drop-in entry (VERDICT r4 "missing" #3): cfg.perturb = 1 with the
    network in train() mode (stratified depth jitter, if_clight_renderer.py:276-283) and cfg.raw_noise_std > 0 (density noise,
    nerf_net_utils.py:39-44), on the draws the REAL reference took (tests/golden/g19_perturb_*.npz, oracle/gen_golden_perturb.py)
    handed in through the batch: every stage reads the jittered depths (hull test, neighbour records, texel lists, deltas),
    the compositing adds the noise on the rays the reference composites.  'small': R' <= 2400 (un-masked), 'large': masked."""
    from util import gold
    from transhuman_amd.config import get_cfg
    g = gold(f"g19_perturb_{tag}")
    H, S, focal, std = int(g["H"]), int(g["n_samples"]), float(g["focal"]), float(g["noise_std"])
    cfg = get_cfg()
    net = make_net(12).to(gpu)
    net.train()
    r = _renderer(net, 300, S, synth_assign(300))
    b = synth.batch_to(synth.make_batch(H, H, 3, seed=0, focal=None if focal < 0 else focal), gpu)
    hit = torch.as_tensor(np.asarray(g["hit"])).bool()
    try:
        plain = {k: v.clone() for k, v in r.render_fast(b).items()}
        cfg.perturb, cfg.raw_noise_std = 1.0, std
        bj = dict(b, t_rand=g["t_rand"][None].to(gpu), raw_noise=g["raw_noise"][None].to(gpu))
        o = r.render_fast(bj)
        assert r.last_stats["hit_rays"] == int(hit.sum())
        assert (r.last_stats["hit_rays"] > 2400) == (tag == "large")
        assert maxdiff(o["rgb_map"][0].cpu(), g["rgb"]) < BAR
        assert maxdiff(o["acc_map"][0].cpu(), g["acc"]) < BAR
        assert maxdiff(o["depth_map"][0].cpu(), g["depth"]) < 1e-3
        assert maxdiff(o["rgb_map"][0].cpu(), plain["rgb_map"][0].cpu()) > 1e-3          # (the randomisations are live)
        # the frame pipeline serves them too, frame by frame
        seq = list(r.render_sequence(iter([bj, bj])))
        for q in seq:
            for k in ("rgb_map", "acc_map", "depth_map"):
                assert torch.equal(q[k], o[k]), k
        # drawn on the device when the batch carries no draws: two calls differ, eval() mode switches the jitter off
        cfg.raw_noise_std = 0.0
        a1, a2 = r.render_fast(b)["rgb_map"].clone(), r.render_fast(b)["rgb_map"].clone()
        assert maxdiff(a1.cpu(), a2.cpu()) > 1e-4
        net.eval()
        e1 = r.render_fast(b)["rgb_map"].clone()
        net.train()
        cfg.perturb = 0.0
        e2 = r.render_fast(b)["rgb_map"].clone()
        assert maxdiff(e2.cpu(), plain["rgb_map"].cpu()) == 0.0
    finally:
        cfg.perturb, cfg.raw_noise_std, cfg.N_samples = 0.0, 0.0, 64
        net.train()


def test_graph_capture_failure_falls_back_to_separate_launches(hip, gpu, monkeypatch):
    """The frame paths replay the stem and TransHE as hipGraphs; a capture that fails (another thread's API call under a global
    capture mode, a runtime that refuses) must leave a renderer that works: the forms are switched off for the process and the
    frame is rendered through the separate launches -- same image, bit for bit."""
    import warnings
    from transhuman_amd.config import get_cfg
    net = make_net(12).to(gpu)
    r = _renderer(net, 300, 32, synth_assign(300))
    b = synth.batch_to(synth.make_batch(64, 64, 3, seed=1, focal=200.0), gpu)
    try:
        with torch.no_grad():
            ref = {k: v.clone() for k, v in r.render_fast(b).items()}
            ref2 = {k: v.clone() for k, v in r.render_fast(b).items()}          # (second call: replayed graphs)
            for k in ref:
                assert torch.equal(ref[k], ref2[k]), k
            # a fresh network: new parameter storage, so both forms capture again -- and this time the capture fails
            net2 = make_net(12).to(gpu)
            r2 = _renderer(net2, 300, 32, synth_assign(300))
            real = torch.cuda.CUDAGraph.capture_begin

            def refuse(self, *a, **k):
                raise RuntimeError("capture refused (test)")
            monkeypatch.setattr(torch.cuda.CUDAGraph, "capture_begin", refuse)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                o1 = {k: v.clone() for k, v in r2.render_fast(b).items()}
            assert any("hipGraph capture failed" in str(x.message) for x in w)
            assert not hip.graphs_enabled()
            monkeypatch.setattr(torch.cuda.CUDAGraph, "capture_begin", real)
            o2 = r2.render_fast(b)
            for k in ref:
                assert torch.equal(o1[k], ref[k]) and torch.equal(o2[k], ref[k]), k
    finally:
        hip._graphs_off[0] = False
        get_cfg().N_samples = 64


def test_graphs_back_off_when_the_weights_change_with_every_frame(hip, gpu):
    """rendering inside a training loop: the weights change between every two frames, so a captured graph would never be replayed
    -- after three such captures the stem and TransHE stay on separate launches (no 70 ms capture per frame), the images stay
    right, and once the weights have settled the graphs come back"""
    from transhuman_amd.config import get_cfg
    net = make_net(12).to(gpu)
    r = _renderer(net, 300, 32, synth_assign(300))
    b = synth.batch_to(synth.make_batch(48, 48, 3, seed=0, focal=150.0), gpu)
    real = torch.cuda.CUDAGraph.capture_begin
    n_cap = [0]

    def counting(self, *a, **k):
        n_cap[0] += 1
        return real(self, *a, **k)
    torch.cuda.CUDAGraph.capture_begin = counting
    try:
        with torch.no_grad():
            per_frame = []
            for i in range(7):
                net.encoder.model.conv1.weight.mul_(1.0)            # (a no-op update: the version counter moves, the values do not)
                net.ViT.norm.weight.mul_(1.0)
                before = n_cap[0]
                o = r.render_fast(b)
                per_frame.append(n_cap[0] - before)
            assert per_frame[0] > 0 and sum(per_frame[4:]) == 0, per_frame        # backed off
            ref = {k: v.clone() for k, v in o.items()}
            for i in range(4):                                                    # settled: captured once more, then replayed
                before = n_cap[0]
                o = r.render_fast(b)
                per_frame.append(n_cap[0] - before)
                for k in ref:
                    assert torch.equal(o[k], ref[k]), (i, k)
            assert sum(per_frame[7:]) > 0 and per_frame[-1] == 0, per_frame
    finally:
        torch.cuda.CUDAGraph.capture_begin = real
        get_cfg().N_samples = 64
