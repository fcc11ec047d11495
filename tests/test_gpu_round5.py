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
    assert len(bad) <= 32, len(bad)
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
