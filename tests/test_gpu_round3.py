"""GPU tests added in round 3: the CPU oracle on the BASELINE configs[1] frame at its own size and on the S-dense
frame (every sample valid), the range guard across a frame pipeline (stem-convolution overflow seen one frame early),
the snapshot ring, and the training entry.  Everything goes through the C ABI (transhuman_amd.hip)."""
import itertools
import warnings

import numpy as np
import pytest
import torch

from oracle import th_oracle as O
from transhuman_amd import synth
from util import make_sd, make_net, synth_assign, csr, can_centres64, can64, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(gpu):
    from transhuman_amd import hip as H
    H.load_library()
    return H


@pytest.fixture(scope="module")
def net(gpu, hip):
    return make_net(12).to(gpu)


def _renderer(net, nc, samples=64):
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_clight_renderer
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = samples, nc
    return if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(nc))


def _oracle_on(bc, pick, assign, samples=64):
    sd = make_sd()
    off, mem = csr(assign)
    sub = dict(bc)
    for k in ("ray_o", "ray_d", "near", "far"):
        sub[k] = bc[k][:, pick]
    with torch.no_grad():
        hol, pix = O.encoder_forward(sd, bc["input_imgs"][0][0])
        ref, _ = O.render_fast(sd, sub, hol, pix, off, mem, can_centres64(assign), n_samples=samples, small_frame_rays=-1)
    return ref


def test_headline_frame_oracle_sample(hip, gpu, net):
    """BASELINE.json configs[1] at its own size (512 x 512 rays x 64 samples, V = 3, N_c = 500: the frame bench.py
    times): the CPU oracle on 96 of its rays (64 of them hits), rgb / acc within 1e-4, through render_fast AND through
    the frame pipeline bench.py's timed loop uses."""
    r = _renderer(net, 500)
    bc = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
    b = synth.batch_to(bc, gpu)
    out = r.render_fast(b, is_train=False)
    st = dict(r.last_stats)
    rgb, acc = out["rgb_map"][0], out["acc_map"][0]
    assert st["hit_rays"] > 30000 and st["valid_samples"] > 1500000 and st["unmasked"] == 0
    seq = r.render_sequence(itertools.repeat(b))
    next(seq)
    piped = next(seq)
    seq.close()
    assert torch.equal(piped["rgb_map"], out["rgb_map"]) and torch.equal(piped["acc_map"], out["acc_map"])
    rs = np.random.RandomState(5)
    hits = torch.nonzero(acc > 0).reshape(-1).cpu().numpy()
    pick = np.sort(np.concatenate([rs.choice(hits, 64, replace=False), rs.choice(512 * 512, 32, replace=False)]))
    ref = _oracle_on(bc, pick, synth_assign(500))
    assert maxdiff(rgb[pick].cpu(), ref["rgb_map"][0]) < 1e-4 and maxdiff(acc[pick].cpu(), ref["acc_map"][0]) < 1e-4
    assert float(ref["acc_map"].max()) > 0.05
    assert not any(v for k, v in hip.guard_state(gpu).items() if k != "epoch")


def test_pipeline_rebuilds_frames_built_before_a_conv_fallback(hip, gpu):
    """Range guard across render_sequence: the stem convolutions' range slot is sticky and written by the side stream
    one or two frames AHEAD, so the overflow of a coming frame is first seen in the snapshot of an earlier one.  Every
    frame whose constants were built before the switch to the stock convolutions must be rebuilt (epoch of its front),
    not only the one whose snapshot showed it: the pipeline's frames equal render_fast's on the fallen-back context."""
    net2 = make_net(12).to(gpu)                       # its own module: the fallback is tied to the uploaded weights
    r = _renderer(net2, 300, samples=32)
    bc = synth.make_batch(64, 64, 3, seed=0, focal=210.0)
    big = dict(bc)
    big["input_imgs"] = [t * 1.0e5 for t in bc["input_imgs"]]   # |x| > 65504 in front of conv1: inf in the fp16 hi planes
    b = synth.batch_to(big, gpu)
    assert not hip.conv_fallback(gpu)
    e0 = hip.range_epoch(gpu)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        seq = r.render_sequence(itertools.repeat(b))
        frames = [next(seq) for _ in range(4)]
        seq.close()
    assert hip.conv_fallback(gpu) and hip.range_epoch(gpu) > e0
    assert any("convolution" in str(x.message) for x in w)
    ref = r.render_fast(b, is_train=False)             # stock convolutions now
    assert torch.isfinite(ref["rgb_map"]).all()
    # (the fallen-back frames go through torch's stock convolutions: MIOpen on two streams is not bit-reproducible from call
    # to call -- one frame in ~50 differs from render_fast's by 1e-15 -- so this comparison is to 1e-6, not torch.equal)
    for f in frames:
        assert torch.isfinite(f["rgb_map"]).all()
        assert maxdiff(f["rgb_map"], ref["rgb_map"]) < 1e-6 and maxdiff(f["acc_map"], ref["acc_map"]) < 1e-6
    # new weights bring the HIP convolutions back (and clear the sticky slot)
    with torch.no_grad():
        net2.alpha_fc.bias.add_(0.0)
    small = synth.batch_to(bc, gpu)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = r.render_fast(small, is_train=False)
    assert not hip.conv_fallback(gpu) and torch.isfinite(out["rgb_map"]).all()
    assert not any(v for k, v in hip.guard_state(gpu).items() if k != "epoch")


def test_stale_range_snapshot_is_reported(hip, gpu, net):
    """The snapshot ring holds the 8 most recent snapshots: an id held across more guarded calls than that is reported
    as overwritten (range_read -> None, the guard answers 'render again') instead of returning another call's maxima."""
    r = _renderer(net, 300, samples=32)
    b = synth.batch_to(synth.make_batch(32, 32, 3, seed=0), gpu)
    frame = r.prepare_frame(b)
    pts = hip.Points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], n_samples=32)
    rgb, acc, dep, st, check = hip.render_rays(net, frame, pts, defer_guard=True)
    assert check() is True
    rgb, acc, dep, st, check = hip.render_rays(net, frame, pts, defer_guard=True)
    for _ in range(9):
        hip.render_rays(net, frame, pts)
    assert check() is False                            # overwritten: unchecked -> the caller renders again
    assert not any(v for k, v in hip.guard_state(gpu).items() if k != "epoch")   # ... without switching any path


def test_training_entry_serves_autograd(hip, gpu, net):
    """Renderer.render is the reference trainer's entry (if_nerf_clight.py:45): with gradients enabled it runs the
    differentiable form (transhuman_amd.networks.autograd_path; round 3 refused the call) and the result equals the HIP
    path's under no_grad within the parity bar."""
    r = _renderer(net, 300, samples=32)
    b = synth.batch_to(synth.make_batch(16, 16, 3, seed=0), gpu)
    assert any(p.requires_grad for p in net.parameters())
    out = r.render(b)
    assert out["rgb_map"].requires_grad and out["rgb_map"].shape == (1, 256, 3)
    out["rgb_map"].sum().backward()
    assert net.fc_0.weight.grad is not None and torch.isfinite(net.fc_0.weight.grad).all()
    net.zero_grad(set_to_none=True)
    with torch.no_grad():
        fast = r.render(b)
    assert not fast["rgb_map"].requires_grad and torch.isfinite(fast["rgb_map"]).all()
    assert maxdiff(fast["rgb_map"], out["rgb_map"].detach()) < 1e-4 and maxdiff(fast["acc_map"], out["acc_map"].detach()) < 1e-4


def _texel_use(b, frame, pts_world, H, W):
    """per view the (x, y) texel indices a bilinear gather at pts_world reads (torch restatement of grid_sample's
    align_corners=True / border coordinates, fp64)"""
    R, T, K = (b[k][0][0].double().cpu() for k in ("input_R", "input_T", "input_K"))
    sc = frame.scale.double().cpu()
    p = pts_world.double().cpu()
    out = []
    for v in range(R.shape[0]):
        cam = p @ R[v].T + T[v].reshape(1, 3)
        uvw = cam @ K[v].T
        u, w = uvw[:, 0] / uvw[:, 2], uvw[:, 1] / uvw[:, 2]
        ix = (u * sc[0] / 2 * (W - 1)).clamp(0, W - 1)
        iy = (w * sc[1] / 2 * (H - 1)).clamp(0, H - 1)
        out.append((ix.floor(), (ix.floor() + 1).clamp(max=W - 1), iy.floor(), (iy.floor() + 1).clamp(max=H - 1)))
    return out


@pytest.mark.parametrize("focal,hw", [(600.0, (512, 512)), (150.0, (64, 48))])
def test_cropped_map_holds_every_texel_the_frame_reads(hip, gpu, net, focal, hw):
    """prepare_frame writes only the per-view texel box within reach of the hull (th_map_box).  (1) inside the box the map
    equals the complete map bit for bit; (2) the box contains every texel read by the gather of every hull-valid sample
    of a full frame and by the painting of the input vertices; (3) tokens and images are the complete map's."""
    r = _renderer(net, 300, samples=32 if hw[0] < 512 else 64)
    H, W = hw
    bc = synth.make_batch(H, W, 3, seed=0, all_rays=True, focal=focal)
    b = synth.batch_to(bc, gpu)
    f_crop = r.prepare_frame(b)
    g_crop = r.last_grouped.clone()
    f_full = r.prepare_frame(b, crop_map=False)
    assert f_crop.map.box is not None and f_full.map.box is None and torch.equal(g_crop, r.last_grouped)
    assert torch.equal(f_crop.tokens, f_full.tokens)
    box = f_crop.map.box.cpu().numpy()
    spans = hip.map_spans(f_crop.map.box, H).cpu()                  # [V,H,2]: the rows' own spans inside the box (round 4)
    lat_c, lat_f, rgb_c, rgb_f = f_crop.map.latents, f_full.map.latents, f_crop.map.rgb0, f_full.map.rgb0
    area, span_area = 0, 0
    xs = torch.arange(W)[None, :]
    for v in range(3):
        x0, y0, x1, y1 = (int(t) for t in box[v])
        assert 0 <= x0 <= x1 <= W - 1 and 0 <= y0 <= y1 <= H - 1
        inside = (xs >= spans[v, :, 0:1]) & (xs <= spans[v, :, 1:2])            # [H,W]
        inside[:y0] = False
        inside[y1 + 1:] = False
        assert bool(inside.any()) and bool((inside[:, :x0] == False).all()) and bool((inside[:, x1 + 1:] == False).all())
        m = inside.to(gpu)
        assert torch.equal(lat_c[v][m], lat_f[v][m]) and torch.equal(rgb_c[v][m], rgb_f[v][m])
        area += (x1 - x0 + 1) * (y1 - y0 + 1)
        span_area += int(inside.sum())
    if focal == 600.0:
        assert area < 0.6 * 3 * H * W                       # the crop is worth something on the headline frame
        assert span_area < 0.8 * area                       # ... and the row spans on top of the box
    print("box area", area, "span area", span_area, "of", 3 * H * W)
    # every valid sample's corner texels and the painted vertices' lie inside the box
    S = get_cfg_samples()
    pts = hip.Points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], n_samples=S)
    mask, hit = hip.hull_mask(pts, b["tar_smpl_vertice"][0])
    t = torch.linspace(0.0, 1.0, S, device=gpu)
    z = b["near"][0][:, None] * (1.0 - t) + b["far"][0][:, None] * t
    world = (b["ray_o"][0][:, None] + b["ray_d"][0][:, None] * z[..., None]).reshape(-1, 3)[mask.reshape(-1).bool()]
    assert world.shape[0] > 1000
    for cloud in (world, b["input_smpl_vertice"][0][0].reshape(-1, 3)):
        for v, (xa, xb, ya, yb) in enumerate(_texel_use(b, f_crop, cloud, H, W)):
            x0, y0, x1, y1 = (int(q) for q in box[v])
            assert xa.min() >= x0 and xb.max() <= x1 and ya.min() >= y0 and yb.max() <= y1, (v, box[v])
            sp = spans[v].to(xa.device)
            for yy in (ya.long(), yb.long()):                 # both texel rows of every gather inside their rows' spans
                assert bool((xa.long() >= sp[yy, 0]).all()) and bool((xb.long() <= sp[yy, 1]).all()), v
    a = r.render_fast(b, frame=f_crop)
    c = r.render_fast(b, frame=f_full)
    assert r.last_stats["valid_samples"] > 1000
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert torch.equal(a[k], c[k]), k


def get_cfg_samples():
    from transhuman_amd.config import get_cfg
    return int(get_cfg().N_samples)


def test_cropped_map_is_completed_for_the_unmasked_branch(hip, gpu, net):
    """R' <= 2400 hit rays: the reference shades EVERY sample of those rays (:551), also the ones far outside the hull,
    whose texels a cropped map does not hold -- the C side writes the rest of the map first (th_frame.map_source).  Same
    for a frame rendered without a hull test (Renderer.render under no_grad builds an un-cropped map)."""
    r = _renderer(net, 300, samples=32)
    bc = synth.make_batch(48, 48, 3, seed=0, all_rays=True)
    b = synth.batch_to(bc, gpu)
    f_crop = r.prepare_frame(b)
    f_full = r.prepare_frame(b, crop_map=False)
    box = f_crop.map.box.cpu().numpy()
    assert any((bx[2] - bx[0] + 1) * (bx[3] - bx[1] + 1) < 48 * 48 for bx in box)     # a real crop
    a = r.render_fast(b, frame=f_crop)
    st = dict(r.last_stats)
    c = r.render_fast(b, frame=f_full)
    assert st["unmasked"] == 1 and 0 < st["hit_rays"] <= 2400
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert torch.equal(a[k], c[k]), k
    assert torch.equal(f_crop.map.interleaved(), f_full.map.interleaved())      # ... and the map is complete now
    with torch.no_grad():
        out = r.render(b)
    assert torch.isfinite(out["rgb_map"]).all()
