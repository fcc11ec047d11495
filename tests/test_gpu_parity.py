"""GPU parity: every HIP kernel (called through the C ABI via transhuman_amd.hip)
against the CPU oracle on the same seeded inputs, and against the committed
golden vectors produced by the real reference.

Tolerances (fp32 path; BASELINE.json north_star asks 1e-4 on rendered RGB/alpha):
  * index / boolean work (hull mask, compaction, 7-NN ids) ....... bit-exact
  * sampling positions ............................................ bit-exact
  * gathers / bilinear / pooling / compositing .................... 2e-6 .. 1e-5
  * DPaRF PE channels (sin at pi*2^9 amplifies 1 ulp of the
    rotated offset to ~1e-5) ...................................... 1e-4
  * MLP / ViT outputs (sum order of the fp32 MFMA chains) ......... 1e-4 absolute on O(1) values
  * rendered rgb / acc ............................................ 1e-4
"""
import numpy as np
import pytest
import torch

from oracle import th_oracle as O
from transhuman_amd import synth
from util import gold, make_sd, make_net, synth_assign, real_assign, csr, can_centres64, can64, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(gpu):
    from transhuman_amd import hip as H
    H.load_library()
    return H


@pytest.fixture(scope="module")
def net(gpu, hip):
    return make_net(12).to(gpu)


def cams_of(b, dev):
    from transhuman_amd import hip as H
    return H.pack_cams(b["input_R"][0][0].to(dev), b["input_T"][0][0].to(dev), b["input_K"][0][0].to(dev))


# ---------------------------------------------------------------------------
def test_linear_mfma_vs_torch(hip, gpu):
    """the fp32 MFMA GEMM, asymmetric operands, ragged M/N/K, all epilogues"""
    rs = np.random.RandomState(0)
    for (M, K, N) in ((1, 4, 1), (63, 255, 256), (200, 283, 128), (129, 384, 384), (70, 768, 192), (33, 192, 576),
                      (5, 128, 3)):
        x = torch.from_numpy(rs.normal(size=(M, K)).astype(np.float32))
        w = torch.from_numpy(rs.normal(size=(N, K)).astype(np.float32) / np.sqrt(K))
        b = torch.from_numpy(rs.normal(size=(N,)).astype(np.float32))
        for act, fn in ((0, lambda t: t), (1, torch.relu), (2, torch.nn.functional.gelu)):
            ref = fn(x.double() @ w.double().t() + b.double())
            out = hip.linear(x.to(gpu), w.to(gpu), b.to(gpu), act).cpu()
            assert out.shape == (M, N)
            assert maxdiff(out, ref) < 2e-5, (M, K, N, act)


def test_sampling_and_hull_mask_bit_exact(hip, gpu):
    b = synth.make_batch(48, 48, 3, seed=0, focal=150.0)
    S = 32
    pts, z = O.sampling_points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], S)
    ref = O.hull_mask(pts.reshape(-1, 3), b["tar_smpl_vertice"][0]).view(-1, S)
    P = hip.Points(b["ray_o"][0].to(gpu), b["ray_d"][0].to(gpu), b["near"][0].to(gpu), b["far"][0].to(gpu), S)
    m, hit = hip.hull_mask(P, b["tar_smpl_vertice"][0].to(gpu))
    assert ref.sum() > 1000
    assert torch.equal(m.cpu(), ref)
    assert torch.equal(hit.cpu(), ref.sum(-1) > 0)
    # explicit points (mesh path) incl. points far outside the vertex AABB
    grid = synth.make_grid_pts(b, 24).reshape(-1, 3)
    far_pts = torch.cat([grid, grid[:50] + 5.0, grid[:50] - 7.0])
    refg = O.hull_mask(far_pts, b["tar_smpl_vertice"][0])
    mg, _ = hip.hull_mask(hip.Points(pts=far_pts.to(gpu)), b["tar_smpl_vertice"][0].to(gpu))
    assert torch.equal(mg.cpu().view(-1), refg)


def test_hull_mask_golden(hip, gpu):
    g = gold("g11_render_large")
    b = synth.make_batch(64, 64, 3, seed=0, focal=210.0)
    P = hip.Points(b["ray_o"][0].to(gpu), b["ray_d"][0].to(gpu), b["near"][0].to(gpu), b["far"][0].to(gpu), 32)
    m, hit = hip.hull_mask(P, b["tar_smpl_vertice"][0].to(gpu))
    assert int(hit.sum()) == int(g["hit_rays"])
    assert np.array_equal(np.packbits(m[hit].cpu().numpy()), g["mask_bits"])


def test_paint_group_and_segment_means(hip, gpu):
    g = gold("g45_paint_group")
    b = synth.make_batch(32, 32, 3, seed=0)
    hol = torch.from_numpy(synth.smooth_noise((3, 192, 32, 32), 21)).to(gpu)
    off, mem = csr(synth_assign(300))
    offd, memd = hip.csr_to_device(off, mem, gpu)
    scale = hip.feat_scale(np.array([32, 32]) / (np.array([32, 32]) - 1) * 2.0, (32, 32), gpu)
    tok, painted = hip.paint_group(hol, b["input_smpl_vertice"][0][0].to(gpu), cams_of(b, gpu), scale,
                                   b["input_vizmaps"][0][0].to(gpu), offd, memd, return_painted=True)
    assert maxdiff(painted[:, :96].cpu(), g["big_head"]) < 2e-5   # 1-ulp uv differences x feature gradient
    assert maxdiff(tok.cpu(), g["grouped"]) < 2e-5
    for k in (500, 1500):
        gg = gold(f"g5_group_real{k}")
        o2, m2 = hip.csr_to_device(*csr(real_assign(k)), gpu)
        assert maxdiff(hip.segment_mean(painted[0], o2, m2).cpu()[None], gg["grouped"]) < 2e-5
    g7 = gold("g7_dparf")
    cen = hip.segment_mean(b["tar_smpl_vertice_smplcoord"][0].to(gpu), offd, memd)
    rot = hip.segment_mean_rot(b["blend_mtx"][0].to(gpu), offd, memd)
    assert maxdiff(cen.cpu(), g7["centres"]) < 5e-7
    assert maxdiff(rot.cpu().view(-1, 3, 3), g7["blend"][:, :3, :3].float()) < 1e-7


def test_vit_vs_oracle_and_golden(hip, gpu, net):
    g = gold("g6_vit")
    grouped = gold("g45_paint_group")["grouped"]
    pe_norm = O.normalize_pe(can_centres64(synth_assign(300))[None].repeat(3, 1, 1))
    pe_tab = net.ViT.get_PE(pe_norm.to(gpu))
    # argument = exact fma; sin itself may differ by an ulp between CPUs (sleef code path)
    assert maxdiff(pe_tab[0].cpu(), g["pe_table"]) < 1e-6
    out = net.ViT(grouped.to(gpu), pe_norm.to(gpu), mask=None).cpu()
    assert maxdiff(out, g["out"]) < 1e-4
    assert maxdiff(out, O.vit_forward(grouped, pe_norm, make_sd(), 12)) < 1e-4
    # ragged token count (not a multiple of the 64-wide tiles), V = 1
    g5 = gold("g6_vit_n500_v1")
    x = torch.from_numpy(synth.smooth_noise((1, 500, 192), 22, passes=0))
    pe5 = O.normalize_pe(can_centres64(synth_assign(500))[None])
    assert maxdiff(net.ViT(x.to(gpu), pe5.to(gpu)).cpu(), g5["out"]) < 1e-4


def test_dparf_vs_golden(hip, gpu):
    g = gold("g7_dparf")
    tok = gold("g6_vit")["out"].to(gpu)
    rot = g["blend"][:, :3, :3].float().reshape(-1, 9)
    out = hip.dparf_encode(g["pts_s"].to(gpu), g["centres"].to(gpu), rot.to(gpu), tok).cpu()   # [P,V,256]
    ref = g["human_rep"].permute(2, 0, 1)                                                      # [P,V,255]
    assert (out[..., 255] == 0).all()
    assert maxdiff(out[..., :195], ref[..., :195]) < 2e-6        # tokens + raw xyz
    assert maxdiff(out[..., 195:255], ref[..., 195:]) < 1e-4     # sin/cos channels
    # 7-NN ids must agree exactly: recompute weights from the oracle and compare the token part
    hr = O.dparf(g["pts_s"], g["centres"], g["blend"], tok.cpu())
    assert maxdiff(out[..., :192], hr[..., :192]) < 2e-6


def test_pixel_gather_vs_golden(hip, gpu):
    g = gold("g9_pixel_aligned")
    b = synth.make_batch(32, 32, 3, seed=0)
    pix = torch.from_numpy(synth.smooth_noise((3, 384, 32, 32), 24)).to(gpu)
    nhwc = hip.nchw_to_nhwc(pix)
    assert torch.equal(nhwc.cpu(), pix.permute(0, 2, 3, 1).contiguous().cpu())
    scale = hip.feat_scale(np.array([32, 32]) / (np.array([32, 32]) - 1) * 2.0, (32, 32), gpu)
    f = hip.pixel_gather(nhwc, g["xyz"].to(gpu), cams_of(b, gpu), scale).cpu()      # [P,V,384]
    assert maxdiff(f.permute(1, 2, 0), g["feat"]) < 2e-5


def _frame_consts():
    b = synth.make_batch(32, 32, 3, seed=0)
    off, mem = csr(synth_assign(300))
    centres = O.segment_mean(b["tar_smpl_vertice_smplcoord"][0], off, mem)
    blend = O.segment_mean(b["blend_mtx"][0], off, mem)
    return centres, blend


def test_network_forward_vs_golden(hip, gpu, net):
    """Network.forward through the reference's own call signature"""
    g = gold("g8_forward")
    centres, blend = _frame_consts()
    tok = gold("g6_vit")["out"]
    pf = torch.from_numpy(synth.smooth_noise((3, 384, 1024), 23, passes=0))
    dd = lambda: {"pts_smplcoord": g["pts_s"][None].to(gpu), "obs_smpl_smplcoord": centres[None].to(gpu),
                  "blend_mtx": blend[None].to(gpu)}
    for tag, mk in (("none", None), ("rand", g["mask"][None].to(gpu)), ("zero", torch.zeros_like(g["mask"])[None].to(gpu))):
        raw = net(pf.to(gpu), g["viewdir"][None].to(gpu), dd(), holder=tok.to(gpu), face_idx=None, pts_mask=mk)
        assert raw.shape == (1, 1024, 4)
        assert maxdiff(raw[0].cpu(), g["raw_" + tag]) < 1e-4, tag
    raw = net(pf[:1].to(gpu), g["viewdir"][None].to(gpu), dd(), holder=tok[:1].to(gpu), pts_mask=g["mask"][None].to(gpu))
    assert maxdiff(raw[0].cpu(), g["raw_v1_rand"]) < 1e-4
    raw = net(pf[:1].to(gpu), g["viewdir"][None].to(gpu), dd(), holder=tok[:1].to(gpu), pts_mask=None)
    assert maxdiff(raw[0].cpu(), g["raw_v1_none"]) < 1e-4


def test_network_forward_chunk_invariance(hip, gpu, net):
    """the network is strictly per-point: > 1 internal chunk (32768) must equal the oracle point for point"""
    centres, blend = _frame_consts()
    tok = gold("g6_vit")["out"]
    rs = np.random.RandomState(9)
    P = 40000
    b = synth.make_batch(32, 32, 3, seed=0)
    vid = rs.randint(0, synth.NV, size=P)
    pts = b["tar_smpl_vertice_smplcoord"][0][vid] + torch.from_numpy(rs.normal(0, 0.05, (P, 3)).astype(np.float32))
    pf = torch.from_numpy(rs.normal(size=(3, 384, P)).astype(np.float32))
    vd = O.view_embed(torch.from_numpy(rs.normal(size=(P, 3)).astype(np.float32)))
    mask = torch.from_numpy(rs.uniform(size=P) < 0.9)
    rot = blend[:, :3, :3].float().reshape(-1, 9)
    args = (net, pf.to(gpu), vd.to(gpu), pts.to(gpu), centres.to(gpu), rot.to(gpu), tok.to(gpu), mask.to(gpu))
    hip.set_chunk_samples(32768)          # 36 k valid samples -> two passes
    raw = hip.network_forward(*args).cpu()
    hip.set_chunk_samples(524288)         # default: one pass
    assert torch.equal(hip.network_forward(*args).cpu(), raw), "result must not depend on the chunk size"
    hip.set_mlp_mode(0)                   # per-layer fp32 MFMA form vs fused fp16-split form
    raw32 = hip.network_forward(*args).cpu()
    hip.set_mlp_mode(1)
    assert maxdiff(raw32, raw) < 5e-5
    sel = torch.cat([torch.arange(0, 700), torch.arange(32500, 33200), torch.arange(P - 600, P)])
    ref = O.network_forward(make_sd(), pf[:, :, sel], vd[sel], pts[sel], centres, blend, tok, mask[sel])
    assert maxdiff(raw[sel], ref) < 1e-4
    assert (raw[~mask] == 0).all()


def test_composite_vs_golden(hip, gpu):
    g = gold("g10_raw2outputs")
    rgb, acc, dep, w = hip.composite(g["raw"].to(gpu), g["z"].to(gpu), g["ray_d"].to(gpu), return_weights=True)
    assert maxdiff(rgb.cpu(), g["rgb"]) < 2e-6 and maxdiff(acc.cpu(), g["acc"]) < 2e-6
    assert maxdiff(dep.cpu(), g["depth"]) < 1e-5 and maxdiff(w.cpu(), g["weights"]) < 2e-6
    assert float(acc[5]) == 0.0 and float(acc[6]) == 0.0
    # S = 64 (one wave per ray) and S = 96 (two passes) against the oracle
    rs = np.random.RandomState(3)
    for S in (64, 96, 7):
        raw = torch.from_numpy(rs.normal(0, 2, (50, S, 4)).astype(np.float32))
        z = torch.sort(torch.from_numpy(rs.uniform(2, 4, (50, S)).astype(np.float32)), dim=1)[0]
        d = torch.from_numpy(rs.normal(size=(50, 3)).astype(np.float32))
        r0, a0, d0, _ = O.raw2outputs(raw, z, d)
        r1, a1, d1 = hip.composite(raw.to(gpu), z.to(gpu), d.to(gpu))
        assert maxdiff(r1.cpu(), r0) < 5e-6 and maxdiff(a1.cpu(), a0) < 5e-6 and maxdiff(d1.cpu(), d0) < 2e-5


def test_view_embed(hip, gpu):
    d = torch.from_numpy(np.random.RandomState(1).normal(size=(300, 3)).astype(np.float32))
    assert maxdiff(hip.view_embed(d.to(gpu)).cpu(), O.view_embed(d)) < 2e-6    # 1 ulp of |d| x octave 8


def _renderer(net, mesh=False):
    from transhuman_amd.networks.renderer import if_clight_renderer, if_mesh_renderer
    mod = if_mesh_renderer if mesh else if_clight_renderer
    return mod.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(300))


def _cfg(S):
    from transhuman_amd.config import get_cfg
    get_cfg().N_samples = S
    get_cfg().num_class = 300


@pytest.mark.parametrize("tag,H,focal", [("small", 32, None), ("large", 64, 210.0)])
def test_render_fast_vs_golden(hip, gpu, net, tag, H, focal):
    """the whole path, Renderer.render_fast, both sides of the R' <= 2400 switch"""
    g = gold(f"g11_render_{tag}")
    _cfg(32)
    b = synth.batch_to(synth.make_batch(H, H, 3, seed=0, focal=focal), gpu)
    r = _renderer(net)
    before = {k: v.clone() for k, v in b.items() if torch.is_tensor(v)}
    out = r.render_fast(b, is_train=False)
    assert all(torch.equal(b[k], v) for k, v in before.items()), "batch must not be mutated"
    assert out["rgb_map"].shape == (1, H * H, 3) and out["acc_map"].shape == (1, H * H)
    assert r.last_stats["hit_rays"] == int(g["hit_rays"])
    assert r.last_stats["unmasked"] == (1 if tag == "small" else 0)
    assert maxdiff(out["rgb_map"][0].cpu(), g["rgb"]) < 1e-4
    assert maxdiff(out["acc_map"][0].cpu(), g["acc"]) < 1e-4
    assert maxdiff(out["depth_map"][0].cpu(), g["depth"]) < 5e-4
    mse = float(((out["rgb_map"][0].cpu().double() - g["rgb"].double()) ** 2).mean())
    assert -10 * np.log10(max(mse, 1e-30)) > 80.0          # PSNR(build, reference) > 80 dB  


@pytest.mark.parametrize("cin,cout,ks,stride,H,W", [(64, 64, 3, 1, 128, 128), (64, 64, 3, 1, 37, 45), (64, 128, 3, 2, 128, 128),
                                                     (64, 128, 3, 2, 33, 47), (128, 128, 3, 1, 64, 64),
                                                     (128, 128, 3, 1, 19, 70), (64, 128, 1, 2, 128, 128),
                                                     (64, 128, 1, 2, 31, 33), (3, 64, 7, 2, 512, 512), (3, 64, 7, 2, 61, 90)])
def test_conv2d_vs_torch(hip, gpu, cin, cout, ks, stride, H, W):
    """K12 (fp16-split MFMA implicit GEMM) against torch's fp64 convolution: the stem shapes, ragged sizes, borders;
    re-packs when the weight changes"""
    torch.manual_seed(cin + cout + ks + H)
    conv = torch.nn.Conv2d(cin, cout, ks, stride, ks // 2, bias=False).to(gpu)
    assert hip.conv2d_supported(conv)
    x = (torch.randn(2, cin, H, W, device=gpu) * 1.5 + 0.3).contiguous()
    for rep in range(2):
        ref = torch.nn.functional.conv2d(x.double(), conv.weight.detach().double(), None, stride, ks // 2)
        got = hip.conv2d(x, conv)
        assert got.shape == ref.shape
        scale = float(ref.detach().abs().max())
        assert maxdiff(got.cpu(), ref.cpu()) < 2e-6 * max(scale, 1.0), (rep, maxdiff(got.cpu(), ref.cpu()), scale)
        with torch.no_grad():
            conv.weight.mul_(-0.37)                       # version bump: the packed image must be rebuilt


def test_maxpool_vs_torch(hip, gpu):
    for (N, C, H, W) in ((3, 64, 256, 256), (1, 5, 7, 9), (2, 3, 33, 32)):
        x = torch.randn(N, C, H, W, device=gpu)
        assert torch.equal(hip.maxpool3x3s2(x), torch.nn.functional.max_pool2d(x, 3, 2, 1))


def test_fused_bn_trunk_equals_stock_modules(hip, gpu):
    """K11 (hip.bn_act: train-mode BatchNorm + residual + ReLU) against torch's modules on the encoder trunk: latents,
    running statistics and counters after two forwards; plus odd shapes / no-affine / no-residual directly"""
    import copy
    from transhuman_amd.networks.encoder import SpatialEncoder
    torch.manual_seed(3)
    a = SpatialEncoder().to(gpu).train()
    b = copy.deepcopy(a)
    for it in range(2):
        x = torch.rand(3, 3, 128, 96, device=gpu)
        la = a.trunk(x, fused_bn=True)
        lb = b.trunk(x, fused_bn=False)
        for u, v in zip(la, lb):
            assert u.shape == v.shape and maxdiff(u.cpu(), v.cpu()) < 2e-5, it
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if "layer3" in k or "layer4" in k:
            continue
        if k.endswith("num_batches_tracked"):
            assert int(sa[k]) == int(sb[k]), k
        elif "running_" in k:
            assert maxdiff(sa[k].cpu(), sb[k].cpu()) < 1e-6, k
    assert int(sa["model.bn1.num_batches_tracked"]) == 2
    rs = np.random.RandomState(0)
    for (N, C, H, W) in ((1, 5, 7, 9), (2, 3, 33, 31), (3, 64, 64, 64), (1, 2, 300, 300)):
        x = torch.from_numpy(rs.normal(size=(N, C, H, W)).astype(np.float32) * 2 + 0.5).to(gpu)
        r = torch.from_numpy(rs.normal(size=(N, C, H, W)).astype(np.float32)).to(gpu)
        for affine, res, relu in ((True, None, True), (True, r, True), (False, r, False)):
            bn1 = torch.nn.BatchNorm2d(C, affine=affine).to(gpu).train()
            if affine:
                with torch.no_grad():
                    bn1.weight.uniform_(0.5, 1.5); bn1.bias.uniform_(-0.5, 0.5)
            bn2 = copy.deepcopy(bn1)
            got = hip.bn_act(x, bn1, residual=res, relu=relu)
            ref = bn2(x)
            if res is not None:
                ref = ref + res
            if relu:
                ref = torch.relu(ref)
            assert maxdiff(got.cpu(), ref.cpu()) < 5e-6, (N, C, H, W, affine, relu)
            assert maxdiff(bn1.running_mean.cpu(), bn2.running_mean.cpu()) < 1e-6
            assert maxdiff(bn1.running_var.cpu(), bn2.running_var.cpu()) < 1e-6


def test_render_sequence_equals_per_frame_render(hip, gpu, net):
    """Renderer.render_sequence (hull stage + constants of the next frames on a second stream under the shading of
    frame i, rotating workspaces, several th_render_prepass tokens pending) returns,
    for every frame of a stream of DIFFERENT frames, the image and statistics of render_fast on that frame; the
    look-ahead never mixes frames up (each frame has its own images, pose and rays) and an exhausted stream ends it"""
    _cfg(32)
    r = _renderer(net)
    frames = [synth.batch_to(synth.make_batch(64, 64, 3, seed=sd, focal=fc), gpu)
              for sd, fc in ((0, 210.0), (1, 190.0), (2, 230.0), (0, 210.0))]
    ref, ref_stats = [], []
    for b in frames:
        o = r.render_fast(b)
        ref.append({k: v.clone() for k, v in o.items()})
        ref_stats.append(dict(r.last_stats))
    assert maxdiff(ref[0]["rgb_map"].cpu(), ref[1]["rgb_map"].cpu()) > 1e-2       # the frames really differ
    for rep, la in enumerate((2, 1, 3)):                                             # look-ahead depths
        got = []
        for i, o in enumerate(r.render_sequence(iter(frames), lookahead=la)):
            assert r.last_stats == ref_stats[i], i
            got.append({k: v.clone() for k, v in o.items()})
        assert len(got) == len(frames)
        torch.cuda.synchronize()
        for i in range(len(frames)):
            for k in ("rgb_map", "acc_map", "depth_map"):
                # (frame constants are recomputed on the side stream: every kernel of the path -- K12 convolutions, K11
                # BatchNorm with fixed-order partial sums, K8, TransHE -- is deterministic, so the images are identical)
                assert torch.equal(got[i][k], ref[i][k]), (rep, i, k)
    assert list(r.render_sequence(iter([]))) == []


def test_render_ray_sharding_equals_full(hip, gpu, net):
    """rays are independent: rendering two interleaved shards == rendering the frame"""
    _cfg(32)
    b = synth.batch_to(synth.make_batch(64, 64, 3, seed=0, focal=210.0), gpu)
    r = _renderer(net)
    frame = r.prepare_frame(b)
    idx = torch.arange(64 * 64, device=gpu)
    # shards of <= 2400 hit rays would fall into the reference's un-masked branch (:551), which also
    # shades out-of-hull samples; sharded rendering therefore pins the switch off for every shard
    full2 = r.render_fast(b, frame=frame, small_frame_rays=-1)["rgb_map"][0]
    parts2 = torch.zeros_like(full2)
    for rank in range(2):
        sel = idx[rank::2]
        bb = dict(b)
        for k in ("ray_o", "ray_d", "near", "far"):
            bb[k] = b[k][:, sel]
        parts2[sel] = r.render_fast(bb, frame=frame, small_frame_rays=-1)["rgb_map"][0]
    # (a shard regroups the valid samples into different 32-sample tiles; the fused kernel blends a tile's token rows on the
    # matrix pipe over the UNION of the tile's neighbour centres (TH_ROWS_NBR), so the position of a sample's seven
    # terms in the accumulation depends on its tile mates: equal to fp32 rounding, not bit for bit)
    assert float((parts2 - full2).abs().max()) < 2e-6
    # with K4 blending the rows in fp32 (th_set_tok_gather(ctx, 0)) the shards ARE the frame, bit for bit -- and the two
    # hand-over forms agree to fp32 rounding
    hip.set_tok_gather(False)
    try:
        full3 = r.render_fast(b, frame=frame, small_frame_rays=-1)["rgb_map"][0]
        parts3 = torch.zeros_like(full3)
        for rank in range(2):
            sel = idx[rank::2]
            bb = dict(b)
            for k in ("ray_o", "ray_d", "near", "far"):
                bb[k] = b[k][:, sel]
            parts3[sel] = r.render_fast(bb, frame=frame, small_frame_rays=-1)["rgb_map"][0]
    finally:
        hip.set_tok_gather(True)
    assert torch.equal(parts3, full3)
    assert float((full3 - full2).abs().max()) < 2e-6
    # the threshold is a property of the CALL, also when the frame constants are handed in: the default
    # 2400 puts the same shard into the reference's un-masked branch
    bb = dict(b)
    for k in ("ray_o", "ray_d", "near", "far"):
        bb[k] = b[k][:, idx[0::2]]
    r.render_fast(bb, frame=frame)
    assert r.last_stats["unmasked"] == 1
    r.render_fast(bb, frame=frame, small_frame_rays=-1)
    assert r.last_stats["unmasked"] == 0


def test_mesh_sigma_cube_vs_golden(hip, gpu, net):
    g = gold("g12_mesh_cube")
    _cfg(32)
    b = synth.make_batch(32, 32, 3, seed=0)
    b["pts"] = synth.make_grid_pts(b, 20)
    b = synth.batch_to(b, gpu)
    out = _renderer(net, mesh=True).render(b)
    cube = torch.from_numpy(out["cube"][10:-10, 10:-10, 10:-10])
    assert out["cube"].shape == (40, 40, 40)
    assert maxdiff(cube, g["cube"]) < 1e-4
    assert torch.equal(cube != 0, g["cube"] != 0)


def test_dense_frame_properties(hip, gpu, net):
    """BASELINE-size properties that do not need the oracle: a 512x512x64 frame renders,
    is finite, acc in [0,1], rays that miss the hull are exactly zero, and the result is
    reproducible run to run (deterministic compaction)."""
    _cfg(64)
    from transhuman_amd.config import get_cfg
    get_cfg().num_class = 500
    from transhuman_amd.networks.renderer import if_clight_renderer
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(500))
    b = synth.batch_to(synth.make_batch(512, 512, 3, seed=0), gpu)
    frame = r.prepare_frame(b)
    o1 = r.render_fast(b, frame=frame)
    st = dict(r.last_stats)
    o2 = r.render_fast(b, frame=frame)
    assert torch.equal(o1["rgb_map"], o2["rgb_map"]) and torch.equal(o1["acc_map"], o2["acc_map"])
    assert torch.isfinite(o1["rgb_map"]).all()
    acc = o1["acc_map"][0]
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    P = hip.Points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], 64)
    m, hit = hip.hull_mask(P, b["tar_smpl_vertice"][0])
    assert int(hit.sum()) == st["hit_rays"] and int(m.sum()) == st["valid_samples"]
    assert (o1["rgb_map"][0][~hit] == 0).all() and (acc[~hit] == 0).all()
    assert st["hit_rays"] > 20000


def test_fused_encoder_tail_equals_reference_order(hip, gpu, net):
    """K8: channels-last map written by one kernel + reduction applied after sampling == encoder() then paint"""
    _cfg(32)
    b = synth.batch_to(synth.make_batch(64, 48, 3, seed=0, focal=150.0), gpu)       # non-square image, too
    r = _renderer(net)
    imgs = b["input_imgs"][0][0]
    with torch.no_grad():
        hol, hs, pix, ps = net.encoder(imgs)
        lat = net.encoder.trunk(imgs)
    nhwc = hip.upsample_concat_nhwc(imgs, lat[0], lat[1], lat[2], net.encoder.upsample_color.weight,
                                    net.encoder.upsample_color.bias)
    assert maxdiff(nhwc.permute(0, 3, 1, 2).cpu(), pix.cpu()) < 2e-5
    f_ref = r.prepare_frame(b, fused_encoder_tail=False)
    g_ref = r.last_grouped.clone()
    for compact in (False, True):
        f_new = r.prepare_frame(b, fused_encoder_tail=True, compact_map=compact)
        assert f_new.map.shape[-1] == (256 if compact else 384)          # compact: hip.SplitMap (256 latents + r g b 0 plane)
        assert maxdiff(r.last_grouped.cpu(), g_ref.cpu()) < 5e-5
        assert maxdiff(f_new.tokens.cpu(), f_ref.tokens.cpu()) < 1e-4
    # compact map = the 256 latent channels of the full map | r g b | 0
    cmp_map = hip.upsample_concat_nhwc(imgs, lat[0], lat[1], lat[2])
    assert torch.equal(cmp_map[..., :256], nhwc[..., :256])
    assert torch.equal(cmp_map[..., 256:259], imgs.permute(0, 2, 3, 1)) and (cmp_map[..., 259] == 0).all()
    g13 = gold("g13_encoder")
    b32 = synth.batch_to(synth.make_batch(32, 32, 3, seed=0), gpu)
    i32 = b32["input_imgs"][0][0]
    lat = net.encoder.trunk(i32)
    n32 = hip.upsample_concat_nhwc(i32, lat[0], lat[1], lat[2], net.encoder.upsample_color.weight,
                                   net.encoder.upsample_color.bias)
    assert maxdiff(n32.permute(0, 3, 1, 2)[:, :, ::8, ::8].cpu(), g13["pixel_px"]) < 1e-4


@pytest.mark.parametrize("mode", [1, 0])
def test_compact_map_equals_full_map(hip, gpu, net, mode):
    """colour-lift fold: rendering from the 260-channel map with colour-folded layers == rendering from the
    reference's 384-channel map (same function, different association of a linear map), in both MLP forms"""
    _cfg(32)
    b = synth.batch_to(synth.make_batch(64, 64, 3, seed=0, focal=210.0), gpu)
    r = _renderer(net)
    hip.set_mlp_mode(mode)
    try:
        outs = []
        for compact in (False, True):
            frame = r.prepare_frame(b, compact_map=compact)
            outs.append(r.render_fast(b, frame=frame))
    finally:
        hip.set_mlp_mode(1)
    assert maxdiff(outs[0]["rgb_map"].cpu(), outs[1]["rgb_map"].cpu()) < 3e-5
    assert maxdiff(outs[0]["acc_map"].cpu(), outs[1]["acc_map"].cpu()) < 3e-5
    g = gold("g11_render_large")
    assert maxdiff(outs[1]["rgb_map"][0].cpu(), g["rgb"]) < 1e-4


def test_pixel_gather_padded_rows(hip, gpu):
    """th_pixel_gather with ldo > C: the sampled channels are unchanged and the tail is zero"""
    torch.manual_seed(3)
    m = torch.randn(2, 24, 20, 260, device=gpu)
    pts = torch.randn(777, 3, device=gpu) * 0.3 + torch.tensor([0.0, 0.0, 3.0], device=gpu)
    b = synth.batch_to(synth.make_batch(24, 20, 2, seed=1), gpu)
    cams = cams_of(b, gpu)
    scale = torch.tensor([2.0 / 20, 2.0 / 24], device=gpu)
    a = hip.pixel_gather(m, pts, cams, scale)
    w = hip.pixel_gather(m, pts, cams, scale, row_floats=272)
    assert w.shape == (777, 2, 272)
    assert torch.equal(w[..., :260], a) and (w[..., 260:] == 0).all()


@pytest.mark.parametrize("case", ["axis", "oblique"])
def test_ray_generation_vs_golden(hip, gpu, case):
    """8f-2 on device: th_gen_rays vs the reference's get_rays / get_near_far outputs.  Directions agree to fp32
    rounding (BLAS summation order / LAPACK inverse are not specified); the float64 box test then decides
    identically except for rays that graze a box edge within that rounding."""
    g = gold("g14_rays")
    H, W = (int(x) for x in g[f"{case}_HW"])
    K, R, T, b = (g[f"{case}_{k}"] for k in ("K", "R", "T", "bounds"))
    dense = hip.gen_rays(K, R, T, b, H, W, device=gpu, compact=False)
    m_ref = g[f"{case}_mask"].bool()
    m = dense["mask_at_box"].cpu()
    assert maxdiff(dense["ray_d"].cpu(), g[f"{case}_ray_d_all"]) < 2e-6
    assert (dense["ray_d"].abs() >= 1e-5).all()
    both = m & m_ref
    assert int((m != m_ref).sum()) <= max(2, H * W // 500)
    near_ref = torch.zeros(H * W); far_ref = torch.zeros(H * W)
    near_ref[m_ref] = g[f"{case}_near"]; far_ref[m_ref] = g[f"{case}_far"]
    assert maxdiff(dense["near"].cpu()[both], near_ref[both]) < 2e-5
    assert maxdiff(dense["far"].cpu()[both], far_ref[both]) < 2e-5
    assert (dense["near"].cpu()[~m] == 0).all()
    c = hip.gen_rays(K, R, T, b, H, W, device=gpu)                # the reference's compacted ray list
    assert c["ray_o"].shape == (int(m.sum()), 3) and torch.equal(c["near"], dense["near"][dense["mask_at_box"]])
    assert maxdiff(c["ray_o"].cpu(), g[f"{case}_ray_o"][:1].expand(c["ray_o"].shape[0], 3)) < 1e-6


def test_generated_rays_render(hip, gpu, net):
    """rays made on device feed Renderer.render_fast exactly like rays from the batch"""
    _cfg(32)
    b = synth.batch_to(synth.make_batch(48, 48, 3, seed=0, all_rays=False), gpu)
    cam = synth.make_cameras(48, 48, 3)
    verts = b["tar_smpl_vertice"][0].cpu().numpy()
    bounds = np.stack([verts.min(0), verts.max(0)]).astype(np.float32)
    bounds[0, 2] -= 0.05; bounds[1, 2] += 0.05                    # can_smpl.py:228-230
    rays = hip.gen_rays(cam["K"].astype(np.float32), cam["R"].astype(np.float32), cam["T"].astype(np.float32), bounds,
                        48, 48, device=gpu)
    bb = dict(b)
    for k in ("ray_o", "ray_d", "near", "far"):
        bb[k] = rays[k][None]
    r = _renderer(net)
    out = r.render_fast(bb)
    assert out["rgb_map"].shape == (1, rays["near"].numel(), 3) and torch.isfinite(out["rgb_map"]).all()
    assert r.last_stats["hit_rays"] > 0
    # dense form (every pixel a ray; rays that miss the box carry near = far = 0 and render as background): the image of
    # the masked list scattered into the frame, with the same frame constants
    frame = r.prepare_frame(bb)
    sparse = r.render_fast(bb, frame=frame, small_frame_rays=-1)
    dense_rays = hip.gen_rays(cam["K"].astype(np.float32), cam["R"].astype(np.float32), cam["T"].astype(np.float32), bounds,
                              48, 48, device=gpu, compact=False)
    bd = dict(b)
    for k in ("ray_o", "ray_d", "near", "far"):
        bd[k] = dense_rays[k][None]
    dense = r.render_fast(bd, frame=frame, small_frame_rays=-1)
    m = dense_rays["mask_at_box"]
    assert torch.equal(m, rays["mask_at_box"]) and 0 < int(m.sum()) < m.numel()
    # (equal to fp32 rounding, not bit for bit: the sample list groups 16 CONSECUTIVE rays, so a ray has other tile mates in the
    # dense list than in the masked one and the token blend on the matrix pipe sums its seven terms in another order, DESIGN 5 iv)
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert float((dense[k][0][m] - sparse[k][0]).abs().max()) < 4e-6
        assert float(dense[k][0][~m].abs().max()) == 0.0


def _render_vs_oracle(hip, gpu, net, V, nc, assign, H=32, S=32, focal=None):
    """whole render_fast on the device vs the CPU oracle on the same synthetic frame (encoder included)"""
    from oracle import th_oracle as O
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_clight_renderer
    get_cfg().N_samples, get_cfg().num_class = S, nc
    b = synth.make_batch(H, H, V, seed=0, focal=focal)
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=assign)
    out = r.render_fast(synth.batch_to(b, gpu), is_train=False)
    sd = make_sd()
    off, mem = csr(assign)
    with torch.no_grad():
        hol, pix = O.encoder_forward(sd, b["input_imgs"][0][0])
        ref, _ = O.render_fast(sd, b, hol, pix, off, mem, can_centres64(assign), n_samples=S)
    assert float(ref["acc_map"].max()) > 0.05, "degenerate frame"
    assert maxdiff(out["rgb_map"].cpu(), ref["rgb_map"]) < 1e-4
    assert maxdiff(out["acc_map"].cpu(), ref["acc_map"]) < 1e-4
    return r.last_stats


def test_render_single_view_vs_oracle(hip, gpu, net):
    """BASELINE configs[0] shape of the path: V = 1 reference view (fused kernel <1,1>), N_c = 300"""
    st = _render_vs_oracle(hip, gpu, net, V=1, nc=300, assign=synth_assign(300))
    assert st["hit_rays"] > 0


def test_render_two_views_vs_oracle(hip, gpu, net):
    st = _render_vs_oracle(hip, gpu, net, V=2, nc=300, assign=synth_assign(300), H=64, focal=210.0)
    assert st["unmasked"] == 0          # large-frame (masked, chunked) branch


def test_render_nc1500_real_kmeans_vs_oracle(hip, gpu, net):
    """N_c = 1500 with the reference's own kmeans cluster file (ragged clusters):
    1500 centres in the DPaRF LDS table, 1500-token ViT attention"""
    st = _render_vs_oracle(hip, gpu, net, V=3, nc=1500, assign=real_assign(1500))
    assert st["hit_rays"] > 0


def test_smpl_lbs_vs_golden(hip, gpu):
    """8f-3 on device: th_smpl_lbs (float64) vs the reference's SMPL._call outputs, both pose input forms"""
    g = gold("g15_smpl")
    model = hip.SmplModel(synth.make_smpl_model(), device=gpu)
    pose, beta = synth.make_smpl_pose()
    for form in (g["R"], torch.from_numpy(pose)):
        v, j, T = model(form, beta)
        assert v.dtype == torch.float64 and T.shape == (6890, 4, 4)
        assert maxdiff(v.cpu(), g["v"]) < 1e-10 and maxdiff(j.cpu(), g["joints"]) < 1e-10
        assert maxdiff(T.cpu()[::16], g["T_sub"]) < 1e-10
        assert maxdiff(T.cpu().sum(0), g["T_sum"]) < 1e-8


def test_prepass_equals_plain_render(hip, gpu, net):
    """th_render_prepass only reorders the queue: with the same frame constants the image and the statistics are
    bit-identical, a stale token (other rays) is ignored, and Renderer.render_fast's automatic
    prepass -> prepare_frame -> shading order gives the frame of the plain order"""
    _cfg(32)
    b = synth.batch_to(synth.make_batch(64, 64, 3, seed=0, focal=210.0), gpu)
    r = _renderer(net)
    frame = r.prepare_frame(b)

    def pts(n=None):
        return hip.Points(b["ray_o"][0][:n], b["ray_d"][0][:n], b["near"][0][:n], b["far"][0][:n], n_samples=32)

    rgb0, acc0, dep0, st0 = hip.render_rays(net, frame, pts())
    P1 = pts()
    hip.render_prepass(P1, b["tar_smpl_vertice"][0], 3)
    torch.zeros(1 << 20, device=gpu).normal_()            # unrelated work queued in between
    rgb1, acc1, dep1, st1 = hip.render_rays(net, frame, P1)
    assert torch.equal(rgb0, rgb1) and torch.equal(acc0, acc1) and torch.equal(dep0, dep1) and st0 == st1
    hip.render_prepass(pts(100), b["tar_smpl_vertice"][0], 3)      # token for OTHER rays: must not be used
    rgb2, _, _, st2 = hip.render_rays(net, frame, pts())
    assert torch.equal(rgb0, rgb2) and st0 == st2
    auto = r.render_fast(b)                               # prepass -> prepare_frame -> shading
    assert r.last_stats == st0
    assert torch.equal(auto["rgb_map"][0], rgb0)          # (frame constants recomputed: deterministic kernels end to end)


def test_weight_updates_are_picked_up(hip, gpu, net):
    """the per-frame weight check (version counters of a cached parameter list) sees in-place updates
    (optimiser step / load_state_dict) and re-uploads the packed images"""
    g = gold("g8_forward")
    torch.manual_seed(0)
    P = 512
    pf = torch.randn(3, 384, P, device=gpu)
    vd = torch.randn(P, 27, device=gpu)
    ps = torch.randn(P, 3, device=gpu) * 0.3
    cen = torch.randn(300, 3, device=gpu) * 0.4
    rot = torch.eye(3, device=gpu).reshape(1, 9).repeat(300, 1)
    tok = torch.randn(3, 300, 192, device=gpu)
    raw0 = hip.network_forward(net, pf, vd, ps, cen, rot, tok)
    with torch.no_grad():
        net.alpha_fc.bias.add_(0.25)
    try:
        raw1 = hip.network_forward(net, pf, vd, ps, cen, rot, tok)
        assert maxdiff((raw1[:, 3] - raw0[:, 3]).cpu(), torch.full((P,), 0.25)) < 1e-5
    finally:
        with torch.no_grad():
            net.alpha_fc.bias.sub_(0.25)
    raw2 = hip.network_forward(net, pf, vd, ps, cen, rot, tok)
    assert torch.equal(raw2, raw0)


def test_render_four_views_vs_oracle(hip, gpu, net):
    """V = 4 reference views: beyond the fused kernel's register budget -> the layer-by-layer fp32 MFMA path
    (plain fp32 rows from K4 / K5) is dispatched automatically"""
    st = _render_vs_oracle(hip, gpu, net, V=4, nc=300, assign=synth_assign(300))
    assert st["hit_rays"] > 0


def test_edge_cases_no_hits_single_ray_empty(hip, gpu, net):
    """frames the reference handles through its early exits: no ray touches the hull (all outputs zero), a single
    ray, an empty ray list"""
    _cfg(32)
    b = synth.batch_to(synth.make_batch(32, 32, 3, seed=0), gpu)
    r = _renderer(net)
    far_away = dict(b)
    far_away["near"] = torch.full_like(b["near"], 40.0)
    far_away["far"] = torch.full_like(b["far"], 41.0)
    out = r.render_fast(far_away)
    assert r.last_stats["hit_rays"] == 0 and r.last_stats["valid_samples"] == 0
    assert float(out["rgb_map"].abs().max()) == 0.0 and float(out["acc_map"].abs().max()) == 0.0
    full = r.render_fast(b)
    hit = int(torch.nonzero(full["acc_map"][0] > 0)[0])
    one = dict(b)
    for k in ("ray_o", "ray_d", "near", "far"):
        one[k] = b[k][:, hit:hit + 1]
    o1 = r.render_fast(one)
    assert o1["rgb_map"].shape == (1, 1, 3) and r.last_stats["hit_rays"] == 1
    # a single hit ray is a "small frame" (R' <= 2400, un-masked branch) exactly like the 32x32 frame it came from
    assert maxdiff(o1["rgb_map"][0, 0].cpu(), full["rgb_map"][0, hit].cpu()) < 1e-5
    none = dict(b)
    for k in ("ray_o", "ray_d", "near", "far"):
        none[k] = b[k][:, :0]
    o0 = r.render_fast(none)
    assert o0["rgb_map"].shape == (1, 0, 3) and o0["acc_map"].shape == (1, 0)


@pytest.mark.parametrize("nc", [300, 500, 1500])
def test_dparf_candidate_grid_is_exact(hip, gpu, net, nc, monkeypatch):
    """the per-cell candidate lists of K4 are supersets of every point's 7 nearest centres: rendering with the grid
    equals rendering with the full N_c scan bit for bit (synthetic and the reference's ragged kmeans clusters)"""
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_clight_renderer
    get_cfg().N_samples, get_cfg().num_class = 32, nc
    # (round 6: the grid's cell follows the token density, 0.075 m cbrt(500 / N_c), lists in slots of 192 -- three densities)
    assign = synth_assign(nc) if nc != 1500 else real_assign(1500)
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=assign)
    b = synth.batch_to(synth.make_batch(96, 96, 3, seed=0, focal=260.0), gpu)
    frame = r.prepare_frame(b)
    with_grid = r.render_fast(b, frame=frame)
    st = dict(r.last_stats)
    monkeypatch.setenv("TH_DPARF_NOGRID", "1")
    full_scan = r.render_fast(b, frame=frame)
    assert st == r.last_stats and st["valid_samples"] > 20000
    assert torch.equal(with_grid["rgb_map"], full_scan["rgb_map"]) and torch.equal(with_grid["acc_map"], full_scan["acc_map"])
