"""GPU tests added in round 2: weight-image ownership of the device context, the range guard of the fp16 hi/lo
split, parity with weights of other magnitudes / torch default initialisation, and full-size (BASELINE
configuration) checks.  Everything goes through the C ABI (transhuman_amd.hip)."""
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


def _mlp_inputs(gpu, P=512, seed=0):
    torch.manual_seed(seed)
    pf = torch.randn(3, 384, P, device=gpu)
    vd = torch.randn(P, 27, device=gpu)
    ps = torch.randn(P, 3, device=gpu) * 0.3
    cen = torch.randn(300, 3, device=gpu) * 0.4
    rot = torch.eye(3, device=gpu).reshape(1, 9).repeat(300, 1)
    tok = torch.randn(3, 300, 192, device=gpu)
    return pf, vd, ps, cen, rot, tok


def test_context_weights_follow_the_calling_module(hip, gpu, net):
    """One th_ctx holds ONE MLP weight image per device.  Rendering with net A, then net B, then A again must
    re-upload A (the round-1 cache keyed uploads by id(module) alone and shaded A with B's weights)."""
    import copy
    a = net
    b = copy.deepcopy(net)
    with torch.no_grad():
        b.alpha_fc.bias.add_(0.5)
        b.rgb_fc.bias.add_(0.125)
    inp = _mlp_inputs(gpu)
    ra0 = hip.network_forward(a, *inp)
    rb0 = hip.network_forward(b, *inp)
    ra1 = hip.network_forward(a, *inp)
    rb1 = hip.network_forward(b, *inp)
    assert torch.equal(ra0, ra1) and torch.equal(rb0, rb1)
    assert maxdiff((rb0[:, 3] - ra0[:, 3]).cpu(), torch.full((ra0.shape[0],), 0.5)) < 1e-5
    # a module that dies and a new one that may land on the same id(): still its own weights
    del b
    c = copy.deepcopy(net)
    with torch.no_grad():
        c.alpha_fc.bias.sub_(0.25)
    rc = hip.network_forward(c, *inp)
    assert maxdiff((rc[:, 3] - ra0[:, 3]).cpu(), torch.full((ra0.shape[0],), -0.25)) < 1e-5
    assert torch.equal(hip.network_forward(a, *inp), ra0)


# ---------------------------------------------------------------------------
# range guard of the fp16 hi/lo split arithmetic
# ---------------------------------------------------------------------------
def _frame_consts():
    from util import gold
    g7 = gold("g7_dparf")
    return g7["centres"], g7["blend"]


def _rescaled_net(net, c):
    """The same function with other hidden magnitudes: relu networks are positively homogeneous, so fc_2 (weight,
    bias) x c with fc_3.weight / feature_fc.weight x 1/c leaves raw unchanged in exact arithmetic while `inter`
    (and its view mean) is c times larger / smaller."""
    import copy
    n2 = copy.deepcopy(net)
    with torch.no_grad():
        n2.fc_2.weight.mul_(c)
        n2.fc_2.bias.mul_(c)
        n2.fc_3.weight.div_(c)
        n2.feature_fc.weight.div_(c)
    return n2


def _forward_case(gpu, P=3000, seed=9):
    from util import gold
    centres, blend = _frame_consts()
    tok = gold("g6_vit")["out"]
    rs = np.random.RandomState(seed)
    b = synth.make_batch(32, 32, 3, seed=0)
    vid = rs.randint(0, synth.NV, size=P)
    pts = b["tar_smpl_vertice_smplcoord"][0][vid] + torch.from_numpy(rs.normal(0, 0.05, (P, 3)).astype(np.float32))
    pf = torch.from_numpy(rs.normal(size=(3, 384, P)).astype(np.float32))
    vd = O.view_embed(torch.from_numpy(rs.normal(size=(P, 3)).astype(np.float32)))
    mask = torch.from_numpy(rs.uniform(size=P) < 0.9)
    rot = blend[:, :3, :3].float().reshape(-1, 9)
    dev_args = (pf.to(gpu), vd.to(gpu), pts.to(gpu), centres.to(gpu), rot.to(gpu), tok.to(gpu), mask.to(gpu))
    cpu_args = (pf, vd, pts, centres, blend, tok, mask)
    return dev_args, cpu_args


def _sd_of(n):
    return {k: v.detach().cpu().clone() for k, v in n.state_dict().items()}


@pytest.mark.parametrize("log2c", [10, -4])
def test_other_magnitudes_inside_the_range_stay_on_the_fused_kernel(hip, gpu, net, log2c):
    """hidden activations of order 1e3 .. 1e4 (c = 2^10) and 1e-2 .. 1e-1 (c = 2^-4): inside what the split resolves
    -> fused kernel, no fallback, raw within 1e-4 of the fp32 oracle"""
    import warnings
    n2 = _rescaled_net(net, 2.0 ** log2c)
    dev_args, cpu_args = _forward_case(gpu)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # a fallback warning would fail the test
        raw = hip.network_forward(n2, *dev_args).cpu()
    vals = hip.last_range
    inter_max = float(np.array([vals[4]], dtype=np.uint16).view(np.float16)[0])
    assert (inter_max > 500.0) if log2c > 0 else (inter_max < 1.0), inter_max
    ref = O.network_forward(_sd_of(n2), *cpu_args)
    assert maxdiff(raw, ref) < 1e-4


@pytest.mark.parametrize("log2c,kind", [(15, "overflow"), (-16, "tiny")])
def test_range_guard_detects_and_falls_back(hip, gpu, net, log2c, kind):
    """c = 2^15: `inter` passes 65504 (fp16 hi halves become inf) -- c = 2^-16: the whole tensor sits below 2^-14 where
    the halves are subnormal.  Both must be DETECTED (launch-wide maxima, th_range_read), reported, and the call
    must come back with the fp32-path result: raw within 1e-4 of the oracle either way."""
    n2 = _rescaled_net(net, 2.0 ** log2c)
    dev_args, cpu_args = _forward_case(gpu)
    with pytest.warns(RuntimeWarning, match="fp16 hi/lo split"):
        raw = hip.network_forward(n2, *dev_args).cpu()
    try:
        assert hip._range_fallback.get(gpu.index or 0), "context must be on the fp32 path now"
        ref = O.network_forward(_sd_of(n2), *cpu_args)
        assert torch.isfinite(raw).all()
        assert maxdiff(raw, ref) < 1e-4, kind
        # the next call with the same weights stays on the fp32 path silently
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            raw2 = hip.network_forward(n2, *dev_args).cpu()
        assert torch.equal(raw2, raw)
    finally:
        # other weights -> back to the fused kernel (the guard checks them afresh)
        raw3 = hip.network_forward(net, *dev_args).cpu()
        assert not hip._range_fallback.get(gpu.index or 0)
    assert maxdiff(raw3, O.network_forward(make_sd(), *cpu_args)) < 1e-4


def test_range_guard_flags_non_finite_inputs(hip, gpu, net):
    """an inf in the pixel features (K5's split rows) is seen by the guard (slot f) instead of silently becoming
    relu(NaN) = 0 downstream"""
    dev_args, cpu_args = _forward_case(gpu, P=600)
    pf = dev_args[0].clone()
    pf[1, 17, 5] = float("inf")
    with pytest.warns(RuntimeWarning, match="fp16 hi/lo split"):
        hip.network_forward(net, pf, *dev_args[1:])
    assert hip.last_range is not None
    hip.set_mlp_mode(1)                              # leave the fallback for the following tests
    assert not hip._range_fallback.get(gpu.index or 0)


def test_torch_default_init_vs_oracle(hip, gpu):
    """the reference modules' own default initialisation (torch.manual_seed(123)) instead of the
    deterministic synthetic weights every golden uses; alpha_fc.bias += 1 so that a good part of the samples has
    sigma > 0 (RGB branch exercised).  Whole render_fast against the CPU oracle."""
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.cross_transformer import Network
    from transhuman_amd.networks.renderer import if_clight_renderer
    cfg = get_cfg()
    cfg.vit_depth, cfg.N_samples, cfg.num_class = 12, 32, 300
    torch.manual_seed(123)
    n = Network()
    with torch.no_grad():
        n.alpha_fc.bias.add_(1.0)
    n.train()
    sd = _sd_of(n)
    n = n.to(gpu)
    assign = synth_assign(300)
    b = synth.make_batch(48, 48, 3, seed=0, focal=150.0)
    r = if_clight_renderer.Renderer(n, vertex_can=can64().numpy(), pc2voxel_ind=assign)
    out = r.render_fast(synth.batch_to(b, gpu), is_train=False)
    off, mem = csr(assign)
    with torch.no_grad():
        hol, pix = O.encoder_forward(sd, b["input_imgs"][0][0])
        ref, _ = O.render_fast(sd, b, hol, pix, off, mem, can_centres64(assign), n_samples=32)
    assert r.last_stats["valid_samples"] > 5000
    assert float(ref["acc_map"].max()) > 0.05, "degenerate frame"
    assert maxdiff(out["rgb_map"].cpu(), ref["rgb_map"]) < 1e-4
    assert maxdiff(out["acc_map"].cpu(), ref["acc_map"]) < 1e-4
    print("default-init range table:", [hex(v) for v in hip.last_range], "fallback:", dict(hip._range_fallback))


def test_bound_2d_mask_equals_oracle(hip, gpu):
    """8f-2: th_bound_mask (the six box faces rasterised like cv2.fillPoly) bit-exact against the oracle's restatement,
    on-axis / oblique cameras and a box that leaves the image"""
    from util import gold
    g = gold("g14_rays")
    cases = []
    for name in ("axis", "oblique"):
        H, W = [int(v) for v in g[f"{name}_HW"]]
        pose = np.concatenate([g[f"{name}_R"].numpy(), g[f"{name}_T"].numpy()], axis=1)
        cases.append((g[f"{name}_bounds"].numpy(), g[f"{name}_K"].numpy(), pose, H, W))
    K = np.array([[90.0, 0.0, 30.0], [0, 90.0, 20.0], [0, 0, 1]], np.float32)
    pose = np.concatenate([np.eye(3, dtype=np.float32), np.array([[1.2], [0.0], [0.0]], np.float32)], axis=1)
    cases.append((np.array([[-0.5, -0.5, 2.0], [0.5, 0.5, 3.0]], np.float32), K, pose, 40, 56))
    K5 = np.array([[600.0, 0.0, 256.0], [0, 600.0, 256.0], [0, 0, 1]], np.float32)
    pose5 = np.concatenate([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)], axis=1)
    cases.append((np.array([[-0.4, -0.8, 2.7], [0.45, 0.95, 3.3]], np.float32), K5, pose5, 512, 512))
    for b, K, pose, H, W in cases:
        m = hip.bound_2d_mask(b, K, pose, H, W, device=gpu).cpu().numpy()
        ref = O.bound_2d_mask(b, K, pose, H, W)
        assert m.shape == ref.shape and np.array_equal(m, ref), (H, W, int((m != ref).sum()))
        assert ref.sum() > 0


def test_marching_cubes_equals_oracle(hip, gpu):
    """8f-4: th_marching_cubes_* bit-exact against the oracle restatement (same vertex / triangle order by
    construction: float64 vertices equal, index lists equal), whole grid and slab by slab (the multi-GPU split),
    incl. the index -> world transform and empty results"""
    from util import gold
    g = gold("g17_mcubes")
    for name in ("ellipsoid", "blobs", "noise_padded"):
        vol, iso = g[name + "_vol"], float(g[name + "_iso"])
        scale, origin = (0.005, 0.006, 0.007), (-0.4, 0.25, 2.6)
        rv, rf = O.marching_cubes(vol.numpy(), iso, scale=scale, origin=origin)
        v, f = hip.marching_cubes(vol.to(gpu), iso, scale=scale, origin=origin)
        assert v.dtype == torch.float64 and f.dtype == torch.int32
        assert tuple(v.shape) == rv.shape and tuple(f.shape) == rf.shape, name
        assert np.array_equal(f.cpu().numpy(), rf.astype(np.int32)), name
        assert np.array_equal(v.cpu().numpy(), rv), name
        # slabs [0,7) [7,15) [15,X): disjoint contiguous ranges that tile the full arrays
        X = vol.shape[0]
        full_v, full_f = torch.zeros_like(v), torch.zeros_like(f)
        ends = [0, 7, 15, X]
        prev_v = prev_t = 0
        for a, b in zip(ends[:-1], ends[1:]):
            sv, sf, (v0, v1), (t0, t1) = hip.marching_cubes(vol.to(gpu), iso, scale=scale, origin=origin, x_range=(a, b))
            assert v0 == prev_v and t0 == prev_t
            full_v[v0:v1], full_f[t0:t1] = sv[v0:v1], sf[t0:t1]
            prev_v, prev_t = v1, t1
        assert prev_v == v.shape[0] and prev_t == f.shape[0]
        assert torch.equal(full_v, v) and torch.equal(full_f, f)
    v, f = hip.marching_cubes(torch.zeros((5, 6, 7), device=gpu), 20.0)
    assert tuple(v.shape) == (0, 3) and tuple(f.shape) == (0, 3)


def test_mesh_renderer_produces_the_iso_surface(hip, gpu, net, tmp_path):
    """if_mesh_renderer.Renderer.render end to end (:46-113): cube (padded by 10 like :101) -> device marching cubes
    at cfg.mesh_th -> world coordinates -> PLY.  The mesh equals the oracle's on the same cube, is closed, and lies
    inside the body box."""
    from transhuman_amd.config import get_cfg
    from transhuman_amd.mesh import read_ply
    from transhuman_amd.networks.renderer import if_mesh_renderer
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = 32, 300
    b = synth.make_batch(32, 32, 3, seed=0)
    b["pts"] = synth.make_grid_pts(b, 40)
    bd = synth.batch_to(b, gpu)
    r = if_mesh_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(300))
    old_th = cfg.mesh_th
    try:
        cfg.mesh_th = 0.5                                  # the synthetic weights' sigma_raw is O(1)
        out = r.render(bd)
    finally:
        cfg.mesh_th = old_th
    cube, mesh = out["cube"], out["mesh"]
    assert cube.shape == (60, 60, 60) and float(np.abs(cube[:10]).max()) == 0.0
    assert mesh.vertices.shape[0] > 100 and mesh.is_watertight
    voxel = np.array(cfg.voxel_size, dtype=np.float64)
    LB = b["can_bounds"][0].numpy().astype(np.float64)[0] - 10 * voxel
    rv, rf = O.marching_cubes(cube, 0.5, scale=voxel, origin=LB)
    assert np.array_equal(mesh.vertices.cpu().numpy(), rv) and np.array_equal(mesh.faces.cpu().numpy(), rf)
    path = mesh.export(str(tmp_path / "0.ply"))
    v2, f2 = read_ply(path)
    assert v2.shape[0] == rv.shape[0] and f2.shape[0] == rf.shape[0]


def test_white_background_vs_reference_golden(hip, gpu, net):
    """with cfg.white_bkgd the reference's render_fast leaves rays that miss the hull BLACK (they are never
    composited, if_clight_renderer.py:459-476) and adds (1 - acc) to the rays that hit it; Renderer.render
    (:486-498) composites every ray.  Golden from the reference itself (oracle/gen_golden_white.py)."""
    from util import gold
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_clight_renderer
    g = gold("g11w_render_white")
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = 32, 300
    b = synth.batch_to(synth.make_batch(64, 64, 3, seed=0, focal=210.0), gpu)
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(300))
    cfg.white_bkgd = True
    try:
        out = r.render_fast(b, is_train=False)
    finally:
        cfg.white_bkgd = False
    assert maxdiff(out["rgb_map"][0].cpu(), g["rgb"]) < 1e-4 and maxdiff(out["acc_map"][0].cpu(), g["acc"]) < 1e-4
    black = (g["rgb"].abs().sum(-1) == 0)
    assert float(out["rgb_map"][0].cpu()[black].abs().max()) == 0.0


# ---------------------------------------------------------------------------
# full-size checks (BASELINE.json configs 4 and 5 at their own sizes)
# ---------------------------------------------------------------------------
def _oracle_frame_bits(b, sd, assign):
    off, mem = csr(assign)
    with torch.no_grad():
        hol, pix = O.encoder_forward(sd, b["input_imgs"][0][0])
    return hol, pix, off, mem, can_centres64(assign)


def test_full_size_frame_nc1500_properties_and_oracle_sample(hip, gpu, net):
    """C4 at full size: 512 x 512 rays x 64 samples, V = 3, N_c = 1500 with the reference's own (ragged) kmeans
    file.  Size-independent properties over the whole frame + the CPU oracle on a sample of its rays."""
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_clight_renderer
    from util import real_assign
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = 64, 1500
    assign = real_assign(1500)
    bc = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
    b = synth.batch_to(bc, gpu)
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=assign)
    out = r.render_fast(b, is_train=False)
    st = dict(r.last_stats)
    rgb, acc = out["rgb_map"][0], out["acc_map"][0]
    assert rgb.shape == (512 * 512, 3) and st["hit_rays"] > 30000 and st["valid_samples"] > 1500000 and st["unmasked"] == 0
    assert torch.isfinite(rgb).all() and torch.isfinite(acc).all()
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5 and float(rgb.min()) >= 0.0
    P = hip.Points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], 64)
    m, hit = hip.hull_mask(P, b["tar_smpl_vertice"][0])
    assert int(hit.sum()) == st["hit_rays"] and int(m.sum()) == st["valid_samples"]
    assert float(acc[~hit].abs().max()) == 0.0 and float(rgb[~hit].abs().max()) == 0.0      # misses are exactly zero
    again = r.render_fast(b, is_train=False)
    assert torch.equal(again["rgb_map"], out["rgb_map"]) and torch.equal(again["acc_map"], out["acc_map"])   # run to run
    # ray-sharded == whole frame (8 ranks' shards)
    from transhuman_amd.dist import shard_ray_indices
    frame = r.prepare_frame(b)
    idx = shard_ray_indices(512, 512, 8, 5, tile_major=True).to(gpu)
    sh = dict(b)
    for k in ("ray_o", "ray_d", "near", "far"):
        sh[k] = b[k][:, idx].contiguous()
    whole = r.render_fast(b, frame=frame, small_frame_rays=-1)
    part = r.render_fast(sh, frame=frame, small_frame_rays=-1)
    # (fp32 rounding, not bit for bit: see test_render_ray_sharding_equals_full)
    assert float((part["rgb_map"][0] - whole["rgb_map"][0][idx]).abs().max()) < 2e-6
    # oracle on 96 rays (64 of them hits)
    rs = np.random.RandomState(11)
    hits = torch.nonzero(hit).reshape(-1).cpu().numpy()
    pick = np.sort(np.concatenate([rs.choice(hits, 64, replace=False), rs.choice(512 * 512, 32, replace=False)]))
    sd = make_sd()
    hol, pix, off, mem, cc = _oracle_frame_bits(bc, sd, assign)
    sub = dict(bc)
    for k in ("ray_o", "ray_d", "near", "far"):
        sub[k] = bc[k][:, pick]
    with torch.no_grad():
        ref, _ = O.render_fast(sd, sub, hol, pix, off, mem, cc, n_samples=64, small_frame_rays=-1)
    assert maxdiff(rgb[pick].cpu(), ref["rgb_map"][0]) < 1e-4 and maxdiff(acc[pick].cpu(), ref["acc_map"][0]) < 1e-4
    assert float(ref["acc_map"].max()) > 0.05


def test_full_size_sigma_grid_256_and_mesh(hip, gpu, net):
    """C5 at full size: sigma on a 256^3 grid (16.8 M voxels) + marching cubes.  Properties (zero exactly outside the
    hull, invariance to how the voxels are split into passes / shards, closed mesh inside the box) + the CPU oracle on
    a sample of voxels."""
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_mesh_renderer
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = 64, 500
    assign = synth_assign(500)
    bc = synth.make_batch(64, 64, 3, seed=0)
    bc["pts"] = synth.make_grid_pts(bc, 256)
    b = synth.batch_to(bc, gpu)
    r = if_mesh_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=assign)
    old = cfg.mesh_th
    cfg.mesh_th = 0.5
    try:
        out = r.render(b)
    finally:
        cfg.mesh_th = old
    cube = torch.from_numpy(out["cube"][10:-10, 10:-10, 10:-10])
    assert cube.shape == (256, 256, 256) and torch.isfinite(cube).all()
    flat = b["pts"].reshape(-1, 3)
    m, _ = hip.hull_mask(hip.Points(pts=flat), b["tar_smpl_vertice"][0])
    m = m.view(-1).cpu()
    assert int(m.sum()) == r.last_stats["valid_samples"] > 200000
    assert float(cube.reshape(-1)[~m].abs().max()) == 0.0
    # a shard (every 8th run of 4096 voxels, like bench.py deals them) equals the full evaluation (to fp32 rounding: the
    # shard's valid samples form different 32-sample tiles, see test_render_ray_sharding_equals_full)
    frame = r.prepare_frame(b)
    run = torch.arange(256 ** 3, device=gpu) // 4096
    mine = torch.nonzero(run % 8 == 3).reshape(-1)
    part = r.render(b, frame=frame, pts_slice=mine)["sigma"]
    full = r.render(b, frame=frame, pts_slice=torch.arange(256 ** 3, device=gpu))["sigma"]
    assert float((part - full[mine]).abs().max()) <= 2e-5 * max(1.0, float(full.abs().max()))
    mesh = out["mesh"]
    assert mesh.vertices.shape[0] > 10000 and mesh.is_watertight
    v = mesh.vertices.cpu().numpy()
    box = bc["can_bounds"][0].numpy().astype(np.float64)
    voxel = np.array(cfg.voxel_size)
    lo = box[0] - 10 * voxel
    assert (v >= lo - 1e-9).all() and (v <= lo + voxel * 275 + 1e-9).all()
    # oracle on 1200 voxels (1000 inside the hull)
    rs = np.random.RandomState(5)
    ins = np.flatnonzero(m.numpy())
    pick = np.sort(np.concatenate([rs.choice(ins, 1000, replace=False), rs.choice(256 ** 3, 200, replace=False)]))
    sd = make_sd()
    hol, pix, off, mem, cc = _oracle_frame_bits(bc, sd, assign)
    with torch.no_grad():
        ref = O.render_sigma_grid(sd, bc, bc["pts"].reshape(-1, 3)[pick].reshape(1, -1, 1, 1, 3), hol, pix, off, mem, cc)
    assert maxdiff(cube.reshape(-1)[pick], ref.reshape(-1)) < 1e-4


def test_vit_fp16_split_gemms_equal_fp32_gemms(hip, gpu, net):
    """TransHE's dense layers on the fp16-split MFMA path (th_gemm_h3: LayerNorm fused, GELU, residual accumulate)
    against the fp32 MFMA GEMMs of the same library and the oracle, N_c = 300 / 500 / 1500 (ragged row tiles), V = 1 / 3"""
    import ctypes as C
    from util import gold
    lib = hip.load_library()
    for V, nc in ((3, 300), (1, 500), (3, 1500)):
        x = torch.from_numpy(synth.smooth_noise((V, nc, 192), 31 + nc, passes=0))
        pe = O.normalize_pe(can_centres64(synth_assign(500) if nc == 500 else synth_assign(300))[None].repeat(V, 1, 1)) \
            if nc != 1500 else (torch.rand(V, nc, 3) * 2 - 1)
        if pe.shape[1] != nc:
            pe = (torch.rand(V, nc, 3, generator=torch.Generator().manual_seed(nc)) * 2 - 1)
        out_h3 = net.ViT(x.to(gpu), pe.to(gpu), mask=None).cpu()
        try:
            hip._check(lib.th_set_vit_mode(hip.ctx(gpu), 0))
            out_f32 = net.ViT(x.to(gpu), pe.to(gpu), mask=None).cpu()
        finally:
            hip._check(lib.th_set_vit_mode(hip.ctx(gpu), 1))
        assert torch.isfinite(out_h3).all()
        assert maxdiff(out_h3, out_f32) < 3e-5, (V, nc, maxdiff(out_h3, out_f32))
        if nc <= 500:
            assert maxdiff(out_h3, O.vit_forward(x, pe, make_sd(), 12)) < 1e-4


def test_split_map_equals_interleaved_map(hip, gpu, net):
    """TH_MAP_SPLIT (latents as 1 KiB rows + an r g b 0 plane, the default) against the interleaved 260-channel map: the
    same texels, so tokens and images are identical bit for bit; non-square image, both MLP forms"""
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer import if_clight_renderer
    get_cfg().N_samples, get_cfg().num_class = 32, 300
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(300))
    b = synth.batch_to(synth.make_batch(64, 48, 3, seed=0, focal=150.0), gpu)
    f_split = r.prepare_frame(b, crop_map=False)         # whole maps are compared below (cropped maps: test_gpu_round3)
    g_split = r.last_grouped.clone()
    f_int = r.prepare_frame(b, compact_map="interleaved")
    assert isinstance(f_split.map, hip.SplitMap) and tuple(f_int.map.shape[-1:]) == (260,)
    assert torch.equal(f_split.map.interleaved(), f_int.map)
    assert torch.equal(g_split, r.last_grouped) and torch.equal(f_split.tokens, f_int.tokens)
    # (row hand-over on both maps: the split map's default, the texel hand-over with folded maps, rounds in another order -- the
    # last block compares it with the interleaved map's rows to 1e-5)
    for mode in (1, 0):
        hip.set_mlp_mode(mode)
        hip.set_tex_rows(False)
        try:
            a = r.render_fast(b, frame=f_split)
            c = r.render_fast(b, frame=f_int)
        finally:
            hip.set_mlp_mode(1)
            hip.set_tex_rows(True)
        assert r.last_stats["valid_samples"] > 5000
        for k in ("rgb_map", "acc_map", "depth_map"):
            assert torch.equal(a[k], c[k]), (mode, k)
    assert f_split.map.fold is not None
    a = r.render_fast(b, frame=f_split)
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert maxdiff(a[k].cpu(), c[k].cpu()) < 1e-5, k
    # gathered rows: 260-wide rows from the split map == rows from the interleaved map (incl. border clamping)
    pts = torch.randn(500, 3, device=gpu) * 0.5 + torch.tensor([0.0, 0.1, 3.0], device=gpu)
    cams = hip.pack_cams(b["input_R"][0][0], b["input_T"][0][0], b["input_K"][0][0])
    rows_s = hip.pixel_gather(f_split.map, pts, cams, f_split.scale, row_floats=272)
    rows_i = hip.pixel_gather(f_int.map, pts, cams, f_int.scale, row_floats=272)
    assert torch.equal(rows_s, rows_i) and float(rows_s[..., 260:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("nc,V,P", [(7, 3, 1000), (64, 1, 2077), (500, 2, 4099), (4096, 3, 3001)])
def test_token_blend_on_the_matrix_pipe_equals_the_fp32_blend(hip, gpu, net, nc, V, P):
    """TH_ROWS_NBR (K4 hands over neighbour slots + per-tile unions, the fused kernel blends the rows of the token table with
    MFMAs) against TH_ROWS_FOLDED (K4 blends them in fp32) on random inputs: scattered points and random centres make the
    unions of a 32-sample tile as large as they get (nc = 4096: ~200 of 224 distinct centres, seven passes of 32 slots),
    nc = 7 is the smallest table, P is never a multiple of the tile / K4 workgroup size, every view count of the fused path."""
    rs = np.random.RandomState(nc + P)
    centres = torch.from_numpy(rs.uniform(-0.8, 0.8, (nc, 3)).astype(np.float32))
    q, _ = np.linalg.qr(rs.normal(size=(nc, 3, 3)))
    rot = torch.from_numpy(q.astype(np.float32).reshape(nc, 9))
    tok = torch.from_numpy(rs.normal(size=(V, nc, 192)).astype(np.float32))
    pts = torch.from_numpy(rs.uniform(-0.9, 0.9, (P, 3)).astype(np.float32))
    pf = torch.from_numpy(rs.normal(size=(V, 384, P)).astype(np.float32))
    vd = O.view_embed(torch.from_numpy(rs.normal(size=(P, 3)).astype(np.float32)))
    mask = torch.from_numpy(rs.uniform(size=P) < 0.85)
    args = (net, pf.to(gpu), vd.to(gpu), pts.to(gpu), centres.to(gpu), rot.to(gpu), tok.to(gpu), mask.to(gpu))
    raw_nbr = hip.network_forward(*args).cpu()
    hip.set_tok_gather(False)
    try:
        raw_rows = hip.network_forward(*args).cpu()
    finally:
        hip.set_tok_gather(True)
    assert torch.isfinite(raw_nbr).all() and torch.equal(hip.network_forward(*args).cpu(), raw_nbr)
    scale = max(1.0, float(raw_rows.abs().max()))
    assert float((raw_nbr - raw_rows).abs().max()) < 1e-5 * scale, (nc, V, P, float((raw_nbr - raw_rows).abs().max()), scale)


@pytest.mark.gpu
def test_pregather_and_two_phase_shading_change_nothing(hip, gpu, net, monkeypatch):
    """th_render_pregather (pixel rows + neighbour records of all chunks queued before TransHE has finished, on the
    current stream; TransHE beside them) and the two-phase shading of render_sequence are pure re-orderings: the frame is
    bit-identical with and without them, in the masked and in the un-masked (R' <= 2400) branch, also with several chunks"""
    from transhuman_amd.networks.renderer import if_clight_renderer
    from transhuman_amd.config import get_cfg
    cfg = get_cfg()
    cfg.N_samples, cfg.num_class = 32, 300
    r = if_clight_renderer.Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(300))
    hip.set_chunk_samples(4096)                       # ~25 k valid samples -> 7 chunks: 5 pre-gathered + 2 interleaved
    try:
        for res, focal in ((64, 210.0), (32, 105.0)):
            b = synth.batch_to(synth.make_batch(res, res, 3, seed=0, focal=focal), gpu)
            monkeypatch.setenv("TH_PREGATHER", "0")
            ref = r.render_fast(b, is_train=False)
            st_ref = dict(r.last_stats)
            seq_ref = [o["rgb_map"].clone() for o in r.render_sequence(iter([b, b]))]
            monkeypatch.delenv("TH_PREGATHER")
            out = r.render_fast(b, is_train=False)
            assert dict(r.last_stats) == st_ref
            for k in ("rgb_map", "acc_map", "depth_map"):
                assert torch.equal(out[k], ref[k]), (res, k)
            seq = [o["rgb_map"].clone() for o in r.render_sequence(iter([b, b]))]
            assert all(torch.equal(a, c) for a, c in zip(seq, seq_ref)) and torch.equal(seq[0], ref["rgb_map"])
    finally:
        hip.set_chunk_samples(524288)
