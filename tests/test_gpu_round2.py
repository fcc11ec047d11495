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
