"""The training entry (Renderer.render with gradients -> transhuman_amd.networks.autograd_path) against a forward /
backward of the REAL reference (tests/golden/g18_train_step.npz, made by oracle/gen_golden_train.py from the reference's
own Renderer.render + the trainer's image loss): outputs, loss and the gradients of 20 parameters spread over the
encoder, TransHE and the per-point network.  The path is plain torch, so this pin runs on CPU; the GPU test checks that
Renderer.render dispatches to it under autograd and to the HIP kernels under no_grad, and that both agree."""
import os

import numpy as np
import pytest
import torch

from transhuman_amd import synth
from transhuman_amd.config import get_cfg
from util import GOLD, can64, synth_assign, SIGMA_BIAS


def _setup(device="cpu"):
    from transhuman_amd.networks.cross_transformer import Network
    from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
    cfg = get_cfg()
    cfg.vit_depth, cfg.N_samples, cfg.num_class, cfg.perturb, cfg.raw_noise_std = 2, 16, 300, 0.0, 0.0
    torch.manual_seed(0)
    net = Network()
    net.load_state_dict(synth.det_state_dict(net.state_dict(), seed=0, sigma_bias=SIGMA_BIAS))
    net.train()
    net = net.to(device)
    r = Renderer(net, vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(300))
    b = synth.batch_to(synth.make_batch(20, 20, 3, seed=0, all_rays=False, focal=62.5), device)
    return cfg, net, r, b


def _loss(ret, target):
    return torch.mean((ret["rgb_map"] - target) ** 2) + 0.1 * ret["acc_map"].mean() + 0.01 * ret["depth_map"].mean()


def test_training_step_matches_the_reference():
    from transhuman_amd.networks import autograd_path
    g = np.load(os.path.join(GOLD, "g18_train_step.npz"))
    cfg, net, r, b = _setup()
    try:
        assert b["ray_o"].shape[1] == int(g["rays"])
        ret = autograd_path.render(r, b)
        for k, name in (("rgb_map", "rgb"), ("acc_map", "acc"), ("depth_map", "depth")):
            d = float((ret[k][0].detach() - torch.from_numpy(g[name])).abs().max())
            assert d < 2e-5, (k, d)
        loss = _loss(ret, torch.from_numpy(g["target"])[None])
        assert abs(float(loss) - float(g["loss"])) < 1e-6
        loss.backward()
        params = dict(net.named_parameters())
        keys = [k[5:] for k in g.files if k.startswith("grad:")]
        assert len(keys) == 20
        for k in keys:
            ref = torch.from_numpy(g["grad:" + k])
            got = params[k].grad
            assert got is not None and got.shape == ref.shape, k
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            assert err < 2e-3, (k, err)
        # every parameter the reference trains receives a gradient (the dead cls / mask tokens and PE buffers do not)
        missing = [k for k, p in params.items() if p.grad is None and not k.endswith(("cls_token", "mask_token"))
                   and ".layer3." not in k and ".layer4." not in k and "PE" not in k]
        assert not missing, missing
    finally:
        cfg.vit_depth, cfg.N_samples = 12, 64


def test_training_randomisations_are_live():
    """cfg.perturb (stratified depth jitter, :276-283) and cfg.raw_noise_std (density noise, nerf_net_utils.py:39-46) change
    the result from call to call in train() mode and leave it alone in eval() / at 0"""
    from transhuman_amd.networks import autograd_path
    cfg, net, r, b = _setup()
    try:
        with torch.no_grad():
            base = autograd_path.render(r, b)["rgb_map"]
            cfg.perturb = 1.0
            a1 = autograd_path.render(r, b)["rgb_map"]
            a2 = autograd_path.render(r, b)["rgb_map"]
            assert float((a1 - a2).abs().max()) > 1e-6 and float((a1 - base).abs().max()) > 1e-6
            net.eval()
            e1 = autograd_path.render(r, b)["rgb_map"]
            e2 = autograd_path.render(r, b)["rgb_map"]
            assert torch.equal(e1, e2)
            net.train()
            cfg.perturb, cfg.raw_noise_std = 0.0, 1.0
            n1 = autograd_path.render(r, b)["rgb_map"]
            n2 = autograd_path.render(r, b)["rgb_map"]
            assert float((n1 - n2).abs().max()) > 1e-7
    finally:
        cfg.perturb, cfg.raw_noise_std, cfg.vit_depth, cfg.N_samples = 0.0, 0.0, 12, 64


@pytest.mark.gpu
def test_render_dispatch_on_the_gpu(gpu):
    """Renderer.render: under autograd the differentiable path (loss.backward() reaches the encoder), under no_grad the
    HIP kernels -- and the two agree within the parity bar on the same patch"""
    cfg, net, r, b = _setup(gpu)
    try:
        ret = r.render(b)
        assert ret["rgb_map"].requires_grad
        target = torch.rand(ret["rgb_map"].shape, device=gpu)
        _loss(ret, target).backward()
        g = net.encoder.model.conv1.weight.grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
        with torch.no_grad():
            fast = r.render(b)
        assert not fast["rgb_map"].requires_grad
        for k in ("rgb_map", "acc_map", "depth_map"):
            assert float((fast[k] - ret[k].detach()).abs().max()) < 1e-4, k
        # the training-time randomisations are served too (stratified sampling in train() mode)
        cfg.perturb = 1.0
        with torch.no_grad():
            j1, j2 = r.render(b)["rgb_map"], r.render(b)["rgb_map"]
        assert float((j1 - j2).abs().max()) > 1e-6
    finally:
        cfg.perturb, cfg.vit_depth, cfg.N_samples = 0.0, 12, 64
