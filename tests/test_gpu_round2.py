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
