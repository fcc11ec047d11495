"""SpatialEncoder -- feeds the hot path.

Mirrors /root/reference/lib/networks/encoder.py:50-155: torchvision-ResNet18
stem (conv1/bn1/relu, maxpool, layer1, layer2), each latent bilinearly
upsampled (align_corners=True) to HxW, concatenated with a 1x1 colour lift
(-> 384 ch ``pixel_feat_map``) and reduced by a 1x1 conv to the 192-ch
``holder_feat_map``.  torchvision is not installed in this image, so the trunk
is restated here with torchvision's parameter names (``encoder.model.*``,
including the never-executed layer3/layer4) to stay ``strict=True``
checkpoint-compatible.

On a GPU the stem runs as hand-written HIP end to end (``trunk``: K12 fp16-split MFMA convolutions + max pooling,
K11 train-mode BatchNorm / ReLU / residual), and Renderer.prepare_frame replaces the tail of ``forward`` (three
upsamples + colour lift + concat + reduction) by K8 / th_paint_group_nhwc; ``forward`` itself (the reference's op
order, stock torch modules for the tail) is kept for callers that want the two NCHW maps and for the CPU tests.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..config import get_cfg


def _graphs_enabled():
    from .. import hip
    return hip.graphs_enabled()


class _PEBuffers(nn.Module):
    """Buffer-only twin of the reference's PositionalEncoding
    (lib/networks/vision_transformer.py:100-122): keeps the `_freqs` /
    `_phases` state-dict entries."""

    def __init__(self, num_freqs, d_in=3, include_input=True):
        super().__init__()
        self.num_freqs = num_freqs
        self.d_out = num_freqs * 2 * d_in + (d_in if include_input else 0)
        self.include_input = include_input
        freqs = np.pi * 2.0 ** torch.arange(0, num_freqs)
        self.register_buffer("_freqs", torch.repeat_interleave(freqs, 2).view(1, -1, 1))
        ph = torch.zeros(2 * num_freqs)
        ph[1::2] = np.pi * 0.5
        self.register_buffer("_phases", ph.view(1, -1, 1))

    def forward(self, x):
        e = x.unsqueeze(1).repeat(1, self.num_freqs * 2, 1)
        e = torch.sin(torch.addcmul(self._phases, e, self._freqs)).view(x.shape[0], -1)
        return torch.cat((x, e), dim=-1) if self.include_input else e


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride=1, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), norm_layer(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class ResNet18Trunk(nn.Module):
    """torchvision.models.resnet18 parameter layout (conv1, bn1, layer1..4)."""

    def __init__(self, pretrained=False, norm_layer=None, **_):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64, 1, norm_layer), BasicBlock(64, 64, 1, norm_layer))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2, norm_layer), BasicBlock(128, 128, 1, norm_layer))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2, norm_layer), BasicBlock(256, 256, 1, norm_layer))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2, norm_layer), BasicBlock(512, 512, 1, norm_layer))
        self.avgpool = nn.Sequential()
        self.fc = nn.Sequential()


class SpatialEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        cfg = get_cfg()
        self.model = ResNet18Trunk()
        # encoder.py:85 is built while cfg.img_feat_size is still the YAML's 256
        # (cross_transformer.py:94 runs before :123) -> 256+128 = 384 inputs
        self.reduction_layer = nn.Conv2d(256 + 128, cfg.embed_size, 1)
        self.PE_color = _PEBuffers(10)                                                 # encoder.py:93 (unused)
        self.upsample_color = nn.Conv2d(3, 128, 1)                                     # encoder.py:95

    def trunk(self, x, fused_bn=None, graph=False):
        """ResNet18 stem -> the three latents (64ch @H/2, 64ch @H/4, 128ch @H/8), encoder.py:114-126.
        On a GPU with the network in train() (run.py:29) the stem is hand-written HIP end to end: convolutions as K12
        (hip.conv2d: fp16-split MFMA implicit GEMMs; TH_STOCK_CONV=1 keeps torch/MIOpen), max pooling, and the
        elementwise tail of every stage -- train-mode BatchNorm, ReLU, residual add -- as K11 (hip.bn_act: one
        statistics pass + one apply pass instead of torch's 3-4 launches per site).  ``fused_bn=False`` or
        TH_STOCK_BN=1 runs the stock modules; running statistics and num_batches_tracked evolve as in them."""
        m = self.model
        if fused_bn is None:
            fused_bn = x.is_cuda and os.environ.get("TH_STOCK_BN") != "1"
        if fused_bn and self._bn_sites_fusable():
            # (the HIP stem carries no autograd in either grad mode: the reference's inference loop runs under no_grad, a caller
            # that forgot it gets the same replayed graph)
            if graph and os.environ.get("TH_STEM_GRAPH", "1") != "0" and _graphs_enabled():
                lat = self._trunk_graphed(x)
                if lat is not None:
                    return lat
            return self._trunk_fused_bn(x)
        x = m.relu(m.bn1(m.conv1(x)))
        lat = [x]
        x = m.layer1(m.maxpool(x))
        lat.append(x)
        x = m.layer2(x)
        lat.append(x)
        return lat

    def _bn_sites(self):
        m = self.model
        sites = [m.bn1]
        for layer in (m.layer1, m.layer2):
            for blk in layer:
                sites += [blk.bn1, blk.bn2] + ([blk.downsample[1]] if blk.downsample is not None else [])
        return sites

    def _bn_sites_fusable(self):
        """train(): batch statistics (momentum given); eval(): running statistics -- both have a HIP form"""
        return all(isinstance(b, nn.BatchNorm2d) and ((b.training and b.momentum is not None) or
                                                      (not b.training and b.track_running_stats and b.running_mean is not None))
                   for b in self._bn_sites())

    def _trunk_fused_bn(self, x):
        from .. import hip
        m = self.model
        stock_conv = os.environ.get("TH_STOCK_CONV") == "1"

        fuse_stats = os.environ.get("TH_BN_STATS_PASS") != "1"

        def conv_bn(mod, bn, t, residual=None, relu=True):
            # K12 for the stem's shapes (its epilogue leaves the BatchNorm partial sums when the module is in train():
            # conv -> bn is two launches, TH_BN_STATS_PASS=1 keeps the separate statistics pass), the stock module otherwise
            if not stock_conv and hip.conv2d_supported(mod):
                if bn.training and fuse_stats:
                    y, st = hip.conv2d(t.contiguous(), mod, stats=True)
                    return hip.bn_act(y, bn, residual=residual, relu=relu, conv_stats=st)
                return hip.bn_act(hip.conv2d(t.contiguous(), mod), bn, residual=residual, relu=relu)
            return hip.bn_act(mod(t), bn, residual=residual, relu=relu)

        x = conv_bn(m.conv1, m.bn1, x)
        lat = [x]
        x = m.maxpool(x) if stock_conv else hip.maxpool3x3s2(x)
        for layer in (m.layer1, m.layer2):
            for blk in layer:
                idt = x if blk.downsample is None else conv_bn(blk.downsample[0], blk.downsample[1], x, relu=False)
                y = conv_bn(blk.conv1, blk.bn1, x)
                x = conv_bn(blk.conv2, blk.bn2, y, residual=idt)
            lat.append(x)
        counters = [b.num_batches_tracked for b in self._bn_sites()
                    if b.training and b.track_running_stats and b.num_batches_tracked is not None]
        if counters:
            torch._foreach_add_(counters, 1)            # one launch for the ten counters
        return lat

    # ---- the stem as a hipGraph ------------------------------------------------------------------------------------------
    # The HIP stem is 21 dependent launches (10 convolutions, 10 BatchNorm passes, the pooling) of 10 - 100 us each: for a frame
    # pipeline that is 0.35 ms of host time per frame (a third of what a rank of an 8-rank job can spend) and 31 launch latencies
    # on the side stream's dependent chain.  Frames of one shape run the same launches on the same parameter storage, so the
    # chain is captured once per instance and replayed: one graph launch per frame.  GRAPH_RING instances rotate (static input
    # copy + latents each): a frame's latents stay valid while the next GRAPH_RING - 1 frames' constants are built -- the frame
    # pipeline holds at most three frames of constants, and a cropped map's completion (th_map_source) reads the latents at
    # shading time.  Any change of the parameters' storage or version, of a BatchNorm's mode or of the range guard's
    # convolution fallback drops the instances.  BatchNorm running statistics and num_batches_tracked advance once per frame
    # exactly as in the eager form (they are written by the captured kernels).  TH_STEM_GRAPH=0 switches it off.
    GRAPH_RING = 4

    def _graph_version(self, x):
        from .. import hip
        m = self.model
        convs = [m.conv1] + [c for layer in (m.layer1, m.layer2) for blk in layer
                             for c in ([blk.downsample[0]] if blk.downsample is not None else []) + [blk.conv1, blk.conv2]]
        if os.environ.get("TH_STOCK_CONV") == "1" or not all(hip.conv2d_supported(c) for c in convs):
            return None
        v = [(c.weight._version, c.weight.data_ptr()) for c in convs]
        for b in self._bn_sites():
            # (momentum, eps and the statistics' storage are baked into the captured th_bn_act launches as arguments: part of the key)
            v.append((b.training, None if b.weight is None else (b.weight.data_ptr(), b.bias.data_ptr()),
                      None if b.running_mean is None else b.running_mean.data_ptr(),
                      None if b.running_var is None else b.running_var.data_ptr(),
                      None if b.num_batches_tracked is None else b.num_batches_tracked.data_ptr(),
                      b.momentum, b.eps, b.track_running_stats))
        return (tuple(x.shape), str(x.device), tuple(v))

    def _trunk_graphed(self, x):
        ver = self._graph_version(x)
        if ver is None or x.dtype != torch.float32:
            return None
        st = getattr(self, "_stem_graphs", None)
        if st is not None and st.get("eager_only"):
            st["same"] = st["same"] + 1 if st["ver"] == ver else 0      # (a version that has settled gets its graphs back)
            st["ver"] = ver
            if st["same"] < 2:
                return None
            st = None
        if st is None or st["ver"] != ver:
            # first frame of a version: eagerly (it also packs the convolution weights, which must not happen under capture),
            # then every instance of the ring is captured at once: the cost of capturing lands in this one frame
            # (weights that change between every two frames -- rendering inside a training loop -- would capture on every call:
            # after three captures that were never replayed the stem stays on separate launches)
            from .. import hip
            thrash = 0 if st is None or st["replays"] > 0 else st["thrash"] + 1
            if thrash >= 3:
                self._stem_graphs = {"eager_only": True, "ver": ver, "same": 0}
                return None
            st = self._stem_graphs = {"ver": ver, "inst": [], "next": 0, "replays": 0, "thrash": thrash}
            lat = self._trunk_fused_bn(x)
            try:
                for _ in range(self.GRAPH_RING):
                    xs = torch.empty_like(x, memory_format=torch.contiguous_format)
                    g, l = hip.graph_capture(lambda: self._trunk_fused_bn(xs))       # (capture records, it does not run)
                    st["inst"].append((g, xs, l))
            except hip.GraphCaptureFailed:
                self._stem_graphs = None
            return lat
        k = st["next"]
        st["next"] = (k + 1) % self.GRAPH_RING
        st["replays"] += 1
        g, xs, lat = st["inst"][k]
        # (an elementwise kernel, not xs.copy_(x): the runtime's device-to-device copy of the 9.4 MB of three 512 x 512 images took
        # 180 - 240 us -- 45 GB/s -- in front of the first convolution, tools/dropin_trace.sh)
        torch.mul(x, 1.0, out=xs)
        g.replay()
        return lat

    @staticmethod
    def feat_scale(H, W):
        """'scales used in projection', encoder.py:148-153 (pixel and holder maps are both HxW)."""
        s = np.array([W, H])
        return s / (s - 1) * 2.0

    def forward(self, x):
        H, W = x.shape[2:]
        x_ori = x
        lat = self.trunk(x)
        lat = [F.interpolate(l, (H, W), mode="bilinear", align_corners=True) for l in lat]
        pixel_feat_map = torch.cat(lat + [self.upsample_color(x_ori)], dim=1)
        holder_feat_map = self.reduction_layer(pixel_feat_map)
        # "scales used in projection", encoder.py:148-153
        s = np.array([pixel_feat_map.shape[-1], pixel_feat_map.shape[-2]])
        s = s / (s - 1) * 2.0
        return holder_feat_map, s, pixel_feat_map, s.copy()
