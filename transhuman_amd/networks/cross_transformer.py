"""Network -- DPaRF point encoding + per-point multi-view colour/density MLP.

Drop-in for /root/reference/lib/networks/cross_transformer.py (`Network`,
:83-353): same constructor (reads the global cfg and writes
cfg.embed_size / cfg.img_feat_size, :91,:123), same sub-module and parameter
names (``ViT``, ``encoder``, ``spatial_key_value_{0,1}.{key,value}_embed``,
``fc_0..4``, ``alpha_fc``, ``feature_fc``, ``view_fc``, ``rgb_fc``,
``alpha_res_0``, ``rgb_res_{0,1}``, ``PE_relative._freqs/_phases``), same
``forward(pixel_feat, sincos_viewdir, DPaRF_param_dict, holder, face_idx,
pts_mask) -> raw[1,P,4]``.  The reference's dead ``xyzc_net`` (spconv, never
executed, :101) is not instantiated; its checkpoint entries are accepted and
ignored so ``load_state_dict(strict=True)`` of a reference checkpoint works.

All arithmetic runs in the HIP library (transhuman_amd/csrc, C ABI in
include/transhuman_hip.h); inference only (no autograd through the kernels).
"""
import os
import sys

# Drop-in loading: the reference instantiates this file through imp.load_source(cfg.<x>_module, cfg.<x>_path)
# (lib/networks/make_network.py:4-11, renderer/make_renderer.py:4-8) under WHATEVER module name the YAML gives --
# 'lib.networks.cross_transformer' if only the path key is changed.  Relative imports would then resolve inside the
# reference's `lib` package, so the package is imported absolutely, found through this file's own location.
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

import torch                                                        # noqa: E402
from torch import nn                                                # noqa: E402

from transhuman_amd.config import get_cfg                           # noqa: E402
from transhuman_amd.networks.encoder import SpatialEncoder, _PEBuffers   # noqa: E402
from transhuman_amd.networks import vision_transformer as ViT       # noqa: E402


class SpatialKeyValue(nn.Module):
    def __init__(self, input_dim=256, att_dim=128, out_dim=256):
        super().__init__()
        self.key_embed = nn.Conv1d(input_dim, att_dim, kernel_size=1, stride=1)
        self.value_embed = nn.Conv1d(input_dim, out_dim, kernel_size=1, stride=1)


class Network(nn.Module):
    def __init__(self):
        super().__init__()
        cfg = get_cfg()
        self.ViT = ViT.vit_tiny(depth=cfg.vit_depth)
        cfg.embed_size = self.ViT.embed_dim                       # :91
        self.encoder = SpatialEncoder()
        self.spatial_key_value_0 = SpatialKeyValue()
        self.spatial_key_value_1 = SpatialKeyValue()
        self.PE_relative = _PEBuffers(cfg.KNN_FREQ)               # :106
        self.fc_0 = nn.Conv1d(cfg.embed_size + self.PE_relative.d_out, 256, 1)
        self.fc_1 = nn.Conv1d(256, 256, 1)
        self.fc_2 = nn.Conv1d(256, 256, 1)
        self.alpha_fc = nn.Conv1d(256, 1, 1)
        self.feature_fc = nn.Conv1d(256, 256, 1)
        self.view_fc = nn.Conv1d(283, 128, 1)
        self.rgb_fc = nn.Conv1d(128, 3, 1)
        self.fc_3 = nn.Conv1d(256, 256, 1)
        self.fc_4 = nn.Conv1d(128, 128, 1)
        cfg.img_feat_size = 256 + 128                             # :123
        self.alpha_res_0 = nn.Conv1d(cfg.img_feat_size, 256, 1)
        self.rgb_res_0 = nn.Conv1d(cfg.img_feat_size, 256, 1)
        self.rgb_res_1 = nn.Conv1d(cfg.img_feat_size, 128, 1)
        if cfg.KNN != 7 or cfg.KNN_FREQ != 10 or abs(cfg.KNN_DIST_ALPHA - 0.5) > 0:
            raise NotImplementedError("the DPaRF kernel is built for KNN=7, KNN_FREQ=10, KNN_DIST_ALPHA=0.5 "
                                      "(configs/train_or_eval.yaml:63-66)")
        if getattr(cfg, "use_truncation", False):
            raise NotImplementedError("cfg.use_truncation=True is a training-only option (cross_transformer.py:249)")

    # -- checkpoint compatibility ------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.startswith("xyzc_net.")}
        return super().load_state_dict(sd, strict=strict, **kw)

    # -- reference API -------------------------------------------------------------
    def forward(self, pixel_feat, sincos_viewdir, DPaRF_param_dict, holder=None, face_idx=None, pts_mask=None):
        """pixel_feat [V,384,P]; sincos_viewdir [1,P,27]; holder [V,N_c,192];
        pts_mask bool [1,P] or None -> raw [1,P,4] (cross_transformer.py:207-271)."""
        from transhuman_amd import hip
        pts = DPaRF_param_dict["pts_smplcoord"]
        centres = DPaRF_param_dict["obs_smpl_smplcoord"]
        blend = DPaRF_param_dict["blend_mtx"]
        assert pts.shape[0] == 1, "B = 1 is assumed (cross_transformer.py:222)"
        rot = blend[0][..., :3, :3].type(torch.float32).reshape(-1, 9)    # :185
        raw = hip.network_forward(self, pixel_feat, sincos_viewdir[0], pts[0], centres[0], rot, holder,
                                  mask=None if pts_mask is None else pts_mask[0])
        return raw.unsqueeze(0)
