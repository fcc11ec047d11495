"""Import the REAL reference hot-path modules on CPU (survey container only).

TEST INFRASTRUCTURE.  /root/reference does not exist on the GPU box; this file
is used solely by oracle/gen_golden.py (here) to produce tests/golden/*.npz and
by the optional ``reference-present`` CPU tests.  Recipe = SURVEY.md Appendix B:
stub the missing third-party packages in sys.modules, fake argv, chdir into the
reference (it uses cwd-relative paths), make ``.cuda()`` a no-op.

The only stand-in that carries arithmetic is ``pytorch3d.ops.knn_points``
(un-vendored, version unpinned): it is bound to oracle.th_oracle.knn_points_exact.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REF, "lib", "networks"))


class _AnyModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x, *a, **k):
        return x


class _Seq(nn.Sequential):
    def __init__(self, *mods):
        super().__init__(*[m for m in mods if isinstance(m, nn.Module)])


class _Stub(types.ModuleType):
    """Module whose unknown attributes resolve to inert classes."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_AnyModule,), {})
        setattr(self, name, cls)
        return cls


def _stub(name, **attrs):
    m = _Stub(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _knn_points(p1, p2, K=1, return_nn=False, **_):
    from oracle.th_oracle import knn_points_exact
    assert p1.shape[0] == 1 and p2.shape[0] == 1
    d2, idx = knn_points_exact(p1[0].float(), p2[0].float(), K)
    nn_pts = p2[0][idx] if return_nn else None
    return d2[None], idx[None], (nn_pts[None] if nn_pts is not None else None)


_LOADED = {}


def load_reference(num_class=300, n_samples=64, vit_depth=12, extra_opts=()):
    """Returns a dict of the reference's modules.  Can only be configured once
    per process (cfg is built at import time, lib/config/config.py:152-167)."""
    if _LOADED:
        return _LOADED
    assert reference_available(), "reference tree not mounted"
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from transhuman_amd.networks.encoder import ResNet18Trunk

    for n in ("open3d", "cv2", "chumpy", "trimesh", "mcubes", "imageio", "plyfile", "termcolor",
              "tensorboardX", "skimage", "skimage.metrics", "skimage.measure", "lpips", "easydict",
              "spconv", "spconv.pytorch", "spconv.pytorch.conv", "spconv.pytorch.core",
              "spconv.pytorch.identity", "spconv.pytorch.modules", "spconv.pytorch.ops",
              "spconv.pytorch.pool", "spconv.pytorch.tables", "spconv.pytorch.utils",
              "pytorch3d", "pytorch3d.ops", "pytorch3d.renderer", "pytorch3d.structures",
              "torchvision", "torchvision.models", "torchvision.transforms", "torchvision.utils"):
        if n not in sys.modules:
            _stub(n)
    sys.modules["spconv.pytorch.modules"].SparseSequential = _Seq
    sys.modules["spconv.pytorch.modules"].SparseModule = _AnyModule
    sys.modules["pytorch3d.ops"].knn_points = _knn_points
    sys.modules["torchvision.models"].resnet18 = lambda pretrained=False, norm_layer=None, **k: ResNet18Trunk(
        norm_layer=norm_layer)
    sys.modules["cv2"].Rodrigues = lambda *a, **k: None

    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = ["x", "--cfg_file", "configs/train_or_eval.yaml", "run_mode", "test", "perturb", "0",
                "num_class", str(num_class), "N_samples", str(n_samples), "vit_depth", str(vit_depth),
                "pretrained", "False", *extra_opts]
    os.chdir(REF)
    sys.path.insert(0, REF)
    # the reference hard-codes .cuda() (if_clight_renderer.py:180-181,194-195)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.current_device = lambda: "cpu"
    torch.cuda.empty_cache = lambda: None
    try:
        from lib.config import cfg
        from lib.networks import cross_transformer, vision_transformer, embedder, encoder
        from lib.networks.renderer import if_clight_renderer, if_mesh_renderer, nerf_net_utils
    finally:
        sys.argv = old_argv
    _LOADED.update(cfg=cfg, cross_transformer=cross_transformer, vision_transformer=vision_transformer,
                   embedder=embedder, encoder=encoder, if_clight_renderer=if_clight_renderer,
                   if_mesh_renderer=if_mesh_renderer, nerf_net_utils=nerf_net_utils, old_cwd=old_cwd)
    return _LOADED


def make_ref_renderer(mods, net, can_verts64, assign, mesh=False):
    """Build the reference Renderer without its constructor (needs the absent
    SMPL pickle, if_clight_renderer.py:43-48) -- set the same attributes by hand."""
    import numpy as np
    mod = mods["if_mesh_renderer"] if mesh else mods["if_clight_renderer"]
    r = mod.Renderer.__new__(mod.Renderer)
    r.net = net
    r.vertex_can = torch.tensor(can_verts64).contiguous()
    r.faces = np.zeros((1, 3), dtype=np.int64)
    r.CR = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    k = int(assign.max()) + 1
    r.dict_voxel2pc_ind = {c: np.where(assign == c)[0].tolist() for c in range(k)}
    r.pc2voxel_ind = torch.tensor(assign).type(torch.int64)
    r.voxel_PE_can = r.voxelization(r.dict_voxel2pc_ind, r.vertex_can)
    return r
