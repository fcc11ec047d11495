"""The drop-in boundary exercised the way the reference exercises it: a global ``lib.config.cfg`` built from
the YAML, ``imp.load_source(cfg.<x>_module, cfg.<x>_path)`` on the two plugin files
(/root/reference/lib/networks/make_network.py:4-11, renderer/make_renderer.py:4-8), ``Network()`` with no arguments,
``Renderer(network)`` reading ``./data/smplx/smpl/SMPL_NEUTRAL.pkl`` and ``./kmeans_dict/kmeans_dict_{N}.npy`` from the
current directory (/root/reference/lib/networks/renderer/if_clight_renderer.py:43-73).  The files in the tmp cwd are
synthetic stand-ins in the reference's formats (the SMPL pickle is licensed and absent; the kmeans pickle carries the
reference's real cluster assignment from tests/golden/kmeans_pc2voxel.npz)."""
import importlib.machinery
import importlib.util
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from util import real_assign, can64

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NET_PATH = os.path.join(REPO, "transhuman_amd", "networks", "cross_transformer.py")
REN_PATH = os.path.join(REPO, "transhuman_amd", "networks", "renderer", "if_clight_renderer.py")
MESH_PATH = os.path.join(REPO, "transhuman_amd", "networks", "renderer", "if_mesh_renderer.py")


def load_source(module, path):
    """imp.load_source (make_network.py:9); `imp` is gone in Python 3.12 -- same loader through importlib there"""
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)
            import imp
        return imp.load_source(module, path)
    except ImportError:
        loader = importlib.machinery.SourceFileLoader(module, path)
        spec = importlib.util.spec_from_loader(module, loader)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[module] = mod
        loader.exec_module(mod)
        return mod


def reference_cfg(num_class=300, module_prefix="lib.networks"):
    """the keys of configs/train_or_eval.yaml the path reads, as a yacs-like attribute bag"""
    return types.SimpleNamespace(
        cross_transformer_network_module=f"{module_prefix}.cross_transformer", cross_transformer_network_path=NET_PATH,
        renderer_module=f"{module_prefix}.renderer.if_clight_renderer", renderer_path=REN_PATH,
        N_samples=32, perturb=0.0, raw_noise_std=0.0, white_bkgd=False, run_mode="test", H=1024, W=1024, ratio=0.5,
        time_steps=1, rasterize=True, depth_map=False, depth_vizmap=False, embed_size=192, img_feat_size=256,
        xyz_res=10, view_res=4, pretrained=False, num_class=num_class, vit_depth=12, KNN=7, KNN_FREQ=10,
        KNN_DIST_ALPHA=0.5, KNN_SIGMA=0.25, use_truncation=False, voxel_size=[0.005, 0.005, 0.005], mesh_th=20,
        exp_name="dropin", data_root="data/zju_mocap")


@pytest.fixture
def reference_env(tmp_path, monkeypatch):
    """a fake `lib.config` in sys.modules + a cwd laid out like the reference tree"""
    cfg = reference_cfg()
    lib = types.ModuleType("lib")
    lib.__path__ = []
    conf = types.ModuleType("lib.config")
    conf.cfg = cfg
    lib.config = conf
    monkeypatch.setitem(sys.modules, "lib", lib)
    monkeypatch.setitem(sys.modules, "lib.config", conf)
    # SMPL pickle: dict with 'v_template' (float64 [6890,3]) and 'f' (faces), latin1-loadable (:44-47)
    smpl_dir = tmp_path / "data" / "smplx" / "smpl"
    smpl_dir.mkdir(parents=True)
    v_template = can64().numpy()
    faces = np.stack([np.arange(0, 6888), np.arange(1, 6889), np.arange(2, 6890)], axis=1).astype(np.uint32)
    with open(smpl_dir / "SMPL_NEUTRAL.pkl", "wb") as f:
        pickle.dump({"v_template": v_template, "f": faces}, f, protocol=2)
    # kmeans pickle in the reference's format (:55): {'pc2voxel_ind': int32 [6890], 'dict_voxel2pc_ind': {int32: list}}
    assign = real_assign(300).astype(np.int32)
    voxel2pc = {np.int32(k): [int(i) for i in np.nonzero(assign == k)[0]] for k in range(300)}
    (tmp_path / "kmeans_dict").mkdir()
    np.save(tmp_path / "kmeans_dict" / "kmeans_dict_300.npy", {"pc2voxel_ind": assign, "dict_voxel2pc_ind": voxel2pc},
            allow_pickle=True)
    monkeypatch.chdir(tmp_path)
    yield cfg
    for name in list(sys.modules):
        if name.startswith("lib.networks"):
            del sys.modules[name]


@pytest.mark.parametrize("prefix", ["lib.networks", "transhuman_amd.networks"])
def test_plugin_files_load_like_the_reference_loads_them(reference_env, prefix):
    """make_network / make_renderer's own three lines, with the YAML's DEFAULT module names (only the path keys
    changed) and with the names INTEGRATION.md suggests"""
    cfg = reference_env
    cfg.cross_transformer_network_module = f"{prefix}.cross_transformer"
    cfg.renderer_module = f"{prefix}.renderer.if_clight_renderer"
    network = load_source(cfg.cross_transformer_network_module, cfg.cross_transformer_network_path).Network()
    assert cfg.embed_size == 192 and cfg.img_feat_size == 384          # side effects of cross_transformer.py:91,:123
    assert len(network.state_dict()) == 310
    network.train()
    renderer = load_source(cfg.renderer_module, cfg.renderer_path).Renderer(network)
    assert renderer.net is network
    assert renderer.faces.shape == (6888, 3)
    assert renderer.vertex_can.dtype == torch.float64 and tuple(renderer.vertex_can.shape) == (6890, 3)
    assert renderer.pc2voxel_ind.dtype == torch.int64 and renderer.num_clusters == 300
    # cluster pooling over the file's lists, in the file's order (:73, :362-369)
    assign = real_assign(300)
    want = torch.stack([renderer.vertex_can[torch.as_tensor(np.nonzero(assign == k)[0])].mean(0) for k in range(300)])
    assert renderer.voxel_PE_can.dtype == torch.float64
    assert torch.equal(renderer.voxel_PE_can, want)
    pe = renderer.normalize_PE(renderer.voxel_PE_can[None])
    assert pe.dtype == torch.float32 and float(pe.abs().max()) <= 1.0
    for name in ("render_fast", "render", "voxelization", "normalize_PE"):
        assert callable(getattr(renderer, name))


def test_mesh_renderer_plugin_and_reference_checkpoint_keys(reference_env):
    cfg = reference_env
    cfg.renderer_module, cfg.renderer_path = "lib.networks.renderer.if_mesh_renderer", MESH_PATH
    network = load_source(cfg.cross_transformer_network_module, cfg.cross_transformer_network_path).Network()
    renderer = load_source(cfg.renderer_module, cfg.renderer_path).Renderer(network)
    assert renderer.num_clusters == 300 and callable(renderer.render)
    # a reference checkpoint carries the dead xyzc_net.* entries: strict load must accept them (run.py:24-28)
    sd = dict(network.state_dict())
    sd["xyzc_net.conv0.0.weight"] = torch.zeros(3, 3, 3, 192, 64)
    network.load_state_dict(sd, strict=True)
    missing = dict(sd)
    del missing["fc_0.weight"]
    with pytest.raises(RuntimeError):
        network.load_state_dict(missing, strict=True)


def test_unsupported_config_raises(reference_env):
    cfg = reference_env
    cfg.KNN = 5
    with pytest.raises(NotImplementedError):
        load_source(cfg.cross_transformer_network_module, cfg.cross_transformer_network_path).Network()


@pytest.mark.gpu
def test_reference_loading_renders_like_the_injected_renderer(reference_env, gpu):
    """the renderer built through the reference's loader from the files in the cwd renders the same frame (bit for
    bit) as the one the rest of the suite builds with injected arrays; run.py's sequence: make_network().cuda(),
    load_state_dict(strict), .train(), make_renderer(), render_fast(batch, is_train=False)"""
    from transhuman_amd import synth
    from util import make_sd
    cfg = reference_env
    network = load_source(cfg.cross_transformer_network_module, cfg.cross_transformer_network_path).Network().cuda()
    network.load_state_dict(make_sd(), strict=True)
    network.train()
    renderer = load_source(cfg.renderer_module, cfg.renderer_path).Renderer(network)
    b = synth.batch_to(synth.make_batch(48, 48, 3, seed=0, focal=150.0), gpu)
    out = renderer.render_fast(b, is_train=False)
    from transhuman_amd.networks.renderer.if_clight_renderer import Renderer as Injected
    ref = Injected(network, vertex_can=can64().numpy(), pc2voxel_ind=real_assign(300)).render_fast(b, is_train=False)
    assert renderer.last_stats["hit_rays"] > 100
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert out[k].shape == ref[k].shape and torch.equal(out[k], ref[k]), k


@pytest.mark.gpu
def test_trainer_wrapper_validation_step(reference_env, gpu):
    """The reference's training-side caller of the boundary, restated: lib/train/trainers/if_nerf_clight.py:24-31 builds
    ``if_clight_renderer.Renderer(self.net)`` directly (module import at :4, not through the YAML path), :45 calls
    ``self.renderer.render(batch)`` and :83-86 forms the image loss on ``batch['mask_at_box']``; Trainer.val
    (trainer.py:131-150) runs that wrapper with ``network.eval()`` under ``torch.no_grad()`` -- evaluation during
    training.  With the renderer module swapped for this repository's both run: the train step through the differentiable
    form of Renderer.render (transhuman_amd.networks.autograd_path: loss.backward() reaches every trained parameter), the
    validation step on the HIP kernels (eval-mode BatchNorm included: th_bn_act_eval)."""
    from transhuman_amd import synth
    from util import make_sd
    cfg = reference_env
    mod = load_source(cfg.renderer_module, cfg.renderer_path)
    network = load_source(cfg.cross_transformer_network_module, cfg.cross_transformer_network_path).Network().cuda()
    network.load_state_dict(make_sd(), strict=True)

    class NetworkWrapper(torch.nn.Module):                       # if_nerf_clight.py:24-104, non-patch branch
        def __init__(self, net):
            super().__init__()
            self.net = net
            self.renderer = mod.Renderer(self.net)               # :29
            self.img2mse = lambda x, y: torch.mean((x - y) ** 2)   # :31

        def forward(self, batch):
            ret = self.renderer.render(batch)                    # :45
            mask = batch["mask_at_box"]                          # :83
            img_loss = self.img2mse(ret["rgb_map"][mask], batch["rgb"][mask])
            return ret, img_loss, {"img_loss": img_loss, "loss": img_loss}, {}

    wrapper = NetworkWrapper(network)
    bc = synth.make_batch(24, 24, 3, seed=0, all_rays=False, focal=75.0)
    R = bc["ray_o"].shape[1]
    bc["rgb"] = torch.rand(1, R, 3)
    bc["mask_at_box"] = torch.ones(1, R, dtype=torch.bool)
    b = synth.batch_to(bc, gpu)
    # Trainer.train (trainer.py:79-86): forward, loss.backward(), gradient clipping, optimizer.step()
    wrapper.train()
    opt = torch.optim.Adam(wrapper.parameters(), lr=1e-4)
    out_t, loss_t, _, _ = wrapper(b)
    opt.zero_grad()
    loss_t.mean().backward()
    torch.nn.utils.clip_grad_value_(wrapper.parameters(), 40)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in network.named_parameters()
               if n.startswith(("fc_", "alpha_", "rgb_", "view_fc", "feature_fc", "spatial_key_value", "encoder.model.conv1",
                                "encoder.model.layer1", "encoder.model.layer2", "ViT.blocks", "ViT.norm")))
    opt.step()
    # Trainer.val (trainer.py:131-150)
    wrapper.eval()
    with torch.no_grad():
        out, loss, stats, _ = wrapper(b)
        out2, loss2, _, _ = wrapper(b)
    assert out["rgb_map"].shape == (1, R, 3) and torch.isfinite(out["rgb_map"]).all() and torch.isfinite(loss)
    # eval mode: no state changes between two calls (the stem runs through torch's stock convolutions here, and MIOpen is not
    # bit-reproducible from call to call: compare to 1e-6, not torch.equal)
    assert float((out["rgb_map"] - out2["rgb_map"]).abs().max()) < 1e-6 and abs(float(loss) - float(loss2)) < 1e-6
    assert float(out["acc_map"].max()) > 0.0
    # the same call with the network in train() (run.py:29's state): batch statistics in the stem, HIP kernels end to end
    wrapper.train()
    with torch.no_grad():
        out3, loss3, _, _ = wrapper(b)
    assert torch.isfinite(out3["rgb_map"]).all() and torch.isfinite(loss3)
