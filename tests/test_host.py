"""CPU: host-side logic -- config mirror, checkpoint compatibility, CSR construction,
synthetic-input generator, ray sharding, FLOP accounting."""
import numpy as np
import torch

from transhuman_amd import synth
from transhuman_amd.dist import shard_ray_indices, shard_lengths
from util import make_net, real_assign, synth_assign


def test_state_dict_keys_match_reference_layout():
    sd = make_net(12).state_dict()
    assert len(sd) == 310                      # reference has these + 85 dead xyzc_net.* entries
    for k in ("ViT.cls_token", "ViT.PE._freqs", "ViT.blocks.11.mlp.fc2.bias", "ViT.norm.weight",
              "encoder.model.layer4.1.bn2.running_var", "encoder.reduction_layer.weight", "encoder.PE_color._phases",
              "spatial_key_value_0.key_embed.weight", "spatial_key_value_1.value_embed.bias", "PE_relative._freqs",
              "fc_0.weight", "alpha_fc.bias", "view_fc.weight", "rgb_res_1.bias"):
        assert k in sd, k
    assert tuple(sd["fc_0.weight"].shape) == (256, 255, 1)
    assert tuple(sd["view_fc.weight"].shape) == (128, 283, 1)
    assert tuple(sd["encoder.reduction_layer.weight"].shape) == (192, 384, 1, 1)


def test_reference_checkpoint_with_dead_spconv_keys_loads_strict():
    net = make_net(2)
    sd = dict(net.state_dict())
    sd["xyzc_net.conv0.0.weight"] = torch.zeros(27, 192, 64)          # spconv tensors of the real checkpoint
    sd["xyzc_net.conv0.1.running_mean"] = torch.zeros(64)
    net.load_state_dict(sd, strict=True)                              # must not raise
    sd["not_a_key"] = torch.zeros(1)
    try:
        net.load_state_dict(sd, strict=True)
        raise AssertionError("unexpected keys must still be rejected")
    except RuntimeError:
        pass


def test_cfg_side_effects_of_network_ctor():
    from transhuman_amd.config import get_cfg
    make_net(2)
    cfg = get_cfg()
    assert cfg.embed_size == 192 and cfg.img_feat_size == 384       # cross_transformer.py:91,:123


def test_csr_matches_dict_order_of_reference_kmeans_files():
    for k in (300, 500, 1500):
        a = real_assign(k)
        off, mem = synth.csr_from_assign(a)
        assert len(off) == k + 1 and off[-1] == 6890
        for c in (0, k // 2, k - 1):
            m = mem[off[c]:off[c + 1]]
            assert (a[m] == c).all() and (np.diff(m) > 0).all()       # ascending vertex ids, like the dict lists
        sizes = np.diff(off)
        assert sizes.min() >= 1


def test_synthetic_clusters_cover_every_vertex_once():
    a = synth_assign(500)
    off, mem = synth.csr_from_assign(a)
    assert sorted(mem.tolist()) == list(range(6890))


def test_make_batch_schema():
    b = synth.make_batch(16, 16, 3, seed=0)
    assert b["ray_o"].shape == (1, 256, 3) and b["near"].shape == (1, 256)
    assert b["blend_mtx"].dtype == torch.float64 and b["blend_mtx"].shape == (1, 6890, 4, 4)
    assert b["input_imgs"][0].shape == (1, 3, 3, 16, 16) and b["input_vizmaps"][0].dtype == torch.bool
    assert b["input_R"][0].shape == (1, 3, 3, 3) and b["input_T"][0].shape == (1, 3, 3, 1)
    assert (b["far"] > b["near"]).all()
    b2 = synth.make_batch(16, 16, 3, seed=0)
    assert torch.equal(b["tar_smpl_vertice"], b2["tar_smpl_vertice"])           # deterministic


def test_det_state_dict_depends_only_on_name_and_shape():
    a = synth.det_tensor("fc_0.weight", (4, 5, 1), "weight", fan_in=5)
    b = synth.det_tensor("fc_0.weight", (4, 5, 1), "weight", fan_in=5)
    c = synth.det_tensor("fc_1.weight", (4, 5, 1), "weight", fan_in=5)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_ray_tiles_partition_the_frame():
    for (H, W, n) in ((512, 512, 8), (64, 48, 3), (40, 40, 5)):
        idx = [shard_ray_indices(H, W, n, r) for r in range(n)]
        allidx = torch.cat(idx)
        assert allidx.numel() == H * W and torch.equal(torch.sort(allidx)[0], torch.arange(H * W))
    lens = shard_lengths(512, 512, 8)
    assert len(set(lens)) == 1 and lens[0] == 512 * 512 // 8             # balanced at the BASELINE frame size
    # interleaving: a 64x64 block of pixels touches every rank
    tid = shard_ray_indices(512, 512, 8, 3)
    ys, xs = tid // 512, tid % 512
    assert ((ys < 64) & (xs < 64)).sum() == 64 * 64 // 8


def test_algorithmic_flop_accounting():
    import bench
    assert bench.algorithmic_mlp_flops(3, 1, 0) == 2 * 1543040
    assert bench.algorithmic_mlp_flops(3, 0, 1) == 2 * 764416
    assert bench.algorithmic_mlp_flops(1, 1, 1) == 2 * 823424


def test_gen_path_virt_vs_reference_golden():
    """the virtual camera orbit of the free-viewpoint video (render_utils.py:318-364): our restatement against the
    reference's own function on the synthetic 21-camera rig (tests/golden/g16_path.npz, oracle/gen_golden_rays.py)"""
    import os
    from transhuman_amd.camera_path import gen_path_virt, synthetic_rig
    from util import GOLD
    g = np.load(os.path.join(GOLD, "g16_path.npz"))
    rig = synthetic_rig()
    assert np.array_equal(np.array(rig), g["rig"]), "the rig generator must reproduce the golden's input"
    before = [m.copy() for m in rig]
    p60 = np.array(gen_path_virt(rig, render_views=60))
    assert all(np.array_equal(a, b) for a, b in zip(rig, before)), "the input list must not be mutated"
    assert p60.shape == (60, 4, 4) and np.abs(p60 - g["path60"]).max() < 1e-12
    p7 = np.array(gen_path_virt(rig, center=g["centre"], render_views=7))
    assert np.abs(p7 - g["path7c"]).max() < 1e-12
    # properties: rigid world-to-camera matrices, a closed orbit around the rig centre
    for m in p60:
        assert np.abs(m[:3, :3] @ m[:3, :3].T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(m[:3, :3]) - 1) < 1e-12
        assert np.array_equal(m[3], [0, 0, 0, 1])
    centres = np.array([-m[:3, :3].T @ m[:3, 3] for m in p60])
    rig_c = np.array([-m[:3, :3].T @ m[:3, 3] for m in rig]).mean(0)
    d = np.linalg.norm(centres - rig_c, axis=1)
    assert d.min() > 1.5 and d.max() < 5.0


def test_evaluator_psnr_images_and_files(tmp_path):
    """lib/evaluators/if_nerf.py:34-37, :41-62, :121-130, :146-170: MSE / PSNR of the masked ray list, the cropped
    full-frame images and the files run.py --type evaluate leaves behind"""
    from PIL import Image
    from transhuman_amd.evaluator import Evaluator, bounding_rect, to_uint8
    from transhuman_amd.config import get_cfg
    from oracle import th_oracle as O
    cfg = get_cfg()
    H = W = 24
    rs = np.random.RandomState(0)
    mask = np.zeros((H, W), bool)
    mask[5:17, 8:20] = rs.uniform(size=(12, 12)) < 0.8
    mask[5, 8] = mask[16, 19] = True
    n = int(mask.sum())
    gt = rs.uniform(size=(n, 3))
    pred = np.clip(gt + rs.normal(0, 0.02, size=(n, 3)), 0, 1)
    batch = {"rgb": torch.from_numpy(gt)[None], "mask_at_box": torch.from_numpy(mask.reshape(-1))[None],
             "human_name": ["CoreView_313"], "frame_index": torch.tensor([7]), "cam_ind": torch.tensor([3])}
    ev = Evaluator(result_dir=str(tmp_path / "res"))
    r = ev.evaluate({"rgb_map": torch.from_numpy(pred)[None]}, batch, H, W)
    assert abs(r["psnr"] - O.psnr_metric(pred, gt)) < 1e-12 and abs(r["mse"] - np.mean((pred - gt) ** 2)) < 1e-15
    assert bounding_rect(mask) == (8, 5, 12, 12)
    img = np.array(Image.open(tmp_path / "res" / "CoreView_313" / "pred" / "frame7_view3.png"))
    assert img.shape == (12, 12, 3) and img.dtype == np.uint8
    full = np.zeros((H, W, 3))
    full[mask] = pred
    assert np.array_equal(img, to_uint8(full[5:17, 8:20]))
    assert (tmp_path / "res" / "CoreView_313" / "gt" / "frame7_view3_gt.png").exists()
    s = ev.summarize()
    assert abs(s["psnr"] - r["psnr"]) < 1e-12 and np.load(tmp_path / "res" / "psnr.npy").shape == (1,)
    old = cfg.white_bkgd
    cfg.white_bkgd = True
    try:
        p2, _ = ev.images(pred, gt, batch, H, W)
        assert (p2[~mask[5:17, 8:20]] == 1.0).all()
    finally:
        cfg.white_bkgd = old


def test_training_entry_needs_a_device_too():
    """Renderer.render with gradients enabled (the reference trainer's call, if_nerf_clight.py:45) is served by the
    differentiable torch form of the path -- which is plain torch and WOULD run on a CPU batch; the method refuses that
    loudly like every inference entry does (no CPU path behind the boundary; tests/test_train_path.py drives the module
    directly for its CPU pin against the reference)."""
    import pytest
    from transhuman_amd.config import get_cfg
    from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
    from util import can64
    get_cfg().num_class = 300
    r = Renderer(make_net(2), vertex_can=can64().numpy(), pc2voxel_ind=synth_assign(300))
    from transhuman_amd import hip
    with pytest.raises(hip.HipError, match="MI355X"):
        r.render(synth.make_batch(8, 8, 1, seed=0))


def test_synthetic_rays_match_the_oracle_restatement_of_the_loader():
    """synth.pixel_rays / box_interval (the generator's own ray set-up) give the numbers of the oracle's restatement of
    the reference data loader (oracle/th_oracle.py, pinned to the reference by g14_rays): the committed goldens were
    made from batches built on exactly these values."""
    from oracle import th_oracle as O
    cams = synth.make_cameras(48, 40, 3, center=(0.03, 0.10, 3.0))
    K, R, T = cams["in_K"][1], cams["in_R"][1], cams["in_T"][1]
    bounds = np.array([[-0.4, -0.8, 2.6], [0.45, 0.95, 3.4]], np.float32)
    K, R, T = K.astype(np.float32), R.astype(np.float32), T.astype(np.float32)
    ref = O.gen_rays(48, 40, K, R, T, bounds)
    o, d = synth.pixel_rays(48, 40, K, R, T)
    o = o.reshape(-1, 3).astype(np.float32)
    d = d.reshape(-1, 3).astype(np.float32)
    near, far, m = synth.box_interval(bounds, o, d)
    assert m.sum() > 100 and np.array_equal(m, ref["mask_at_box"])
    assert np.array_equal(near.astype(np.float32), ref["near"][m]) and np.array_equal(far.astype(np.float32), ref["far"][m])
    assert np.array_equal(o, ref["ray_o"])
    assert np.array_equal(np.where(np.abs(d) < 1e-5, np.float32(1e-5), d), ref["ray_d"])


def test_bench_reads_the_committed_hbm_traffic():
    """bench.py takes roofline.traffic from profiles/hbm_traffic.json (written by tools/collect_profiles.sh): the file must parse
    and carry the fused MLP's bytes per launch from a summary that is itself under profiles/"""
    import os
    import bench
    t = bench.hbm_traffic()
    assert t is not None and t["mlp_fused_bytes_per_launch"] > 1e9 and t["launch_samples"] >= 524288
    assert t["source"].startswith("profiles/") and os.path.exists(os.path.join(os.path.dirname(bench.__file__), t["source"]))
    # bytes per launch scale with the samples of a launch: one launch over n samples = n / launch_samples of the measured one
    n = 2 * t["launch_samples"]
    blk = bench.roofline_block(1, 5.0e14, 9.4e12, 17.9, 1.0, n_valid=n)
    assert abs(blk["traffic"] - 2 * t["mlp_fused_bytes_per_launch"]) < 1.0
    assert "mlp_fused8_kernel<3>" in blk["kernel"] or "mlp_fused_kernel<3,1" in blk["kernel"]
    blk4 = bench.roofline_block(1, 5.0e14, 9.4e12, 17.9, 4.0, n_valid=n)
    assert abs(blk4["traffic"] - 0.5 * t["mlp_fused_bytes_per_launch"]) < 1.0
    assert blk["traffic_per_frame"]["measured_K4_K5_K6_bytes"] > 2 * t["mlp_fused_bytes_per_launch"]
