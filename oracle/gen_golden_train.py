"""Golden for the training entry (Renderer.render with gradients): the REAL reference, imported here on CPU, runs one
forward / backward of its trainer's call pattern (lib/train/trainers/if_nerf_clight.py:45 ``self.renderer.render(batch)``,
:83-86 image loss, trainer.py:83 ``loss.backward()``) on a synthetic patch of <= 2400 rays; outputs, loss and the
gradients of a spread of parameters are stored in tests/golden/g18_train_step.npz.

TEST INFRASTRUCTURE, runs only where /root/reference is mounted:
    python -m oracle.gen_golden_train
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_harness as rh          # noqa: E402
from transhuman_amd import synth              # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SIGMA_BIAS = -1.7
# parameters whose gradients are stored (a spread over the encoder, TransHE and the per-point network)
GRAD_KEYS = ("encoder.model.conv1.weight", "encoder.model.layer2.0.conv1.weight", "encoder.model.layer1.0.bn1.weight",
             "encoder.upsample_color.weight", "encoder.reduction_layer.bias", "ViT.blocks.0.attn.qkv.weight",
             "ViT.blocks.1.mlp.fc2.bias", "ViT.norm.weight", "fc_0.weight", "alpha_res_0.weight",
             "spatial_key_value_0.key_embed.weight", "spatial_key_value_1.value_embed.bias", "fc_2.weight", "fc_3.bias",
             "alpha_fc.weight", "feature_fc.weight", "view_fc.weight", "rgb_res_1.weight", "fc_4.weight", "rgb_fc.bias")


def main():
    torch.set_num_threads(8)
    mods = rh.load_reference(num_class=300, n_samples=16, vit_depth=2)
    cfg = mods["cfg"]
    cfg.vit_depth, cfg.N_samples, cfg.perturb, cfg.raw_noise_std = 2, 16, 0.0, 0.0
    torch.manual_seed(0)
    net = mods["cross_transformer"].Network()
    net.load_state_dict(synth.det_state_dict(net.state_dict(), seed=0, sigma_bias=SIGMA_BIAS))
    net.train()
    body, _ = synth.make_body(0)
    assign = np.load(os.path.join(REPO, "tests", "golden", "synth_assign.npz"))["assign_300"].astype(np.int64)
    can64 = body.astype(np.float64) * 1.02 + 0.001
    r = rh.make_ref_renderer(mods, net, can64, assign)
    b = synth.make_batch(20, 20, 3, seed=0, all_rays=False, focal=62.5)
    R = b["ray_o"].shape[1]
    assert 100 < R <= 2400
    rs = np.random.RandomState(18)
    target = torch.from_numpy(rs.uniform(size=(1, R, 3)).astype(np.float32))
    ret = r.render({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    # NetworkWrapper.forward :83-86 (img_loss on mask_at_box = every ray of the patch) + terms that reach acc / depth too
    loss = torch.mean((ret["rgb_map"] - target) ** 2) + 0.1 * ret["acc_map"].mean() + 0.01 * ret["depth_map"].mean()
    loss.backward()
    params = dict(net.named_parameters())
    out = {"rays": np.int64(R), "rgb": ret["rgb_map"][0].detach().numpy(), "acc": ret["acc_map"][0].detach().numpy(),
           "depth": ret["depth_map"][0].detach().numpy(), "loss": np.float64(loss.item()), "target": target[0].numpy()}
    for k in GRAD_KEYS:
        g = params[k].grad
        assert g is not None and float(g.abs().max()) > 0.0, k
        out["grad:" + k] = g.numpy()
        print(f"  {k:50s} |g| max {float(g.abs().max()):.3e}")
    os.chdir(mods["old_cwd"])
    np.savez_compressed(os.path.join(OUT, "g18_train_step.npz"), **out)
    print("rays", R, "loss", float(loss), "acc max", float(ret["acc_map"].max()),
          os.path.getsize(os.path.join(OUT, "g18_train_step.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
