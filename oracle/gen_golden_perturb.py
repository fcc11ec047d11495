"""Golden for the two sampling randomisations of the reference's inference entry (VERDICT r4 "missing" #3): the REAL reference
Renderer.render_fast with cfg.perturb = 1 in train() mode (stratified depth jitter, if_clight_renderer.py:276-283) and
cfg.raw_noise_std = 0.4 (density noise, nerf_net_utils.py:39-44).  The draws the reference takes from torch.rand / torch.randn
are recorded (the two functions are wrapped for the duration of the call) and stored with the result, so that the oracle and the
HIP path can be run on the same draws.  Two frames: 32 x 32 x 32 (R' <= 2400: the un-masked branch) and 64 x 64 x 16 (masked).
Writes tests/golden/g19_perturb_{small,large}.npz.

    python -m oracle.gen_golden_perturb        (survey container only: needs /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_harness as rh          # noqa: E402
from oracle import th_oracle as O             # noqa: E402
from oracle.gen_golden import save, SIGMA_BIAS  # noqa: E402
from transhuman_amd import synth              # noqa: E402


def main():
    torch.set_num_threads(8)
    mods = rh.load_reference(num_class=300, n_samples=32)
    cfg = mods["cfg"]
    cfg.vit_depth = 12
    torch.manual_seed(0)
    net = mods["cross_transformer"].Network()
    net.load_state_dict(synth.det_state_dict(net.state_dict(), seed=0, sigma_bias=SIGMA_BIAS))
    net.train()
    body, _ = synth.make_body(0)
    assign = np.load(os.path.join(REPO, "tests", "golden", "synth_assign.npz"))["assign_300"].astype(np.int64)
    can64 = body.astype(np.float64) * 1.02 + 0.001
    r = rh.make_ref_renderer(mods, net, can64, assign)
    out = {}
    for tag, H, S, focal in (("small", 32, 32, None), ("large", 64, 16, 210.0)):
        cfg.N_samples = S
        bb = synth.make_batch(H, H, 3, seed=0, focal=focal)
        draws = {"rand": [], "randn": []}
        rand0, randn0 = torch.rand, torch.randn

        def rand(*a, **k):
            t = rand0(*a, **k)
            draws["rand"].append(t.clone())
            return t

        def randn(*a, **k):
            t = randn0(*a, **k)
            draws["randn"].append(t.clone())
            return t
        cfg.perturb, cfg.raw_noise_std = 1.0, 0.4
        torch.manual_seed(7)
        torch.rand, torch.randn = rand, randn
        try:
            with torch.no_grad():
                ret = r.render_fast({k: (v.clone() if torch.is_tensor(v) else v) for k, v in bb.items()}, is_train=False)
        finally:
            torch.rand, torch.randn = rand0, randn0
            cfg.perturb, cfg.raw_noise_std = 0.0, 0.0
        assert len(draws["rand"]) == 1 and len(draws["randn"]) == 1, (len(draws["rand"]), len(draws["randn"]))
        t_rand = draws["rand"][0][0]                                   # [R, S]
        R = t_rand.shape[0]
        # which rays the reference composited (the rows of its [R', S] noise): the hull test on the jittered points, through the
        # same exact K = 1 search the harness binds pytorch3d's knn_points to
        pts, z = O.sampling_points(bb["ray_o"][0], bb["ray_d"][0], bb["near"][0], bb["far"][0], S, t_rand)
        hit = O.hull_mask(pts.reshape(-1, 3), bb["tar_smpl_vertice"][0]).view(R, S).sum(-1) > 0
        noise_c = draws["randn"][0] * 0.4
        assert noise_c.shape == (int(hit.sum()), S), (noise_c.shape, int(hit.sum()))
        draw_full = torch.zeros(R, S)
        draw_full[hit] = draws["randn"][0]
        print(tag, "rays", R, "hit", int(hit.sum()), "rgb max", float(ret["rgb_map"].abs().max()))
        out[tag] = dict(t_rand=t_rand, raw_noise=draw_full, z_vals=z, hit=hit.to(torch.uint8), rgb=ret["rgb_map"][0],
                        acc=ret["acc_map"][0], depth=ret["depth_map"][0], noise_std=torch.tensor(0.4), n_samples=torch.tensor(S),
                        H=torch.tensor(H), focal=torch.tensor(-1.0 if focal is None else focal))
    os.chdir(mods["old_cwd"])
    for tag, d in out.items():
        save(f"g19_perturb_{tag}", **d)


if __name__ == "__main__":
    main()
