"""Golden for render_fast with cfg.white_bkgd = True: the REAL reference Renderer.render_fast on the
64x64 "large" frame of g11 (masked / chunked branch).  Rays that miss the hull are never composited by the reference
(if_clight_renderer.py:459-476: only the hit rays go through _render, the rest of the image stays zeros), so with a
white background a miss is BLACK, a hit ray gets rgb + (1 - acc).  Writes tests/golden/g11w_render_white.npz.

    python -m oracle.gen_golden_white        (survey container only: needs /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_harness as rh          # noqa: E402
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
    cfg.N_samples = 32
    r = rh.make_ref_renderer(mods, net, can64, assign)
    bb = synth.make_batch(64, 64, 3, seed=0, focal=210.0)
    cfg.white_bkgd = True
    try:
        with torch.no_grad():
            ret = r.render_fast({k: (v.clone() if torch.is_tensor(v) else v) for k, v in bb.items()}, is_train=False)
    finally:
        cfg.white_bkgd = False
    hit = ret["acc_map"][0] > 0
    print("rays", 64 * 64, "rays with acc > 0:", int(hit.sum()), "max rgb of the others:", float(ret["rgb_map"][0][~hit].abs().max()))
    os.chdir(mods["old_cwd"])
    save("g11w_render_white", rgb=ret["rgb_map"][0], acc=ret["acc_map"][0], depth=ret["depth_map"][0])


if __name__ == "__main__":
    main()
