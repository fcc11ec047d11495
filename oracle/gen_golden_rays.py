"""Golden vectors for the ray-generation row, produced by the REFERENCE's own functions
(lib/utils/if_nerf/if_nerf_data_utils.py:11-30 get_rays, :65-97 get_near_far, test split of
sample_ray_h36m :271-283) imported in the survey container.  Writes tests/golden/g14_rays.npz.

    python oracle/gen_golden_rays.py        (needs /root/reference; test infrastructure, never shipped)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_harness  # noqa: E402


def cases():
    """(name, H, W, K, R, T, bounds) -- float32 like can_smpl.py:640-645 / :216-232 produce them."""
    out = []
    # 1: camera on the optical axis of the box: the centre pixel has d = (0,0,1) -> the |d| < 1e-5 clamp (:70)
    K = np.array([[75.0, 0, 24.0], [0, 75.0, 32.0], [0, 0, 1]], np.float32)
    R = np.eye(3, dtype=np.float32)
    T = np.zeros((3, 1), np.float32)
    b = np.array([[-0.35, -0.9, 2.6], [0.4, 0.85, 3.3]], np.float32)
    out.append(("axis", 64, 48, K, R, T, b))
    # 2: oblique camera, box partly outside the image, non-square pixels
    a, e = 0.7, -0.25
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(e), -np.sin(e)], [0, np.sin(e), np.cos(e)]])
    R2 = (Rx @ Ry).astype(np.float32)
    c = np.array([0.05, 0.0, 3.0])
    T2 = (-(R2.astype(np.float64) @ c) + np.array([0.3, -0.1, 2.4])).reshape(3, 1).astype(np.float32)
    K2 = np.array([[90.0, 0.4, 30.5], [0, 84.0, 20.25], [0, 0, 1]], np.float32)
    out.append(("oblique", 40, 56, K2, R2, T2, b))
    return out


def main():
    mods = ref_harness.load_reference()
    from lib.utils.if_nerf import if_nerf_data_utils as du      # the reference module
    arrs = {}
    for name, H, W, K, R, T, b in cases():
        ray_o, ray_d = du.get_rays(H, W, K, R, T)
        # test split of sample_ray_h36m (:271-283)
        ray_o = ray_o.reshape(-1, 3).astype(np.float32)
        ray_d = ray_d.reshape(-1, 3).astype(np.float32)
        near, far, mask = du.get_near_far(b, ray_o, ray_d)      # mutates ray_d (|d| < 1e-5 -> 1e-5)
        arrs.update({f"{name}_K": K, f"{name}_R": R, f"{name}_T": T, f"{name}_bounds": b,
                     f"{name}_HW": np.array([H, W]), f"{name}_ray_o": ray_o[mask], f"{name}_ray_d": ray_d[mask],
                     f"{name}_ray_d_all": ray_d, f"{name}_near": near.astype(np.float32),
                     f"{name}_far": far.astype(np.float32), f"{name}_mask": mask})
        print(name, H, W, "rays in box:", int(mask.sum()), "of", H * W, "near", near.min(), near.max())
    os.chdir(mods["old_cwd"])
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "g14_rays.npz")
    np.savez_compressed(out, **arrs)
    print("wrote", out, os.path.getsize(out), "bytes")
    # virtual camera path of the free-viewpoint video (lib/utils/render_utils.py:318-364, numpy only): the
    # reference's own function on the synthetic capture rig
    from lib.utils import render_utils as ru
    from transhuman_amd.camera_path import synthetic_rig
    rig = synthetic_rig()
    path60 = np.array(ru.gen_path_virt([m.copy() for m in rig], render_views=60))
    centre = np.array([0.0, 0.1, 3.0])
    path7c = np.array(ru.gen_path_virt([m.copy() for m in rig], center=centre, render_views=7))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "g16_path.npz")
    np.savez_compressed(out, rig=np.array(rig), path60=path60, centre=centre, path7c=path7c)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
