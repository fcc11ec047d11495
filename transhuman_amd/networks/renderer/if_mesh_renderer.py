"""Mesh-reconstruction renderer: density cube over a voxel grid.

Drop-in for /root/reference/lib/networks/renderer/if_mesh_renderer.py
(`Renderer.render`, :46-113).  The hull mask, DPaRF, pixel gather and the
sigma branch of the MLP run through th_eval_sigma_grid; the reference also
evaluates (and discards) RGB for sigma>0 voxels (:84-99) -- `cube` only needs
sigma_raw, which is what is produced here.  Marching cubes (PyMCubes, :103)
stays a host step outside the hot path: it runs only if `mcubes` is importable.
"""
import os
import sys

# Drop-in loading: the reference instantiates this file through imp.load_source(cfg.<x>_module, cfg.<x>_path)
# (lib/networks/make_network.py:4-11, renderer/make_renderer.py:4-8) under WHATEVER module name the YAML gives --
# 'lib.networks.cross_transformer' if only the path key is changed.  Relative imports would then resolve inside the
# reference's `lib` package, so the package is imported absolutely, found through this file's own location.
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

import numpy as np                                                  # noqa: E402

from transhuman_amd.config import get_cfg                           # noqa: E402
from transhuman_amd import hip                                      # noqa: E402
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer as Base_Renderer   # noqa: E402


class Renderer(Base_Renderer):
    def render(self, batch, frame=None, pts_slice=None):
        cfg = get_cfg()
        pts = batch["pts"]                                           # [1,X,Y,Z,3]
        sh = pts.shape
        frame = frame if frame is not None else self.prepare_frame(batch)
        flat = pts.reshape(-1, 3)
        if pts_slice is not None:
            flat = flat[pts_slice]
        sigma, stats = hip.eval_sigma_grid(self.net, frame, flat)
        self.last_stats = stats
        if pts_slice is not None:
            return {"sigma": sigma}
        cube = sigma.view(*sh[1:4]).detach().cpu().numpy()           # :99-100
        cube = np.pad(cube, 10, mode="constant")                     # :101
        mesh = None
        try:                                                         # :103-109 (host, third-party)
            import mcubes
            import trimesh
            vertices, triangles = mcubes.marching_cubes(cube, cfg.mesh_th)
            can_bounds = batch["can_bounds"][0].cpu().numpy()
            LB = (can_bounds[0] - 10 * np.array(cfg.voxel_size))[None, ...]
            mesh = trimesh.Trimesh(vertices * np.array(cfg.voxel_size)[None, ...] + LB, triangles)
        except ImportError:
            pass
        return {"cube": cube, "mesh": mesh}
