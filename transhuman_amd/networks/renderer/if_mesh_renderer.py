"""Mesh-reconstruction renderer: density cube over a voxel grid.

Drop-in for /root/reference/lib/networks/renderer/if_mesh_renderer.py
(`Renderer.render`, :46-113).  The hull mask, DPaRF, pixel gather and the
sigma branch of the MLP run through th_eval_sigma_grid; the reference also
evaluates (and discards) RGB for sigma>0 voxels (:84-99) -- `cube` only needs
sigma_raw, which is what is produced here.  Marching cubes (PyMCubes, :103)
stays a host step outside the hot path: it runs only if `mcubes` is importable.
"""
import numpy as np

from ...config import get_cfg
from ... import hip
from .if_clight_renderer import Renderer as Base_Renderer


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
