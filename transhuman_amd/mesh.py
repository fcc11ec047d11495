"""Triangle mesh container + binary PLY writer + the evaluator's PSNR -- the consumers of the mesh / eval workloads
(SURVEY 8f-4).

The reference hands `mcubes.marching_cubes`' output to `trimesh.Trimesh(vertices_world, triangles)`
(lib/networks/renderer/if_mesh_renderer.py:109) and the visualiser calls `mesh.export('<frame>.ply')`
(lib/visualizers/if_nerf_mesh.py:25-35); neither package is installed here.  ``Mesh`` keeps the two attributes the
reference's code touches (``vertices`` float64 [nv,3], ``faces`` int64 [nt,3]) and ``export`` writes the same file
layout trimesh's PLY exporter produces (binary little-endian, float x/y/z per vertex, uchar-counted int32 index
lists per face), so downstream tools read either.  ``psnr_metric`` is lib/evaluators/if_nerf.py:34-37 on device
tensors (one reduction kernel, no host copy of the image).
"""
import numpy as np
import torch


class Mesh:
    def __init__(self, vertices, faces):
        self.vertices = vertices        # [nv,3] float64 (device tensor or ndarray)
        self.faces = faces              # [nt,3] integer

    def _host(self):
        v = self.vertices.detach().cpu().numpy() if torch.is_tensor(self.vertices) else np.asarray(self.vertices)
        f = self.faces.detach().cpu().numpy() if torch.is_tensor(self.faces) else np.asarray(self.faces)
        return np.asarray(v, dtype=np.float64).reshape(-1, 3), np.asarray(f).reshape(-1, 3)

    @property
    def is_watertight(self):
        """every undirected edge is shared by exactly two triangles"""
        _, f = self._host()
        if f.shape[0] == 0:
            return False
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64)
        e.sort(axis=1)
        _, counts = np.unique(e, axis=0, return_counts=True)
        return bool((counts == 2).all())

    def export(self, path):
        """binary little-endian PLY (trimesh's layout: float32 positions, `list uchar int` faces)"""
        v, f = self._host()
        header = ("ply\nformat binary_little_endian 1.0\ncomment transhuman_amd marching cubes\n"
                  f"element vertex {v.shape[0]}\nproperty float x\nproperty float y\nproperty float z\n"
                  f"element face {f.shape[0]}\nproperty list uchar int vertex_indices\nend_header\n")
        face_rec = np.empty(f.shape[0], dtype=[("n", "u1"), ("idx", "<i4", (3,))])
        face_rec["n"] = 3
        face_rec["idx"] = f.astype("<i4")
        with open(path, "wb") as fh:
            fh.write(header.encode("ascii"))
            fh.write(v.astype("<f4").tobytes())
            fh.write(face_rec.tobytes())
        return path


def read_ply(path):
    """reader for the files Mesh.export writes (tests / tools) -> (vertices float32 [nv,3], faces int32 [nt,3])"""
    with open(path, "rb") as fh:
        nv = nt = 0
        while True:
            line = fh.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                nv = int(line.split()[-1])
            elif line.startswith("element face"):
                nt = int(line.split()[-1])
            elif line == "end_header":
                break
        v = np.frombuffer(fh.read(12 * nv), dtype="<f4").reshape(nv, 3)
        rec = np.frombuffer(fh.read(13 * nt), dtype=[("n", "u1"), ("idx", "<i4", (3,))])
    assert (rec["n"] == 3).all()
    return v, rec["idx"]


def psnr_metric(img_pred, img_gt):
    """lib/evaluators/if_nerf.py:34-37: -10 log10(mean((pred - gt)^2)) -- tensors (any device) or ndarrays"""
    if torch.is_tensor(img_pred):
        d = img_pred.to(torch.float64) - torch.as_tensor(img_gt, device=img_pred.device).to(torch.float64)
        mse = float((d * d).mean())
    else:
        mse = float(np.mean((np.asarray(img_pred, np.float64) - np.asarray(img_gt, np.float64)) ** 2))
    return -10.0 * np.log(mse) / np.log(10.0)
