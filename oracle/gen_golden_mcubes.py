"""Golden data for the marching-cubes row.  PyMCubes (the reference's mcubes.marching_cubes,
if_mesh_renderer.py:103) is not installed anywhere in this image; scikit-image 0.18.3 -- an independent third-party
implementation of the same published algorithm (classic Lorensen & Cline mode, method='_lorensen') -- happens to be
present in the image's /opt/conda Python 3.9.  It is run HERE (survey container) on small seeded volumes and its
output is committed as tests/golden/g17_mcubes.npz: vertex set (sorted), triangle count, surface area, enclosed
volume.  The classic mode splits the quads of some cases along the other diagonal than the Bourke-table traversal
PyMCubes uses, so triangles are compared through those invariants, vertices exactly.

    python oracle/gen_golden_mcubes.py      (test infrastructure; needs /opt/conda/bin/python3.9 with scikit-image)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from transhuman_amd import synth  # noqa: E402


def volumes():
    out = {}
    x, y, z = np.mgrid[-1:1:24j, -1:1:20j, -1:1:28j]
    out["ellipsoid"] = ((0.7 - np.sqrt(x * x + 1.3 * y * y + z * z)).astype(np.float32), 0.0)
    # two blobs joined by a thin neck + a hole: ambiguous faces / saddle cases occur
    x, y, z = np.mgrid[-1.2:1.2:30j, -1:1:26j, -1:1:22j]
    f = np.exp(-((x - 0.45) ** 2 + y * y + z * z) * 6) + np.exp(-((x + 0.45) ** 2 + y * y + z * z) * 6) \
        - 0.8 * np.exp(-((x) ** 2 + (y - 0.1) ** 2 + z * z) * 40)
    out["blobs"] = (f.astype(np.float32), 0.35)
    # smooth noise at a density-like level (many components, cells of every case), padded like the sigma cube (:101)
    n = synth.smooth_noise((1, 1, 20 * 18, 16), 77)[0, 0].reshape(20, 18, 16) * 30.0 + 18.0
    out["noise_padded"] = (np.pad(n.astype(np.float32), 3, mode="constant"), 20.0)
    return out


SK = r"""
import sys, numpy as np
from skimage.measure import marching_cubes
d = np.load(sys.argv[1]); out = {}
for k in d.files:
    if k.endswith('_iso'): continue
    v, f = marching_cubes(d[k], float(d[k + '_iso']), method='_lorensen')
    out[k + '_v'] = v; out[k + '_f'] = f
np.savez(sys.argv[2], **out)
"""


def main():
    vols = volumes()
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(inp, **{k: v for k, (v, _) in vols.items()}, **{k + "_iso": np.float64(i) for k, (_, i) in vols.items()})
        subprocess.run(["/opt/conda/bin/python3.9", "-W", "ignore", "-c", SK, inp, outp], check=True)
        sk = np.load(outp)
        arrs = {}
        for k, (vol, iso) in vols.items():
            v, f = sk[k + "_v"].astype(np.float64), sk[k + "_f"].astype(np.int64)
            a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
            area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
            volume = abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
            order = np.lexsort((v[:, 2], v[:, 1], v[:, 0]))
            arrs.update({k + "_vol": vol, k + "_iso": np.float64(iso), k + "_verts_sorted": v[order].astype(np.float32),
                         k + "_ntri": np.int64(f.shape[0]), k + "_area": np.float64(area), k + "_volume": np.float64(volume)})
            print(k, vol.shape, "verts", v.shape[0], "tris", f.shape[0], "area", area, "volume", volume)
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "g17_mcubes.npz")
    np.savez_compressed(out, **arrs)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
