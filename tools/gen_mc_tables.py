#!/usr/bin/env python
"""Writes transhuman_amd/csrc/mc_tables.h from tests/golden/mc_case_table.npz (the published 256 x 16 marching-cubes
triangle table, Bourke / Bloyd numbering; the .npz was extracted from scikit-image 0.18.3's copy of it,
skimage/measure/_marching_cubes_lewiner_luts.py CASESCLASSIC, and checked for consistency with the corner / edge
numbering: every case lists exactly the edges its corner pattern cuts)."""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tri = np.load(os.path.join(ROOT, "tests", "golden", "mc_case_table.npz"))["tri"]
EV = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
for c in range(256):
    used = set(int(e) for e in tri[c] if e >= 0)
    cut = set(e for e, (a, b) in enumerate(EV) if ((c >> a) & 1) != ((c >> b) & 1))
    assert used == cut, c
out = ["// Triangle table of the classic marching-cubes algorithm (Lorensen & Cline 1987) in the corner / edge numbering of",
       "// Paul Bourke's \"Polygonising a scalar field\" (1994; table by Cory Gene Bloyd) -- the table PyMCubes'",
       "// mcubes.marching_cubes (the call at if_mesh_renderer.py:103) and scikit-image's classic mode ship.  Published",
       "// data, generated into this header by tools/gen_mc_tables.py; row c lists, three at a time, the cut edges that",
       "// form the triangles of case c (bit m of c set <=> corner m is at or below the iso level), -1 terminated.",
       "#pragma once", "static const signed char MC_TRI_TABLE[256][16] = {"]
out += ["    {" + ", ".join(f"{int(v):2d}" for v in tri[c]) + "}," for c in range(256)]
nt = [int((tri[c] >= 0).sum()) // 3 for c in range(256)]
out += ["};", "static const unsigned char MC_NUM_TRIS[256] = {"]
out += ["    " + ", ".join(str(v) for v in nt[r:r + 32]) + "," for r in range(0, 256, 32)]
out += ["};"]
open(os.path.join(ROOT, "transhuman_amd", "csrc", "mc_tables.h"), "w").write("\n".join(out) + "\n")
