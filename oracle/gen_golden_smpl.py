"""Golden vectors for SMPL linear blend skinning, produced by the REFERENCE's SMPL._call
(lib/utils/SMPL.py:114-186) imported in the survey container, on the synthetic body model of
transhuman_amd/synth.py (the SMPL pickle is absent).  The rotation-matrix input form (:131-132) is used so that
no stand-in for cv2.Rodrigues is involved.  Writes tests/golden/g15_smpl.npz (outputs only; every input is
regenerated from seeds).

    python oracle/gen_golden_smpl.py        (needs /root/reference; test infrastructure, never shipped)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_harness, th_oracle as O  # noqa: E402
from transhuman_amd import synth  # noqa: E402


def main():
    mods = ref_harness.load_reference()
    from lib.utils import SMPL as ref_smpl                      # the reference module (cv2 is a stub: unused here)
    m = synth.make_smpl_model()
    s = ref_smpl.SMPL.__new__(ref_smpl.SMPL)                    # the constructor needs the absent pickle (:81-82)
    s.J_regressor, s.weights, s.posedirs = m["J_regressor"], m["weights"], m["posedirs"]
    s.v_template, s.shapedirs = m["v_template"], m["shapedirs"]
    s.parent = m["parent"][1:]
    pose, beta = synth.make_smpl_pose()
    R = np.array([O.rodrigues(p) for p in pose.reshape(-1, 3)], dtype="float32")
    v, joints, T = s(R, beta)                                   # SMPL.__call__ -> _call (:107-112)
    os.chdir(mods["old_cwd"])
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "g15_smpl.npz")
    np.savez_compressed(out, R=R, v=v, joints=joints, T_sub=T[::16], T_sum=T.sum(0), v_sum=v.sum(0))
    print("v", v.shape, v.dtype, "T", T.shape, T.dtype, "wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
