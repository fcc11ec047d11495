"""Virtual camera path of the free-viewpoint video workload (BASELINE configs[2]).

``gen_path_virt`` restates /root/reference/lib/utils/render_utils.py:318-364 (used once per sequence by
lib/datasets/light_stage/can_smpl_perform.py:40-42: the target camera of frame i is
``render_w2c[i % len(render_w2c)]``): an elliptical orbit fitted to the rig's camera centres -- the average "up"
of the rig, the 80th percentile of the camera offsets x 1.3 as radii -- looking 1.3 units above the rig centre
along the up axis.  Host numpy like the reference's (60 4x4 matrices once per sequence: nothing to put on a
device); pinned by goldens produced with the reference's own function (oracle/gen_golden_rays.py,
tests/golden/g16_path.npz).  ``synthetic_rig`` is the stand-in for annots.npy's 21 light-stage cameras.
"""
import math

import numpy as np


def _unit(x):
    return x / np.linalg.norm(x)


def _look(z, up, pos):
    """camera-to-world [3,4] with the given backward axis z, an approximate up vector and a position (:225-231)"""
    z = _unit(z)
    x = _unit(np.cross(z, up))
    y = _unit(np.cross(x, z))
    return np.stack([y, x, z, pos], 1)


def gen_path_virt(RT, center=None, render_views=None):
    """RT: sequence of world-to-camera 4x4 matrices (OpenCV convention, x_cam = R x + T) of the capture rig.
    -> list of ``render_views`` world-to-camera 4x4 float64 matrices (:318-364)."""
    bottom = np.array([[0.0, 0.0, 0.0, 1.0]])
    c2w_all = np.linalg.inv(np.array(RT, dtype=np.float64))                     # camera-to-world of every rig camera
    # (x, y, z) -> (y, x, -z): the path is built in the "up first" convention of the LLFF spiral code (:325-326)
    c2w_all = np.concatenate([c2w_all[:, :, 1:2], c2w_all[:, :, 0:1], -c2w_all[:, :, 2:3], c2w_all[:, :, 3:4]], 2)
    up = _unit(c2w_all[:, :3, 0].sum(0))
    z0 = _unit(c2w_all[0, :3, 2])
    a1 = _unit(np.cross(z0, up))
    a2 = _unit(np.cross(up, a1))
    lift = 0.0
    if center is None:
        center = c2w_all[:, :3, 3].mean(0)
        lift = 1.3
    frame = np.stack([up, a1, a2, center], 1)                                   # [3,4]: orbit frame, origin = rig centre
    # rig camera centres in the orbit frame -> radii (80th percentile of |offset| per axis, x 1.3)
    local = np.matmul(frame[:3, :3].T, (c2w_all[:, :3, 3] - frame[:3, 3])[..., np.newaxis])[..., 0].T
    rads = np.percentile(np.abs(local), 80, -1) * 1.3
    rads = np.array(list(rads) + [1.0])
    out = []
    for theta in np.linspace(0.0, 2 * np.pi, render_views + 1)[:-1]:
        pos = np.dot(frame[:3, :4], np.array([0, np.sin(theta), np.cos(theta), 1]) * rads)
        target = np.dot(frame[:3, :4], np.array([lift, 0, 0, 1.0]))
        m = _look(pos - target, up, pos)
        m = np.concatenate([m[:, 1:2], m[:, 0:1], -m[:, 2:3], m[:, 3:4]], 1)    # back to (x, y, z)
        out.append(np.linalg.inv(np.concatenate([m, bottom], 0)))
    return out


def synthetic_rig(n=21, centre=(0.03, 0.10, 3.0), radius=2.8, height=1.3, seed=5):
    """A light-stage-like ring of ``n`` world-to-camera 4x4 matrices around ``centre`` (the synthetic body of
    transhuman_amd.synth stands at (0.03, 0.10, 3.0), world +y pointing down like the image rows): cameras on a
    slightly irregular circle mounted ``height`` above the subject's centre and looking down at it, y down / z
    forward like the ZJU-MoCap calibration -- gen_path_virt aims its orbit 1.3 units "below" the mean camera
    position (:331-333, :352-353), i.e. at the subject for such a rig."""
    rs = np.random.RandomState(seed)
    c = np.asarray(centre, np.float64)
    out = []
    for i in range(n):
        ang = 2 * math.pi * i / n + rs.uniform(-0.04, 0.04)
        r = radius * (1.0 + rs.uniform(-0.05, 0.05))
        pos = c + np.array([r * math.sin(ang), -height * (1.0 + rs.uniform(-0.08, 0.08)), -r * math.cos(ang)])
        fwd = _unit(c - pos)
        right = _unit(np.cross(np.array([0.0, -1.0, 0.0]), fwd) * -1.0)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], 0)                                     # rows = camera axes in world coords
        T = -R @ pos
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = R, T
        out.append(m)
    return out
