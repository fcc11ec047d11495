"""Seeded synthetic inputs for the rendering hot path (no ZJU-MoCap / SMPL here).

Everything is generated with numpy's legacy ``RandomState`` (bit-stable across
numpy versions) so the same inputs can be rebuilt (a) in the survey container
next to the imported reference when golden vectors are produced
(oracle/gen_golden.py), (b) in the CPU tests, and (c) on the GPU box by
bench.py -- without shipping large fixtures.

Shapes follow the reference's ``batch`` dict (built at
/root/reference/lib/datasets/light_stage/can_smpl.py:537-594).
"""
import math
import zlib

import numpy as np
import torch

NV = 6890  # SMPL vertex count (lib/config/config.py:27)


# --------------------------------------------------------------------------
# deterministic weights
# --------------------------------------------------------------------------
_SKIP_SUFFIX = ("_freqs", "_phases", "num_batches_tracked")


def det_tensor(name, shape, kind="weight", fan_in=None, seed=0):
    """Deterministic fp32 tensor that depends only on (name, shape, seed)."""
    rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    n = int(np.prod(shape)) if len(shape) else 1
    u = rs.uniform(-1.0, 1.0, size=n).astype(np.float64)
    if kind == "weight":
        # He-uniform: keeps the activation scale through the ReLU stack so the
        # sigma head sees both signs (a 1/sqrt(fan_in) bound lets the signal die)
        a = math.sqrt(6.0 / max(1, fan_in))
        v = u * a
    elif kind == "norm":
        v = 1.0 + 0.1 * u
    elif kind == "var":
        v = 1.0 + 0.5 * u
    else:  # bias
        v = 0.1 * u
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def det_state_dict(state_dict, seed=0, sigma_bias=0.0):
    """Replace every learnable tensor of ``state_dict`` by a deterministic one.

    Buffers that are constants of the architecture (PE frequencies/phases,
    BN batch counters) are left untouched.  ``sigma_bias`` is added to
    ``alpha_fc.bias`` so a chosen fraction of samples has sigma > 0
    (SURVEY.md 8d "Weights").
    """
    out = {}
    for k, v in state_dict.items():
        if k.endswith(_SKIP_SUFFIX) or not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            t = det_tensor(k, shape, "var", seed=seed)
        elif k.endswith("running_mean"):
            t = det_tensor(k, shape, "bias", seed=seed)
        elif v.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = det_tensor(k, shape, "weight", fan_in=fan_in, seed=seed)
        elif k.endswith("weight"):
            t = det_tensor(k, shape, "norm", seed=seed)
        else:
            t = det_tensor(k, shape, "bias", seed=seed)
        if k == "alpha_fc.bias":
            t = t + float(sigma_bias)
        out[k] = t.to(v.dtype)
    return out


def heavy_tailed_state_dict(state_dict, seed=0, df=3.0, spike=3.0, sigma_gain=40.0, sigma_bias=0.0):
    """Weights shaped like a TRAINED network's rather than like an initialisation (the fp16 hi/lo split of the MFMA
    kernels loses bits on exactly what an initialisation lacks: a few entries / directions far above the bulk).
    Every matrix gets Student-t entries (``df`` = 3: single entries 5-20 x the typical one) scaled to the He
    variance plus a rank-2 term whose two singular values sit ``spike`` x above the bulk edge; biases are Student-t
    too.  ``alpha_fc`` is scaled by ``sigma_gain`` so sigma_raw spans roughly -200 .. 200 instead of O(1) (with the
    frame's sample spacing of ~5 mm that gives per-sample alphas from 0 to ~0.6: rays saturate inside the hull like a
    trained density does), ``sigma_bias`` is added to its bias.  Deterministic in (name, shape, seed)."""
    out = {}
    for k, v in state_dict.items():
        if k.endswith(_SKIP_SUFFIX) or not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        shape = tuple(v.shape)
        rs = np.random.RandomState((zlib.crc32(k.encode()) + 104729 * seed + 17) & 0x7FFFFFFF)
        if k.endswith("running_var"):
            t = 1.0 + 0.5 * rs.uniform(-1, 1, size=shape)
        elif k.endswith("running_mean"):
            t = 0.1 * rs.uniform(-1, 1, size=shape)
        elif v.dim() >= 2:
            n_out, fan_in = shape[0], int(np.prod(shape[1:]))
            w = rs.standard_t(df, size=(n_out, fan_in))
            w *= math.sqrt(2.0 / max(1, fan_in)) / math.sqrt(df / (df - 2.0))          # He variance
            # two dominant directions: singular value = spike x the bulk edge sigma (sqrt(n_out) + sqrt(fan_in))
            edge = math.sqrt(2.0 / max(1, fan_in)) * (math.sqrt(n_out) + math.sqrt(fan_in))
            for _ in range(2):
                a = rs.normal(size=n_out); a /= np.linalg.norm(a) + 1e-30
                b = rs.normal(size=fan_in); b /= np.linalg.norm(b) + 1e-30
                w += spike * edge * np.outer(a, b) * (1.0 if min(n_out, fan_in) > 8 else 0.0)
            t = w.reshape(shape)
        elif k.endswith("weight"):
            t = 1.0 + 0.2 * rs.standard_t(df, size=shape).clip(-4, 4)
        else:
            t = 0.1 * rs.standard_t(df, size=shape)
        if k == "alpha_fc.weight":
            t = t * float(sigma_gain)
        if k == "alpha_fc.bias":
            t = t * float(sigma_gain) + float(sigma_bias)
        out[k] = torch.from_numpy(np.asarray(t, dtype=np.float32).reshape(shape)).to(v.dtype)
    return out


# --------------------------------------------------------------------------
# body
# --------------------------------------------------------------------------
def _capsules():
    """16 capsules (p0, p1, radius) of a ~1.7 m standing figure, SMPL coords
    (y up, origin near the pelvis)."""
    c = []
    add = lambda a, b, r: c.append((np.array(a, np.float64), np.array(b, np.float64), r))
    add([0, -0.05, 0], [0, 0.25, 0], 0.13)      # lower torso
    add([0, 0.25, 0], [0, 0.48, 0], 0.14)       # upper torso
    add([0, 0.52, 0], [0, 0.60, 0.01], 0.05)    # neck
    add([0, 0.66, 0.02], [0, 0.76, 0.02], 0.09)  # head
    for s in (-1.0, 1.0):
        add([0.17 * s, 0.45, 0], [0.42 * s, 0.40, 0], 0.045)     # upper arm
        add([0.42 * s, 0.40, 0], [0.66 * s, 0.36, 0.03], 0.035)  # fore arm
        add([0.66 * s, 0.36, 0.03], [0.76 * s, 0.35, 0.04], 0.03)  # hand
        add([0.09 * s, -0.08, 0], [0.11 * s, -0.48, 0.01], 0.07)   # thigh
        add([0.11 * s, -0.48, 0.01], [0.12 * s, -0.88, 0], 0.05)   # shin
        add([0.12 * s, -0.90, 0], [0.13 * s, -0.92, 0.12], 0.04)   # foot
    return c


def make_body(seed=0, nv=NV):
    """nv points on the capsule union + per-point bone id.  float32 [nv,3]."""
    rs = np.random.RandomState(seed)
    caps = _capsules()
    area = np.array([2 * math.pi * r * np.linalg.norm(b - a) + 4 * math.pi * r * r for a, b, r in caps])
    cnt = np.floor(area / area.sum() * nv).astype(int)
    cnt[0] += nv - cnt.sum()
    pts, bone = [], []
    for bi, ((a, b, r), n) in enumerate(zip(caps, cnt)):
        L = np.linalg.norm(b - a)
        ax = (b - a) / L
        tmp = np.array([1.0, 0, 0]) if abs(ax[0]) < 0.9 else np.array([0, 1.0, 0])
        e1 = np.cross(ax, tmp); e1 /= np.linalg.norm(e1)
        e2 = np.cross(ax, e1)
        side = 2 * math.pi * r * L
        cap = 4 * math.pi * r * r
        on_side = rs.uniform(size=n) < side / (side + cap)
        t = rs.uniform(size=n) * L
        phi = rs.uniform(size=n) * 2 * math.pi
        p_side = a[None] + ax[None] * t[:, None] + r * (np.cos(phi)[:, None] * e1[None] + np.sin(phi)[:, None] * e2[None])
        g = rs.normal(size=(n, 3)); g /= np.linalg.norm(g, axis=1, keepdims=True)
        along = g @ ax
        p_cap = np.where(along[:, None] > 0, b[None], a[None]) + r * g
        pts.append(np.where(on_side[:, None], p_side, p_cap))
        bone.append(np.full(n, bi))
    pts = np.concatenate(pts).astype(np.float32)
    bone = np.concatenate(bone)
    return pts, bone


def _rodrigues(v):
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def make_blend_mtx(verts, seed=1):
    """Per-vertex float64 4x4 'blend matrices' (stand-in for SMPL LBS output,
    /root/reference/lib/utils/SMPL.py:114-186): softmax-blended rigid bone
    transforms with |axis-angle| <= 0.5."""
    rs = np.random.RandomState(seed)
    caps = _capsules()
    nb = len(caps)
    G = np.zeros((nb, 4, 4))
    for b in range(nb):
        aa = rs.uniform(-1, 1, size=3)
        aa = aa / np.linalg.norm(aa) * rs.uniform(0, 0.5)
        G[b, :3, :3] = _rodrigues(aa)
        G[b, :3, 3] = rs.uniform(-0.05, 0.05, size=3)
        G[b, 3, 3] = 1.0
    v = verts.astype(np.float64)
    d = np.zeros((v.shape[0], nb))
    for b, (a, c, r) in enumerate(caps):
        ac = c - a
        t = np.clip(((v - a) @ ac) / (ac @ ac), 0, 1)
        d[:, b] = np.linalg.norm(v - (a[None] + t[:, None] * ac[None]), axis=1)
    w = np.exp(-(d - d.min(1, keepdims=True)) / 0.05)
    w /= w.sum(1, keepdims=True)
    return np.einsum("vb,bij->vij", w, G)  # float64 [nv,4,4]


def kmeans_assign(verts, k, seed=3, iters=8):
    """Small deterministic k-means (farthest-point init + Lloyd) -> int32[nv]
    cluster id per vertex, every cluster non-empty, ids 0..k-1.  Stand-in for
    the reference's pre-computed kmeans_dict_*.npy on the synthetic body."""
    v = verts.astype(np.float64)
    n = v.shape[0]
    rs = np.random.RandomState(seed)
    centers = np.empty((k, 3))
    idx = rs.randint(n)
    dmin = np.full(n, np.inf)
    for i in range(k):
        centers[i] = v[idx]
        dmin = np.minimum(dmin, ((v - centers[i]) ** 2).sum(1))
        idx = int(np.argmax(dmin))
    for _ in range(iters):
        a = _nearest(v, centers)
        for i in range(k):
            m = a == i
            if m.any():
                centers[i] = v[m].mean(0)
    a = _nearest(v, centers)
    # repair empty clusters by stealing the farthest member of the largest one
    for i in range(k):
        if not (a == i).any():
            big = np.bincount(a, minlength=k).argmax()
            cand = np.where(a == big)[0]
            far = cand[np.argmax(((v[cand] - centers[big]) ** 2).sum(1))]
            a[far] = i
    return a.astype(np.int32)


def _nearest(v, c, blk=2048):
    out = np.empty(v.shape[0], np.int64)
    for s in range(0, v.shape[0], blk):
        d = ((v[s:s + blk, None, :] - c[None]) ** 2).sum(-1)
        out[s:s + blk] = d.argmin(1)
    return out


def csr_from_assign(assign, k=None):
    """pc2voxel_ind -> CSR (offsets int32[k+1], members int32[nv]).  Members of
    a cluster are in ascending vertex order, exactly the order of the lists in
    the reference's dict_voxel2pc_ind (verified for kmeans_dict_{300,500,1500})."""
    assign = np.asarray(assign).astype(np.int64)
    if k is None:
        k = int(assign.max()) + 1
    members = np.argsort(assign, kind="stable").astype(np.int32)
    counts = np.bincount(assign, minlength=k)
    offsets = np.zeros(k + 1, np.int32)
    offsets[1:] = np.cumsum(counts)
    return offsets, members


# --------------------------------------------------------------------------
# cameras / rays
# --------------------------------------------------------------------------
def make_cameras(H, W, V=3, center=(0.0, 0.0, 3.0), dist=3.0, focal=None):
    """Target camera (identity pose) + V reference cameras on a ring looking at
    ``center``.  Convention x_cam = R x + T (if_clight_renderer.py:123-126)."""
    f = focal if focal is not None else 600.0 * W / 512.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float64)
    c = np.array(center, np.float64)
    Rs, Ts = [], []
    for v in range(V):
        ang = 2 * math.pi * v / max(V, 1) + 0.3
        ca, sa = math.cos(ang), math.sin(ang)
        R = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])
        T = -R @ c + np.array([0, 0, dist])
        Rs.append(R); Ts.append(T.reshape(3, 1))
    return dict(K=K, R=np.eye(3), T=np.zeros((3, 1)),
                in_K=np.stack([K] * V), in_R=np.stack(Rs), in_T=np.stack(Ts))


def pixel_rays(H, W, K, R, T):
    """Origin and (un-normalised) direction of the ray through every pixel centre of an H x W pinhole camera with
    x_cam = R x + T: the synthetic batch's stand-in for the data loader's ray set-up (the reference's is
    lib/utils/if_nerf/if_nerf_data_utils.py:11-30; the hot-path version is csrc/k_rays.hip, K9).  The arithmetic
    keeps that function's operation order (float32 pixel grid, float64 products), because tests/golden/* were
    produced from batches made of these numbers."""
    Kinv_t = np.linalg.inv(K).T
    shift = T.ravel()
    eye = -np.dot(R.T, T).ravel()
    cols = np.arange(W, dtype=np.float32)
    rows = np.arange(H, dtype=np.float32)
    homog = np.empty((H, W, 3), np.float32)
    homog[..., 0] = cols[None, :]
    homog[..., 1] = rows[:, None]
    homog[..., 2] = 1.0
    world = np.dot(np.dot(homog, Kinv_t) - shift, R)
    dirs = world - eye[None, None]
    return np.broadcast_to(eye, dirs.shape), dirs


def box_interval(box, origin, direction):
    """Entry / exit depth of every ray against the axis-aligned box `box` ([2,3]: min corner, max corner) grown by
    1 cm, in units of |direction|, and the mask of rays that cross it (exactly two of the six face planes are met
    inside the box, tolerance 1e-6; direction components below 1e-5 in magnitude are set to 1e-5 first).  Same rule
    as the reference's data loader (if_nerf_data_utils.py:65-97), evaluated face by face."""
    grown = box + np.array([-0.01, 0.01])[:, None]
    direction = np.where(np.abs(direction) < 1e-5, 1e-5, direction)
    lo, hi = grown[0] - 1e-6, grown[1] + 1e-6
    length = np.linalg.norm(direction, axis=1)
    n = origin.shape[0]
    met = np.zeros((n, 6), bool)
    depth = np.zeros((n, 6), np.float64)
    for side in range(2):
        for axis in range(3):
            t = (grown[side, axis] - origin[:, axis]) / direction[:, axis]
            point = t[:, None] * direction + origin
            inside = np.ones(n, bool)
            for a in range(3):
                inside &= (point[:, a] >= lo[a]) & (point[:, a] <= hi[a])
            k = 3 * side + axis
            met[:, k] = inside
            depth[:, k] = np.linalg.norm(point - origin, axis=1) / length
    crosses = met.sum(1) == 2
    near = np.where(met, depth, np.inf).min(1)[crosses]
    far = np.where(met, depth, -np.inf).max(1)[crosses]
    return near, far, crosses


def smooth_noise(shape, seed, passes=2):
    """~N(0,1) low-pass-filtered noise, float32 (feature-map stand-in)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal(size=shape).astype(np.float32)
    for _ in range(passes):
        x = (x + np.roll(x, 1, -1) + np.roll(x, -1, -1)) / 3.0
        x = (x + np.roll(x, 1, -2) + np.roll(x, -1, -2)) / 3.0
    x /= x.std()
    return x.astype(np.float32)


# --------------------------------------------------------------------------
# the batch dict
# --------------------------------------------------------------------------
def make_batch(H=64, W=64, V=3, seed=0, all_rays=True, dense=False, nv=NV, focal=None, dilate=3):
    """Synthetic ``batch`` with the reference's keys/shapes/dtypes (B=1).

    all_rays=True  : every pixel is a ray (benchmark convention);
                     rays that miss the bbox get near=far=body depth.
    dense=True     : 'S-dense' regime -- near/far of every ray clamped to a thin
                     slab hugging the front surface so (nearly) every sample of
                     hit rays lies inside the 0.1 m hull.
    dilate         : (dense) radius in pixels within which a ray takes the depth of the nearest
                     projected vertex.  With a long lens on the torso (focal=6000: the 512^2 window is
                     0.26 m wide at the body) and dilate=64 EVERY ray gets such a slab: the
                     "S-dense" frame of SURVEY 8d, all R x S samples valid.
    """
    verts_s, _ = make_body(seed, nv)                    # SMPL coords, posed
    blend = make_blend_mtx(verts_s, seed + 1)           # float64 [nv,4,4]
    ang = 0.15
    Rh = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]], np.float32)
    Th = np.array([[0.03, 0.10, 3.0]], np.float32)
    # world2smpl is q=(p-Th)Rh  (if_clight_renderer.py:289-295)  => p=q Rh^-1+Th
    verts_w = (verts_s.astype(np.float64) @ np.linalg.inv(Rh.astype(np.float64)) + Th).astype(np.float32)
    cams = make_cameras(H, W, V, center=tuple(Th[0].tolist()), focal=focal)
    ray_o, ray_d = pixel_rays(H, W, cams["K"], cams["R"], cams["T"])
    ray_o = ray_o.reshape(-1, 3).astype(np.float32)
    ray_d = ray_d.reshape(-1, 3).astype(np.float32)
    bmin, bmax = verts_w.min(0), verts_w.max(0)
    bounds = np.stack([bmin - 0.05, bmax + 0.05]).astype(np.float32)
    near_b, far_b, at_box = box_interval(bounds.astype(np.float64), ray_o.astype(np.float64), ray_d.astype(np.float64))
    R = ray_o.shape[0]
    if all_rays:
        near = np.full(R, float(Th[0, 2]) - 0.5, np.float32)
        far = np.full(R, float(Th[0, 2]) + 0.5, np.float32)
        near[at_box] = near_b.astype(np.float32)
        far[at_box] = far_b.astype(np.float32)
    else:
        ray_o, ray_d = ray_o[at_box], ray_d[at_box]
        near, far = near_b.astype(np.float32), far_b.astype(np.float32)
        R = ray_o.shape[0]
    if dense:
        # thin slab right in front of the body surface along each ray: depth of
        # the nearest vertex within 3 px of the ray, +-0.04 m
        uvw = (cams["K"] @ verts_w.T.astype(np.float64)).T
        px = np.round(uvw[:, 0] / uvw[:, 2]).astype(int).clip(0, W - 1)
        py = np.round(uvw[:, 1] / uvw[:, 2]).astype(int).clip(0, H - 1)
        zbuf = np.full((H, W), np.inf)
        np.minimum.at(zbuf, (py, px), uvw[:, 2])
        for _ in range(dilate):  # dilate
            z2 = zbuf.copy()
            for dy, dx in ((0, 1), (1, 0), (0, -1), (-1, 0)):
                z2 = np.minimum(z2, np.roll(zbuf, (dy, dx), (0, 1)))
            zbuf = z2
        zb = zbuf.reshape(-1)
        if not all_rays:
            zb = zb[at_box]
        hit = np.isfinite(zb)
        near = np.where(hit, zb - 0.03, near).astype(np.float32)
        far = np.where(hit, zb + 0.05, far).astype(np.float32)
    vizmap = np.zeros((V, nv), bool)
    c = Th[0].astype(np.float64)
    for v in range(V):
        rel = (cams["in_R"][v] @ (verts_w.astype(np.float64) - c).T).T
        vizmap[v] = rel[:, 2] < 0.03
    imgs = (smooth_noise((V, 3, H, W), seed + 11, passes=3) * 0.25 + 0.5).clip(0, 1).astype(np.float32)
    t = torch.from_numpy
    batch = {
        "ray_o": t(ray_o)[None], "ray_d": t(ray_d)[None],
        "near": t(near)[None], "far": t(far)[None],
        "tar_smpl_vertice": t(verts_w)[None],
        "tar_smpl_vertice_smplcoord": t(verts_s)[None],
        "Rh": t(Rh)[None], "Th": t(Th)[None],
        "blend_mtx": t(blend)[None],
        "input_imgs": [t(imgs)[None]],
        "input_vizmaps": [t(vizmap)[None]],
        "input_R": [t(cams["in_R"].astype(np.float32))[None]],
        "input_T": [t(cams["in_T"].astype(np.float32))[None]],
        "input_K": [t(cams["in_K"].astype(np.float32))[None]],
        "input_smpl_vertice": [t(verts_w)[None]],
        "input_blend_mtx": [t(blend)[None]],
        "input_smpl_vertice_smplcoord": [t(verts_s)[None]],
        "can_bounds": t(bounds)[None],
        "mask_at_box": t(at_box)[None],
        "H": H, "W": W,
    }
    return batch


def make_grid_pts(batch, n=32):
    """[1,n,n,n,3] world-space voxel centres over the body AABB (mesh workload,
    /root/reference/lib/datasets/light_stage/can_smpl_mesh.py:25-97)."""
    b = batch["can_bounds"][0].numpy().astype(np.float64)
    ax = [np.linspace(b[0, i], b[1, i], n) for i in range(3)]
    g = np.stack(np.meshgrid(*ax, indexing="ij"), -1).astype(np.float32)
    return torch.from_numpy(g)[None]


def batch_to(batch, device):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.to(device)
        elif isinstance(v, list):
            out[k] = [x.to(device) if torch.is_tensor(x) else x for x in v]
        else:
            out[k] = v
    return out


# --------------------------------------------------------------------------
# synthetic SMPL-shaped body model (the real SMPL pickle is not redistributable / absent)
# --------------------------------------------------------------------------
SMPL_PARENT = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int64)


def make_smpl_model(seed=5, nv=NV):
    """Arrays with the shapes / dtypes of the SMPL pickle fields lib/utils/SMPL.py:83-89 reads (float64):
    v_template [nv,3], shapedirs [nv,3,10], posedirs [nv,3,207], J_regressor [24,nv] (rows sum to 1),
    weights [nv,24] (rows sum to 1), parent [24] (the SMPL kinematic tree).  Geometry = the capsule body."""
    rs = np.random.RandomState(seed)
    v, _ = make_body(0, nv)
    v = v.astype(np.float64)
    # 24 joint seeds spread over the body: farthest-point sample of the vertices, ordered root-first by height
    idx = [int(np.argmin(np.abs(v[:, 1] - np.median(v[:, 1])) + np.abs(v[:, 0])))]
    dmin = ((v - v[idx[0]]) ** 2).sum(1)
    for _ in range(23):
        idx.append(int(np.argmax(dmin)))
        dmin = np.minimum(dmin, ((v - v[idx[-1]]) ** 2).sum(1))
    jpos = v[idx]
    d = np.linalg.norm(v[:, None, :] - jpos[None], axis=2)                 # [nv,24]
    w = np.exp(-(d - d.min(1, keepdims=True)) / 0.06)
    w[w < 1e-4] = 0.0
    w /= w.sum(1, keepdims=True)
    jr = np.exp(-(d.T / 0.05) ** 2)                                        # [24,nv]
    jr[jr < 1e-3] = 0.0
    jr /= jr.sum(1, keepdims=True)
    shapedirs = (rs.standard_normal((nv, 3, 10)) * 0.004 + v[:, :, None] * rs.uniform(-0.02, 0.02, (1, 1, 10)))
    posedirs = rs.standard_normal((nv, 3, 207)) * 0.002
    return dict(v_template=v, shapedirs=shapedirs, posedirs=posedirs, J_regressor=jr, weights=w,
                parent=SMPL_PARENT.copy())


def make_smpl_pose(seed=7, scale=0.35):
    """(poses [1,72] axis-angle float32 like the annots, betas [10] float64)"""
    rs = np.random.RandomState(seed)
    pose = (rs.uniform(-1, 1, (24, 3)) * scale).astype(np.float32)
    pose[5] = 0.0                                                        # one exact-zero rotation (the theta = 0 branch)
    beta = rs.uniform(-1.5, 1.5, 10)
    return pose.reshape(1, 72), beta
