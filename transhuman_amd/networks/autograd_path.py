"""The training entry of the boundary: ``Renderer.render`` with gradients.

The reference trains through ``Renderer.render(batch)`` (lib/train/trainers/if_nerf_clight.py:45, trainer.py:79-86:
``loss.backward()`` through the whole path, patches of <= 2400 rays).  The HIP kernels of this repository carry no
autograd, so a call with gradients enabled (or with the training-time randomisations ``cfg.perturb`` /
``cfg.raw_noise_std``) is served by this module instead: the same forward composed from differentiable torch operators
on the module's own parameters, on the device the batch lives on (torch's ROCm kernels on an MI355X).  It is NOT the
rendering hot path and not a fallback of it -- inference calls (``render_fast``, ``render_sequence``, ``render`` under
``no_grad``) never come here; SURVEY 7 hard-part 6 planned exactly this split.  No backward HIP kernels exist.

Pinned by ``oracle/gen_golden_train.py`` (the real reference imported in the survey container: outputs and parameter
gradients of one training step's forward/backward on a synthetic patch -> ``tests/golden/g18_train_step.npz``) and
``tests/test_train_path.py``.  Every function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

from ..config import get_cfg


# ---- sampling ----------------------------------------------------------------------------------------------------------
def sample_depths(near, far, n_samples, perturb):
    """if_clight_renderer.py:271-287: z = near (1 - t) + far t; with ``perturb`` one uniform sample per stratum."""
    t = torch.linspace(0.0, 1.0, steps=n_samples).to(near)
    z = near[..., None] * (1.0 - t) + far[..., None] * t
    if perturb:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * torch.rand(z.shape).to(upper)
    return z


def embed_view(ray_d, view_res):
    """if_clight_renderer.py:525-526 + embedder.py:9-35: [v, sin(2^k v), cos(2^k v)], k < view_res."""
    v = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
    out = [v]
    for k in range(view_res):
        out += [torch.sin(v * (2.0 ** k)), torch.cos(v * (2.0 ** k))]
    return torch.cat(out, -1)


# ---- encoder, painting, grouping ----------------------------------------------------------------------------------------
def encode(enc, images):
    """SpatialEncoder.forward (encoder.py:97-155) through torch's own modules (autograd)."""
    H, W = images.shape[2:]
    lat = enc.trunk(images, fused_bn=False)
    lat = [F.interpolate(l, (H, W), mode="bilinear", align_corners=True) for l in lat]
    pix = torch.cat(lat + [enc.upsample_color(images)], dim=1)
    return enc.reduction_layer(pix), pix


def project(x, R, T, K):
    """x [N,3] -> uv [V,N,2] (if_clight_renderer.py:123-126, :228-232)."""
    cam = torch.einsum("vij,nj->vni", R, x) + T[:, None, :, 0]
    pix = torch.einsum("vij,vnj->vni", K, cam)
    return pix[..., :2] / pix[..., 2:]


def sample_map(feat, uv, enc, image_shape):
    """sample_from_feature_map (:186-208): grid_sample, bilinear, align_corners, border.  feat [V,C,H,W] -> [V,C,N]."""
    from .. import hip
    H, W = feat.shape[2:]
    scale = hip.feat_scale(enc.feat_scale(H, W), image_shape, feat.device)       # :193-195 (float64 divide, fp32 cast)
    g = (uv * scale - 1.0).unsqueeze(2)
    return F.grid_sample(feat, g, align_corners=True, mode="bilinear", padding_mode="border")[:, :, :, 0]


_pool_cache = {}


def pooling_matrix_cached(renderer, n_verts, device, dtype):
    """pooling_matrix of the renderer's clustering, built once per (renderer, device, dtype): the dense [N_c, 6890] matrix is
    a constant of the renderer, not of the call"""
    key = (id(renderer), int(n_verts), str(device), dtype)
    ent = _pool_cache.get(key)
    if ent is None or ent[0]() is not renderer:
        import weakref
        if len(_pool_cache) > 16:
            _pool_cache.clear()
        ent = (weakref.ref(renderer), pooling_matrix(renderer.csr_offsets, renderer.csr_members, n_verts, device, dtype))
        _pool_cache[key] = ent
    return ent[1]


def pooling_matrix(offsets, members, n_verts, device, dtype):
    """voxelization (:356-371) as one matrix: row c holds 1 / |cluster c| at the cluster's vertices"""
    rows = torch.repeat_interleave(torch.arange(len(offsets) - 1), torch.as_tensor(offsets[1:] - offsets[:-1]))
    cols = torch.as_tensor(members, dtype=torch.long)
    M = torch.zeros((len(offsets) - 1, n_verts), dtype=dtype)
    M[rows, cols] = 1.0
    return (M / M.sum(1, keepdim=True)).to(device)


# ---- TransHE -------------------------------------------------------------------------------------------------------------
def vit_forward(vit, x, pe_xyz):
    """VisionTransformer.forward (vision_transformer.py:257-307, :362-383), mask = None."""
    x = x + vit.get_PE(pe_xyz).to(x.dtype)
    V, N, C = x.shape
    h = vit.num_heads
    for blk in vit.blocks:
        y = blk.norm1(x)
        qkv = blk.attn.qkv(y).reshape(V, N, 3, h, C // h).permute(2, 0, 3, 1, 4)
        a = (qkv[0] @ qkv[1].transpose(-2, -1)) * blk.attn.scale
        y = (a.softmax(dim=-1) @ qkv[2]).transpose(1, 2).reshape(V, N, C)
        x = x + blk.attn.proj(y)
        x = x + blk.mlp.fc2(F.gelu(blk.mlp.fc1(blk.norm2(x))))
    return vit.norm(x)


# ---- per-point network ---------------------------------------------------------------------------------------------------
def _lin(mod, x):
    """a Conv1d(k = 1) applied to rows: x [..., in] -> [..., out]"""
    return F.linear(x, mod.weight.reshape(mod.weight.shape[0], -1), mod.bias)


def human_representation(net, pts_s, centres, rot, tokens, K, alpha):
    """get_human_representation (cross_transformer.py:158-205) -> [P,V,255]"""
    d2 = ((pts_s[:, None, :] - centres[None]) ** 2).sum(-1)
    d2k, idx = torch.topk(d2, K, dim=1, largest=False)
    w = F.softmax(-d2k.clamp_min(0).sqrt() / alpha, dim=1)
    rel = pts_s[:, None, :] - centres[idx]
    de = torch.matmul(rel.unsqueeze(-2), rot[idx]).squeeze(-2)
    pe = net.PE_relative(de.reshape(-1, 3)).view(pts_s.shape[0], K, -1)
    out = [torch.sum(w.unsqueeze(-1) * torch.cat([tokens[v][idx], pe], -1), dim=1) for v in range(tokens.shape[0])]
    return torch.stack(out, dim=1)


def point_network(net, h, f, viewdir):
    """MLP_forward_ori (cross_transformer.py:273-353): h [P,V,255], f [P,V,384], viewdir [P,27] -> raw [P,4]"""
    V = h.shape[1]
    s = F.relu(_lin(net.fc_0, h))
    p = F.relu(_lin(net.alpha_res_0, f))
    kp, vp = _lin(net.spatial_key_value_0.key_embed, p), _lin(net.spatial_key_value_0.value_embed, p)
    ks, vs = _lin(net.spatial_key_value_1.key_embed, s), _lin(net.spatial_key_value_1.value_embed, s)
    A = F.softmax(torch.einsum("pjc,pic->pji", kp, ks) / math.sqrt(kp.shape[-1]), dim=1)          # :141-144
    n = vs + torch.einsum("pjc,pji->pic", vp, A)
    inter = F.relu(_lin(net.fc_2, F.relu(_lin(net.fc_1, n))))
    sigma = _lin(net.alpha_fc, F.relu(_lin(net.fc_3, inter.mean(dim=1))))
    feat = _lin(net.feature_fc, inter) + _lin(net.rgb_res_0, f)
    feat = torch.cat([feat, viewdir[:, None, :].expand(-1, V, -1)], dim=-1)
    c = F.relu(_lin(net.view_fc, feat)) + _lin(net.rgb_res_1, f)
    rgb = _lin(net.rgb_fc, F.relu(_lin(net.fc_4, c.mean(dim=1))))
    return torch.cat([rgb, sigma], dim=1)


def composite(raw, z, ray_d, raw_noise_std, white_bkgd):
    """raw2outputs (nerf_net_utils.py:14-59)"""
    d = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], -1)
    d = d * torch.norm(ray_d[..., None, :], dim=-1)
    noise = torch.randn(raw[..., 3].shape).to(raw) * raw_noise_std if raw_noise_std > 0.0 else 0.0
    a = 1.0 - torch.exp(-F.relu(raw[..., 3] + noise) * d)
    T = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1.0 - a + 1e-10], -1), -1)[:, :-1]
    w = a * T
    rgb = torch.sum(w[..., None] * torch.sigmoid(raw[..., :3]), -2)
    acc = torch.sum(w, -1)
    if white_bkgd:
        rgb = rgb + (1.0 - acc[..., None])
    return rgb, acc, torch.sum(w * z, -1)


# ---- the entry -------------------------------------------------------------------------------------------------------------
def render(renderer, batch, chunk=32768):
    """Renderer.render -> _render (if_clight_renderer.py:486-605) with pts_mask = None: every sample of every ray through
    the network, RGB everywhere.  Differentiable with respect to every parameter of ``renderer.net``."""
    cfg = get_cfg()
    net = renderer.net
    assert cfg.time_steps == 1
    ray_o, ray_d = batch["ray_o"][0], batch["ray_d"][0]
    near, far = batch["near"][0], batch["far"][0]
    dev = ray_o.device
    S = int(cfg.N_samples)
    z = sample_depths(near, far, S, float(getattr(cfg, "perturb", 0.0)) > 0.0 and net.training)
    xyz = (ray_o[:, None] + ray_d[:, None] * z[..., None]).reshape(-1, 3)
    pts_s = torch.matmul(xyz - batch["Th"][0].reshape(1, 3), batch["Rh"][0])                    # world2smpl :289-295
    viewdir = embed_view(ray_d, int(getattr(cfg, "view_res", 4)))[:, None, :].expand(-1, S, -1).reshape(-1, 27)

    images = batch["input_imgs"][0].reshape(-1, *batch["input_imgs"][0].shape[2:])
    V = images.shape[0]
    R_in, T_in, K_in = (batch[k][0].reshape(V, *sh) for k, sh in (("input_R", (3, 3)), ("input_T", (3, 1)), ("input_K", (3, 3))))
    image_shape = batch["input_imgs"][0].shape[-2:]
    hol, pix = encode(net.encoder, images)
    verts_in = batch["input_smpl_vertice"][0][0]
    painted = sample_map(hol, project(verts_in, R_in, T_in, K_in), net.encoder, image_shape).permute(0, 2, 1)
    if cfg.rasterize:
        painted = painted * batch["input_vizmaps"][0][0][..., None].to(painted.dtype)              # :181-182
    nv = verts_in.shape[0]
    M = pooling_matrix_cached(renderer, nv, dev, torch.float32)
    tokens = vit_forward(net.ViT, torch.einsum("cn,vnd->vcd", M, painted), renderer._pe_norm(V, dev))
    centres = M @ batch["tar_smpl_vertice_smplcoord"][0]
    blend = batch["blend_mtx"][0]
    M64 = pooling_matrix_cached(renderer, nv, dev, blend.dtype)        # (float64 mean, :544)
    rot = (M64 @ blend.reshape(nv, 16)).reshape(-1, 4, 4)[:, :3, :3].to(torch.float32)             # cross_transformer.py:185

    raws = []
    for s0 in range(0, xyz.shape[0], chunk):                       # batchify_rays :607-656 without a mask
        x = xyz[s0:s0 + chunk]
        f = sample_map(pix, project(x, R_in, T_in, K_in), net.encoder, image_shape).permute(2, 0, 1)
        h = human_representation(net, pts_s[s0:s0 + chunk], centres, rot, tokens, int(cfg.KNN), float(cfg.KNN_DIST_ALPHA))
        raws.append(point_network(net, h, f, viewdir[s0:s0 + chunk]))
    raw = torch.cat(raws, 0).view(-1, S, 4)
    rgb, acc, depth = composite(raw, z, ray_d, float(getattr(cfg, "raw_noise_std", 0.0)), bool(cfg.white_bkgd))
    return {"rgb_map": rgb[None], "acc_map": acc[None], "depth_map": depth[None]}
