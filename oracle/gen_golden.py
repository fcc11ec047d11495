"""Generate tests/golden/*.npz by running the REAL reference modules on CPU.

TEST INFRASTRUCTURE, runs only where /root/reference is mounted:
    python -m oracle.gen_golden
Inputs come from transhuman_amd.synth (seeded, rebuilt identically by the
tests); weights are synth.det_state_dict() loaded into the reference Network,
so only outputs (+ the few small inputs that are not regenerated) are stored.
Every fixture is data: arrays in, arrays out.  See SURVEY.md 8c for the list.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_harness as rh          # noqa: E402
from transhuman_amd import synth              # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SIGMA_BIAS = -1.7


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"  {name}.npz  {sz/1024:.0f} KiB  keys={list(conv)}")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    mods = rh.load_reference(num_class=300, n_samples=32)
    cfg = mods["cfg"]
    CT, VT = mods["cross_transformer"], mods["vision_transformer"]

    # ---- reference runtime data -> pickle-free fixture ---------------------
    km = {}
    for k in (300, 500, 1500):
        d = np.load(os.path.join(rh.REF, "kmeans_dict", f"kmeans_dict_{k}.npy"), allow_pickle=True).item()
        a = np.asarray(d["pc2voxel_ind"]).astype(np.int16)
        lists = d["dict_voxel2pc_ind"]
        assert all(list(lists[c]) == np.where(a == c)[0].tolist() for c in range(k))
        km[f"pc2voxel_{k}"] = a
    save("kmeans_pc2voxel", **km)

    # ---- synthetic body + its own clusterings --------------------------------
    body, _ = synth.make_body(0)
    assign = {k: synth.kmeans_assign(body, k) for k in (300, 500)}
    save("synth_assign", **{f"assign_{k}": v.astype(np.int16) for k, v in assign.items()})
    can64 = body.astype(np.float64) * 1.02 + 0.001      # "canonical template" stand-in (float64)

    def ref_net(depth=12, seed=0):
        cfg.vit_depth = depth
        torch.manual_seed(0)
        net = CT.Network()
        sd = synth.det_state_dict(net.state_dict(), seed=seed, sigma_bias=SIGMA_BIAS)
        net.load_state_dict(sd)
        net.train()                                     # run.py:29
        return net

    net = ref_net()

    # ---- G1 sampling ---------------------------------------------------------
    b = synth.make_batch(16, 16, 3, seed=0)
    cfg.N_samples = 32
    r = rh.make_ref_renderer(mods, net, can64, assign[300])
    sel = slice(40, 77)
    pts, z = r.get_sampling_points(b["ray_o"][:, sel], b["ray_d"][:, sel], b["near"][:, sel], b["far"][:, sel])
    save("g1_sampling", ray_o=b["ray_o"][0, sel], ray_d=b["ray_d"][0, sel], near=b["near"][0, sel],
         far=b["far"][0, sel], pts=pts[0], z=z[0])

    # ---- G4/G5 paint + grouping ---------------------------------------------
    b = synth.make_batch(32, 32, 3, seed=0)
    hol = torch.from_numpy(synth.smooth_noise((3, 192, 32, 32), 21))
    scale = np.array([32, 32]) / (np.array([32, 32]) - 1) * 2.0
    _, big = r.paint_neural_human(b, 0, hol, scale)
    grouped = r.can_body_grouping(big)
    save("g45_paint_group", big_head=big[:, :96], big_sum=big.double().sum((0, 1)), grouped=grouped)
    # real kmeans dicts, arithmetic only
    for k in (500, 1500):
        rr = rh.make_ref_renderer(mods, net, can64, km[f"pc2voxel_{k}"].astype(np.int64))
        g = rr.can_body_grouping(big[:1])
        save(f"g5_group_real{k}", grouped=g, pe_can=rr.voxel_PE_can,
             pe_norm=rr.normalize_PE(rr.voxel_PE_can[None]))

    # ---- G6 ViT ---------------------------------------------------------------
    pe_can = r.voxel_PE_can.unsqueeze(0).repeat(3, 1, 1)
    pe_norm = r.normalize_PE(pe_can)
    with torch.no_grad():
        tok = net.ViT(grouped.clone(), pe_norm, mask=None)
        pe_tab = net.ViT.get_PE(pe_norm, None, grouped.shape)
    save("g6_vit", pe_norm=pe_norm[0], pe_table=pe_tab[0], out=tok)
    r5 = rh.make_ref_renderer(mods, net, can64, assign[500])
    x5 = torch.from_numpy(synth.smooth_noise((1, 500, 192), 22, passes=0))
    with torch.no_grad():
        tok5 = net.ViT(x5.clone(), r5.normalize_PE(r5.voxel_PE_can[None]), mask=None)
    save("g6_vit_n500_v1", out=tok5)

    # ---- G7 DPaRF -------------------------------------------------------------
    centres = r.voxelization(r.dict_voxel2pc_ind, b["tar_smpl_vertice_smplcoord"][0])
    blend = r.voxelization(r.dict_voxel2pc_ind, b["blend_mtx"][0])
    rs = np.random.RandomState(5)
    vid = rs.randint(0, synth.NV, size=256)
    pts_s = b["tar_smpl_vertice_smplcoord"][0][vid] + torch.from_numpy(rs.normal(0, 0.04, (256, 3)).astype(np.float32))
    with torch.no_grad():
        hr, _ = net.get_human_representation(pts_s[None], centres[None], blend[None], tok)
    save("g7_dparf", pts_s=pts_s, centres=centres, blend=blend, human_rep=hr)

    # ---- G8 Network.forward ---------------------------------------------------
    P = 1024
    vid = rs.randint(0, synth.NV, size=P)
    pts8 = b["tar_smpl_vertice_smplcoord"][0][vid] + torch.from_numpy(rs.normal(0, 0.05, (P, 3)).astype(np.float32))
    pf = torch.from_numpy(synth.smooth_noise((3, 384, P), 23, passes=0))
    vdir = torch.from_numpy(rs.normal(size=(P, 3)).astype(np.float32))
    vd = mods["embedder"].view_embedder(vdir / torch.norm(vdir, dim=1, keepdim=True))[None]
    mask = torch.from_numpy(rs.uniform(size=P) < 0.6)[None]
    g8 = dict(pts_s=pts8, viewdir=vd[0], mask=mask[0])
    with torch.no_grad():
        for tag, mk in (("none", None), ("rand", mask), ("zero", torch.zeros_like(mask))):
            dd = {"pts_smplcoord": pts8[None], "obs_smpl_smplcoord": centres[None], "blend_mtx": blend[None]}
            g8["raw_" + tag] = net(pf, vd, dd, holder=tok, face_idx=None, pts_mask=mk)[0]
        dd = {"pts_smplcoord": pts8[None], "obs_smpl_smplcoord": centres[None], "blend_mtx": blend[None]}
        g8["raw_v1_rand"] = net(pf[:1], vd, dd, holder=tok[:1], face_idx=None, pts_mask=mask)[0]
        dd = {"pts_smplcoord": pts8[None], "obs_smpl_smplcoord": centres[None], "blend_mtx": blend[None]}
        g8["raw_v1_none"] = net(pf[:1], vd, dd, holder=tok[:1], face_idx=None, pts_mask=None)[0]
    save("g8_forward", **g8)

    # ---- G9 pixel-aligned gather incl. out-of-image points -------------------
    pix = torch.from_numpy(synth.smooth_noise((3, 384, 32, 32), 24))
    xyz = b["tar_smpl_vertice"][0][rs.randint(0, synth.NV, size=96)].clone()
    xyz[:24] += torch.from_numpy(rs.normal(0, 1.5, (24, 3)).astype(np.float32))   # many land outside
    with torch.no_grad():
        pfe = r.get_pixel_aligned_feature(b, xyz[None], pix, scale, t=0)
    save("g9_pixel_aligned", xyz=xyz, feat=pfe)

    # ---- G10 raw2outputs -------------------------------------------------------
    raw = torch.from_numpy(rs.normal(0, 2, (64, 32, 4)).astype(np.float32))
    raw[5] = 0; raw[6, :, 3] = -1.0; raw[7, -1, 3] = 5.0; raw[7, :-1, 3] = 0
    zz = torch.sort(torch.from_numpy(rs.uniform(2, 4, (64, 32)).astype(np.float32)), dim=1)[0]
    rd = torch.from_numpy(rs.normal(size=(64, 3)).astype(np.float32))
    rgb, disp, acc, w, dep = mods["nerf_net_utils"].raw2outputs(raw, zz, rd, 0, False)
    save("g10_raw2outputs", raw=raw, z=zz, ray_d=rd, rgb=rgb, acc=acc, depth=dep, weights=w)

    # ---- G11 full render_fast (both branches of the R'<=2400 switch) ---------
    for tag, (H, focal) in (("small", (32, None)), ("large", (64, 210.0))):
        bb = synth.make_batch(H, H, 3, seed=0, focal=focal)
        cap = {}
        orig = r._render

        def spy(batch, pts, z_vals, is_train=True, pts_mask=None, _o=orig, _c=cap):
            _c["mask"] = pts_mask.clone()
            _c["rays"] = batch["ray_o"].shape[1]
            return _o(batch, pts, z_vals, is_train=is_train, pts_mask=pts_mask)
        r._render = spy
        bb2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in bb.items()}
        with torch.no_grad():
            ret = r.render_fast(bb2, is_train=False)
            hmap, _, pmap, _ = net.encoder(bb["input_imgs"][0][0])
        r._render = orig
        print(f"  render_fast[{tag}] rays={H*H} hit={cap['rays']} valid_pts={int(cap['mask'].sum())}")
        save(f"g11_render_{tag}", rgb=ret["rgb_map"][0], acc=ret["acc_map"][0], depth=ret["depth_map"][0],
             hit_rays=np.int64(cap["rays"]), mask_bits=np.packbits(cap["mask"][0].numpy()),
             holder_map_sum=hmap.double().sum((0, 2, 3)), pixel_map_sum=pmap.double().sum((0, 2, 3)),
             holder_map_px=hmap[:, :, 7, 9], pixel_map_px=pmap[:, :, 7, 9])

    # ---- G12 mesh sigma cube ---------------------------------------------------
    sys.modules["mcubes"].marching_cubes = lambda cube, th: (np.zeros((0, 3)), np.zeros((0, 3), int))
    mods["if_mesh_renderer"].mcubes = sys.modules["mcubes"]
    cfg.voxel_size = [0.005, 0.005, 0.005]      # configs/reconstruction.yaml:14-15
    cfg.mesh_th = 20
    rm = rh.make_ref_renderer(mods, net, can64, assign[300], mesh=True)
    bb = synth.make_batch(32, 32, 3, seed=0)
    bb["pts"] = synth.make_grid_pts(bb, 20)
    with torch.no_grad():
        out = rm.render(bb)
    cube = out["cube"][10:-10, 10:-10, 10:-10]
    print("  mesh cube", cube.shape, "nonzero", int((cube != 0).sum()))
    save("g12_mesh_cube", cube=cube)

    # ---- f-1 encoder on its own -------------------------------------------------
    imgs = bb["input_imgs"][0][0]
    with torch.no_grad():
        hmap, hs, pmap, ps = net.encoder(imgs)
    save("g13_encoder", holder_px=hmap[:, :, ::8, ::8], pixel_px=pmap[:, :, ::8, ::8], scale=hs)
    print("done")


if __name__ == "__main__":
    main()
