"""Renderer -- orchestration of the volumetric rendering hot path on MI355X.

Drop-in for /root/reference/lib/networks/renderer/if_clight_renderer.py
(`Renderer`, :37-656): ``Renderer(net)``, ``render_fast(batch, is_train)``
(:429-484) and ``render(batch, is_train)`` (:486-498) return
``{'rgb_map' [1,R,3], 'acc_map' [1,R], 'depth_map' [1,R]}``.

What runs where
  * encoder (ResNet18 stem)            th_conv2d / th_bn_act / th_maxpool3x3s2  (K12, K11)
  * paint + cluster pooling            th_paint_group        (K2)
  * TransHE                            th_vit_forward        (K3)
  * DPaRF tables (centres, rotations)  th_segment_mean_*     (K2)
  * sampling, hull mask, compaction,
    DPaRF, pixel gather, MLP,
    compositing                        th_render_rays        (K1,K4,K5,K6,K7)
Unlike the reference nothing here mutates ``batch`` (:459-462).
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
import pickle                                                       # noqa: E402

import numpy as np                                                  # noqa: E402
import torch                                                        # noqa: E402

from transhuman_amd.config import get_cfg                           # noqa: E402
from transhuman_amd import hip, synth                               # noqa: E402


class Renderer:
    def __init__(self, net, vertex_can=None, pc2voxel_ind=None):
        """``vertex_can`` (float64 [6890,3]) / ``pc2voxel_ind`` (int [6890]) may be
        injected; otherwise they are read from the reference's cwd-relative files
        (./data/smplx/smpl/SMPL_NEUTRAL.pkl :43-48, ./kmeans_dict/kmeans_dict_{N}.npy :55)."""
        cfg = get_cfg()
        self.net = net
        self.faces = None
        if vertex_can is None:
            with open("./data/smplx/smpl/SMPL_NEUTRAL.pkl", "rb") as f:
                data = pickle.load(f, encoding="latin1")
            vertex_can = np.asarray(data["v_template"])
            self.faces = data["f"]
        self.vertex_can = torch.as_tensor(np.asarray(vertex_can)).contiguous()      # float64 like :48
        self.CR = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])                  # :50
        voxel2pc = None
        if pc2voxel_ind is None:
            num_voxel = cfg.num_class
            path = f"./kmeans_dict/kmeans_dict_{num_voxel}.npy"
            pc2voxel_ind, voxel2pc = np.load(path, allow_pickle=True).item().values()     # :55 (same unpacking)
            pc2voxel_ind = np.asarray(pc2voxel_ind)
        self.pc2voxel_ind = torch.as_tensor(np.asarray(pc2voxel_ind)).type(torch.int64)
        if voxel2pc is not None:
            # CSR straight from the file's cluster lists, in the file's order (the reference pools over
            # dict_voxel2pc_ind.values(), :73 / :362-369)
            lists = [np.asarray(v, dtype=np.int64).reshape(-1) for v in voxel2pc.values()]
            self.csr_offsets = np.concatenate([[0], np.cumsum([len(v) for v in lists])]).astype(np.int64)
            self.csr_members = np.concatenate(lists).astype(np.int64)
        else:
            # injected assignment: members ascending inside a cluster == the order of the lists in the reference's files
            self.csr_offsets, self.csr_members = synth.csr_from_assign(self.pc2voxel_ind.numpy())
        self.num_clusters = len(self.csr_offsets) - 1
        self.voxel_PE_can = self._host_segment_mean(self.vertex_can)               # :73  (float64 [N_c,3])
        self._dev = {}

    # ---- host-side helpers (constants of the renderer) ----------------------------
    def _host_segment_mean(self, x):
        out = [x[torch.as_tensor(self.csr_members[self.csr_offsets[c]:self.csr_offsets[c + 1]], dtype=torch.long)].mean(0)
               for c in range(self.num_clusters)]
        return torch.stack(out)

    def normalize_PE(self, PE, CR=None):
        """:373-383 -- float64 in, float32 out."""
        assert len(PE.shape) == 3
        CR = self.CR if CR is None else CR
        mn, mx = CR[:3][None, None, :].to(PE.device), CR[3:][None, None, :].to(PE.device)
        return ((((PE - mn) / (mx - mn)) - 0.5) * 2).type(torch.float32)

    def voxelization(self, src):
        """:356-371 on device: per-cluster mean of per-vertex rows."""
        off, mem = self._csr(src.device)
        return hip.segment_mean(src, off, mem)

    def _csr(self, device):
        key = ("csr", str(device))
        if key not in self._dev:
            self._dev[key] = hip.csr_to_device(self.csr_offsets, self.csr_members, device)
        return self._dev[key]

    def _pe_norm(self, V, device):
        key = ("pe", V, str(device))
        if key not in self._dev:
            pe = self.voxel_PE_can.unsqueeze(0).repeat(V, 1, 1)                    # :536
            self._dev[key] = self.normalize_PE(pe).to(device)
        return self._dev[key]

    # ---- per-frame constants ---------------------------------------------------------
    def prepare_frame(self, batch, hull_thresh=None, fused_encoder_tail=True, compact_map=True, token_exchange=None,
                      pregather=None, defer_tokens=False, stem_exchange=None, crop_map=None, demand=None, stem_graph=False):
        """paint -> group -> TransHE -> DPaRF tables (:531-547).  Returns hip.Frame.

        fused_encoder_tail=True (default): the ResNet stem runs through SpatialEncoder.trunk (K12 / K11), its tail
        (3 upsamples + colour lift + concat, encoder.py:133-146) is ONE HIP kernel that writes the
        384-channel map channels-last, and holder_feat_map is never materialised -- the 384->192
        reduction_layer is applied to the 3 x 6890 sampled vertex rows instead (it commutes with the
        bilinear sampling).  False: the reference's op order through ``net.encoder(images)``.
        compact_map=True (default, only with the fused tail): the last 128 map channels are
        upsample_color(img) = Wc rgb + bc, a linear lift of 3 numbers, so the map keeps r,g,b instead
        (split planes [V,H,W,256] + [V,H,W,4], hip.SplitMap; compact_map="interleaved": one [V,H,W,260] tensor)
        and the lift is folded into the four layers that read those channels (alpha_res_0,
        rgb_res_0, rgb_res_1, reduction_layer: W' = [W_lat | W_col Wc], b' = b + W_col bc) -- a third less
        map/gather/staging traffic and 12 % fewer MLP MACs, same function.
        All forms give the same tokens / pixels to fp32 rounding (tests/test_gpu_parity.py).
        crop_map (default: on for the split compact map, TH_MAP_CROP=0 switches it off): the reference writes the map over
        the whole image and then reads it only at samples within the hull threshold of a target vertex (:440-444) and at
        the projected input vertices (:168-172) -- here only the per-view texel box those reads can touch is written
        (hip.map_box: projected corners of the threshold-sized cube around every vertex; about a third of a 512 x 512
        view for a standing body).  Same pixels: the box is a superset by construction, and a call that leaves its premise
        (un-masked small-frame branch, no hull test) makes the C side write the rest first (th_frame.map_source).
        token_exchange (multi-GPU, transhuman_amd.dist.TokenExchange): callable(compute, shape, device) that either
        runs ``compute`` (paint -> group -> TransHE) here or receives the tokens from the rank that did.
        stem_exchange (multi-GPU, transhuman_amd.dist.StemExchange): the ResNet stem runs on one rank per frame and its
        three low-resolution latents (69 MB) are broadcast; the upsample / concat into the 0.82 GB map stays local.
        pregather=(points, slot) (render_fast, behind its hull prepass): the pixel-feature gather and the neighbour
        records of the frame's first chunks (hip.render_pregather: they need the map and the token centres, not the
        tokens) are queued on the current stream and TransHE runs BESIDE them on a second stream instead of in front.
        defer_tokens=True (render_sequence): everything up to the grouped vertex features; ``frame.finish_tokens()`` runs
        TransHE later (the frame pipeline issues it at the start of the next shading window, see render_sequence).
        demand=(buffer, event) (``self.predemand``, behind the frame's hull prepass): the map and its fold are written only at
        the texels the prepass's valid samples (and, where this rank paints, the input vertices) read -- the frame is then
        complete for exactly that prepass; any other use of it writes the rest first (th_map_source.demand)."""
        cfg = get_cfg()
        assert cfg.time_steps == 1                                                  # :412
        t = 0
        # the weight image (and with it the sticky range slot of the stem convolutions, which belongs to the same
        # parameter set) is brought up to date BEFORE the stem runs and on the stream that runs it: a frame's own
        # convolutions are then never queued in front of the clear that new weights trigger
        hip._sync_weights(self.net, "mlp")
        images = batch["input_imgs"][t]
        images = images.reshape(-1, *images.shape[2:])                              # :397
        dev = images.device
        cams = hip.pack_cams(batch["input_R"][t].reshape(-1, 3, 3), batch["input_T"][t].reshape(-1, 3, 1),
                             batch["input_K"][t].reshape(-1, 3, 3))
        image_shape = batch["input_imgs"][t].shape[-2:]
        off, mem = self._csr(dev)
        viz = batch["input_vizmaps"][t][0] if cfg.rasterize else None               # :103-119
        enc = self.net.encoder
        fold_done = None
        if fused_encoder_tail and hasattr(enc, "trunk"):
            H, W = images.shape[2:]
            V = images.shape[0]
            # (a device whose stem has been switched to the stock convolutions -- range guard, or non-finite latents received
            # from another rank -- computes its own latents from here on: the owner rank may not have switched yet)
            # -- it still takes part in the exchange (a collective: the other ranks wait for its turn as owner) but drops what
            # it receives
            # (stem_graph: the stem's 31 launches replayed as one hipGraph -- the callers that render a stream of frames)
            trunk = (lambda im: enc.trunk(im, graph=True)) if stem_graph else enc.trunk
            lat = trunk(images) if stem_exchange is None else stem_exchange.latents(trunk, images)
            stem_flag = None if stem_exchange is None else stem_exchange.last_flag
            if stem_exchange is not None and hip.conv_fallback(dev) and not stem_exchange.last_mine:
                lat, stem_flag = enc.trunk(images), None
            cw, cb = enc.upsample_color.weight, enc.upsample_color.bias
            scale = hip.feat_scale(enc.feat_scale(H, W), image_shape, dev)
            thr = cfg_hull() if hull_thresh is None else hull_thresh
            if crop_map is None:
                crop_map = os.environ.get("TH_MAP_CROP") != "0"
            if compact_map == "interleaved":            # A/B: one [V,H,W,260] tensor (1040-byte texel rows)
                map_nhwc = hip.upsample_concat_nhwc(images, lat[0], lat[1], lat[2])
            elif compact_map and demand is not None:
                torch.cuda.current_stream(dev).wait_event(demand[1])       # (the marks were made on the hull stage's stream)
                map_nhwc = hip.upsample_concat_split(images, lat[0], lat[1], lat[2], demand=demand[0])
            elif compact_map and crop_map and thr >= 0:
                reach = float(thr) * 1.001 + 1e-6
                box = hip.map_box(batch["tar_smpl_vertice"][0], batch["input_smpl_vertice"][t][0], cams, scale, H, W, reach)
                map_nhwc = hip.upsample_concat_split(images, lat[0], lat[1], lat[2], box=box, reach=reach)
            elif compact_map:
                map_nhwc = hip.upsample_concat_split(images, lat[0], lat[1], lat[2])
            else:
                map_nhwc = hip.upsample_concat_nhwc(images, lat[0], lat[1], lat[2], cw, cb)
            # texel hand-over (hip.set_tex_rows, default): the layers that read the pixel-aligned features (cross_transformer.py
            # :316, :334, :346) are applied to the map's texels here, once per frame -- bilinear sampling commutes with them
            if (isinstance(map_nhwc, hip.SplitMap) and hip.tex_rows_enabled(dev) and V <= 3 and V * H * W < (1 << 22)
                    and hip.mlp_is_fused(dev)):
                if pregather is not None and os.environ.get("TH_FOLD_STREAM", "1") != "0":
                    # a single frame (render_fast) is bound by its chain of dependent stages: stem -> map -> paint / group ->
                    # TransHE -> fused MLP.  The fold (0.35 ms) is needed by the fused MLP only: beside that chain on a stream
                    # of its own, not inside it.  (A stream of frames is bound by the chip's total work: render_sequence keeps
                    # the fold on the side stream, measured -- no gain there, LOG.md.)
                    cur = torch.cuda.current_stream(dev)
                    fs = self._dev.get(("fold_stream", str(dev)))
                    if fs is None:
                        fs = self._dev[("fold_stream", str(dev))] = torch.cuda.Stream(dev)
                    # (the packed MLP image is uploaded on the CURRENT stream if it is due -- first frame, new weights: uploaded on `fs`
                    # by map_fold's own check, nothing would order a map completion that th_render_pregather queues on `cur`
                    # behind it)
                    hip._sync_weights(self.net, "mlp")
                    fs.wait_stream(cur)
                    with torch.cuda.stream(fs):
                        fold = hip.map_fold(self.net, map_nhwc)
                        fold_done = torch.cuda.Event()
                        fold_done.record(fs)
                    fold.record_stream(cur)
                else:
                    hip.map_fold(self.net, map_nhwc)

            def group():
                return hip.paint_group_nhwc(map_nhwc, batch["input_smpl_vertice"][t][0], cams, scale, viz,
                                            enc.reduction_layer.weight, enc.reduction_layer.bias, off, mem,
                                            color_w=cw if compact_map else None, color_b=cb if compact_map else None)
            pix_scale = scale
        else:
            holder_map, holder_scale, pixel_map, pixel_scale = enc(images)          # :399
            V, _, H, W = pixel_map.shape

            def group():
                return hip.paint_group(holder_map, batch["input_smpl_vertice"][t][0], cams,
                                       hip.feat_scale(holder_scale, image_shape, dev), viz, off, mem)
            map_nhwc = hip.nchw_to_nhwc(pixel_map)
            pix_scale = hip.feat_scale(pixel_scale, image_shape, dev)

        def make_tokens():
            self.last_grouped = group()
            return self.net.ViT(self.last_grouped, self._pe_norm(V, dev), mask=None, graph=stem_graph)    # :538

        centres = hip.segment_mean(batch["tar_smpl_vertice_smplcoord"][0], off, mem)   # :543
        rot = hip.segment_mean_rot(batch["blend_mtx"][0], off, mem)                 # :544 + cross_transformer.py:185
        mk_frame = lambda tok: hip.Frame(batch["tar_smpl_vertice"][0], batch["Rh"][0], batch["Th"][0], cams, pix_scale,
                                         map_nhwc, tok, centres, rot,
                                         hull_thresh=cfg_hull() if hull_thresh is None else hull_thresh,
                                         small_frame_rays=2400)
        if token_exchange is not None:
            frame = mk_frame(token_exchange(make_tokens, (V, self.num_clusters, get_cfg().embed_size), dev))
        elif defer_tokens:
            self.last_grouped = grouped = group()
            pe_norm = self._pe_norm(V, dev)
            frame = mk_frame(None)
            frame.finish_tokens = lambda: frame.set_tokens(self.net.ViT(grouped, pe_norm, mask=None, graph=stem_graph))   # :538
        elif pregather is None or os.environ.get("TH_PREGATHER") == "0":
            frame = mk_frame(make_tokens())
        else:
            pts_pg, slot_pg = pregather
            cur = torch.cuda.current_stream(dev)
            self.last_grouped = grouped = group()
            grouped_ready = torch.cuda.Event()
            grouped_ready.record(cur)
            frame = mk_frame(None)
            vs = self._dev.get(("vit_stream", str(dev)))
            if vs is None:
                vs = self._dev[("vit_stream", str(dev))] = torch.cuda.Stream(dev)
            # TransHE is ISSUED first (render_pregather waits on the host for the hull stage's sample count: the 63
            # launches must not queue behind that wait); on the device it runs beside K5 + K4 of the first chunks
            with torch.cuda.stream(vs):
                vs.wait_event(grouped_ready)
                tokens = self.net.ViT(grouped, self._pe_norm(V, dev), mask=None, graph=stem_graph)    # :538
            hip.render_pregather(self.net, frame, pts_pg, slot_pg)
            grouped.record_stream(vs)
            tokens.record_stream(cur)
            cur.wait_stream(vs)
            frame.set_tokens(tokens)
        if fold_done is not None:
            torch.cuda.current_stream(dev).wait_event(fold_done)
        # (range guard, hip.render_rays: the same constants again -- through the stock convolutions -- if the stem's
        # input left the fp16 range)
        # (a rebuilt frame computes its own tokens: the exchange's frame counter must not advance twice)
        frame.rebuild = lambda: self.prepare_frame(batch, hull_thresh, fused_encoder_tail, compact_map, crop_map=crop_map)
        # (device bool: the stem latents this frame was built from were not finite -- dist.StemExchange; read in finish())
        frame.stem_flag = stem_flag if (fused_encoder_tail and hasattr(enc, "trunk")) else None
        return frame

    def predemand(self, batch, pts, token_exchange=None, sharded=False):
        """Behind ``hip.render_prepass(pts, ...)``, on the current stream (the prepass's): the demand buffer of the frame's map
        (hip.render_predemand) + the event prepare_frame(demand=...) waits for, or None where the demand-driven map does not apply
        (TH_MAP_DEMAND=0, no fused encoder tail, an image width that is not a multiple of 64, more than three views)."""
        # Default: for the shards of a multi-rank job only.  On one GPU the frame reads ~70 % of the box's row spans: marking 2 M
        # samples costs more side-stream time (0.6 ms) than the smaller map and fold give back, and the side stream is not what
        # bounds that frame; a rank of 8 reads a sixth of the spans and IS bound by its per-frame front.
        # TH_MAP_DEMAND=1 / 0 forces it on / off.
        mode = os.environ.get("TH_MAP_DEMAND", "auto")
        if mode == "0" or (mode != "1" and not sharded) or not hasattr(self.net.encoder, "trunk"):
            return None
        t = 0
        imgs = batch["input_imgs"][t]
        V, H, W = int(np.prod(imgs.shape[:-3])), int(imgs.shape[-2]), int(imgs.shape[-1])
        dev = imgs.device
        if not (hip.tex_rows_enabled(dev) and hip.mlp_is_fused(dev)) or V > 3 or V * H * W >= (1 << 22):
            return None                     # (frames that take K5's rows read the latents of every corner: keep the box)
        cams = hip.pack_cams(batch["input_R"][t].reshape(-1, 3, 3), batch["input_T"][t].reshape(-1, 3, 1),
                             batch["input_K"][t].reshape(-1, 3, 3))
        scale = hip.feat_scale(self.net.encoder.feat_scale(H, W), imgs.shape[-2:], dev)
        paints = token_exchange is None or token_exchange.will_compute()
        buf = hip.render_predemand(pts, cams, scale, V, H, W,
                                   verts_paint=batch["input_smpl_vertice"][t][0] if paints else None)
        if buf is None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return buf, ev

    # ---- reference API -------------------------------------------------------------------
    def _own_stream(self, dev):
        """The stream a frame's shading chain is queued on when the caller sits on the device's DEFAULT stream (every
        unchanged caller of the reference's loop does): the default (null) stream orders itself against work it has nothing to do
        with -- measured on the single-frame chain: the neighbour records (K4, on the context's second stream behind an event of
        the calling stream) started 0.6 ms late, with TransHE's last launches on a third stream, whenever the calling stream was
        the null stream; on a stream of our own they start when their inputs are ready (render_fast 18.3 -> 17.6 ms on one box,
        17.85 -> 17.65 on another, profiles/r05_m).  render_sequence stays on the caller's stream: its steady state was 0.17 ms per
        frame SLOWER on a stream of its own (14.79 -> 14.96 ms, same profile).  None = stay on the caller's stream (it is not the
        default stream, or TH_OWN_STREAM=0)."""
        if not torch.cuda.is_available() or os.environ.get("TH_OWN_STREAM", "1") == "0":
            return None
        if torch.cuda.current_stream(dev) != torch.cuda.default_stream(dev):
            return None
        m = self._dev.get(("main_stream", str(dev)))
        if m is None:
            m = self._dev[("main_stream", str(dev))] = torch.cuda.Stream(dev)
        return m

    def render_fast(self, batch, is_train=True, frame=None, ray_slice=None, small_frame_rays=2400):
        """:429-484.  ``frame`` lets callers reuse per-frame constants; ``ray_slice`` renders a sub-range of
        rays (multi-GPU sharding); ``small_frame_rays`` is the R' threshold of :551 (-1 pins the masked
        branch, used when a frame is sharded).
        Without a ready ``frame`` the ray-only stage (hull mask, compaction) is queued first, then the
        per-frame constants, then the shading: the sample count is on the host by the time it is needed.
        Called on the device's default stream the frame is queued on a stream of the renderer's own (``_own_stream``), ordered
        behind everything the caller has queued; the caller's stream waits for the frame before this returns."""
        dev = batch["ray_o"].device
        own = self._own_stream(dev) if dev.type == "cuda" else None
        if own is None:
            return self._render_fast(batch, frame, ray_slice, small_frame_rays)
        caller = torch.cuda.current_stream(dev)
        own.wait_stream(caller)
        with torch.cuda.stream(own):
            out = self._render_fast(batch, frame, ray_slice, small_frame_rays)
        caller.wait_stream(own)
        for v in out.values():
            v.record_stream(caller)
        return out

    def _render_fast(self, batch, frame, ray_slice, small_frame_rays):
        cfg = get_cfg()
        sl = slice(None) if ray_slice is None else ray_slice
        pts = hip.Points(batch["ray_o"][0][sl], batch["ray_d"][0][sl], batch["near"][0][sl], batch["far"][0][sl],
                         n_samples=cfg.N_samples, **self._sampling_randoms(batch, sl, cfg))
        if frame is None:
            V = batch["input_imgs"][0].reshape(-1, *batch["input_imgs"][0].shape[2:]).shape[0]
            if V <= 4 and pts.R > 0:
                # the ray-only stage (hull mask, compaction: ~10 launches, 0.85 ms) runs on a second stream beside the
                # per-frame constants (encoder, paint, TransHE: ~110 latency-bound launches) instead of in front of them;
                # th_render_rays waits for its event
                dev = pts.ray_o.device
                side = self._dev.get(("hull_stream", str(dev)))
                if side is None:
                    side = self._dev[("hull_stream", str(dev))] = torch.cuda.Stream(dev)
                if os.environ.get("TH_HULL_SAME_STREAM") == "1":       # A/B switch: the ray-only stage in front, same stream
                    side = torch.cuda.current_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    hip.render_prepass(pts, batch["tar_smpl_vertice"][0], V, cfg_hull(), small_frame_rays,
                                       n_clusters=len(self.csr_offsets) - 1)
                    dm = self.predemand(batch, pts)
                for t in (pts.ray_o, pts.ray_d, pts.near, pts.far):
                    t.record_stream(side)
                frame = self.prepare_frame(batch, pregather=(pts, 0), demand=dm, stem_graph=True)
            else:
                frame = self.prepare_frame(batch)
        # (the threshold applies to THIS call whether or not the frame constants were handed in)
        rgb, acc, depth, stats = hip.render_rays(self.net, frame, pts, white_bkgd=bool(cfg.white_bkgd),
                                                 small_frame_rays=small_frame_rays)
        self.last_stats = stats
        return {"depth_map": depth[None], "rgb_map": rgb[None], "acc_map": acc[None]}

    def _sampling_randoms(self, batch, sl, cfg):
        """The reference's two randomisations of the sampling, as keyword arguments of hip.Points (empty when both are off, which
        is what run.py renders with: cfg.perturb = 0 at run.py:22,68,123 although it keeps network.train(); the YAML default is
        perturb: 1):
          * cfg.perturb > 0 with the network in train() mode: stratified jitter of the sample depths, get_sampling_points :276-283
            -- the same torch expressions on the device, the kernels read the depths from ``z_vals`` instead of computing
            near (1 - t) + far t;
          * cfg.raw_noise_std > 0: randn * std added to sigma in front of raw2alpha's relu (nerf_net_utils.py:39-44), on every
            sample of the composited rays.
        The draws come from torch's generator of the batch's device (the reference draws from its default device's: streams of
        different devices are not reproducible against each other either).  A caller that wants given draws puts them into the
        batch: ``batch["t_rand"]`` ([1,R,S] uniform draws) / ``batch["raw_noise"]`` ([1,R,S] standard-normal draws) -- the
        parity tests do."""
        kw = {}
        S = int(cfg.N_samples)
        near, far = batch["near"][0][sl], batch["far"][0][sl]
        if float(getattr(cfg, "perturb", 0.0)) > 0.0 and self.net.training:
            t_vals = torch.linspace(0., 1., steps=S).to(near)                                      # :273
            z_vals = near[..., None] * (1. - t_vals) + far[..., None] * t_vals                     # :274
            mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])                                       # :278
            upper = torch.cat([mids, z_vals[..., -1:]], -1)
            lower = torch.cat([z_vals[..., :1], mids], -1)
            t_rand = batch["t_rand"][0][sl].to(upper) if "t_rand" in batch else torch.rand(z_vals.shape, device=near.device).to(upper)
            kw["z_vals"] = lower + (upper - lower) * t_rand                                        # :283
        std = float(getattr(cfg, "raw_noise_std", 0.0))
        if std > 0.0:
            shape = (near.shape[0], S)
            draw = batch["raw_noise"][0][sl].to(near) if "raw_noise" in batch else torch.randn(shape, device=near.device, dtype=near.dtype)
            kw["sigma_noise"] = draw * std                                                         # nerf_net_utils.py:41
        return kw

    def render_fast_sharded(self, batch, my_idx, gatherer, hit_sum, frame=None):
        """One rank's part of a ray-sharded frame with the reference's WHOLE-FRAME R' <= 2400 rule (:551): the shard
        is rendered in the (overwhelmingly common) masked mode, the per-rank hit-ray counts are summed
        (``hit_sum``: transhuman_amd.dist.DeferredSum, its own communicator / stream) while the image is assembled
        (``gatherer``: dist.ImageGatherer over ``my_idx``), and only if the frame total is <= 2400 the shard is
        rendered again un-masked and gathered again -- every sharded caller gets the reference's branch.
        -> dense [R, 5] image (rgb | acc | depth) on every rank."""
        sh = dict(batch)
        for k in ("ray_o", "ray_d", "near", "far"):
            sh[k] = batch[k][:, my_idx].contiguous()
        if frame is None:
            frame = self.prepare_frame(batch)
        out = self.render_fast(sh, frame=frame, small_frame_rays=-1)
        hit_sum.start(self.last_stats["hit_rays"])
        cat = lambda o: torch.cat([o["rgb_map"][0], o["acc_map"][0][:, None], o["depth_map"][0][:, None]], dim=1)
        img = gatherer(cat(out))
        if hit_sum.result() <= 2400:
            out = self.render_fast(sh, frame=frame, small_frame_rays=1 << 30)
            img = gatherer(cat(out))
        return img

    def render_sequence(self, batches, ray_slice=None, small_frame_rays=2400, lookahead=1, token_exchange=None,
                        stem_exchange=None):
        """A stream of frames (free-viewpoint video / evaluation loops: the reference calls render_fast once per
        dataset item, run.py:96-118) as a two-stage software pipeline on two HIP streams:

            current stream :                     shading + compositing(i)       -> shading + compositing(i+1) ...
            side stream    : front(i), front(i+1) -> front(i+2) = hull stage + frame constants -> front(i+3) ...

        The ray-only hull stage and the per-frame constants (encoder, paint/group, TransHE: ~110 mostly
        latency-bound launches) of the next ``lookahead`` frames run while the fused MLP of frame i fills the chip,
        instead of in front of it, and the sample count of a frame is on the host long before its shading is
        queued: the host never waits for the device.  lookahead + 1 render workspaces rotate (hull stages write
        while earlier frames shade; a look-ahead beyond 1 only helps when the time per frame is very uneven).  Every frame executes exactly the work of ``render_fast`` -- same kernels, same
        order per frame.  Generator: yields render_fast's dict per batch; ``self.last_batch`` / ``self.last_frame`` /
        ``self.last_stats`` describe the frame just yielded.  ``lookahead`` frames are taken from ``batches`` ahead
        of the one being shaded; the iterator is advanced with the side stream current, so device work it issues
        for a coming frame (ray generation, SMPL skinning, uploads) also runs under the shading of the current one.
        ``token_exchange`` (transhuman_amd.dist.TokenExchange, multi-GPU): TransHE of frame j runs on rank j % world
        only and its tokens are broadcast from the side stream; ``stem_exchange`` (dist.StemExchange): the same for the
        encoder stem's latents.
        The stem and TransHE of these frames are replayed hipGraphs (encoder.trunk(graph=True), hip.vit_forward(graph=True)):
        the latents and tokens of ``self.last_frame`` live in the graphs' rotating buffers and stay valid until three more
        frames have been yielded -- ``self.last_frame.rebuild()`` returns a frame that owns its memory."""
        import collections
        cfg = get_cfg()
        sl = slice(None) if ray_slice is None else ray_slice
        it = iter(batches)
        lookahead = max(1, min(int(os.environ.get("TH_LOOKAHEAD", lookahead)), 3))    # (th_render_prepass keeps at most 4 tokens)
        if lookahead >= 3 and hip.graphs_enabled():
            # the stem / TransHE graph rings hold 4 instances = pipeline depth 2 + the frame being shaded + the frame just handed
            # out: at depth 3 the instance `last_frame` points at has been replayed for a later frame by the time it is yielded
            lookahead = 2
        # split front (single rank, TH_SPLIT_FRONT=0 switches it off): the front of a frame is issued in two pieces -- A =
        # hull stage, encoder, paint, group (chip-filling kernels) and B = TransHE (63 small dependent launches).  In the
        # shading window of frame i the side stream runs B(i+1) FIRST and then A(i+2): the latency-bound launches of
        # TransHE find free CUs beside the producers of frame i instead of queueing, one by one, behind MLP tiles.
        split = token_exchange is None and os.environ.get("TH_SPLIT_FRONT", "1") != "0"
        if split:
            lookahead = max(lookahead, 2)
        elif os.environ.get("TH_PREGATHER_EARLY", "1") != "0" and "TH_LOOKAHEAD" not in os.environ:
            # multi-rank job: two frames ahead as well, so that the front of frame i+1 is complete before the shading of
            # frame i is queued and its neighbour records can start beside frame i's compositing and image gather
            # (th_render_pregather_early below; emulated rank of 8: 2.835 -> 2.80 ms per frame)
            lookahead = max(lookahead, 2)
        nslots = lookahead + 1

        def front(b, j, side):
            """side stream: hull stage of b's rays into workspace 1 + j % nslots (0 is render_fast's), then b's frame
            constants"""
            # the range-guard epoch these constants are built under: a frame whose front was issued before the guard
            # switched a path (fp32 MLP, stock convolutions, fp32 TransHE GEMMs) is rebuilt before it is handed out
            ep = hip.range_epoch(b["ray_o"].device)
            with torch.cuda.stream(side):
                pts = hip.Points(b["ray_o"][0][sl], b["ray_d"][0][sl], b["near"][0][sl], b["far"][0][sl],
                                 n_samples=cfg.N_samples, **self._sampling_randoms(b, sl, cfg))
                V = b["input_imgs"][0].reshape(-1, *b["input_imgs"][0].shape[2:]).shape[0]
                if V <= 4 and pts.R > 0:
                    # the ray-only hull stage (grid, hull test, compaction: ~10 dependent launches, 0.3 ms) shares nothing with
                    # the frame constants: on a stream of its own beside them, not in front of them -- the side stream's chain of
                    # ~55 dependent launches is what bounds a rank of 8 (2.7 ms against a 2.3 ms shard of the fused MLP)
                    hs = hull_side if hull_side is not None else side
                    hs.wait_stream(side)
                    with torch.cuda.stream(hs):
                        hip.render_prepass(pts, b["tar_smpl_vertice"][0], V, cfg_hull(), small_frame_rays,
                                           n_clusters=len(self.csr_offsets) - 1, slot=1 + j % nslots)
                        dm = self.predemand(b, pts, token_exchange, sharded=token_exchange is not None or ray_slice is not None)
                    for t in (pts.ray_o, pts.ray_d, pts.near, pts.far):
                        t.record_stream(hs)
                else:
                    dm = None
                frame = self.prepare_frame(b, token_exchange=token_exchange, defer_tokens=split, stem_exchange=stem_exchange,
                                           demand=dm, stem_graph=True)
                if V <= 4 and pts.R > 0:
                    side.wait_stream(hs)
                if V <= 4 and pts.R > 0 and os.environ.get("TH_PREGRID", "1") != "0":
                    hip.render_pregrid(frame, pts)         # K4's candidate grid: here, not in front of K4
                ready = torch.cuda.Event()
                ready.record(side)
            return [b, pts, frame, ready, ep, ready, False]      # [5]: piece A's event (tokens() replaces [3]); [6]: see below

        def tokens(ent, side):
            """side stream: piece B of an entry whose piece A has been issued"""
            fin = getattr(ent[2], "finish_tokens", None)
            if fin is None:
                return
            with torch.cuda.stream(side):
                fin()
                ent[2].finish_tokens = None
                ent[3] = torch.cuda.Event()
                ent[3].record(side)

        first = next(it, None)
        if first is None:
            return
        dev = first["ray_o"].device
        side = self._dev.get(("side_stream", str(dev)))
        if side is None:
            # TH_SIDE_PRIORITY=-1: high-priority side stream (A/B switch; measured: see DESIGN.md 6)
            side = self._dev[("side_stream", str(dev))] = torch.cuda.Stream(dev, priority=int(os.environ.get("TH_SIDE_PRIORITY", "0")))
        side.wait_stream(torch.cuda.current_stream(dev))
        hull_side = None
        if os.environ.get("TH_HULL_STREAM", "1") != "0":
            hull_side = self._dev.get(("hull_side_stream", str(dev)))
            if hull_side is None:
                hull_side = self._dev[("hull_side_stream", str(dev))] = torch.cuda.Stream(dev)
        if token_exchange is not None and os.environ.get("TH_GRAPH_PRIME", "1") != "0":
            # multi-rank job: a rank runs TransHE (and, with the stem exchange, the stem) only for the frames it owns, so the
            # first-call capture of their graphs (a device synchronisation + a garbage collection per instance: tens of
            # milliseconds) would land on its first OWNED frame -- frame r of rank r, inside a short run's timed frames.  One
            # local set of frame constants up front (no exchange, result dropped) captures both rings on every rank at once.
            # (Train-mode BatchNorm running statistics advance by this one extra frame; they do not enter the rendering.)
            # (keyed on the shape AND the graph epoch: instances dropped for new weights are captured again the same way)
            shape = (tuple(first["input_imgs"][0].shape), hip.graph_epoch())
            primed = self._dev.setdefault(("graphs_primed", str(dev)), set())
            if shape not in primed and hip.graphs_enabled():
                primed.add(shape)
                with torch.cuda.stream(side):
                    # the extra frame must not show in the network's state: train-mode BatchNorm statistics are put back
                    bns = [m for m in self.net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
                    keep = [(m, None if m.running_mean is None else m.running_mean.clone(),
                             None if m.running_var is None else m.running_var.clone(),
                             None if m.num_batches_tracked is None else m.num_batches_tracked.clone()) for m in bns]
                    self.prepare_frame(first, stem_graph=True)
                    for m, rm, rv, nb in keep:
                        if rm is not None: m.running_mean.copy_(rm)
                        if rv is not None: m.running_var.copy_(rv)
                        if nb is not None: m.num_batches_tracked.copy_(nb)
        queue = collections.deque([front(first, 0, side)])
        tokens(queue[0], side)
        queued, more = 1, True

        def pull():
            nonlocal queued, more
            if not more:
                return
            with torch.cuda.stream(side):
                b = next(it, None)
            if b is None:
                more = False
                return
            queue.append(front(b, queued, side))
            queued += 1

        for _ in range(lookahead - 1):
            pull()
        # A frame is handed out one step late: its range-guard snapshot (hip.render_rays, defer_guard) is read back
        # -- a host wait for that frame -- only after the shading of the NEXT frame has been queued, so the device
        # never waits for the host.  A frame whose snapshot is not clean (or that was queued before an earlier frame
        # switched the context to the fp32 path) is rendered again before it is yielded.
        shaded = collections.deque()

        def finish(ent):
            rgb, acc, depth, stats, frame, cur, pts, check, epoch = ent
            ok = check()
            bad_stem = getattr(frame, "stem_flag", None) is not None and bool(frame.stem_flag)
            if bad_stem:
                hip.force_conv_fallback(dev, "the stem latents received for this frame are not finite (fp16 overflow in "
                                             "the owner rank's convolutions)")
            # (bad_stem on its own is a reason to rebuild: once the device is in conv fallback the epoch does not move again)
            if not ok or bad_stem or epoch != hip.range_epoch(dev):
                # (look-ahead 3: five frames are alive when this one is finished, one more than the graphs' rings of four hold --
                # its latents / tokens may have been handed to the frame three ahead: constants again in that case too)
                if bad_stem or hip.conv_fallback(dev) or hip.vit_fallback(dev) or lookahead >= 3:
                    frame = frame.rebuild()                 # constants again, through the paths the guard switched to
                rgb, acc, depth, stats = hip.render_rays(self.net, frame, pts, white_bkgd=bool(cfg.white_bkgd),
                                                         small_frame_rays=small_frame_rays)
            self.last_stats, self.last_frame, self.last_batch = stats, frame, cur
            return {"depth_map": depth[None], "rgb_map": rgb[None], "acc_map": acc[None]}

        while queue:
            cur, pts, frame, ready, epoch, _, early = queue.popleft()
            main = torch.cuda.current_stream(dev)
            main.wait_event(ready)
            # everything queued so far (inputs of coming batches, the shading of the previous frame -- the last user
            # of the workspace the next hull stage writes) is ordered before the side stream's next piece of work;
            # this frame's shading is not
            fence = torch.cuda.Event()
            fence.record(main)
            # two-phase shading: the texture-path-bound producers (pixel gather, neighbour records: ONE launch each over the
            # frame's valid samples, up to TH_PRE_SAMPLES = 2.6 M, into region A of the shading pool -- 3.6 KB per sample)
            # first, then the fused MLP over the same samples in one launch (what lies beyond: producers and MLP alternate
            # in 512 Ki-sample chunks as in render_fast).  The side stream's front of the next frame starts with this
            # frame's shading: its ~130 small launches share the chip with the producers (which leave LDS / registers /
            # the matrix pipe free) instead of time-slicing with MLP tiles that own whole CUs.
            # (Built and removed in round 5: the NEXT frame's producers on a stream of their own BESIDE this frame's fused MLP,
            # every frame in flight with its own pool -- the window between two MLP launches closes, and the MLP slows down by
            # exactly what K4 / K5t take: they fill the chip, the window was never idle time.  profiles/r05_h.)
            if os.environ.get("TH_PREGATHER") != "0":
                hip.render_pregather(self.net, frame, pts, early=early)
            # The NEXT frame's neighbour records (K4, the long pole of its producers) may start the moment this frame's
            # per-sample stage is done, beside this frame's compositing and the consumer's image assembly instead of behind
            # them (th_render_pregather_early): for that, the current stream is ordered behind piece A of the next frame's
            # front BEFORE this frame's shading is queued -- it was issued a whole frame ago (split front: lookahead >= 2).
            if queue and os.environ.get("TH_PREGATHER_EARLY", "1") != "0":
                main.wait_event(queue[0][5])
                queue[0][6] = True
            rgb, acc, depth, stats, check = hip.render_rays(self.net, frame, pts, white_bkgd=bool(cfg.white_bkgd),
                                                            defer_guard=True, small_frame_rays=small_frame_rays)
            side.wait_event(fence)
            if queue:
                tokens(queue[0], side)        # B(i+1) first ...
            pull()                            # ... then A(i+lookahead)
            shaded.append((rgb, acc, depth, stats, frame, cur, pts, check, epoch))
            if len(shaded) > 1:
                yield finish(shaded.popleft())
        while shaded:
            yield finish(shaded.popleft())

    def render(self, batch, is_train=True):
        """:486-498 -- no hull mask, every sample shaded, RGB everywhere.
        Inference (``torch.no_grad()``, the reference's ``Trainer.val``): the HIP path, forward only.
        Training (the reference's trainer calls this entry with gradients enabled and back-propagates through it,
        lib/train/trainers/if_nerf_clight.py:45, trainer.py:79-86; ``cfg.perturb`` / ``cfg.raw_noise_std`` are its
        randomisations): the HIP kernels carry no autograd, so such a call is served by
        ``transhuman_amd.networks.autograd_path`` -- the same forward composed from differentiable torch operators on
        the batch's device.  It is a separate entry for training, not a fallback of the rendering hot path: without a GPU
        this method raises like every other."""
        cfg = get_cfg()
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters())
        randomised = (float(getattr(cfg, "perturb", 0.0)) > 0.0 and self.net.training) or \
            float(getattr(cfg, "raw_noise_std", 0.0)) > 0.0
        if wants_grad or randomised:
            if not batch["ray_o"].is_cuda:
                raise hip.HipError("Renderer.render needs the batch on an MI355X (there is no CPU path)")
            from transhuman_amd.networks import autograd_path
            n_rays = int(batch["ray_o"].shape[1])
            if wants_grad and not randomised and n_rays > 4 * 2400:
                # the reference trains on patches of <= 2400 rays (:551); a whole frame with gradients enabled is almost
                # always an evaluation loop that forgot torch.no_grad() -- it would build the autograd graph of every sample
                import warnings
                warnings.warn(f"Renderer.render: {n_rays} rays with gradients enabled take the differentiable torch path "
                              "(training entry), not the HIP kernels; wrap inference in torch.no_grad()", RuntimeWarning)
            out = autograd_path.render(self, batch)
            self.last_stats = dict(hit_rays=int(batch["ray_o"].shape[1]), valid_samples=int(batch["ray_o"].shape[1]) * int(cfg.N_samples),
                                   unmasked=1)
            return out
        frame = self.prepare_frame(batch, hull_thresh=-1.0)
        pts = hip.Points(batch["ray_o"][0], batch["ray_d"][0], batch["near"][0], batch["far"][0],
                         n_samples=cfg.N_samples)
        rgb, acc, depth, stats = hip.render_rays(self.net, frame, pts, white_bkgd=bool(cfg.white_bkgd))
        self.last_stats = stats
        return {"rgb_map": rgb[None], "acc_map": acc[None], "depth_map": depth[None]}


def cfg_hull():
    from transhuman_amd.config import cfg_get
    return float(cfg_get("hull_dist", 0.1))
