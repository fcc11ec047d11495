"""Multi-GPU ray sharding: one process per GPU, torch.distributed (RCCL over xGMI).

The reference renders on a single GPU (every script passes `gpus "${CARD},"`);
its only collectives are DDP training ones.  Rays are independent
given the per-frame constants, so inference shards them with no data-path
exchange; the one collective is the image gather (R/N x 5 fp32 per rank,
655 KB at N=8 for a 512x512 frame -- latency-bound, far below the 153 GB/s of
an xGMI link).  Hull rays are spatially clustered, hence the interleaved
8x8-pixel tile deal instead of contiguous row blocks (hard part 8).
"""
import torch


def _tile_skew(world):
    """row-to-row shift of the tile dealing: coprime with `world`, so that the tiles of a rank form diagonals"""
    import math
    for c in (3, 5, 7, 11, 13):
        if math.gcd(c, world) == 1:
            return c
    return 1


def shard_ray_indices(H, W, world, rank, tile=8, tile_major=False):
    """Indices (into the row-major H*W ray list) of the rays owned by `rank`:
    pixel tile (ty, tx) belongs to rank (tx + skew * ty) % world -- diagonals of tiles.  (Dealing the row-major tile
    index t % world hands every rank the SAME tile columns in every tile row whenever the tiles-per-row count is a
    multiple of `world` (512 / 8 = 64 tiles, 8 ranks): vertical stripes, and a body that covers some stripes more than
    others: 19 % more samples on the busiest of 8 ranks than on average.)
    tile_major=False: ascending ray index.  True: tile after tile (row-major inside a tile), so that rays
    that are consecutive in the shard are 2-D neighbours in the image: their samples project to
    neighbouring pixels of the reference views and the pixel gather re-uses cache lines in both directions."""
    ty = (H + tile - 1) // tile
    tx = (W + tile - 1) // tile
    y = torch.arange(H)
    x = torch.arange(W)
    tid = (y[:, None] // tile) * tx + (x[None, :] // tile)
    own = (((x[None, :] // tile) + _tile_skew(world) * (y[:, None] // tile)) % world) == rank
    idx = torch.nonzero(own.reshape(-1), as_tuple=False).reshape(-1)
    if tile_major:
        key = tid.reshape(-1)[idx] * (H * W) + idx          # (tile, row-major position): unique
        idx = idx[torch.argsort(key)]
    return idx


def shard_lengths(H, W, world, tile=8):
    return [int(shard_ray_indices(H, W, world, r, tile).numel()) for r in range(world)]


def _into_tensor_ok(t, group=None):
    """all_gather_into_tensor is the RCCL path (device tensors, backend nccl); gloo -- CPU tests, or the one-GPU
    multi-rank check of bench.py (TH_DIST_BACKEND=gloo) -- gathers a list"""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "nccl"


def gather_image(local, my_idx, n_rays, world, H=None, W=None, tile=8, group=None):
    """all_gather the per-rank [n_local, C] results into the dense [n_rays, C] image on
    every rank.  Shards may differ in length (ragged tile counts): padded to the max."""
    import torch.distributed as dist
    C = local.shape[1]
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    lens = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(lens, n_local, group=group)
    lens = [int(l) for l in lens]
    mx = max(lens)
    pad_val = torch.zeros((mx, C), dtype=local.dtype, device=local.device)
    pad_val[: local.shape[0]] = local
    pad_idx = torch.full((mx,), -1, dtype=torch.int64, device=local.device)
    pad_idx[: my_idx.numel()] = my_idx
    all_val = torch.empty((world * mx, C), dtype=local.dtype, device=local.device)
    all_idx = torch.empty((world * mx,), dtype=torch.int64, device=local.device)
    if _into_tensor_ok(local, group):
        dist.all_gather_into_tensor(all_val, pad_val, group=group)
        dist.all_gather_into_tensor(all_idx, pad_idx, group=group)
    else:  # gloo (CPU tests)
        vs = [torch.empty_like(pad_val) for _ in range(world)]
        is_ = [torch.empty_like(pad_idx) for _ in range(world)]
        dist.all_gather(vs, pad_val, group=group)
        dist.all_gather(is_, pad_idx, group=group)
        all_val, all_idx = torch.cat(vs), torch.cat(is_)
    keep = all_idx >= 0
    img = torch.zeros((n_rays, C), dtype=local.dtype, device=local.device)
    img[all_idx[keep]] = all_val[keep]
    return img


class ImageGatherer:
    """gather_image with the shard layout exchanged ONCE: the ray -> rank assignment is a function of the camera
    resolution only, so the per-frame exchange is a single all_gather_into_tensor of the padded [max_local, C]
    blocks (one collective, no host synchronisation) followed by one index_copy into the dense image."""

    def __init__(self, my_idx, n_rays, world, channels=5, group=None):
        import torch.distributed as dist
        self.world, self.n_rays, self.group, self.C = world, n_rays, group, channels
        dev = my_idx.device
        n_local = torch.tensor([my_idx.numel()], dtype=torch.int64, device=dev)
        lens = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(lens, n_local, group=group)
        self.lens = [int(l) for l in lens]
        self.mx = max(self.lens)
        pad_idx = torch.full((self.mx,), -1, dtype=torch.int64, device=dev)
        pad_idx[: my_idx.numel()] = my_idx
        idxs = [torch.empty_like(pad_idx) for _ in range(world)]
        dist.all_gather(idxs, pad_idx, group=group)
        all_idx = torch.cat(idxs)
        self.keep = torch.nonzero(all_idx >= 0, as_tuple=False).reshape(-1)
        self.dst = all_idx[self.keep]
        self.n_local = my_idx.numel()
        self.pad = torch.zeros((self.mx, channels), dtype=torch.float32, device=dev)
        self.all_val = torch.empty((world * self.mx, channels), dtype=torch.float32, device=dev)

    def __call__(self, local):
        import torch.distributed as dist
        assert local.shape == (self.n_local, self.C)
        self.pad[: self.n_local] = local
        if _into_tensor_ok(local, self.group):
            dist.all_gather_into_tensor(self.all_val, self.pad, group=self.group)
            vals = self.all_val
        else:  # gloo (CPU tests)
            vs = [torch.empty_like(self.pad) for _ in range(self.world)]
            dist.all_gather(vs, self.pad, group=self.group)
            vals = torch.cat(vs)
        img = torch.zeros((self.n_rays, self.C), dtype=local.dtype, device=local.device)
        img.index_copy_(0, self.dst, vals.index_select(0, self.keep))
        return img


class TokenExchange:
    """TransHE (the ViT over the N_c tokens, vision_transformer.py:371-383) computed by ONE rank per frame instead
    of by every rank: frame j belongs to rank j % world, which paints / groups / runs the ViT and broadcasts the
    [V, N_c, 192] tokens (1.15 MB at N_c = 500) -- the other ranks skip that work for frame j.  Ray sharding leaves
    the per-frame constants replicated; of those the ViT is the one piece whose OUTPUT is small, so it is the one
    worth moving over xGMI (the 0.82 GB feature map is not: recomputing its 0.8 ms encoder is cheaper than any
    exchange).  At N = 8 a rank runs the ViT for every 8th frame only.

    All ranks see the frames in the same order, so the broadcast roots agree without negotiation.  The collective
    has its own communicator (never queued behind the image gather of the render stream) and is issued from
    whatever stream is current -- Renderer.render_sequence calls it from its side stream, i.e. under the shading of
    the previous frame, ``lookahead`` frames ahead of use.

    ``emulate=(world, rank)`` (one GPU, no process group): the owner test is applied, non-owned frames re-use the
    last tokens this rank computed -- the per-rank WORK of an N-rank job for timing, not its image."""

    def __init__(self, world=None, rank=None, group=None, emulate=None, shift=0):
        self.emulate = emulate
        self.shift = int(shift)     # owner of frame j = (j + shift) % world: a second exchange (StemExchange) is staggered
        if emulate is not None:
            self.world, self.rank = emulate
            self.group = None
        else:
            import torch.distributed as dist
            self.world = dist.get_world_size() if world is None else world
            self.rank = dist.get_rank() if rank is None else rank
            self.group = dist.new_group() if group is None else group
        self.frame = 0
        self.computed = 0           # frames whose tokens this rank produced itself
        self.last_mine = False      # the last frame handed out was computed here
        self._last = None

    def owner(self, j):
        return (j + self.shift) % self.world

    def will_compute(self):
        """whether the NEXT call runs ``compute`` on this rank (its frame is this rank's, or the emulation has nothing to re-use)"""
        return self.owner(self.frame) == self.rank or (self.emulate is not None and self._last is None)

    def __call__(self, compute, shape, device, dtype=torch.float32):
        """tokens of the next frame: ``compute()`` on the owner, a receive buffer elsewhere, then the broadcast"""
        j = self.frame
        self.frame += 1
        mine = self.last_mine = self.owner(j) == self.rank
        if mine:
            tok = compute().contiguous()
            assert tuple(tok.shape) == tuple(shape) and tok.dtype == dtype
            self.computed += 1
        if self.emulate is not None:
            if mine or self._last is None:
                if not mine:
                    tok = compute().contiguous()
                self._last = tok
            return self._last
        if not mine:
            tok = torch.empty(tuple(shape), dtype=dtype, device=device)
        import torch.distributed as dist
        dist.broadcast(tok, src=self.owner(j), group=self.group)
        return tok


class _HostFlag:
    """A device-side boolean on its way to the host: copied into pinned memory on the stream that computed it (the side
    stream, frames ahead of its use) and read behind an event -- ``bool(flag)`` never waits for anything but that copy.
    (``bool(device_tensor)`` would copy on the CURRENT stream and drain the whole shading queue first.)"""

    def __init__(self, flag):
        if flag.is_cuda:
            self.host = torch.empty(1, dtype=torch.bool, pin_memory=True)
            self.host.copy_(flag.reshape(1), non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(flag.device))
        else:
            self.host, self.event = flag.reshape(1).clone(), None

    def __bool__(self):
        if self.event is not None:
            self.event.synchronize()
        return bool(self.host[0])


class StemExchange(TokenExchange):
    """The ResNet stem of SpatialEncoder (encoder.py:114-126: 31 GFLOP of convolutions + ten train-mode BatchNorms, 0.5 ms
    on one MI355X) computed by ONE rank per frame as well: its output, the three low-resolution latents ([V,64,H/2,W/2],
    [V,64,H/4,W/4], [V,128,H/8,W/8]: 69 MB at V = 3, 512^2), is 12 times smaller than the 0.82 GB map built from it, so
    every rank keeps the (local, 0.3 ms) upsample / concat and receives the latents.  Same protocol as TokenExchange (own
    communicator, issued from the side stream a frame or two ahead of use); the owner is staggered by half the world size so
    that a rank does not own the stem and TransHE of the same frame.
    What it buys is an ESTIMATE until a multi-GPU node has run it: at N = 8 a rank's frame is 3.5 ms of which the replicated
    stem is 0.5 ms of chip time on 7 frames in 8; the broadcast moves 69 MB per frame and rank (21 GB/s inbound at 3.2 ms per
    frame, against 7 xGMI links of ~153 GB/s) from a stream of its own.  OFF by default (round 4): the first run on a
    real multi-GPU node measures the safe variant; TH_STEM_EXCHANGE=1 switches it on (tools/run_scale.sh runs both and
    prints the A/B), and only a hardware line that shows it paying flips the default.
    Side effect: the stem's BatchNorm running statistics advance only on the owner of a frame (they do not enter the
    train()-mode forward the renderer runs, run.py:29)."""

    def __init__(self, world=None, rank=None, group=None, emulate=None):
        w = emulate[0] if emulate is not None else (world if world is not None else None)
        if w is None:
            import torch.distributed as dist
            w = dist.get_world_size()
        super().__init__(world, rank, group, emulate, shift=w // 2)
        self.last_flag = None

    @staticmethod
    def wanted(world):
        import os
        return os.environ.get("TH_STEM_EXCHANGE") == "1" and world >= 2

    @staticmethod
    def latent_shapes(V, H, W):
        h1, w1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1          # conv1 7x7 / 2, padding 3
        h2, w2 = (h1 + 2 - 3) // 2 + 1, (w1 + 2 - 3) // 2 + 1        # maxpool 3x3 / 2, padding 1 (layer1 keeps the size)
        h3, w3 = (h2 + 2 - 3) // 2 + 1, (w2 + 2 - 3) // 2 + 1        # layer2: 3x3 / 2, padding 1
        return [(V, 64, h1, w1), (V, 64, h2, w2), (V, 128, h3, w3)]

    def latents(self, trunk, images):
        """-> [lat0, lat1, lat2] of ``images`` ([V,3,H,W]): ``trunk(images)`` on the owner of this frame, received elsewhere"""
        V, _, H, W = images.shape
        shapes = self.latent_shapes(V, H, W)
        sizes = [int(torch.Size(sh).numel()) for sh in shapes]

        def compute():
            lat = trunk(images)
            assert [tuple(l.shape) for l in lat] == shapes, "unexpected latent shapes"
            return torch.cat([l.reshape(-1) for l in lat])
        flat = self(compute, (sum(sizes),), images.device)
        # the stem's fp16-range word lives on the owner of the frame: a receiver would otherwise learn of an overflow (inf /
        # NaN in the hi planes of the owner's convolutions) only through its own MLP guard and fall back to the 7x slower
        # fp32 MLP for good.  Every rank looks at what it is about to use instead (one 69 MB reduction, asynchronous); the
        # frame pipeline reads the flag when the frame is finished and rebuilds the frame locally through the stock
        # convolutions (Renderer.render_sequence -> hip.force_conv_fallback).
        self.last_flag = _HostFlag(~torch.isfinite(flat).all())
        out, o = [], 0
        for sh, n in zip(shapes, sizes):
            out.append(flat[o:o + n].view(sh))
            o += n
        return out


class DeferredSum:
    """Sum of one integer per rank (the whole-frame hit-ray count of the reference's R' <= 2400 rule,
    if_clight_renderer.py:551) that never stalls the render stream: the 8-byte all-reduce runs on its own
    communicator and (on a GPU) its own stream, ``start`` is called as soon as the rank knows its count and
    ``result`` after the frame's work has been queued -- it waits for that small collective only, not for the
    shading queued on the render stream in between."""

    def __init__(self, device):
        import torch.distributed as dist
        self.group = dist.new_group()                      # own communicator: never queued behind the image gather
        self.buf = torch.zeros(1, dtype=torch.int64, device=device)
        self.stream = torch.cuda.Stream(device) if torch.device(device).type == "cuda" else None
        self.work = None

    def _ctx(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def start(self, n):
        import torch.distributed as dist
        with self._ctx():
            self.buf.fill_(int(n))
            self.work = dist.all_reduce(self.buf, group=self.group, async_op=True)

    def result(self):
        with self._ctx():
            self.work.wait()
            return int(self.buf)
