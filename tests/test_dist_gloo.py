"""CPU: the N>1 path (ray-tile sharding + image gather) with torch.distributed `gloo`,
world_size 2 and 3 (ragged shards), one process per rank like the GPU launch.

The per-rank "renderer" here is the CPU oracle's compositing on random raw values: the point is
the sharding / gather plumbing (transhuman_amd/dist.py), which is backend-agnostic; on the GPU box
the same code runs over RCCL (bench.py --gpus N)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from transhuman_amd.dist import ImageGatherer, DeferredSum, TokenExchange, StemExchange, gather_image, shard_ray_indices, _tile_skew


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)
        full = torch.rand((H * W, 5), generator=g)            # what a single GPU would render
        idx = shard_ray_indices(H, W, world, rank, tile=8)
        local = full[idx].clone()                              # this rank's rays
        # global decision of the reference's R' <= 2400 switch: sum of per-rank hit counts
        hits = torch.tensor([int((local[:, 3] > 0.5).sum())])
        dist.all_reduce(hits)
        img = gather_image(local, idx, H * W, world)
        ok = torch.equal(img, full) and int(hits) == int((full[:, 3] > 0.5).sum())
        # tile-major shard order (what bench.py uses) through the cached-layout gatherer, two frames
        idx2 = shard_ray_indices(H, W, world, rank, tile=8, tile_major=True)
        ga = ImageGatherer(idx2, H * W, world)
        for scale in (1.0, 2.0):
            ok = ok and torch.equal(ga(full[idx2] * scale), full * scale)
        ok = ok and torch.equal(torch.sort(idx2).values, idx)
        # bench.py's step order: start the whole-frame count, gather the image on the default group, then read the
        # count (own communicator: the two collectives never queue behind each other), for a few frames
        ds = DeferredSum(torch.device("cpu"))
        for fr in range(3):
            mine = int((local[:, 3] > 0.5).sum()) + fr * (rank + 1)
            ds.start(mine)
            img2 = ga(full[idx2] * (fr + 1.0))
            want = int((full[:, 3] > 0.5).sum()) + fr * world * (world + 1) // 2
            ok = ok and ds.result() == want and torch.equal(img2, full * (fr + 1.0))
        # TransHE sharded over the frames of a stream: frame j's tokens come from rank j % world (its own
        # communicator), every rank ends up with the owner's tensor and runs the "ViT" for its own frames only;
        # interleaved with the image gather of the render stream exactly like Renderer.render_sequence + bench.step
        tx = TokenExchange()
        calls = []
        for fr in range(2 * world + 1):
            def vit(fr=fr):
                calls.append(fr)
                return torch.full((3, 7, 192), float(fr)) + torch.arange(192.0) * (rank + 1)   # rank-specific payload
            tok = tx(vit, (3, 7, 192), torch.device("cpu"))
            want_tok = torch.full((3, 7, 192), float(fr)) + torch.arange(192.0) * (fr % world + 1)
            ok = ok and torch.equal(tok, want_tok)
            ok = ok and torch.equal(ga(full[idx2] * (fr + 1.0)), full * (fr + 1.0))
        ok = ok and calls == [fr for fr in range(2 * world + 1) if fr % world == rank] and tx.computed == len(calls)
        # the encoder stem sharded over the frames as well (its own communicator, owner staggered by world // 2): every rank
        # receives the owner's three latents, interleaved with the token exchange and the image gather like in a frame
        sx = StemExchange()
        tx2 = TokenExchange()
        stem_calls = []
        imgs = torch.zeros(3, 3, 16, 24)
        shapes = StemExchange.latent_shapes(3, 16, 24)
        ok = ok and shapes == [(3, 64, 8, 12), (3, 64, 4, 6), (3, 128, 2, 3)]
        for fr in range(2 * world + 1):
            def trunk(x, fr=fr):
                stem_calls.append(fr)
                return [torch.full(sh, float(fr) + 0.25 * k) + (rank + 1) for k, sh in enumerate(shapes)]
            lat = sx.latents(trunk, imgs)
            own = (fr + world // 2) % world
            ok = ok and all(tuple(l.shape) == sh and bool((l == float(fr) + 0.25 * k + own + 1).all())
                            for k, (l, sh) in enumerate(zip(lat, shapes)))
            tok = tx2(lambda fr=fr: torch.full((3, 7, 192), float(fr) + rank), (3, 7, 192), torch.device("cpu"))
            ok = ok and bool((tok == float(fr) + fr % world).all())
            ok = ok and torch.equal(ga(full[idx2] * (fr + 1.0)), full * (fr + 1.0))
        ok = ok and stem_calls == [fr for fr in range(2 * world + 1) if (fr + world // 2) % world == rank]
        # mesh workload: voxel runs of 4096 dealt to the ranks (bench.py run_secondary), sigma gathered back
        g = 32
        nvox = g * g * g
        sig_full = torch.rand(nvox, generator=torch.Generator().manual_seed(7))
        vox = torch.arange(nvox)
        run = vox // 4096
        per_slice = max(1, (g * g) // 4096)
        mine = vox[((run % per_slice + _tile_skew(world) * (run // per_slice)) % world) == rank]
        sig = gather_image(sig_full[mine][:, None], mine, nvox, world)[:, 0]
        ok = ok and torch.equal(sig, sig_full)
        cover = torch.zeros(nvox, dtype=torch.int64)
        cover[mine] = 1
        dist.all_reduce(cover)
        ok = ok and int(cover.min()) == 1 and int(cover.max()) == 1
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run(world, H, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(world))
    assert all(ok for _, ok in res)


def test_gather_world2_balanced():
    _run(2, 64, 64)


def test_gather_world3_ragged_tiles():
    _run(3, 40, 56)          # 5 x 7 tiles over 3 ranks: unequal shard lengths -> padded all_gather


def test_shards_are_balanced_over_a_centred_subject():
    """the tile dealing must not hand a rank the same tile columns in every row (512 / 8 = 64 tiles per row and 8
    ranks did exactly that with t % world: vertical stripes, +19 % samples on the busiest rank): pixels of an
    off-centre ellipse (a stand-in for the body's silhouette) are shared out within a few per cent, for every
    world size, and the shards still partition the image"""
    H = W = 512
    y, x = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    inside = (((x - 250.0) / 70.0) ** 2 + ((y - 270.0) / 190.0) ** 2 <= 1.0).reshape(-1)
    for world in (2, 3, 4, 8):
        seen = torch.zeros(H * W, dtype=torch.int32)
        counts = []
        for r in range(world):
            idx = shard_ray_indices(H, W, world, r, tile=8, tile_major=True)
            seen[idx] += 1
            counts.append(int(inside[idx].sum()))
        assert int(seen.min()) == 1 and int(seen.max()) == 1
        mean = sum(counts) / world
        assert max(counts) <= 1.04 * mean and min(counts) >= 0.96 * mean, (world, counts)


def test_token_exchange_emulation_counts_the_owned_frames():
    """bench.py --emulate-world N: the per-rank work of an N-rank job on one device (no process group) -- the ViT runs
    for the frames this rank owns (and once more if its first frame is not its own)"""
    for world, rank in ((8, 0), (8, 5), (3, 2)):
        tx = TokenExchange(emulate=(world, rank))
        calls = []
        for fr in range(17):
            def vit(fr=fr):
                calls.append(fr)
                return torch.full((1, 2, 192), float(fr))
            before = len(calls)
            predicted = tx.will_compute()            # (the renderer marks the painted vertices' texels only on such frames)
            tok = tx(vit, (1, 2, 192), torch.device("cpu"))
            assert tok.shape == (1, 2, 192)
            assert predicted == (len(calls) > before), fr
        owned = [fr for fr in range(17) if fr % world == rank]
        assert calls == ([0] if rank != 0 else []) + owned


def test_stem_exchange_emulation_and_switch(monkeypatch):
    """emulation: the stem is computed for the frames this rank owns only (owner staggered by world // 2); the default is
    off unless TH_STEM_EXCHANGE=1 asks for it (never measured on a multi-GPU node: the safe variant is the default)"""
    world, rank = 8, 3
    sx = StemExchange(emulate=(world, rank))
    shapes = StemExchange.latent_shapes(1, 32, 32)
    calls = []
    for fr in range(20):
        lat = sx.latents(lambda x, fr=fr: (calls.append(fr), [torch.zeros(sh) for sh in shapes])[1], torch.zeros(1, 3, 32, 32))
        assert [tuple(l.shape) for l in lat] == shapes
    assert calls == [0] + [fr for fr in range(1, 20) if (fr + 4) % 8 == 3]      # (frame 0: nothing to re-use yet)
    assert sx.last_flag is not None and not bool(sx.last_flag)                  # finite latents: no flag
    sx2 = StemExchange(emulate=(2, 1))
    bad = [torch.zeros(sh) for sh in shapes]
    bad[1][0, 0, 0, 0] = float("inf")                                           # an fp16 overflow on the owner shows up like this
    sx2.latents(lambda x: bad, torch.zeros(1, 3, 32, 32))
    assert bool(sx2.last_flag)                                                   # ... and every rank that uses the latents sees it
    monkeypatch.delenv("TH_STEM_EXCHANGE", raising=False)
    assert not StemExchange.wanted(2) and not StemExchange.wanted(4) and not StemExchange.wanted(8)   # default OFF (round 4)
    monkeypatch.setenv("TH_STEM_EXCHANGE", "0")
    assert not StemExchange.wanted(8)
    monkeypatch.setenv("TH_STEM_EXCHANGE", "1")
    assert StemExchange.wanted(2)
