"""GPU: the N-rank job of bench.py end to end on ONE device -- `torch.distributed.run` with two / three ranks that all use
cuda:0 over gloo (RCCL refuses two ranks per device; `TH_DIST_BACKEND=gloo TH_ONE_GPU=1`).  Everything but the transport is
what an 8-GPU node runs: diagonal 8x8 ray-tile shards (ragged at three ranks), TransHE owned by rank j mod N and broadcast
from the side stream, the deferred whole-frame hit count, the cached-layout image gather.  The gathered frame must equal
the single-rank frame to fp32 rounding (rays are independent given the per-frame constants: SURVEY.md 8e)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n, path, res, extra=(), env_extra=None):
    env = dict(os.environ, TH_SAVE_IMAGE=path, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    if n == 1:
        cmd = [sys.executable, "bench.py", "--res", str(res)] + ARGS + list(extra)
    else:
        env.update(TH_DIST_BACKEND="gloo", TH_ONE_GPU="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(n), "--res", str(res)] + ARGS + list(extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


@pytest.mark.gpu
@pytest.mark.parametrize("res", [128, 96])
def test_two_and_three_rank_frames_equal_the_single_rank_frame(tmp_path, res):
    """res 128: ~3200 hit rays (masked branch); res 96: ~1800 <= 2400 -- the reference's small-frame rule fires on the
    WHOLE-frame count (all-reduced) and every rank re-renders its shard un-masked (if_clight_renderer.py:551)"""
    one = _run(1, str(tmp_path / "n1.npy"), res)
    hits = int((one[:, 3] > 0).sum())
    assert one.shape == (res * res, 5) and ((res == 128 and hits > 2400) or (res == 96 and 500 < hits <= 2400)), hits
    for n in (2, 3):
        img = _run(n, str(tmp_path / f"n{n}.npy"), res)
        # (equal to fp32 rounding: a rank's shard regroups the valid samples into other 32-sample tiles, and the fused kernel's
        # token blend accumulates over the union of a tile's neighbour centres)
        assert one.shape == img.shape and float(np.abs(one[:, :4] - img[:, :4]).max()) < 2e-6, (n, float(np.abs(one - img).max()))
        assert float(np.abs(one[:, 4] - img[:, 4]).max()) < 2e-5 * max(1.0, float(np.abs(one[:, 4]).max()))


@pytest.mark.gpu
def test_stem_exchange_frames_equal_the_single_rank_frame(tmp_path):
    """dist.StemExchange forced on (it is off below 4 ranks by default): the ResNet stem of frame j runs on one rank, its
    three latents are broadcast from the side stream next to the token broadcast, every rank builds the map locally --
    the gathered frames of 2- and 3-rank jobs equal the single-rank frame"""
    one = _run(1, str(tmp_path / "n1.npy"), 128)
    for n in (2, 3):
        img = _run(n, str(tmp_path / f"s{n}.npy"), 128, env_extra={"TH_STEM_EXCHANGE": "1"})
        assert one.shape == img.shape and float(np.abs(one[:, :4] - img[:, :4]).max()) < 2e-6, (n, float(np.abs(one - img).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra", [("orbit", ()), ("mesh", ("--grid", "48"))])
def test_secondary_workloads_two_ranks(tmp_path, workload, extra):
    """C3 (orbit along gen_path_virt, rays generated on device, pixel tiles dealt to the ranks) and C5 (sigma grid, voxel
    runs dealt to the ranks): the gathered result of a two-rank job equals the single-rank result"""
    ex = ("--workload", workload) + tuple(extra)
    one = _run(1, str(tmp_path / "n1.npy"), 192, ex)
    two = _run(2, str(tmp_path / "n2.npy"), 192, ex)
    assert one.shape == two.shape and float(np.abs(one).max()) > 0
    assert float(np.abs(one - two).max()) <= 2e-5 * max(1.0, float(np.abs(one).max())), float(np.abs(one - two).max())


@pytest.mark.gpu
def test_one_rank_job_over_rccl_equals_the_plain_frame(tmp_path):
    """the transport the gloo runs above replace: `TH_FORCE_DIST=1` makes bench.py take its N-rank path with world size 1
    on the real backend ("nccl" = RCCL): init_process_group on the device, the deferred hit-count all_reduce on its own
    communicator and stream, the token broadcast from the side stream, all_gather_into_tensor of the image.  One rank owns
    every ray tile in ascending order, so the gathered frame is the plain frame bit for bit."""
    one = _run(1, str(tmp_path / "n1.npy"), 128)
    env = dict(os.environ, TH_SAVE_IMAGE=str(tmp_path / "rccl.npy"), TH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", TH_STEM_EXCHANGE="1")      # (+ the latent broadcast)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TH_DIST_BACKEND", "TH_ONE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--res", "128"] + ARGS, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    img = np.load(str(tmp_path / "rccl.npy"))
    assert img.shape == one.shape and np.array_equal(img, one), float(np.abs(img - one).max())
