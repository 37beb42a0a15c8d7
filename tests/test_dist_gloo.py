"""Multi-process path on CPU (gloo, world_size 2 and 3): interleaved 8-row tile partition,
gather to rank 0 and de-interleave.  The per-rank renderer is the CPU oracle standing in for the
HIP executor (same tile_first / tile_step contract), so the test covers exactly the code
bench.py runs around the executor for N > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as rdist
from raytracer_amd import scenegen


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _render_shard(orc, W, H, depth, rank, world):
    """What HipExecutor.render_device produces: [T_max*8][W][4] with this rank's tiles back to back."""
    tmax = rdist.max_local_tiles(H, world)
    buf = torch.zeros((tmax * 8, W, 4), dtype=torch.float32)
    for j in range(rdist.local_tiles(H, rank, world)):
        k = rank + j * world
        y0, rows = 8 * k, min(8, H - 8 * k)
        r = orc.render_region(W, H, 0, y0, W, rows, max_depth=depth)
        buf[8 * j:8 * j + rows, :, :3] = torch.from_numpy(r["rgb"])
        buf[8 * j:8 * j + rows, :, 3] = 1.0
    return buf


def _worker(rank, world, port, W, H, depth, out_path):
    from oracle.binding import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle(scenegen.small_variant("c2").to_flat())
        local = _render_shard(orc, W, H, depth, rank, world)
        frame = rdist.gather_frame(local, H)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 44), (3, 52), (2, 8)])
def test_sharded_render_gathers_to_the_unsharded_frame(tmp_path, world, H):
    from oracle.binding import Oracle
    W, depth = 24, 1
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, depth, out), nprocs=world, join=True)
    got = np.load(out)
    ref = Oracle(scenegen.small_variant("c2").to_flat()).render(W, H, max_depth=depth)["rgb"]
    assert got.shape == (H, W, 4)
    assert np.array_equal(got[..., :3].view(np.int32), ref.view(np.int32))
    assert (got[..., 3] == 1).all()


def test_tile_bookkeeping():
    for H in (1, 7, 8, 9, 1080, 2160, 43):
        t = rdist.total_tiles(H)
        for G in (1, 2, 3, 4, 8):
            counts = [rdist.local_tiles(H, r, G) for r in range(G)]
            assert sum(counts) == t and max(counts) == rdist.max_local_tiles(H, G)
            # de-interleave puts tile k = j*G + r back at rows [8k, 8k+8)
            tmax = max(counts)
            g = torch.zeros((G, tmax * 8, 2, 1))
            for r in range(G):
                for j in range(counts[r]):
                    g[r, 8 * j:8 * j + 8] = float(j * G + r)
            f = rdist.deinterleave(g, H)
            assert f.shape[0] == H
            assert torch.equal(f[:, 0, 0], torch.arange(H) // 8 * 1.0)


@pytest.mark.gpu
def test_rccl_single_rank_collectives_on_the_gpu(tmp_path):
    """The box has one GPU, so the N > 1 gather cannot run there; what can be checked is that the RCCL backend comes up
    in this process layout (one process per GPU, device bound before init) and moves a rendered tile buffer through the
    same collectives bench.py issues (all_reduce of the rank count, gather to rank 0) at world size 1."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from raytracer_amd import dist as rdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
t = torch.ones(1, device=dev); dist.all_reduce(t); assert int(t.item()) == 1
local = torch.arange(16 * 8 * 4, dtype=torch.float32, device=dev).reshape(16, 8, 4)
buf = [torch.empty_like(local)]
dist.gather(local, gather_list=buf, dst=0)
torch.cuda.synchronize()
assert torch.equal(buf[0], local)
assert torch.equal(rdist.gather_frame(local, 13), local[:13])
dist.barrier(); dist.destroy_process_group()
print("rccl ok")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-1500:]
