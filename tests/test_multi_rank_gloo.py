"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` processes shard the 32x32 work-groups
round-robin, each fills its tile buffer (the oracle stands in for the kernel here), ONE gather
brings the buffers to rank 0, which de-interleaves them — and the result equals the single-rank
image bit for bit.  Same host arithmetic as bench.py uses with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, bounce, out_path):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py
    from tdt4230_project_raytracing_amd import host, tiles
    scene = host.Scene.demo()
    cam = host.camera_reference_pose(W, H, spp, bounce)
    cw, ch = tiles.cover(W, H, W + 1, H + 1)
    tx, ty, total = tiles.tile_grid(cw, ch)
    cap = tiles.tiles_per_rank(total, world)
    orc = oracle_py.Oracle()
    # render only the pixel rows this rank needs, then keep only its own tiles
    mine = [rank + k * world for k in range(tiles.owned_tiles(total, rank, world))]
    full = np.zeros((H, W, 4), np.float32)
    for gy in sorted({t // tx for t in mine}):
        orc.render(scene, cam, rows=(gy * 32, gy * 32 + 32), threads=1, image=full)
    buf = torch.from_numpy(tiles.pack_tiles(full, cw, ch, rank, world, cap))
    gl = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gl, dst=0)
    if rank == 0:
        img = tiles.assemble(torch.stack(gl).numpy(), W, H, cw, ch, world)
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_render_equals_single_rank(tmp_path, oracle, world):
    from tdt4230_project_raytracing_amd import host
    W, H, spp, bounce = 160, 100, 1, 4
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, bounce, out), nprocs=world, join=True)
    got = np.load(out)
    ref = oracle.render(host.Scene.demo(), host.camera_reference_pose(W, H, spp, bounce), threads=4)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
