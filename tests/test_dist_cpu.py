"""N > 1 path on CPU: tiling arithmetic and the gather/de-interleave with gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tiles_partition_the_image():
    from rust_raytrace_amd import dist as rd
    for H in (1, 7, 16, 33, 64, 100, 2048):
        for world in (1, 2, 3, 4, 8):
            for S in (1, 4, 16, 32):
                seen = np.concatenate([rd.tile_rows_for(r, world, H, S) for r in range(world)])
                assert sorted(seen.tolist()) == list(range(H)), (H, world, S)
                for r in range(world):
                    t = rd.rank_tile(r, world, H, S)
                    assert np.array_equal(rd.tile_rows(t, H), rd.tile_rows_for(r, world, H, S))


def _worker_u8(rank, world, port, H, W, S, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from rust_raytrace_amd import dist as rd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = rd.tile_rows_for(rank, world, H, S)
    local = torch.zeros((len(rows), W, 3), dtype=torch.uint8)
    for i, r in enumerate(rows):
        local[i, :, 0] = int(r) % 251
        local[i, :, 1] = (torch.arange(W) * 7 % 256).to(torch.uint8)
        local[i, :, 2] = rank + 1
    frame = rd.gather_frame(local, rank, world, H, W, S)
    if rank == 0:
        q.put(frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_quantised_bands_gloo_world2():
    import torch.multiprocessing as mp
    world, W, H, S = 2, 5, 37, 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_u8, args=(r, world, port, H, W, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert frame.shape == (H, W, 3) and frame.dtype == np.uint8
    assert np.array_equal(frame[:, 0, 0], (np.arange(H) % 251).astype(np.uint8))
    assert np.array_equal(frame[:, 0, 2], ((np.arange(H) // S) % world + 1).astype(np.uint8))


def _worker(rank, world, port, H, W, S, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from rust_raytrace_amd import dist as rd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = rd.tile_rows_for(rank, world, H, S)
    # stand-in for the render: pixel value encodes (row, col) so that any misplaced row shows
    local = torch.zeros((len(rows), W, 4), dtype=torch.float32)
    for i, r in enumerate(rows):
        local[i, :, 0] = float(r)
        local[i, :, 1] = torch.arange(W, dtype=torch.float32)
        local[i, :, 2] = float(rank)
    frame = rd.gather_frame(local, rank, world, H, W, S)
    if rank == 0:
        q.put(frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,S", [(64, 16), (50, 16), (9, 4)])
def test_gather_frame_gloo_world2(H, S):
    import torch.multiprocessing as mp
    world, W = 2, 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert frame.shape == (H, W, 4)
    assert np.array_equal(frame[:, 0, 0], np.arange(H, dtype=np.float32))
    assert np.array_equal(frame[3, :, 1], np.arange(W, dtype=np.float32))
    owner = (np.arange(H) // S) % world
    assert np.array_equal(frame[:, 0, 2], owner.astype(np.float32))


def test_bench_self_launches_its_ranks(monkeypatch):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE) must start the N ranks itself -- torch.distributed.run
    as a CHILD process, before the parent touches a GPU -- and exit with their status (round 1 raised SystemExit here,
    which would have failed the driver's scaling run)."""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the children's status is the parent's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
