"""world_size-2 (and 3) gloo test of the multi-GPU path's host logic: row-stripe tiles + the
reduce(sum) of per-tile radiance reassemble the single-process frame bit-for-bit.
No GPU here, so each rank's tile is rendered by the CPU oracle standing in for the HIP launch
(the launch itself is covered by tests/test_gpu_parity.py::test_tile_union_equals_full_frame)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    import importlib
    import torch
    import torch.distributed as dist
    for p in (str(ROOT), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_py
    hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp = 64, 50, 2
    scene = hrt.scenes.mixed_test_scene(400, 10, 3, W, H, spp)
    tile = hrt.tile_for_rank(H, rank, world, stripe_rows=4)
    rows = np.array([y for y in range(tile.y_begin, tile.y_end) if (y // tile.stripe_rows) % tile.stripe_period == tile.stripe_phase], np.uint32)
    states = oracle_py.rng_init(W, H, hrt.scenes.SEED_SALT)            # global pixel index = RNG stream: every rank inits the same array
    part = oracle_py.OracleScene(scene).render(W, H, states, spp, rows=rows)["color"]
    frame = torch.from_numpy(part.copy())
    hrt.reduce_tiles(frame, dst=0)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_split_reduce_gloo(world, tmp_path, hrt, oracle):
    import torch.multiprocessing as mp
    out = tmp_path / "frame.npy"
    mp.spawn(_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    got = np.load(out)
    W, H, spp = 64, 50, 2
    scene = hrt.scenes.mixed_test_scene(400, 10, 3, W, H, spp)
    want = oracle.OracleScene(scene).render(W, H, oracle.rng_init(W, H, hrt.scenes.SEED_SALT), spp)["color"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
