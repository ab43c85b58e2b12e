"""N > 1 path on CPU: two gloo ranks render their stripes (with the oracle, this is a CPU test), gather them
through the same helper bench.py uses, and the re-assembled image must equal the unsharded one bit for bit."""
import os
import socket

import numpy as np
import pytest

from tests.scenes import SCENES, bits, scene_path

drt = pytest.importorskip("dustraytracer_amd")
W, H, STRIPE, FRAMES = 64, 44, 8, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    import oracle
    from dustraytracer_amd.sharding import gather_shards, padded_rows, shard_row_map
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name = "cornell_box"
    _, pos, fwd, _ = SCENES[name]
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    cam = oracle.default_camera(position=pos, forward=fwd)
    st = oracle.default_settings(ray_bounce_limit=3)
    full, _, _ = oracle.render(osc, cam, st, W, H, 1, FRAMES, threads=2, stripe_rows=STRIPE, rank=rank, world=world)
    rows = shard_row_map(H, STRIPE, rank, world)
    assert len(rows) == drt.shard_rows(H, STRIPE, rank, world)          # Python map == C ABI bookkeeping
    pad = padded_rows(H, STRIPE, world)
    local = torch.zeros((pad, W, 4), dtype=torch.float32)
    local[: len(rows)] = torch.from_numpy(full[rows])                   # compact, stripe after stripe
    gathered = torch.zeros((world, pad, W, 4), dtype=torch.float32) if rank == 0 else None
    gather_shards(local, gathered, rank)
    if rank == 0:
        image = np.zeros((H, W, 4), np.float32)
        for r in range(world):
            rr = shard_row_map(H, STRIPE, r, world)
            image[rr] = gathered[r, : len(rr)].numpy()
        np.save(out_path, image)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_gather_reassembles_image(tmp_path, world):
    import torch.multiprocessing as mp

    import oracle
    out = str(tmp_path / "image.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    name = "cornell_box"
    _, pos, fwd, _ = SCENES[name]
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
    ref, _, _ = oracle.render(osc, oracle.default_camera(position=pos, forward=fwd),
                              oracle.default_settings(ray_bounce_limit=3), W, H, 1, FRAMES, threads=2)
    assert np.array_equal(bits(np.load(out)), bits(ref))


def test_shard_bookkeeping():
    from dustraytracer_amd.sharding import shard_row_map
    for height, stripe, world in ((1080, 8, 8), (1080, 8, 3), (70, 5, 4), (7, 8, 2), (2160, 16, 8)):
        seen = np.concatenate([shard_row_map(height, stripe, r, world) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(height))             # a partition of the rows
        for r in range(world):
            assert drt.shard_rows(height, stripe, r, world) == len(shard_row_map(height, stripe, r, world))
