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


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("W,H", [(64, 44), (7, 8), (33, 1), (16, 129)])
def test_group_gather_offsets_put_every_stripe_in_place(world, W, H):
    """drt_shard_stripe is what drt_group_render_batch builds its ncclSend / ncclRecv offsets from (csrc/drt_group.cpp): sending
    every stripe of every rank to its destination offset must rebuild the image, short last stripe included, and a rank's
    stripes must tile its compact shard exactly (CPU check of the multi-GPU entry point's bookkeeping; worlds 2, 3 and 8)."""
    rng = np.random.default_rng(W * 1000 + H + world)
    image = rng.random((H, W, 4), dtype=np.float32)
    rebuilt = np.full(H * W * 4, np.nan, np.float32)
    for rank in range(world):
        rows = [y for y in range(H) if (y // STRIPE) % world == rank]
        shard = image[rows].reshape(-1)                                 # what the rank's renderer holds (compact)
        assert len(rows) == drt.shard_rows(H, STRIPE, rank, world)
        covered, k = 0, 0
        while True:
            st = drt.shard_stripe(W, H, STRIPE, rank, world, k)
            if st is None:
                break
            src, dst, cnt = st
            assert src == covered and cnt > 0                           # stripes tile the shard with no gap
            rebuilt[dst:dst + cnt] = shard[src:src + cnt]
            covered += cnt
            k += 1
        assert covered == shard.size
        assert drt.shard_stripe(W, H, STRIPE, rank, world, k + 1) is None
    assert np.array_equal(bits(rebuilt.reshape(H, W, 4)), bits(image))
    assert drt.shard_stripe(W, H, 0, 0, world, 0) is None and drt.shard_stripe(W, H, STRIPE, world, world, 0) is None


@pytest.mark.gpu
def test_group_of_one_device_equals_the_single_renderer():
    """The C ABI's multi-GPU entry point (drt_group_*) with the one device this box has: same frames, same bits as drt_renderer_*,
    blocking and asynchronous, across a resize; a second batch continues the accumulation."""
    import dustraytracer_amd as d
    sc = d.Scene(); sc.loadGLTFmodel(scene_path("cornell_box"))
    b = d.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    _, pos, fwd, _ = SCENES["cornell_box"]
    cam = d.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    g, r = d.RendererGroup([0]), d.Renderer(0)
    assert g.size() == 1
    for w, h in ((160, 90), (61, 37)):
        for x in (g, r):
            x.m_RendererSettings = d.RendererSettings(ray_bounce_limit=4, max_samples=100)
            x.ResizeBuffer(w, h)
            x.resetAccumulationBuffer()
        g.RenderBatch(cam, sc, 3); r.RenderBatch(cam, sc, 3)
        assert g.getSampleCount() == r.getSampleCount() == 4
        assert np.array_equal(bits(g.GetRenderTargetImage()), bits(r.GetRenderTargetImage()))
        g.RenderBatchAsync(cam, sc, 2); r.RenderBatch(cam, sc, 2)
        assert g.Wait() >= 0
        assert np.array_equal(bits(g.GetRenderTargetImage()), bits(r.GetRenderTargetImage()))
    assert g.kernelInfo().startswith("path_pool")
    with pytest.raises(d.DrtError):
        d.RendererGroup([0, 0])                                          # a device twice
    with pytest.raises(d.DrtError):
        d.RendererGroup([97])


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("W,H", [(64, 44), (7, 8), (16, 129)])
def test_group_gather_of_whole_shards_rebuilds_the_image(world, W, H):
    """The gather of drt_group_render_batch (csrc/drt_group.cpp, round 3): every peer's shard goes, whole and contiguous, to its
    slot [rank][padded_rows][W] of a staging buffer on device 0, and one pass copies row y of the image from slot
    (y / 8) % world, shard row (y / 8 / world) * 8 + y % 8 (drt_assemble_shards).  CPU statement of that bookkeeping for worlds
    2, 3 and 8, short last stripe included: the count sent is the shard's size, slots do not overlap, the image is rebuilt."""
    from dustraytracer_amd.sharding import shard_row_map
    rng = np.random.default_rng(W * 977 + H + world)
    image = rng.random((H, W, 4), dtype=np.float32)
    padded = drt.shard_rows(H, STRIPE, 0, world)
    staging = np.full((world, padded, W, 4), np.nan, np.float32)
    for rank in range(world):
        rows = shard_row_map(H, STRIPE, rank, world)
        n = drt.shard_rows(H, STRIPE, rank, world)
        assert n == len(rows) <= padded
        staging[rank].reshape(-1)[:n * W * 4] = image[rows].reshape(-1)     # ONE contiguous transfer of n * W * 4 floats
    rebuilt = np.empty_like(image)
    for y in range(H):
        s = y // STRIPE
        rebuilt[y] = staging[s % world, (s // world) * STRIPE + y % STRIPE]
    assert np.array_equal(bits(rebuilt), bits(image))


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["shards", "stripes"])
def test_group_gather_through_rccl_on_one_device(monkeypatch, gather):
    """DRT_GROUP_FORCE_RCCL=1: the group of one device loads RCCL (dlopen), creates its communicator and moves its shard with
    ncclSend / ncclRecv to itself -- whole, into a second staging slot that the assemble pass then reads (the default), or stripe
    by stripe at the offsets of drt_shard_stripe (DRT_GROUP_GATHER=stripes, round 2's gather) -- the plumbing of the N-GPU
    gather, on this box.  More than one device has never run here."""
    import dustraytracer_amd as d
    monkeypatch.setenv("DRT_GROUP_FORCE_RCCL", "1")
    monkeypatch.setenv("DRT_GROUP_GATHER", gather)
    try:
        g = d.RendererGroup([0])
    except d.DrtError as e:
        pytest.skip("RCCL cannot be loaded here: %s" % e)
    monkeypatch.delenv("DRT_GROUP_FORCE_RCCL")
    monkeypatch.delenv("DRT_GROUP_GATHER")
    r = d.Renderer(0)
    sc = d.Scene(); sc.loadGLTFmodel(scene_path("room"))
    b = d.BVHBuilder(); b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8; b.buildIterative(sc)
    _, pos, fwd, _ = SCENES["room"]
    cam = d.Camera(pos); cam.m_Forward_dir = np.array(fwd, np.float32)
    for size in ((200, 61), (96, 40)):                                   # 7 full stripes + one of 5 rows; then a resize with work done before
        for x in (g, r):
            x.m_RendererSettings = d.RendererSettings(ray_bounce_limit=3, max_samples=100)
            x.ResizeBuffer(*size)
            x.RenderBatch(cam, sc, 2)
        assert np.array_equal(bits(g.GetRenderTargetImage()), bits(r.GetRenderTargetImage()))
        g.RenderBatchAsync(cam, sc, 1); r.RenderBatch(cam, sc, 1)        # left in flight: the next resize (or the destructor) drains it
    del g
