#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one full progressive render of the frame: reset the accumulation buffer, trace every sample
of the workload (W x H x spp paths: ray generation, BVH traversal, shading, accumulate, resolve) and,
for N > 1, gather the rank-local framebuffer stripes on rank 0 over RCCL and assemble the image.
Inputs (scene, BVH, textures) are resident in HBM before the timed region.  One JSON line on rank 0.

Sharding (N > 1): the frame is cut into 8-row stripes dealt round-robin over the ranks
(drt_renderer_set_shard); RNG seeds use the global pixel index so the image is bit-identical to the
1-GPU one.  Total work is fixed as N grows ("scaling": "strong").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
STRIPE_ROWS = 8

# name: (scene key in tests/scenes.py, W, H, spp, depth)   -- BASELINE.json "configs"
WORKLOADS = {
    "cornell_box_1080p_8spp_d8": ("cornell_box", 1920, 1080, 8, 8),          # configs[1]: the metric config
    "cornell_box_256_1spp_d4": ("cornell_box", 256, 256, 1, 4),              # configs[0]
    "suzanne_plane_1080p_8spp_d2": ("suzanne_plane", 1920, 1080, 8, 2),      # configs[2]
    "dense_monkey_1080p_16spp_d2": ("dense_monkey", 1920, 1080, 16, 2),      # configs[3]
    "room_4k_64spp_d16": ("room", 3840, 2160, 64, 16),                       # configs[4]
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cornell_box_1080p_8spp_d8", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-roofline-counters", action="store_true", help="skip the counting launch (roofline = null)")
    return ap.parse_args()


def cpu_baseline(scene_key, W, H, depth, target_seconds):
    """The oracle (oracle/drt_oracle.c, our CPU restatement = kind "port") timed on this host's cores on a
    bounded sample of the same workload: full-width stripes of the same frame, frame index 1.."""
    import oracle
    from tests.scenes import SCENES, scene_path
    _, pos, fwd, _ = SCENES[scene_key]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    osc = oracle.Scene.load_glb(scene_path(scene_key)).build_bvh(20, 8)
    cam = oracle.default_camera(position=pos, forward=fwd)
    st = oracle.default_settings(ray_bounce_limit=depth)
    # whole frames, one frame index per call, until about target_seconds of CPU work are done (per-sample cost does not
    # depend on the frame index, so this is the same workload as the GPU's, just fewer samples)
    samples, frames, t_total = 0, 0, 0.0
    while t_total < target_seconds and frames < 256:
        t0 = time.perf_counter()
        oracle.render(osc, cam, st, W, H, frames + 1, 1, threads=cores)
        t_total += time.perf_counter() - t0
        frames += 1
        samples += W * H
    return {"value": round(samples / t_total / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%s %dx%d depth %d: whole frames, frame indices 1..%d = %d samples in %.1f s, %d threads (oracle/drt_oracle.c, gcc -O2)"
                      % (scene_key, W, H, depth, frames, samples, t_total, cores)}


def load_traffic(workload):
    """HBM bytes per launch from PMC counters, if a profile for this workload was collected
    (tools/pmc_traffic.py writes profiles/traffic_<workload>.json); else None."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    try:
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    import dustraytracer_amd as drt
    from dustraytracer_amd.sharding import gather_shards
    from tests.scenes import SCENES, scene_path

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    # Rehearsal switch for a 1-GPU box (RCCL refuses two ranks on one device): every rank renders on device 0 and the
    # gather goes through gloo on host copies.  Never set by the driver; the JSON line says so when it is used.
    rehearsal = os.environ.get("DRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    scene_key, W, H, spp, depth = WORKLOADS[args.workload]
    _, pos, fwd, _ = SCENES[scene_key]
    scene = drt.Scene()
    scene.loadGLTFmodel(scene_path(scene_key))
    builder = drt.BVHBuilder()
    builder.m_TargetLeafPrimitivesCount, builder.m_BinCount = 20, 8         # EditorLayer.cpp:53-54
    builder.buildIterative(scene)
    cam = drt.Camera(pos)
    cam.m_Forward_dir = np.array(fwd, np.float32)

    r = drt.Renderer(local_rank)
    r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
    r.setShard(STRIPE_ROWS, rank, world)
    r.ResizeBuffer(W, H)
    local_rows = r.getLocalRows()
    padded = max(drt.shard_rows(H, STRIPE_ROWS, k, world) for k in range(world))
    accum = torch.zeros((padded, W, 3), dtype=torch.float32, device=dev)
    rgba = torch.zeros((padded, W, 4), dtype=torch.float32, device=dev)
    r.bindBuffers(accum.data_ptr(), rgba.data_ptr())
    stream = torch.cuda.current_stream()
    r.setStream(stream.cuda_stream)
    gathered = image = None
    if world > 1 and rank == 0:
        gathered = torch.empty((world, padded, W, 4), dtype=torch.float32, device=dev)
        image = torch.empty((H, W, 4), dtype=torch.float32, device=dev)

    kernel_ms = []

    def step():
        r.resetAccumulationBuffer()
        kernel_ms.append(r.RenderBatch(cam, scene, spp))          # blocking; ms from HIP events on the launch stream
        if world > 1:
            if rehearsal:
                host = torch.empty((world,) + tuple(rgba.shape), dtype=torch.float32) if rank == 0 else None
                gather_shards(rgba.cpu(), host, rank)
                if rank == 0:
                    gathered.copy_(host)
            else:
                gather_shards(rgba, gathered, rank)
            if rank == 0:
                drt.assemble_shards(gathered.data_ptr(), image.data_ptr(), W, H, STRIPE_ROWS, world, padded,
                                    torch.cuda.current_stream().cuda_stream)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # exact algorithmic work of one launch on this rank (counting variant of the same kernel, untimed)
    alg_bytes = None
    counters = None
    if not args.no_roofline_counters:
        r.setCounting(True)
        r.resetAccumulationBuffer()
        r.RenderBatch(cam, scene, spp)
        c = r.getCounters()
        counters = c.as_dict()
        alg_bytes = c.algorithmic_bytes()
        r.setCounting(False)

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_samples = W * H * spp
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples * args.steps / elapsed / 1e6
        avg_kernel_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        roofline = None
        if alg_bytes is not None and avg_kernel_ms > 0:
            achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_traffic(args.workload) if world == 1 else None,
                        "kernel": r.kernelInfo(), "kernel_ms": round(avg_kernel_ms, 4),
                        "algorithmic_bytes_per_launch": int(alg_bytes),
                        "bytes_per_sample": round(alg_bytes / max(counters["samples"], 1), 1)}
        out = {"metric": "Msamples/sec at 1920x1080, 8spp, cornell_box" if args.workload == "cornell_box_1080p_8spp_d8" else "Msamples/sec",
               "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on device 0, gloo gather)",
               "config": {"workload": args.workload, "scene": SCENES[scene_key][0], "width": W, "height": H, "spp": spp,
                          "depth": depth, "bvh": "leaf20/bins8", "pose": {"pos": list(pos), "fwd": list(fwd)},
                          "parallelism": "stripes%dx%d" % (STRIPE_ROWS, world) if world > 1 else "single"},
               "roofline": roofline}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(scene_key, W, H, depth, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        if rank == 0 and os.environ.get("DRT_BENCH_CHECK") == "1":
            # self-check of the sharded path: the assembled image must equal an unsharded render bit for bit
            torch.cuda.synchronize()
            full = drt.Renderer(local_rank)
            full.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
            full.ResizeBuffer(W, H)
            full.RenderBatch(cam, scene, spp)
            same = np.array_equal(full.GetRenderTargetImage().view(np.uint32), image.cpu().numpy().view(np.uint32))
            print("sharded image == single-device image:", same, file=sys.stderr, flush=True)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
