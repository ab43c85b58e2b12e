#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one full progressive render of the frame: reset the accumulation buffer, trace every sample
of the workload (W x H x spp paths: ray generation, BVH traversal, shading, accumulate, resolve) and,
for N > 1, gather the rank-local framebuffer stripes on rank 0 over RCCL and assemble the image.
Inputs (scene, BVH, textures) are resident in HBM before the timed region.  One JSON line on rank 0.
Steps are independent frames; --frames-in-flight (default 4; 1 for the half-second steps of room 4K) of them are enqueued at a time, each on its own
renderer + HIP stream, so the end-of-launch drain of one frame and its gather overlap the next frame's
ramp-up.  Every step still does all of its work inside the timed region (drained before the closing sync).

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

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one
# queue do not run side by side: with four or more frames in flight (each on its own stream, next to the null stream and
# torch's) launches that should overlap serialised -- 4.1 ms per frame instead of 2.6, 0.58 ms per 1/8 shard instead of 0.36.
# Ask for eight before the runtime starts (a process-level setting of the ROCm runtime, like the stream count itself).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
STRIPE_ROWS = 8

# name: (scene key in tests/scenes.py, W, H, spp, depth)   -- BASELINE.json "configs"
WORKLOADS = {
    "cornell_box_1080p_8spp_d8": ("cornell_box", 1920, 1080, 8, 8),          # configs[1]: the metric config
    "cornell_box_256_1spp_d4": ("cornell_box", 256, 256, 1, 4),              # configs[0]
    "suzanne_plane_1080p_8spp_d2": ("suzanne_plane", 1920, 1080, 8, 2),      # configs[2]
    "dense_monkey_1080p_16spp_d2": ("dense_monkey", 1920, 1080, 16, 2),      # configs[3]
    "room_4k_64spp_d16": ("room", 3840, 2160, 64, 16),                       # configs[4]
    "cs16_dust_1080p_8spp_d5": ("cs16_dust", 1920, 1080, 8, 5),              # large closed map: scene served from L2/HBM, not LDS
    # the one configuration the reference shows a timing for (editor screenshot beforeBVHbuildrefactor_col.png: 843x460,
    # 5 bounces, sample 50, 7.232 ms per frame index = 53.6 Msamples/s on the author's GTX 1650); alpha cut-out scene
    "mc_transparency_843x460_50spp_d5": ("mc_transparency", 843, 460, 50, 5),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", default="cornell_box_1080p_8spp_d8", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent frames (steps) kept in flight, each on its own renderer + HIP stream: the drain of one "
                         "frame overlaps the ramp-up of the next (1 = strictly one after the other).  Default 4; 2 for the "
                         "workload whose step is a second long (room 4K / 64 spp: 800 ms with 2 in flight, 896 with 3)")
    ap.add_argument("--emulate-shard", default="", help="R/W: render only rank R's stripes of a W-way split on ONE GPU, no gather "
                    "(what one GPU of a W-GPU run computes; for tuning small-shard behaviour on a 1-GPU box)")
    ap.add_argument("--no-roofline-counters", action="store_true", help="skip the counting launch (roofline = null)")
    ap.add_argument("--group", action="store_true", help="ONE process drives all --gpus devices through the C ABI's drt_group_* (stripes "
                    "per device, gathered into device 0's image with RCCL send / recv inside the library) instead of one rank per GPU")
    return ap.parse_args()


def usable_cores():
    """Threads worth starting: the affinity mask, capped by the cgroup CPU quota (a 256-thread host that grants this
    container 16 cores' worth of time only gets slower with 256 runnable threads)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(scene_key, W, H, depth, target_seconds):
    """The oracle (oracle/drt_oracle.c, our CPU restatement = kind "port") timed on this host's cores on a
    bounded sample of the same workload: full-width stripes of the same frame, frame index 1.."""
    import oracle
    from tests.scenes import SCENES, scene_path
    _, pos, fwd, _ = SCENES[scene_key]
    cores = usable_cores()
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


def load_profile(workload):
    """Per-launch counter averages of the tracing kernel on this workload (SQ instruction / lane counters, HBM bytes), written
    by tools/roofline_from_profiles.py from rocprofv3 --pmc passes of this same bench (tools/collect_profiles.sh); else None."""
    for rnd in ("r03", "r02"):
        path = os.path.join(ROOT, "profiles", "%s_roofline_%s.json" % (rnd, workload))
        try:
            with open(path) as f:
                d = json.load(f)
            d["_file"] = os.path.relpath(path, ROOT)
            return d
        except (OSError, ValueError):
            pass
    return None


VALU_PEAK_GSLOTS = 1024 * 2.4e9 / 2 / 1e9     # 256 CUs x 4 SIMDs, 2.4 GHz, one wave64 add / mul / min / max issues every 2 cycles


def algorithmic_slots(counters):
    """VALU issue slots the REFERENCE's arithmetic needs for this launch's exact work (counted by the measured kernel itself):
    sum over the reference's operations of (work counter x issue slots of one call of that operation in the shipped arithmetic,
    profiles/r03_isa_slots.json, made by tools/isa_by_phase.py --slots-json from tools/probes/isa_probes.hip), per lane; / 64 =
    wave instructions at full lane utilisation.  Queues, state traffic, address arithmetic and everything else the kernel adds
    are NOT in it: algorithmic_frac = this / kernel time / peak is the fraction of a VALU-bound ideal."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_isa_slots.json")) as f:
            per_call = {k: v["slots"] for k, v in json.load(f)["per_call"].items()}
    except (OSError, ValueError, KeyError):
        return None
    c = counters
    terms = {
        "triangle tests (Intersection.cu:4-36)": (c["tri_tests"] + c["tri_tests_shadow"]) * per_call["tri_test"],
        "child-box slab tests (Bounds.cu:18-41), two per interior visit": 2 * (c["inner_visits"] + c["inner_visits_shadow"]) * per_call["box_test"],
        "bounce-direction candidates (Random.cu:50-58)": c.get("sampler_tries", 0) * per_call["sampler_try"],
        "closest-hit frame + throughput (ClosestHit.cuh:13-24, RayGen.cuh:112,121)": (c["hits_textured"] + c["hits_flat"]) * per_call["shade_hit"],
        "texel fetch + decode (Texture.cu:33-58)": c["hits_textured"] * per_call["texture_fetch"],
        "ray set-up: 1/dir + the root's slab test (Ray.cuh:7-9, BVHTraversal.cuh:22-26)": c["rays"] * per_call["ray_setup"],
        "shadow-ray set-up (RayGen.cuh:124-125)": c["shadow_rays"] * per_call["shadow_ray_setup"],
        "per sample: camera ray, sky, tone map, gamma (Camera.cu:82-123, RayGen.cuh:23-61)": c["samples"] * per_call["sample"],
    }
    lane_slots = float(sum(terms.values()))
    return {"wave_issue_slots_per_launch": lane_slots / 64.0, "lane_slots_per_sample": round(lane_slots / max(c["samples"], 1), 1),
            "share": {k: round(v / lane_slots, 4) for k, v in terms.items() if v}, "slots_per_call": per_call,
            "slots_from": "profiles/r03_isa_slots.json"}


def main_group(args):
    """--group: the in-process multi-GPU path (include/drt.h drt_group_*).  Same step, same JSON line; the gather is part of
    every step (drt_group_wait returns when device 0 holds the whole frame)."""
    import numpy as np
    import dustraytracer_amd as drt
    from tests.scenes import SCENES, scene_path
    scene_key, W, H, spp, depth = WORKLOADS[args.workload]
    _, pos, fwd, _ = SCENES[scene_key]
    scene = drt.Scene()
    scene.loadGLTFmodel(scene_path(scene_key))
    builder = drt.BVHBuilder()
    builder.m_TargetLeafPrimitivesCount, builder.m_BinCount = 20, 8
    builder.buildIterative(scene)
    cam = drt.Camera(pos)
    cam.m_Forward_dir = np.array(fwd, np.float32)
    groups = [drt.RendererGroup(list(range(args.gpus))) for _ in range(max(1, args.frames_in_flight))]
    for g in groups:
        g.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
        g.ResizeBuffer(W, H)
    busy = [False] * len(groups)

    def step(i):
        k = i % len(groups)
        if busy[k]:
            groups[k].Wait()
        groups[k].resetAccumulationBuffer()
        groups[k].RenderBatchAsync(cam, scene, spp)
        busy[k] = True

    def drain():
        for k, g in enumerate(groups):
            if busy[k]:
                g.Wait()
                busy[k] = False

    for i in range(args.warmup):
        step(i)
    drain()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    elapsed = time.perf_counter() - t0
    out = {"metric": "Msamples/sec at 1920x1080, 8spp, cornell_box" if args.workload == "cornell_box_1080p_8spp_d8" else "Msamples/sec",
           "value": round(W * H * spp * args.steps / elapsed / 1e6, 3), "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": args.workload, "scene": SCENES[scene_key][0], "width": W, "height": H, "spp": spp, "depth": depth,
                      "bvh": "leaf20/bins8", "pose": {"pos": list(pos), "fwd": list(fwd)},
                      "parallelism": "group: one process, stripes%dx%d, RCCL send/recv into device 0" % (STRIPE_ROWS, args.gpus),
                      "frames_in_flight": len(groups), "kernel": groups[0].kernelInfo()},
           "roofline": None}
    print(json.dumps(out), flush=True)


def main():
    args = parse_args()
    if args.frames_in_flight <= 0:
        # 4 (on the 8 hardware queues asked for above); 1 for the half-second steps of room 4K.  Launches of a millisecond or less are
        # mostly ramp-up and drain, and each one more in flight hides more of that: 256^2 x 1 spp 0.083 -> 0.065 ms, the 1/8 shard
        # 0.40 -> 0.37, suzanne 1.46 -> 1.40; cornell 1080p x 8 spp is the same with 3 and 4 (2.59 ms), 6 is worse everywhere.
        # (room 4K, half a second per launch on full grids: one at a time -- 1 015 Msamples/s with 1 or 2 in flight over four steps, 785 with 3)
        args.frames_in_flight = 1 if args.workload == "room_4k_64spp_d16" else 4
    if args.group:
        return main_group(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # started bare: be our own launcher (a child process, nothing here has touched the GPU yet), one rank per GPU
            import socket
            import subprocess
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.run(cmd).returncode)
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
    # DRT_BENCH_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): take the multi-GPU code path -- process group on
    # nccl (= RCCL), dist.gather, assemble kernel, barrier, all_reduce -- with a world of one rank, on a one-GPU box
    dist_on = world > 1 or (os.environ.get("DRT_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
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

    shard_rank, shard_world = rank, world
    if args.emulate_shard and world == 1:
        shard_rank, shard_world = (int(v) for v in args.emulate_shard.split("/"))
    padded = max(drt.shard_rows(H, STRIPE_ROWS, k, shard_world) for k in range(shard_world))

    class Slot:
        """One frame in flight: a renderer with its own HIP stream and buffers."""

        def __init__(self):
            self.r = drt.Renderer(local_rank)
            self.r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=depth, max_samples=spp + 1)
            self.r.setShard(STRIPE_ROWS, shard_rank, shard_world)
            self.r.setFramesInFlight(max(1, args.frames_in_flight))
            self.r.ResizeBuffer(W, H)
            self.accum = torch.zeros((padded, W, 3), dtype=torch.float32, device=dev)
            self.rgba = torch.zeros((padded, W, 4), dtype=torch.float32, device=dev)
            self.r.bindBuffers(self.accum.data_ptr(), self.rgba.data_ptr())
            self.stream = torch.cuda.Stream(device=dev)
            self.r.setStream(self.stream.cuda_stream)
            self.gathered = self.image = self.host = None
            if dist_on and rank == 0:
                self.gathered = torch.empty((world, padded, W, 4), dtype=torch.float32, device=dev)
                self.image = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
            self.busy = False

    slots = [Slot() for _ in range(max(1, args.frames_in_flight))]
    r = slots[0].r
    kernel_ms = []       # per step: stream-event time of its launches (includes time queued behind other frames in flight)
    span_ms = []         # per step: the tracing kernel's own execution span (device clock) = what rocprofv3 calls its duration

    def retire(slot):
        if slot.busy:
            kernel_ms.append(slot.r.Wait())          # device time of that step's launches (HIP events on its stream)
            span_ms.append(slot.r.kernelSpanMs())
            slot.busy = False

    def step(i):
        slot = slots[i % len(slots)]
        retire(slot)                                  # the step that used this slot before must be done
        with torch.cuda.stream(slot.stream):
            slot.r.resetAccumulationBuffer()
            slot.r.RenderBatchAsync(cam, scene, spp)
            slot.busy = True
            if dist_on:
                if rehearsal:
                    host = torch.empty((world,) + tuple(slot.rgba.shape), dtype=torch.float32) if rank == 0 else None
                    gather_shards(slot.rgba.cpu(), host, rank)
                    if rank == 0:
                        slot.gathered.copy_(host)
                else:
                    gather_shards(slot.rgba, slot.gathered, rank)
                if rank == 0:
                    drt.assemble_shards(slot.gathered.data_ptr(), slot.image.data_ptr(), W, H, STRIPE_ROWS, world, padded,
                                        slot.stream.cuda_stream)

    def drain():
        for slot in slots:
            retire(slot)

    def sync():
        torch.cuda.synchronize()
        if dist_on:
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

    for i in range(args.warmup):
        step(i)
    drain()
    kernel_ms.clear()
    span_ms.clear()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    sync()
    elapsed = time.perf_counter() - t0
    image = slots[(args.steps - 1) % len(slots)].image
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The tracing kernel ALONE: with several frames in flight the launches share the GPU and neither a stream-event interval
    # nor an execution span of one launch is a kernel time.  A few more steps, one launch at a time, blocking, untimed.
    iso_span, iso_events, iso_launches = [], [], 0
    if True:                                   # (every rank: the ranks stay in step; their launches are their shards)
        iso = slots[0].r
        iso.setFramesInFlight(1)
        for _ in range(max(1, min(args.steps, 10 if spp * W * H < 10 ** 8 else 2))):
            torch.cuda.synchronize()
            with torch.cuda.stream(slots[0].stream):
                iso.resetAccumulationBuffer()
                iso_events.append(iso.RenderBatch(cam, scene, spp))
            iso_span.append(iso.kernelSpanMs())
        iso_launches = iso.launchesOfLastBatch()
        iso.setFramesInFlight(max(1, args.frames_in_flight))

    if rank == 0:
        total_samples = W * H * spp if shard_world == world else W * drt.shard_rows(H, STRIPE_ROWS, shard_rank, shard_world) * spp
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples * args.steps / elapsed / 1e6
        roofline = None
        if iso_span and iso_launches > 0:
            kernel_ms = sum(iso_span) / len(iso_span) / iso_launches          # per launch, device clock, first wave in .. last wave out
            events_ms = sum(iso_events) / len(iso_events) / iso_launches      # HIP events on the launch stream (tracing + resolve kernel)
            if kernel_ms <= 0:                                                # pixel_walk (DRT_KERNEL=pixel_walk) has no device-side span
                kernel_ms = events_ms
            prof = load_profile(args.workload)
            same_kernel = bool(prof) and r.kernelInfo().split("<")[0] in prof.get("kernel", "")
            share = total_samples / float(W * H * spp)        # a shard's launch does that part of the profiled full-frame launch's work
            roofline = {"bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_GSLOTS, 1), "unit": "G VALU issue slots/s (1 slot = 2 SIMD cycles)",
                        "frac": None, "traffic": None, "kernel": r.kernelInfo(), "kernel_ms": round(kernel_ms, 4),
                        "kernel_ms_stream_events": round(events_ms, 4), "launches_per_step": iso_launches,
                        "kernel_ms_in_flight": round(sum(span_ms) / max(len(span_ms), 1) / iso_launches, 4),
                        "note": ("branchy scalar fp32 / u32 code, scene and path state in LDS: vector-ALU issue bounds the kernel, HBM does not. "
                                 "frac = frac_alone: achieved = (SQ_INSTS_VALU + SQ_INSTS_VALU_FMA_F32 + 3 SQ_INSTS_VALU_TRANS_F32) per launch [rocprofv3 --pmc, profiles/] / kernel_ms "
                                 "[this run: the tracing kernel alone, device clock; = rocprofv3 --kernel-trace duration]; issue costs 2 / 4 / 8 cycles measured "
                                 "(tools/microbench/valu_issue.hip); compares and conversions priced at 2, so frac is a lower bound of the pipe's occupancy. "
                                 "kernel_ms_in_flight: the same span while %d frames share the GPU (not a kernel time). frac_timed: the same slots x launches per step / ms_per_step (the timed region). "
                                 "algorithmic_frac: the issue slots of the reference's arithmetic alone (exact work counters x per-operation slots of the shipped ISA) at full lane "
                                 "utilisation, over the same times" % len(slots))}
            if same_kernel and prof.get("valu_issue_slots_per_launch"):
                ach = prof["valu_issue_slots_per_launch"] * share / (kernel_ms * 1e-3) / 1e9
                # the same slots over the TIMED region: all launches of a step / ms_per_step (several frames in flight share the chip)
                ach_timed = prof["valu_issue_slots_per_launch"] * share * iso_launches / (ms_per_step * 1e-3) / 1e9
                roofline.update({"achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_GSLOTS, 4), "frac_alone": round(ach / VALU_PEAK_GSLOTS, 4),
                                 "frac_timed": round(ach_timed / VALU_PEAK_GSLOTS, 4),
                                 "lane_utilisation": round(prof.get("lane_utilisation", 0.0), 4),
                                 "valu_wave_instructions_per_launch": int(prof["per_launch"].get("SQ_INSTS_VALU", 0) * share),
                                 "counters_from": prof["_file"] + ("" if share == 1.0 else " (full-frame launch, scaled by this rank's share of the samples)")})
            if same_kernel and prof.get("hbm_bytes_per_launch"):
                roofline["traffic"] = int(prof["hbm_bytes_per_launch"] * share)
            if same_kernel:
                for key in ("wave_time", "l2"):
                    if prof.get(key):
                        roofline[key] = prof[key]
            if counters is not None:
                alg = algorithmic_slots(counters)
                if alg:
                    alg_launch = alg["wave_issue_slots_per_launch"] / iso_launches
                    alg["wave_issue_slots_per_launch"] = int(alg_launch)
                    roofline["algorithmic_frac"] = round(alg_launch / (kernel_ms * 1e-3) / 1e9 / VALU_PEAK_GSLOTS, 4)
                    roofline["algorithmic_frac_timed"] = round(alg_launch * iso_launches / (ms_per_step * 1e-3) / 1e9 / VALU_PEAK_GSLOTS, 4)
                    if roofline.get("valu_wave_instructions_per_launch"):
                        alg["of_issued_slots"] = round(alg_launch / (prof["valu_issue_slots_per_launch"] * share), 4)
                    roofline["algorithmic"] = alg
            if alg_bytes is not None:
                alg_launch = alg_bytes / iso_launches          # (counted on this rank's own launches)
                hbm = {"algorithmic_bytes_per_launch": int(alg_launch), "bytes_per_sample": round(alg_bytes / max(counters["samples"], 1), 1),
                       "achieved": round(alg_launch / (kernel_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(alg_launch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "served_from": "lds" if "lds-scene" in r.kernelInfo() else "l2"}
                if same_kernel and prof.get("hbm_bytes_per_launch_trace"):
                    hbm["hbm_measured_frac"] = round(prof["hbm_bytes_per_launch_trace"] * share / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                hbm["note"] = ("SURVEY 8(d)'s algorithmic bytes over the kernel time; a fraction above 1 means those bytes are not served by HBM "
                               "(served_from); hbm_measured_frac = PMC FETCH_SIZE x 2 + WRITE_SIZE of the tracing kernel over the same time")
                roofline["hbm"] = hbm
        out = {"metric": "Msamples/sec at 1920x1080, 8spp, cornell_box" if args.workload == "cornell_box_1080p_8spp_d8" else "Msamples/sec",
               "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on device 0, gloo gather)",
               "config": {"workload": args.workload, "scene": SCENES[scene_key][0], "width": W, "height": H, "spp": spp,
                          "depth": depth, "bvh": "leaf20/bins8", "pose": {"pos": list(pos), "fwd": list(fwd)},
                          "parallelism": "stripes%dx%d" % (STRIPE_ROWS, world) if world > 1 else "single",
                          "frames_in_flight": len(slots)},
               "roofline": roofline}
        if args.workload == "mc_transparency_843x460_50spp_d5":
            out["config"]["reference_screenshot"] = {"value": 53.6, "unit": "Msamples/s", "hardware": "GTX 1650 (presumed)",
                                                     "source": "DustRayTracer/beforeBVHbuildrefactor_col.png", "ratio": round(value / 53.6, 1)}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(scene_key, W, H, depth, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist_on:
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
