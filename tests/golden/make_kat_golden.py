#!/usr/bin/env python3
"""Generates tests/golden/kat_ref.npz from the REFERENCE's own leaf functions.

Run in the build container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_kat_golden.py

Every expected output in the file was produced by oracle/_ref/ref_kat, i.e. by
the reference's Bounds.cu / Random.cu / Intersection.cu / Camera.cu /
Texture.cu compiled from /root/reference (recipe: oracle/Makefile).  The file
holds inputs and outputs only (data, no source).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_kat as rk  # noqa: E402


def kat_inputs(seed=20240807, n=4096):
    """Deterministic input sets incl. the edge cases SURVEY.md 8(c) lists."""
    rng = np.random.default_rng(seed)
    d = {}
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    seeds[:10] = [0, 1, 2, 3, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 12345, 1920 * 1080, 0xDEADBEEF]
    d["seeds"] = seeds

    # rays: origin, direction (un-normalised, some axis-aligned => invDir = +-inf)
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    dr = rng.normal(0, 1, (n, 3)).astype(np.float32)
    dr[:64, 0] = 0.0
    dr[64:128, 1] = -0.0
    dr[128:160, 0] = 0.0
    dr[128:160, 2] = 0.0
    lo = rng.uniform(-2, 1, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 2.5, (n, 3))).astype(np.float32)
    hi[160:192, 1] = lo[160:192, 1]                      # flat boxes
    o[192:256] = ((lo[192:256] + hi[192:256]) * np.float32(0.5))   # origins inside
    aim = (lo + (hi - lo) * rng.uniform(0, 1, (n, 3))).astype(np.float32)
    dr[256:3072] = (aim - o)[256:3072] * rng.uniform(0.05, 3.0, (2816, 1)).astype(np.float32)   # aimed at the box
    d["slab_rays"] = np.concatenate([o, dr], 1)
    d["slab_boxes"] = np.concatenate([lo, hi], 1)
    # NaN slab lanes (origin exactly on a slab plane with a zero direction component) are
    # kept OUT of the reference comparison: helper_math.cuh:58-66's host fminf/fmaxf differ
    # from the device ones there.  They are covered by a hand-derived test instead.

    tri = rng.uniform(-2, 2, (n, 3, 3)).astype(np.float32)
    ro = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    bary = rng.dirichlet([1, 1, 1], n).astype(np.float32)
    target = np.einsum("nk,nkc->nc", bary, tri).astype(np.float32)
    rd = (target - ro).astype(np.float32)                # most rays hit
    rd[:256] = rng.normal(0, 1, (256, 3)).astype(np.float32)       # most of these miss
    # exact edge / vertex hits and parallel rays on an axis-aligned triangle
    tri[256:320] = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    ro[256:320] = np.array([0.25, 0.25, 1], np.float32)
    rd[256:320] = np.array([0, 0, -1], np.float32)
    ro[256:272, 0] = 0.0                                  # u == 0 edge ... etc
    ro[272:288, 1] = 0.0
    ro[288:296] = np.array([0.5, 0.5, 1], np.float32)     # u+v == 1
    ro[296:304] = np.array([0, 0, 1], np.float32)         # vertex
    rd[304:312] = np.array([1, 0, 0], np.float32)         # parallel to the plane
    ro[312:320] = np.array([0.25, 0.25, -1], np.float32)  # behind (t < 0)
    rd[320:352] *= np.float32(1e-4)                       # small determinants
    rd[352:384] *= np.float32(1e3)
    d["isect_rays"] = np.concatenate([ro, rd], 1)
    d["isect_tris"] = tri.reshape(n, 9)

    d["uv"] = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    d["cams"] = np.array([
        # exposure, vfov_rad, defocus_angle, focus_dist, pos3, fwd3, width, height
        [1, np.float32(60) * (np.float32(3.14159265359) / np.float32(180)), 0, 10, 0, 2, 5, 0, 0, -1, 1920, 1080],
        [1, np.float32(60) * (np.float32(3.14159265359) / np.float32(180)), 0, 10, 3.6, 1.25, 0, -1, 0, 0, 256, 256],
        [2, 0.7, 1.5, 4.0, 0, 1.2, 4.5, 0, -0.15, -1, 640, 360],
        [1, 1.9, 0.4, 25.0, -2, 0.5, 1, 0.3, -0.2, 0.9, 843, 460],
    ], np.float32)

    tex3 = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    tex4 = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8)
    tex4[..., 3] = np.where(rng.uniform(size=(16, 16)) < 0.5, 255, tex4[..., 3])
    tuv = rng.uniform(-3, 3, (n, 2)).astype(np.float32)
    tuv[:8] = [[0, 0], [1, 1], [-1, -1], [0.999999, 0.5], [-1e-9, 0.5], [0.5, -1e-9], [2.5, -2.5], [1e-9, 1e-9]]
    d["tex3"], d["tex4"], d["tex_uv"] = tex3, tex4, tuv

    # ClosestHit: rays as above, hit distances, unit face normals; a block with the normal exactly perpendicular to the
    # ray (dot == 0 -> front face) and one with tiny directions (normalize of a denormal-ish vector)
    fn = rng.normal(size=(n, 3)).astype(np.float32)
    fn /= np.linalg.norm(fn, axis=1, keepdims=True).astype(np.float32)
    ch_rays = np.concatenate([rng.uniform(-5, 5, (n, 3)), rng.uniform(-1, 1, (n, 3))], 1).astype(np.float32)
    ch_rays[:64, 3:] = np.array([1, 0, 0], np.float32)
    fn[:32] = np.array([0, 1, 0], np.float32)
    fn[32:64] = np.array([0, 0, -1], np.float32)
    ch_rays[64:128, 3:] *= np.float32(1e-18)
    d["ch_rays"], d["ch_t"], d["ch_fn"] = ch_rays, rng.uniform(1e-3, 50, n).astype(np.float32), fn

    # Camera::OnUpdate + Camera::Rotate, 512 editor frames in sequence (mouse deltas as sin/cos of small angles)
    m = 512
    ax, ay = rng.uniform(-0.05, 0.05, m).astype(np.float32), rng.uniform(-0.05, 0.05, m).astype(np.float32)
    steps = np.zeros((m, 8), np.float32)
    steps[:, 0], steps[:, 1], steps[:, 2], steps[:, 3] = np.sin(ax), np.cos(ax), np.sin(ay), np.cos(ay)
    steps[:, 4:7] = rng.integers(-1, 2, (m, 3)).astype(np.float32)
    steps[:, 7] = rng.uniform(0.001, 0.03, m).astype(np.float32)
    d["cam_steps"] = steps
    d["cam_start"] = np.array([0, 2, 5, 0, 0, -1, 0, 1, 0, 1, 0, 0, 10], np.float32)   # pos, fwd, up, right, speed
    return d


def main():
    if not rk.available():
        sys.exit("oracle/_ref/ref_kat missing: run `make -C oracle ref` in the build container first")
    d = kat_inputs()
    out = dict(d)
    out["pcg"] = rk.pcg(d["seeds"])
    out["randfloat"], seed_end = rk.randfloat(12345, 1024)
    out["randfloat_seed_end"] = np.uint32(seed_end)
    out["unitvec"], out["unitvec_seed"] = rk.unitvec(d["seeds"])
    out["unitsphere"], out["unitsphere_seed"] = rk.unitsphere(d["seeds"])
    out["unitdisk"], out["unitdisk_seed"] = rk.unitdisk(d["seeds"])
    out["slab"] = rk.slab(d["slab_rays"], d["slab_boxes"])
    out["isect_tuvw"], out["isect_hit"] = rk.intersect(d["isect_rays"], d["isect_tris"])
    out["ch_pos"], out["ch_normal"], out["ch_front"] = rk.closesthit(d["ch_rays"], d["ch_t"], d["ch_fn"])
    out["centroid"] = rk.centroid(d["isect_tris"])
    out["sa_count"] = (np.arange(len(d["slab_boxes"])) % 5).astype(np.int32)          # every 5th node is empty -> area 0
    out["surfacearea"] = rk.surfacearea(d["slab_boxes"], out["sa_count"])
    c = d["cam_start"]
    out["cam_track"] = rk.cammove(c[0:3], c[3:6], c[6:9], c[9:12], c[12], d["cam_steps"])
    rays, rseeds = [], []
    for cam in d["cams"]:
        r, s = rk.getray((cam[0], cam[1], cam[2], cam[3], cam[4:7], cam[7:10]), cam[10], cam[11], d["uv"], d["seeds"])
        rays.append(r)
        rseeds.append(s)
    out["getray"], out["getray_seed"] = np.stack(rays), np.stack(rseeds)
    out["texpixel3"] = rk.texpixel(d["tex3"], d["tex_uv"])
    out["texpixel4"] = rk.texpixel(d["tex4"], d["tex_uv"])
    out["texalpha4"] = rk.texalpha(d["tex4"], d["tex_uv"])
    out["texalpha3"] = rk.texalpha(d["tex3"], d["tex_uv"])
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_layout.json"), "w") as f:
        json.dump(rk.layout(), f, indent=1, sort_keys=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;",
          "hits", int(out["isect_hit"].sum()), "/", len(out["isect_hit"]),
          "slab hits", int((out["slab"] >= 0).sum()))


if __name__ == "__main__":
    main()
