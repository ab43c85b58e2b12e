#!/usr/bin/env python3
"""Generates tests/golden/kat_ref2.npz: the remaining reference functions that compile without thrust -- Interval::surrounds
(Interval.cuh:24-26), Miss (Kernel/Shaders/Miss.cuh:2-6), Bounds3f::getCentroid (Bounds.cu:12-15) -- run through
oracle/_ref/ref_kat (the reference's own sources compiled where they lie, oracle/Makefile target `ref`).
Build container only:   make -C oracle ref && python tests/golden/make_kat_golden2.py
The file holds inputs and outputs only (data, no source)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_kat as rk  # noqa: E402


def main():
    if not rk.available():
        sys.exit("oracle/_ref/ref_kat missing: run `make -C oracle ref` in the build container first")
    rng = np.random.default_rng(20241004)
    n = 2048
    FLT_MAX = np.finfo(np.float32).max
    iv = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    iv[:, 1] = iv[:, 0] + np.abs(iv[:, 1])
    iv[:64, 0], iv[:64, 1] = -1.0, FLT_MAX                       # the traversal's interval (RayGen.cuh:78)
    iv[:64, 2] = np.float32([-1.0, FLT_MAX, 0.0, -0.0, np.inf, -np.inf, np.nan, -1.0000001] * 8)
    iv[64:128, 2] = iv[64:128, 0]                                 # x == min: not surrounded
    iv[128:192, 2] = iv[128:192, 1]                               # x == max
    iv[192:200, 0], iv[192:200, 1] = FLT_MAX, -FLT_MAX            # Interval::empty, the default of a shadow ray
    rays = rng.normal(size=(n, 6)).astype(np.float32)
    col = rng.uniform(0, 3, (n, 3)).astype(np.float32)
    lo = rng.uniform(-1e3, 1e3, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 1e3, (n, 3))).astype(np.float32)
    lo[:16], hi[:16] = FLT_MAX, -FLT_MAX                          # a fresh BVHNode's box
    lo[16:32] *= np.float32(1e35); hi[16:32] *= np.float32(1e35)  # halves do not overflow where the sum would
    out = {"iv": iv, "surrounds": rk.surrounds(iv), "miss_rays": rays, "miss_color": col, "boxes": np.concatenate([lo, hi], 1)}
    out["miss_out_color"], out["miss_t"], out["miss_has_prim"], out["miss_front"] = rk.miss(rays, col)
    out["boundscentroid"] = rk.boundscentroid(out["boxes"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_ref2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; surrounded", int(out["surrounds"].sum()), "/", n)


if __name__ == "__main__":
    main()
