"""-m gpu: the DEVICE leaf functions (RNG, slab, triangle test, camera ray) against tests/golden/kat_ref.npz,
i.e. against outputs of the reference's own compiled sources (generator: tests/golden/make_kat_golden.py).
Bit-exact.  Also the RNG cycle guard, which only exists because pcg_hash has short cycles."""
import numpy as np
import pytest

import oracle
from tests.scenes import bits

pytestmark = pytest.mark.gpu
drt = pytest.importorskip("dustraytracer_amd")


def test_device_unit_vectors_match_reference(kat_golden):
    g = kat_golden
    out = drt.debug_kat(0, g["seeds"])
    assert np.array_equal(out[:, :3], bits(g["unitvec"]))
    assert np.array_equal(out[:, 3], g["unitvec_seed"])
    out = drt.debug_kat(1, g["seeds"])
    assert np.array_equal(out[:, :3], bits(g["unitsphere"]))          # incl. the sqrt-free acceptance test
    assert np.array_equal(out[:, 3], g["unitsphere_seed"])
    out = drt.debug_kat(5, g["seeds"])
    assert np.array_equal(out[:, :2], bits(g["unitdisk"]))
    assert np.array_equal(out[:, 2], g["unitdisk_seed"])


def test_device_sphere_sampler_wide(kat_golden):
    """200k more seeds, against the oracle (itself pinned to the reference on the golden set)."""
    rng = np.random.default_rng(11)
    seeds = rng.integers(0, 2 ** 32, 200_000, dtype=np.uint64).astype(np.uint32)
    v, s, it = oracle.kat_unitsphere(seeds)
    out = drt.debug_kat(1, seeds)
    assert np.array_equal(out[:, :3], bits(v)) and np.array_equal(out[:, 3], s)
    assert np.array_equal(out[:, 4].astype(np.int64), it.astype(np.int64))


def test_device_slab_matches_reference(kat_golden):
    g = kat_golden
    out = drt.debug_kat(2, np.concatenate([g["slab_rays"], g["slab_boxes"]], axis=1).astype(np.float32))
    assert np.array_equal(out[:, 0], bits(g["slab"]))


def test_device_triangle_test_matches_reference(kat_golden):
    g = kat_golden
    out = drt.debug_kat(3, np.concatenate([g["isect_rays"], g["isect_tris"]], axis=1).astype(np.float32))
    hit = out[:, 4].astype(bool)
    assert np.array_equal(hit, g["isect_hit"].astype(bool))
    assert np.array_equal(out[hit, :4], bits(g["isect_tuvw"])[hit])      # t, U, V, W of the straight-line test + exact_rcp


def test_device_closest_hit_frame_matches_reference(kat_golden):
    g = kat_golden
    inp = np.concatenate([g["ch_rays"], g["ch_t"][:, None], g["ch_fn"]], axis=1).astype(np.float32)
    out = drt.debug_kat(6, inp)
    assert np.array_equal(out[:, :3], bits(g["ch_pos"]))
    assert np.array_equal(out[:, 3:6], bits(g["ch_normal"]))
    assert np.array_equal(out[:, 6].astype(np.int32), g["ch_front"])


def test_device_camera_ray_matches_reference(kat_golden):
    g = kat_golden
    for k, c in enumerate(g["cams"]):
        cam = drt.Camera([float(v) for v in c[4:7]])
        cam.exposure, cam.vfov_rad, cam.defocus_angle, cam.focus_dist = (float(c[0]), float(c[1]), float(c[2]), float(c[3]))
        cam.m_Forward_dir = np.array(c[7:10], np.float32)
        inp = np.zeros((len(g["seeds"]), 3), np.uint32)
        inp[:, :2] = bits(g["uv"])
        inp[:, 2] = g["seeds"]
        out = drt.debug_kat(4, inp, cam=cam, width=int(c[10]), height=int(c[11]))
        assert np.array_equal(out[:, 6], g["getray_seed"][k]), "camera %d seed stream" % k
        assert np.array_equal(out[:, :6], bits(g["getray"][k])), "camera %d" % k


def test_rng_hash_has_short_cycles_and_the_guard_terminates():
    """pcg_hash (Random.cu:6-11) is a permutation with short cycles; on some, every candidate of the rejection
    sampler is rejected and the reference's loop (Random.cu:50-58) never ends.  Kernel and oracle stop after 1024
    candidates and agree on the result."""
    cycles = drt.debug_hash_cycles(max_len=64)
    lengths = sorted({n for _, n in cycles})
    assert lengths[:4] == [4, 8, 10, 13]
    by_len = {}
    for v, n in cycles:
        by_len.setdefault(n, []).append(v)
    assert all(len(v) == n for n, v in by_len.items())                # a cycle of length n has n members
    seeds = np.array([v for v, _ in cycles], np.uint32)
    # check the cycle property with the oracle's hash
    for v, n in cycles[:8]:
        x = v
        for _ in range(n):
            x = int(oracle.kat_pcg([x])[0])
        assert x == v
    vec, s_out, tries = oracle.kat_unitsphere(seeds)
    out = drt.debug_kat(1, seeds)
    assert np.array_equal(out[:, :3], bits(vec)) and np.array_equal(out[:, 3], s_out)
    assert np.array_equal(out[:, 4].astype(np.int64), tries.astype(np.int64))
    assert tries.max() == 1024, "expected at least one start on a short cycle that never accepts (the reference would hang)"
    stuck = tries == 1024
    assert np.isfinite(vec[stuck]).all() and np.allclose(np.linalg.norm(vec[stuck].astype(np.float64), axis=1), 1.0, atol=1e-6)
