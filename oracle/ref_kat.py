"""Python side of oracle/_ref/ref_kat (the reference's own leaf functions).

TEST INFRASTRUCTURE ONLY.  Used by tests/ and by tests/golden/make_kat_golden.py.
"""
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_KAT = os.path.join(_HERE, "_ref", "ref_kat")


def available():
    return os.path.isfile(REF_KAT) and os.access(REF_KAT, os.X_OK)


def _run(fn, payload):
    with tempfile.TemporaryDirectory() as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fi, "wb") as f:
            f.write(payload)
        subprocess.run([REF_KAT, fn, fi, fo], check=True)
        with open(fo, "rb") as f:
            return f.read()


def pcg(values):
    return np.frombuffer(_run("pcg", np.asarray(values, "<u4").tobytes()), "<u4").copy()


def randfloat(seed, n):
    raw = _run("randfloat", np.array([seed, n], "<u4").tobytes())
    return np.frombuffer(raw[: 4 * n], "<f4").copy(), int(np.frombuffer(raw[4 * n:], "<u4")[0])


def _vec_seed(fn, seeds, width):
    rec = np.dtype([("v", "<f4", width), ("seed", "<u4")])
    out = np.frombuffer(_run(fn, np.asarray(seeds, "<u4").tobytes()), rec)
    return out["v"].copy(), out["seed"].copy()


def unitvec(seeds):
    return _vec_seed("unitvec", seeds, 3)


def unitsphere(seeds):
    return _vec_seed("unitsphere", seeds, 3)


def unitdisk(seeds):
    return _vec_seed("unitdisk", seeds, 2)


def slab(rays6, boxes6):
    data = np.concatenate([np.asarray(rays6, "<f4").reshape(-1, 6), np.asarray(boxes6, "<f4").reshape(-1, 6)], axis=1)
    return np.frombuffer(_run("slab", np.ascontiguousarray(data).tobytes()), "<f4").copy()


def intersect(rays6, tris9):
    data = np.concatenate([np.asarray(rays6, "<f4").reshape(-1, 6), np.asarray(tris9, "<f4").reshape(-1, 9)], axis=1)
    rec = np.dtype([("tuvw", "<f4", 4), ("hit", "<i4")])
    out = np.frombuffer(_run("intersect", np.ascontiguousarray(data).tobytes()), rec)
    return out["tuvw"].copy(), out["hit"].copy()


def closesthit(rays6, t, face_n3):
    data = np.concatenate([np.asarray(rays6, "<f4").reshape(-1, 6), np.asarray(t, "<f4").reshape(-1, 1),
                           np.asarray(face_n3, "<f4").reshape(-1, 3)], axis=1)
    rec = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("front", "<i4")])
    out = np.frombuffer(_run("closesthit", np.ascontiguousarray(data).tobytes()), rec)
    return out["pos"].copy(), out["normal"].copy(), out["front"].copy()


def cammove(pos, fwd, up, right, speed, steps8):
    """steps8 = n x (sin_x, cos_x, sin_y, cos_y, vel3, delta); returns n x (pos3, fwd3, right3) after each OnUpdate + Rotate."""
    head = np.array([*pos, *fwd, *up, *right, speed], "<f4")
    out = np.frombuffer(_run("cammove", head.tobytes() + np.asarray(steps8, "<f4").reshape(-1, 8).tobytes()), "<f4")
    return out.reshape(-1, 9).copy()


def layout():
    """{key: int} -- struct sizes / offsets and default member values (floats as bit patterns) of the reference's headers."""
    return {k: int(v) for k, v in (line.split("=") for line in _run("layout", b"").decode().splitlines() if line)}


def centroid(tris9):
    return np.frombuffer(_run("centroid", np.asarray(tris9, "<f4").reshape(-1, 9).tobytes()), "<f4").reshape(-1, 3).copy()


def surfacearea(boxes6, counts):
    rec = np.dtype([("box", "<f4", 6), ("count", "<i4")])
    data = np.zeros(len(counts), rec)
    data["box"] = np.asarray(boxes6, "<f4").reshape(-1, 6)
    data["count"] = np.asarray(counts, "<i4")
    return np.frombuffer(_run("surfacearea", data.tobytes()), "<f4").copy()


def stbload(file_bytes):
    """The reference's image decoder (vendored stb_image, as Texture.cu:23 calls it): HxWxC uint8, or None."""
    raw = _run("stbload", bytes(file_bytes))
    w, h, n = np.frombuffer(raw[:12], "<i4")
    if w == 0:
        return None
    return np.frombuffer(raw[12:], np.uint8).reshape(h, w, n).copy()


def getray(cam, width, height, uv2, seeds):
    """cam = (exposure, vfov_rad, defocus_angle, focus_dist, pos3, fwd3)"""
    head = np.array([cam[0], cam[1], cam[2], cam[3], *cam[4], *cam[5], width, height], "<f4")
    rec = np.dtype([("uv", "<f4", 2), ("seed", "<u4")])
    body = np.zeros(len(seeds), rec)
    body["uv"] = np.asarray(uv2, "<f4").reshape(-1, 2)
    body["seed"] = np.asarray(seeds, "<u4")
    orec = np.dtype([("ray", "<f4", 6), ("seed", "<u4")])
    out = np.frombuffer(_run("getray", head.tobytes() + body.tobytes()), orec)
    return out["ray"].copy(), out["seed"].copy()


def _tex(fn, texels, uv2):
    h, w, c = texels.shape
    uv = np.asarray(uv2, "<f4").reshape(-1, 2)
    padded = np.zeros((h * w + w + 1) * c, np.uint8)
    padded[: h * w * c] = texels.reshape(-1)
    payload = np.array([w, h, c, len(uv)], "<i4").tobytes() + uv.tobytes() + padded.tobytes()
    return np.frombuffer(_run(fn, payload), "<f4").copy()


def texpixel(texels, uv2):
    return _tex("texpixel", texels, uv2).reshape(-1, 3)


def texalpha(texels, uv2):
    return _tex("texalpha", texels, uv2)


def surrounds(min_max_x):
    """Interval(min, max).surrounds(x) for n x (min, max, x)"""
    return np.frombuffer(_run("surrounds", np.asarray(min_max_x, "<f4").reshape(-1, 3).tobytes()), "<i4").copy()


def miss(rays6, color3):
    """Miss(ray, colour) -> (colour3, hit_distance, has_prim, front_face)"""
    data = np.concatenate([np.asarray(rays6, "<f4").reshape(-1, 6), np.asarray(color3, "<f4").reshape(-1, 3)], axis=1)
    rec = np.dtype([("color", "<f4", 3), ("t", "<f4"), ("has_prim", "<i4"), ("front", "<i4")])
    out = np.frombuffer(_run("miss", np.ascontiguousarray(data).tobytes()), rec)
    return out["color"].copy(), out["t"].copy(), out["has_prim"].copy(), out["front"].copy()


def boundscentroid(boxes6):
    return np.frombuffer(_run("boundscentroid", np.asarray(boxes6, "<f4").reshape(-1, 6).tobytes()), "<f4").reshape(-1, 3).copy()
