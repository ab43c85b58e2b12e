"""Writes tests/golden/jpeg/*.jpg (small synthetic JPEGs, encoded by Pillow) and tests/golden/jpeg_ref.json: what the
REFERENCE's decoder (its vendored stb_image, compiled into oracle/_ref/ref_kat, called as Texture.cu:23 calls it) makes
of each of them and of the JPEG embedded in models/test/sunshadowTest.glb -- size, channel count, SHA-256 of the texels,
and the first and last 48 bytes.  Run in the build container (needs /root/reference for `make -C oracle ref`)."""
import hashlib, io, json, os, struct, sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.ref_kat as rk                                              # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "jpeg")


def glb_images(path):
    b = open(path, "rb").read()
    jlen = struct.unpack("<I", b[12:16])[0]
    js = json.loads(b[20:20 + jlen])
    base = 20 + jlen + 8
    for im in js.get("images", []):
        bv = js["bufferViews"][im["bufferView"]]
        off = base + bv.get("byteOffset", 0)
        yield b[off:off + bv["byteLength"]]


def synth(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 7.0 + seed) * np.cos(y / 5.0), 255.0 * x / max(w - 1, 1), 255.0 * y / max(h - 1, 1)], -1)
    img += rng.normal(0, 12, img.shape)
    img[h // 3: h // 3 + 5, :, :] = 255                                  # hard edges: large AC coefficients, clamping
    img[:, w // 2: w // 2 + 3, :] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    if not rk.available():
        sys.exit("oracle/_ref/ref_kat missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    cases = [("rgb_444_64x48_q90", (64, 48), dict(quality=90, subsampling=0)),
             ("rgb_422_50x37_q75", (50, 37), dict(quality=75, subsampling=1)),
             ("rgb_420_77x53_q85", (77, 53), dict(quality=85, subsampling=2)),
             ("rgb_420_16x16_q30", (16, 16), dict(quality=30, subsampling=2)),
             ("rgb_420_1x1_q90", (1, 1), dict(quality=90, subsampling=2)),
             ("rgb_420_129x7_q95_opt", (129, 7), dict(quality=95, subsampling=2, optimize=True)),
             ("gray_33x31_q80", (33, 31), dict(quality=80)),
             ("rgb_420_96x80_q60_rst", (96, 80), dict(quality=60, subsampling=2, restart_marker_blocks=3))]
    ref = {}
    for k, (name, (w, h), kw) in enumerate(cases):
        arr = synth(w, h, k + 1)
        im = Image.fromarray(arr[..., 0] if name.startswith("gray") else arr)
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        data = buf.getvalue()
        open(os.path.join(OUT, name + ".jpg"), "wb").write(data)
        ref[name + ".jpg"] = describe(rk.stbload(data))
    # PNG is lossless, but the decoder still decides channel counts: palette (+tRNS) expansion, 16 -> 8 bits, 1-bit grey
    png_dir = os.path.join(ROOT, "tests", "golden", "png")
    os.makedirs(png_dir, exist_ok=True)
    rng = np.random.default_rng(3)
    pal = Image.fromarray(rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)).quantize(16)
    pngs = {"gray8": (Image.fromarray(rng.integers(0, 256, (13, 17), dtype=np.uint8)), {}),
            "graya8": (Image.fromarray(rng.integers(0, 256, (13, 17, 2), dtype=np.uint8), "LA"), {}),
            "rgb8": (Image.fromarray(rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)), {}),
            "rgba8": (Image.fromarray(rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)), {}),
            "gray16": (Image.fromarray(rng.integers(0, 65536, (13, 17), dtype=np.uint16)), {}),
            "pal16": (pal, {}), "pal16_trns": (pal, dict(transparency=3)),
            "bit1": (Image.fromarray(rng.integers(0, 2, (13, 17), dtype=np.uint8) * 255).convert("1"), {})}
    for name, (im, kw) in pngs.items():
        buf = io.BytesIO()
        im.save(buf, "PNG", **kw)
        open(os.path.join(png_dir, name + ".png"), "wb").write(buf.getvalue())
        ref["png/" + name + ".png"] = describe(rk.stbload(buf.getvalue()))
    big = list(glb_images(os.path.join(ROOT, "models", "test", "sunshadowTest.glb")))[0]
    ref["sunshadowTest.glb#image0"] = describe(rk.stbload(big))
    json.dump(ref, open(os.path.join(ROOT, "tests", "golden", "jpeg_ref.json"), "w"), indent=1, sort_keys=True)
    for k, v in ref.items():
        print(k, v["shape"], v["sha256"][:16])


def describe(px):
    flat = px.reshape(-1)
    return {"shape": list(px.shape), "sha256": hashlib.sha256(px.tobytes()).hexdigest(),
            "head": flat[:48].tolist(), "tail": flat[-48:].tolist()}


if __name__ == "__main__":
    main()
