"""Build container only: every image embedded in every .glb under /root/reference, decoded by the product's decoders and
by the reference's own (vendored stb_image inside oracle/_ref/ref_kat); prints the mismatches."""
import glob, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
from make_jpeg_golden import glb_images
import oracle.ref_kat as rk
import dustraytracer_amd as drt
seen, n, bad = set(), 0, []
for p in sorted(glob.glob("/root/reference/**/*.glb", recursive=True)):
    try:
        imgs = list(glb_images(p))
    except Exception:
        continue
    for k, data in enumerate(imgs):
        h = hashlib.sha1(data).hexdigest()
        if h in seen:
            continue
        seen.add(h)
        ref = rk.stbload(data)
        try:
            mine = drt.debug_decode_image(data)
        except Exception as e:
            bad.append((os.path.basename(p), k, str(e)[:60]))
            continue
        n += 1
        if ref is None or ref.shape != mine.shape or not np.array_equal(ref, mine):
            bad.append((os.path.basename(p), k, None if ref is None else ref.shape, mine.shape))
print(n, "distinct images compared with the reference decoder; mismatches:", bad)
