"""glTF(.glb) -> de-indexed vertex streams, restating the reference loader.

TEST INFRASTRUCTURE ONLY (oracle/): never imported by dustraytracer_amd/.

Follows /root/reference/DustRayTracer/src/Core/Scene/Scene.cu *effective*
semantics (parity unpinned: Scene.cu needs tinygltf, an un-vendored submodule,
.gitmodules:15-17, so it cannot be built here):

  * loadGLTFmodel  Scene.cu:181-317  iterates model.nodes (not scenes), one mesh
    per node, node transforms ignored, a node without a mesh is an error.
  * parseMesh      Scene.cu:120-178  indices are read as u16 from byte 0 of the
    *indices'* buffer over [bv.byteOffset/2, (bv.byteOffset+bv.byteLength)/2);
    POSITION/NORMAL/TEXCOORD_0 are read as tightly packed float3/float3/float2
    at their bufferView.byteOffset in that same buffer (accessor offsets and
    strides ignored); a missing attribute maps to accessor 0
    (std::map::operator[]).
  * loadMaterials  Scene.cu:59-86    baseColorFactor rgb + baseColorTexture.index
    used directly as image index.
  * loadTextures   Scene.cu:88-117   one texture per *image*, decoded with the
    file's native channel count (stb_image semantics, Texture.cu:21-30).

PNG decoding is done by Pillow -- an implementation independent of the product's
own from-scratch inflate/unfilter, which is the point.
"""
import io
import json
import struct

import os

import numpy as np


def read_glb(path):
    with open(path, "rb") as f:
        blob = f.read()
    magic, version, length = struct.unpack_from("<4sII", blob, 0)
    if magic != b"glTF" or version != 2:
        raise ValueError("not a glTF 2 binary: %s" % path)
    off = 12
    gltf = None
    bin_chunk = b""
    while off + 8 <= min(length, len(blob)):
        clen, ctype = struct.unpack_from("<I4s", blob, off)
        data = blob[off + 8: off + 8 + clen]
        if ctype == b"JSON":
            gltf = json.loads(data.decode("utf-8"))
        elif ctype == b"BIN\x00" and not bin_chunk:
            bin_chunk = data
        off += 8 + clen + ((4 - clen % 4) % 4)
    if gltf is None:
        raise ValueError("GLB without JSON chunk")
    return gltf, bin_chunk


def _decode_image(data):
    """stb_image-like: 8-bit samples, native channel count (palette expanded).  PNG is lossless, so Pillow's decode is
    the reference's; JPEG is not, and goes through the restatement of stb_image's arithmetic (jpeg_stb.py)."""
    from . import jpeg_stb
    if jpeg_stb.looks_like_jpeg(data):
        # pure-Python Huffman decode: ~12 s for the 4.6 Mpixel texture of the reference's test scenes, so keep the result
        import hashlib
        import os
        import tempfile
        cache = os.path.join(tempfile.gettempdir(), "drt_oracle_jpeg_%s.npy" % hashlib.sha256(data).hexdigest()[:24])
        if os.path.exists(cache):
            try:
                return np.load(cache)
            except (OSError, ValueError):
                pass
        arr = jpeg_stb.decode(data)
        try:
            tmp = cache + ".%d.tmp.npy" % os.getpid()
            np.save(tmp, arr)
            os.replace(tmp, cache)
        except OSError:
            pass
        return arr
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.load()
    if im.mode == "P":
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    elif im.mode in ("1",):
        im = im.convert("L")
    elif im.mode.startswith("I;16") or im.mode == "I":
        raise ValueError("16-bit images are not covered by the oracle")
    comps = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4}[im.mode]
    arr = np.asarray(im, dtype=np.uint8).reshape(im.height, im.width, comps)
    return np.ascontiguousarray(arr)


def flatten(path):
    """Returns dict(pos[n*3,3] f32, nrm[n*3,3] f32, uv[n*3,2] f32, mat[n] i32,
    materials=[(albedo3, tex_index)], textures=[uint8 HxWxC], meshes=[(offset, count)])."""
    is_binary = path.rsplit(".", 1)[-1] == "glb"                  # Scene.cu:32-41
    base_dir = os.path.dirname(os.path.abspath(path))
    if is_binary:
        gltf, bin_chunk = read_glb(path)
    else:
        with open(path, "rb") as f:
            gltf, bin_chunk = json.loads(f.read().decode("utf-8")), b""
    accessors = gltf.get("accessors", [])
    views = gltf.get("bufferViews", [])

    def from_uri(uri, dirs):
        if uri.startswith("data:"):
            import base64
            return base64.b64decode(uri.split(",", 1)[1])
        for d in dirs:
            try:
                with open(os.path.join(d, uri), "rb") as f:
                    return f.read()
            except OSError:
                continue
        raise FileNotFoundError(uri)

    buffers = []
    for b in gltf.get("buffers", []):
        data = from_uri(b["uri"], [base_dir]) if "uri" in b else bin_chunk     # tinygltf resolves buffers next to the file
        buffers.append(data[: b["byteLength"]])

    textures = []
    for img in gltf.get("images", []):
        if is_binary:                                                          # Scene.cu:100-104
            bv = views[img["bufferView"]]
            start = bv.get("byteOffset", 0)
            data = buffers[bv["buffer"]][start: start + bv["byteLength"]]
        else:                                                                  # Scene.cu:90,111: "../models/" + uri from the cwd
            data = from_uri(img["uri"], [os.path.join("..", "models"), base_dir])
        textures.append(_decode_image(data))

    materials = []
    for m in gltf.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        col = pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])
        tex = pbr.get("baseColorTexture", {}).get("index", -1)
        materials.append((np.array(col[:3], dtype=np.float64).astype(np.float32), int(tex)))

    pos_all, nrm_all, uv_all, mat_all, meshes = [], [], [], [], []
    n_prims = 0
    for node in gltf.get("nodes", []):
        if "mesh" not in node:
            raise ValueError("node without mesh: the reference indexes meshes[-1] here (Scene.cu:199-200)")
        mesh = gltf["meshes"][node["mesh"]]
        offset = n_prims
        for prim in mesh["primitives"]:
            attrs = prim.get("attributes", {})
            a_pos = accessors[attrs.get("POSITION", 0)]
            a_nrm = accessors[attrs.get("NORMAL", 0)]
            a_uv = accessors[attrs.get("TEXCOORD_0", 0)]
            a_idx = accessors[prim["indices"]]
            v_pos, v_nrm, v_uv, v_idx = (views[a["bufferView"]] for a in (a_pos, a_nrm, a_uv, a_idx))
            buf = buffers[v_idx["buffer"]]
            u16 = np.frombuffer(buf, dtype="<u2", count=len(buf) // 2)
            i0 = v_idx.get("byteOffset", 0) // 2
            i1 = (v_idx["byteLength"] + v_idx.get("byteOffset", 0)) // 2
            idx = u16[i0:i1].astype(np.int64)

            def stream(view, width):
                o = view.get("byteOffset", 0)
                flat = np.frombuffer(buf, dtype="<f4", offset=o, count=(len(buf) - o) // 4)
                usable = (flat.size // width) * width
                return flat[:usable].reshape(-1, width)

            pos_all.append(stream(v_pos, 3)[idx])
            nrm_all.append(stream(v_nrm, 3)[idx])
            uv_all.append(stream(v_uv, 2)[idx])
            mat_all.append(np.full(a_idx["count"] // 3, prim.get("material", -1), dtype=np.int32))
        pos_n = sum(len(p) for p in pos_all) // 3
        n_prims = pos_n
        meshes.append((offset, n_prims - offset))

    pos = np.ascontiguousarray(np.concatenate(pos_all), dtype=np.float32) if pos_all else np.zeros((0, 3), np.float32)
    nrm = np.ascontiguousarray(np.concatenate(nrm_all), dtype=np.float32) if nrm_all else np.zeros((0, 3), np.float32)
    uv = np.ascontiguousarray(np.concatenate(uv_all), dtype=np.float32) if uv_all else np.zeros((0, 2), np.float32)
    mat = np.concatenate(mat_all) if mat_all else np.zeros((0,), np.int32)
    if len(pos) % 3 or len(mat) != len(pos) // 3:
        raise ValueError("index/material bookkeeping mismatch (Scene.cu:166-175 would misindex)")
    return dict(pos=pos, nrm=nrm, uv=uv, mat=mat, materials=materials, textures=textures, meshes=meshes)
