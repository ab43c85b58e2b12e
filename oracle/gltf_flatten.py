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


def flatten(path, strict=False):
    """Returns dict(pos[n*3,3] f32, nrm[n*3,3] f32, uv[n*3,2] f32, mat[n] i32,
    materials=[(albedo3, tex_index)], textures=[uint8 HxWxC], meshes=[(offset, count)]).
    strict=True: geometry and texture indices as the glTF 2.0 specification defines them (flatten_strict below)."""
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

    materials, materials_ext = [], []
    for m in gltf.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        col = pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])
        tex = pbr.get("baseColorTexture", {}).get("index", -1)
        materials.append((np.array(col[:3], dtype=np.float64).astype(np.float32), int(tex)))
        # Scene.cu:71-75: loaded by the reference, read by nothing; tinygltf defaults 1 / 1 / (0, 0, 0)
        materials_ext.append((np.array(m.get("emissiveFactor", [0.0, 0.0, 0.0]), np.float64).astype(np.float32),
                              np.float32(pbr.get("roughnessFactor", 1.0)), int(pbr.get("metallicFactor", 1.0) > 0)))

    if strict:
        jtex = gltf.get("textures", [])
        materials = [(alb, jtex[t].get("source", -1) if t >= 0 else -1) for alb, t in materials]     # spec: texture -> image (textures[i].source)
        geo = flatten_strict(gltf, buffers, n_file_materials=len(materials))
        if geo.pop("default_material_used"):
            materials.append((np.float32([1, 1, 1]), -1))                                           # the specification's default material
            materials_ext.append((np.float32([0, 0, 0]), np.float32(1.0), 1))
        return dict(materials=materials, materials_ext=materials_ext, textures=textures, **geo)

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
    return dict(pos=pos, nrm=nrm, uv=uv, mat=mat, materials=materials, materials_ext=materials_ext, textures=textures, meshes=meshes)


# ------------------------------------------------------------------------------------------------------------------------
# The same file read as the glTF 2.0 SPECIFICATION says (what DRT_LOAD_STRICT promises), restated independently of the
# product's C++ (dustraytracer_amd/csrc/scene_host.cpp load_strict_geometry): numpy, float64.  Rules, with the section of
# the specification each one follows:
#   3.5.2  the scene to show is `scene` (default 0); without scenes every root node; nodes are visited depth first, a node
#          before its children, children in listed order
#   3.5.3  a node's local transform is `matrix` (column-major) or T * R * S; the world transform is parent * local
#   3.6.2  accessor: bufferView.byteOffset + accessor.byteOffset, element stride = bufferView.byteStride or the element size,
#          componentType 5120..5126, `normalized` integers map to [-1, 1] / [0, 1]
#   3.7.2  primitive: mode 4 (triangles; points and lines are skipped), indices optional, NORMAL optional (flat normals from
#          the geometry, counter-clockwise front), TEXCOORD_0 optional (0, 0), material optional (default material)
#   3.7.2.1 positions transform by the world matrix, normals by its inverse transpose (re-normalised)
# Every float64 expression is written out in one fixed order (sums left to right), so that another implementation that
# follows the same order gets the same float32 results after the final rounding.
# ------------------------------------------------------------------------------------------------------------------------
_COMPONENT = {5120: ("<i1", 1, 127.0), 5121: ("<u1", 1, 255.0), 5122: ("<i2", 2, 32767.0), 5123: ("<u2", 2, 65535.0),
              5125: ("<u4", 4, None), 5126: ("<f4", 4, None)}
_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def read_accessor(gltf, buffers, index):
    """-> float64 array [count, width] (integers converted; normalized ones scaled as 3.6.2.5 says)"""
    a = gltf["accessors"][index]
    if "sparse" in a:
        raise ValueError("sparse accessors")
    v = gltf["bufferViews"][a["bufferView"]]
    dt, size, scale = _COMPONENT[a["componentType"]]
    width = _WIDTH[a["type"]]
    elem = size * width
    stride = v.get("byteStride", 0) or elem
    start = v.get("byteOffset", 0) + a.get("byteOffset", 0)
    count = a["count"]
    raw = np.frombuffer(buffers[v["buffer"]], np.uint8)
    if count and a.get("byteOffset", 0) + stride * (count - 1) + elem > v["byteLength"]:
        raise ValueError("accessor runs past its bufferView")
    rows = np.stack([raw[start + i * stride: start + i * stride + elem] for i in range(count)]) if count else np.zeros((0, elem), np.uint8)
    vals = np.ascontiguousarray(rows).view(dt).reshape(count, width).astype(np.float64)
    if a.get("normalized") and scale is not None:
        vals = vals / scale
        if a["componentType"] in (5120, 5122):
            vals = np.maximum(vals, -1.0)
    return vals


def _local_matrix(node):
    if "matrix" in node and len(node["matrix"]) == 16:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T          # column-major in the file; M[row, col] here
    t = np.array(node.get("translation", [0, 0, 0]), np.float64)
    x, y, z, w = np.array(node.get("rotation", [0, 0, 0, 1]), np.float64)
    sx, sy, sz = np.array(node.get("scale", [1, 1, 1]), np.float64)
    m = np.identity(4)
    m[0, 0] = (1 - 2 * (y * y + z * z)) * sx; m[1, 0] = (2 * (x * y + z * w)) * sx; m[2, 0] = (2 * (x * z - y * w)) * sx
    m[0, 1] = (2 * (x * y - z * w)) * sy; m[1, 1] = (1 - 2 * (x * x + z * z)) * sy; m[2, 1] = (2 * (y * z + x * w)) * sy
    m[0, 2] = (2 * (x * z + y * w)) * sz; m[1, 2] = (2 * (y * z - x * w)) * sz; m[2, 2] = (1 - 2 * (x * x + y * y)) * sz
    m[0:3, 3] = t
    return m


def _matmul(a, b):
    r = np.zeros((4, 4))
    for c in range(4):
        for row in range(4):
            s = 0.0
            for k in range(4):
                s += a[row, k] * b[k, c]
            r[row, c] = s
    return r


def _normal_matrix(w):
    """inverse transpose of the upper 3x3: cofactors / determinant"""
    a, b, c = w[0, 0], w[0, 1], w[0, 2]
    d, e, f = w[1, 0], w[1, 1], w[1, 2]
    g, h, i = w[2, 0], w[2, 1], w[2, 2]
    co = np.array([[e * i - f * h, f * g - d * i, d * h - e * g],
                   [c * h - b * i, a * i - c * g, b * g - a * h],
                   [b * f - c * e, c * d - a * f, a * e - b * d]])
    det = a * co[0, 0] + b * co[0, 1] + c * co[0, 2]
    return co * (1.0 / det if det != 0 else 1.0)


def flatten_strict(gltf, buffers, n_file_materials):
    nodes, meshes = gltf.get("nodes", []), gltf.get("meshes", [])
    scenes = gltf.get("scenes", [])
    if scenes:
        which = max(0, gltf.get("scene", 0))
        roots = list(scenes[which if which < len(scenes) else 0].get("nodes", []))
    else:
        children = {c for n in nodes for c in n.get("children", [])}
        roots = [i for i in range(len(nodes)) if i not in children]
    pos_all, nrm_all, uv_all, mat_all, mesh_ranges = [], [], [], [], []
    default_used = False

    def visit(index, parent, depth):
        nonlocal default_used
        if depth > 256:
            raise ValueError("node hierarchy deeper than 256 levels")
        node = nodes[index]
        world = _matmul(parent, _local_matrix(node))
        if "mesh" in node:
            identity = np.array_equal(world, np.identity(4))
            nm = _normal_matrix(world)
            first = sum(len(m) for m in mat_all)
            for prim in meshes[node["mesh"]]["primitives"]:
                mode = prim.get("mode", 4)
                if mode < 4:
                    continue
                if mode != 4:
                    raise ValueError("triangle strips / fans")
                attrs = prim.get("attributes", {})
                if "POSITION" not in attrs:
                    continue
                p = read_accessor(gltf, buffers, attrs["POSITION"])
                idx = read_accessor(gltf, buffers, prim["indices"])[:, 0].astype(np.int64) if "indices" in prim else np.arange(len(p))
                idx = idx[: (len(idx) // 3) * 3]
                if len(idx) and idx.max() >= len(p):
                    raise ValueError("vertex index out of range")
                x, y, z = p[idx, 0], p[idx, 1], p[idx, 2]
                if identity:
                    wp = np.stack([x, y, z], 1)
                else:
                    wp = np.stack([((world[r, 0] * x + world[r, 1] * y) + world[r, 2] * z) + world[r, 3] for r in range(3)], 1)
                if "NORMAL" in attrs:
                    n = read_accessor(gltf, buffers, attrs["NORMAL"])[idx]
                    if not identity:
                        w3 = np.stack([(nm[r, 0] * n[:, 0] + nm[r, 1] * n[:, 1]) + nm[r, 2] * n[:, 2] for r in range(3)], 1)
                        ln = np.sqrt((w3[:, 0] * w3[:, 0] + w3[:, 1] * w3[:, 1]) + w3[:, 2] * w3[:, 2])
                        n = np.where(ln[:, None] > 0, w3 / np.where(ln > 0, ln, 1.0)[:, None], w3)
                else:
                    t = wp.reshape(-1, 3, 3)
                    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
                    g = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                                  e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], 1)
                    ln = np.sqrt((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2])
                    g = np.where(ln[:, None] > 0, g / np.where(ln > 0, ln, 1.0)[:, None], 0.0)
                    n = np.repeat(g, 3, axis=0)
                uv = read_accessor(gltf, buffers, attrs["TEXCOORD_0"])[idx] if "TEXCOORD_0" in attrs else np.zeros((len(idx), 2))
                material = prim.get("material", -1)
                if material < 0:
                    default_used = True
                    material = n_file_materials
                pos_all.append(wp.astype(np.float32)); nrm_all.append(n.astype(np.float32)); uv_all.append(uv.astype(np.float32))
                mat_all.append(np.full(len(idx) // 3, material, np.int32))
            count = sum(len(m) for m in mat_all) - first
            if count:
                mesh_ranges.append((first, count))
        for c in node.get("children", []):
            visit(c, world, depth + 1)

    for r in roots:
        visit(r, np.identity(4), 0)
    cat = lambda parts, w: np.ascontiguousarray(np.concatenate(parts), np.float32) if parts else np.zeros((0, w), np.float32)
    return dict(pos=cat(pos_all, 3), nrm=cat(nrm_all, 3), uv=cat(uv_all, 2),
                mat=np.concatenate(mat_all) if mat_all else np.zeros((0,), np.int32), meshes=mesh_ranges, default_material_used=default_used)
