"""Host prep of the product (C++ glTF flattening, PNG decode, binned-SAH builder inside libdrt_hip.so)
against the oracle's independent restatement (Python GLB reader + Pillow PNG + C builder).
No GPU needed: these entry points never touch HIP.  Everything is bit-exact."""
import os

import numpy as np
import pytest

import oracle
from tests.scenes import SCENES, bits, scene_path

drt = pytest.importorskip("dustraytracer_amd")


def load_both(name, leaf=20, bins=8):
    sc = drt.Scene()
    sc.loadGLTFmodel(scene_path(name))
    osc = oracle.Scene.load_glb(scene_path(name))
    return sc, osc


@pytest.mark.parametrize("name", sorted(SCENES))
def test_flattened_triangles_match_oracle(name):
    sc, osc = load_both(name)
    tris = sc.m_PrimitivesBuffer
    assert len(tris) == len(osc.tris) and len(tris) > 0
    assert np.array_equal(bits(tris["centroid"]), bits(osc.tris["centroid"]))
    assert np.array_equal(bits(tris["vertex"]["position"]), bits(osc.tris["p"]))
    assert np.array_equal(bits(tris["vertex"]["normal"]), bits(osc.tris["n"]))
    assert np.array_equal(bits(tris["vertex"]["uv"]), bits(osc.tris["uv"]))
    assert np.array_equal(bits(tris["face_normal"]), bits(osc.tris["face_n"]))
    assert np.array_equal(tris["material"], osc.tris["material"])
    mats = sc.m_Material
    assert len(mats) == osc.n_mats
    assert np.array_equal(bits(mats["albedo"]), bits(osc.mats["albedo"][: osc.n_mats]))
    assert np.array_equal(mats["albedo_tex"], osc.mats["albedo_tex"][: osc.n_mats])
    meshes = sc.m_Meshes
    assert [(int(m["primitives_offset"]), int(m["tris_count"])) for m in meshes] == list(osc.meshes)


@pytest.mark.parametrize("name", ["cornell_box", "dense_monkey", "uv_texture_test"])
def test_png_decoder_matches_pillow(name):
    sc, osc = load_both(name)
    texs = sc.m_Textures
    assert len(texs) == len(osc.textures) and len(texs) > 0
    for mine, ref in zip(texs, osc.textures):
        assert mine.shape == ref.shape
        assert np.array_equal(mine, ref)


def test_uv_texture_test_has_alpha_channel():
    sc, _ = load_both("uv_texture_test")
    comps = sorted(t.shape[2] for t in sc.m_Textures)
    assert 4 in comps            # the only fixture that exercises AnyHit's alpha cut-out
    alpha = [t for t in sc.m_Textures if t.shape[2] == 4][0][..., 3]
    assert alpha.min() < 255 and alpha.max() == 255


@pytest.mark.parametrize("name,leaf,bins", [(n, 20, 8) for n in sorted(SCENES)] +
                         [("suzanne_plane", 6, 8), ("suzanne_plane", 4, 4), ("room", 1, 16), ("dense_monkey", 8, 12)])
def test_bvh_matches_oracle(name, leaf, bins):
    sc, osc = load_both(name)
    b = drt.BVHBuilder()
    b.m_TargetLeafPrimitivesCount, b.m_BinCount = leaf, bins
    try:
        osc.build_bvh(leaf, bins)
    except RuntimeError:
        with pytest.raises(drt.DrtError) as e:
            b.buildIterative(sc)
        assert e.value.code == drt.ERR_BVH        # the reference never terminates on this input
        return
    b.buildIterative(sc)
    nodes, onodes = sc.m_BVHNodes, osc.nodes
    assert len(nodes) == len(onodes)
    assert np.array_equal(nodes["is_leaf"].astype(np.int32), onodes["is_leaf"])
    for f in ("child1", "child2", "prim_count", "prim_start"):
        assert np.array_equal(nodes[f], onodes[f]), f
    assert np.array_equal(bits(nodes["bmin"]), bits(onodes["bmin"]))
    assert np.array_equal(bits(nodes["bmax"]), bits(onodes["bmax"]))
    # triangles were reordered identically (libstdc++ std::partition swap sequence)
    assert np.array_equal(bits(sc.m_PrimitivesBuffer["centroid"]), bits(osc.tris["centroid"]))
    assert sc.bvh_depth == oracle.tree_depth(onodes)


def test_bvh_shape_of_baseline_scenes():
    """Node counts / depths measured by SURVEY.md 8(a) T1 with the reference's own builder sources."""
    expect = {"cornell_box": (34, 5, 3), "room": (330, 51, 8), "suzanne_plane": (970, 137, 10), "dense_monkey": (15744, 2347, 15)}
    for name, (tris, nodes, depth) in expect.items():
        sc, _ = load_both(name)
        b = drt.BVHBuilder()
        b.m_TargetLeafPrimitivesCount, b.m_BinCount = 20, 8
        b.buildIterative(sc)
        assert (len(sc.m_PrimitivesBuffer), len(sc.m_BVHNodes), sc.bvh_depth) == (tris, nodes, depth), name
        n = sc.m_BVHNodes
        root = n[-1]                                   # root is the last node (BVHBuilder.cu:85)
        assert root["prim_start"] == 0 and root["prim_count"] == tris
        leaves = n[n["is_leaf"] == 1]
        assert leaves["prim_count"].sum() == tris and leaves["prim_count"].max() <= 20


def test_small_scene_becomes_single_leaf():
    sc = drt.Scene()
    sc.addMaterial((0.5, 0.5, 0.5))
    pos = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), 3)[None]
    sc.setGeometry(pos, nrm, np.zeros((1, 6), np.float32), [0])
    b = drt.BVHBuilder()
    b.buildIterative(sc)
    n = sc.m_BVHNodes
    assert len(n) == 1 and n[0]["is_leaf"] == 1 and n[0]["prim_count"] == 1     # BVHBuilder.cu:34-43
    t = sc.m_PrimitivesBuffer[0]
    assert np.allclose(t["face_normal"], [0, 0, 1]) and np.allclose(t["centroid"], [1 / 3, 1 / 3, 0])


def test_loader_errors_are_reported_not_fatal(tmp_path):
    sc = drt.Scene()
    with pytest.raises(drt.DrtError) as e:
        sc.loadGLTFmodel(str(tmp_path / "missing.glb"))
    assert e.value.code == drt.ERR_IO
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + b"\x02\x00\x00\x00" + b"\x20\x00\x00\x00" + b"\x04\x00\x00\x00JSON{{{{" + b"\0" * 8)
    with pytest.raises(drt.DrtError) as e:
        sc.loadGLTFmodel(str(bad))
    assert e.value.code == drt.ERR_PARSE
    with pytest.raises(drt.DrtError) as e:
        sc.loadGLTFmodel(str(tmp_path / "scene.gltf"))
    assert e.value.code == drt.ERR_IO
    # a scene that failed to load keeps the handle usable
    sc.loadGLTFmodel(scene_path("room"))
    assert len(sc.m_PrimitivesBuffer) == 330


def test_degenerate_bvh_input_is_an_error_not_a_hang():
    sc = drt.Scene()
    sc.addMaterial((1, 1, 1))
    n = 30                                              # > leaf target, all centroids identical
    pos = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], np.float32), (n, 1))
    nrm = np.tile(np.array([0, 0, 1] * 3, np.float32), (n, 1))
    sc.setGeometry(pos, nrm, np.zeros((n, 6), np.float32), np.zeros(n, np.int32))
    b = drt.BVHBuilder()
    with pytest.raises(drt.DrtError) as e:
        b.buildIterative(sc)
    assert e.value.code == drt.ERR_BVH


def test_camera_host_logic_matches_reference(kat_golden):
    """Camera::OnUpdate + Camera::Rotate (Camera.cu:44-80) over 512 editor frames: position, forward and right equal the
    reference's own compiled code bit for bit (tests/golden/kat_ref.npz `cam_track`, made by oracle/_ref/ref_kat)."""
    import dustraytracer_amd as drt
    g = kat_golden
    c = g["cam_start"]
    cam = drt.Camera(c[0:3])
    cam.m_Forward_dir, cam.m_Up_dir, cam.m_Right_dir = c[3:6].copy(), c[6:9].copy(), c[9:12].copy()
    cam.m_movement_speed = float(c[12])
    for k, st in enumerate(g["cam_steps"]):
        cam.OnUpdate(st[4:7], float(st[7]))                 # EditorLayer.cpp:400-401 order
        cam.Rotate(st[0:4])
        got = np.concatenate([cam.m_Position, cam.m_Forward_dir, cam.m_Right_dir]).astype(np.float32)
        assert np.array_equal(bits(got), bits(g["cam_track"][k])), k
    assert np.linalg.norm(g["cam_track"][-1][:3] - c[0:3]) > 0.1


def _jpeg_ref():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_ref.json")))


def test_jpeg_decoders_match_the_reference_decoder():
    """Baseline JPEG is lossy and its decoders differ (IDCT, chroma upsampling, colour conversion); texels feed the shading,
    so both decoders here -- the product's C++ one and the oracle's numpy one -- must equal the reference's decoder (its
    vendored stb_image, run from oracle/_ref/ref_kat to produce tests/golden/jpeg_ref.json) on every byte: 4:4:4, 4:2:2,
    4:2:0, greyscale, restart intervals, optimised tables, sizes that are not MCU multiples, a 1x1 image."""
    import glob
    import hashlib
    import dustraytracer_amd as drt
    from oracle import jpeg_stb
    ref = _jpeg_ref()
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg", "*.jpg")))
    assert len(files) == 8
    for f in files:
        data = open(f, "rb").read()
        want = ref[os.path.basename(f)]
        for who, px in (("product", drt.debug_decode_image(data)), ("oracle", jpeg_stb.decode(data))):
            assert list(px.shape) == want["shape"], (who, f)
            assert px.reshape(-1)[:48].tolist() == want["head"], (who, f)
            assert hashlib.sha256(px.tobytes()).hexdigest() == want["sha256"], (who, f)


def test_png_colour_types_decode_like_the_reference_decoder():
    """Grey, grey+alpha, RGB, RGBA, 16-bit, palette, palette+tRNS, 1-bit: channel count and texels as the reference's
    stb_image returns them (golden from oracle/_ref/ref_kat), for the product's decoder and the oracle's (Pillow-based)."""
    import glob
    import hashlib
    import dustraytracer_amd as drt
    from oracle.gltf_flatten import _decode_image
    ref = _jpeg_ref()
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png", "*.png")))
    assert len(files) == 8
    for f in files:
        data = open(f, "rb").read()
        want = ref["png/" + os.path.basename(f)]
        got = {"product": drt.debug_decode_image(data)}
        if "gray16" not in f:                       # the oracle's Pillow path declines 16-bit images (none in the reference's scenes)
            got["oracle"] = _decode_image(data)
        for who, px in got.items():
            assert list(px.shape) == want["shape"], (who, f)
            assert hashlib.sha256(px.tobytes()).hexdigest() == want["sha256"], (who, f)


def test_reference_jpeg_textured_scene_loads_like_the_reference():
    """models/test/sunshadowTest.glb (and three more of the reference's test scenes) embed one 2164x2152 baseline 4:2:0
    JPEG: product loader and oracle loader both reproduce the reference decoder's 14 MB of texels exactly."""
    import hashlib
    import dustraytracer_amd as drt
    want = _jpeg_ref()["sunshadowTest.glb#image0"]
    sc = drt.Scene()
    sc.loadGLTFmodel(scene_path("sunshadow_test"))
    tex = sc.m_Textures
    assert len(tex) == 1 and list(tex[0].shape) == want["shape"]
    assert hashlib.sha256(tex[0].tobytes()).hexdigest() == want["sha256"]
    osc = oracle.Scene.load_glb(scene_path("sunshadow_test"))
    assert hashlib.sha256(osc.textures[0].tobytes()).hexdigest() == want["sha256"]


def test_unsupported_jpeg_flavours_are_errors():
    import io
    import dustraytracer_amd as drt
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(buf, "JPEG", progressive=True)
    with pytest.raises(drt.DrtError):
        drt.debug_decode_image(buf.getvalue())
    with pytest.raises(drt.DrtError):
        drt.debug_decode_image(buf.getvalue()[:100])


def test_ascii_gltf_with_data_uris_and_external_files(tmp_path):
    """The ASCII branch (Scene.cu:38-41): buffers from a file next to the .gltf or from a base64 data: URI, images from the
    uri (the reference's "../models/" + uri first, then next to the file).  The same geometry packed three ways must load
    to the same triangles as the .glb."""
    import base64
    import json
    import shutil
    import struct
    src = scene_path("uv_texture_test")                         # a GLB with one PNG
    blob = open(src, "rb").read()
    jlen = struct.unpack_from("<I", blob, 12)[0]
    gltf = json.loads(blob[20:20 + jlen])
    bin_chunk = blob[20 + jlen + 8:]
    want = drt.Scene()
    want.loadGLTFmodel(src)
    # external files: every image's bufferView becomes a .png file, the buffer a .bin file
    def image_bytes(k):
        bv = gltf["bufferViews"][gltf["images"][k]["bufferView"]]
        return bin_chunk[bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]]
    n_img = len(gltf["images"])
    g = json.loads(json.dumps(gltf))
    (tmp_path / "tex").mkdir()
    for k in range(n_img):
        g["images"][k] = {"uri": "tex/picture%d.png" % k}
        (tmp_path / "tex" / ("picture%d.png" % k)).write_bytes(image_bytes(k))
    g["buffers"][0]["uri"] = "geometry.bin"
    (tmp_path / "geometry.bin").write_bytes(bin_chunk[: gltf["buffers"][0]["byteLength"]])
    (tmp_path / "a.gltf").write_text(json.dumps(g))
    # data URIs for both
    g2 = json.loads(json.dumps(g))
    for k in range(n_img):
        g2["images"][k] = {"uri": "data:image/png;base64," + base64.b64encode(image_bytes(k)).decode()}
    g2["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bin_chunk[: gltf["buffers"][0]["byteLength"]]).decode()
    (tmp_path / "b.gltf").write_text(json.dumps(g2))
    for name in ("a.gltf", "b.gltf"):
        sc = drt.Scene()
        sc.loadGLTFmodel(str(tmp_path / name))
        assert np.array_equal(sc.m_PrimitivesBuffer.view(np.uint8), want.m_PrimitivesBuffer.view(np.uint8)), name
        assert all(np.array_equal(a, b) for a, b in zip(sc.m_Textures, want.m_Textures)) and len(sc.m_Textures) == n_img, name
        osc = oracle.Scene.load_glb(str(tmp_path / name))
        assert all(np.array_equal(a, b) for a, b in zip(osc.textures, want.m_Textures)) and len(osc.tris) == len(want.m_PrimitivesBuffer)
    shutil.rmtree(tmp_path / "tex")
    with pytest.raises(drt.DrtError) as e:                      # the image file is gone
        drt.Scene().loadGLTFmodel(str(tmp_path / "a.gltf"))
    assert e.value.code == drt.ERR_IO


def _tri_bytes(sc):
    return np.ascontiguousarray(sc.m_PrimitivesBuffer).view(np.uint8)


@pytest.mark.parametrize("name", ["bvh_split_test", "cornell_box", "cornell_box_gltf", "cs16_dust", "dense_monkey", "mc_transparency",
                                  "multi_material", "sunshadow_test", "suzanne_plane", "uv_texture_gltf", "uv_texture_test"])
def test_strict_loading_equals_reference_loading_where_the_assumptions_hold(name):
    """DRT_LOAD_STRICT reads the file as the glTF specification says; the default reads it as Scene.cu does.  For files
    that meet the reference's assumptions (u16 indices at the start of their buffer view, tightly packed attributes, no
    transforms, texture i = image i, flat node list) the two must give the same 128-byte triangle records, bit for bit."""
    a, b = drt.Scene(), drt.Scene()
    a.loadGLTFmodel(scene_path(name))
    b.loadGLTFmodel(scene_path(name), strict=True)
    assert np.array_equal(_tri_bytes(a), _tri_bytes(b))
    assert np.array_equal(a.m_Material["albedo_tex"], b.m_Material["albedo_tex"])


def _write_assumption_breaking_glb(tmp_path, want_locals=False):
    """The GLB of test_strict_loading_honours_what_the_reference_ignores (see there)."""
    import json
    import struct
    rng = np.random.default_rng(11)
    # mesh 0: a quad, interleaved (pos3, nrm3, uv2 = 32 bytes) with 8 bytes of padding per vertex (stride 40), u32 indices
    quad_p = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
    quad_n = np.float32([[0, 0, 1]] * 4)
    quad_t = np.float32([[0, 0], [1, 0], [1, 1], [0, 1]])
    inter = np.zeros((4, 10), np.float32)
    inter[:, 0:3], inter[:, 3:6], inter[:, 6:8] = quad_p, quad_n, quad_t
    idx32 = np.uint32([0, 1, 2, 0, 2, 3])
    # mesh 1: one non-indexed triangle with positions only
    tri_p = np.float32([[0, 0, 0], [2, 0, 0], [0, 2, 0]])
    chunks, views = [], []

    def add_view(data, stride=None):
        off = sum(len(c) for c in chunks)
        pad = (-off) % 4
        if pad:
            chunks.append(b"\0" * pad)
            off += pad
        chunks.append(bytes(data))
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        return len(views) - 1
    v_inter = add_view(inter.tobytes(), 40)
    v_idx = add_view(b"\xAA" * 12 + idx32.tobytes())            # 12 junk bytes, then the indices (accessor byteOffset 12)
    v_tri = add_view(tri_p.tobytes())
    png = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png", "rgb8.png"), "rb").read()
    png2 = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png", "rgba8.png"), "rb").read()
    v_png, v_png2 = add_view(png), add_view(png2)
    q = np.array([0.0, 0.0, np.sin(np.pi / 8), np.cos(np.pi / 8)])          # 45 degrees about z
    child_matrix = np.eye(4)
    child_matrix[:3, 3] = [0, 0, 5]
    child_matrix[0, 0] = 2
    gltf = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 3]}],
            "nodes": [{"children": [1, 2], "translation": [1, 2, 3], "rotation": q.tolist(), "scale": [2, 2, 2]},
                      {"mesh": 0, "matrix": child_matrix.T.reshape(-1).tolist()},
                      {"name": "empty"},
                      {"mesh": 0}, ],
            "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0},
                                       {"attributes": {"POSITION": 4}}]}],
            "accessors": [{"bufferView": v_inter, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                          {"bufferView": v_inter, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC3"},
                          {"bufferView": v_inter, "byteOffset": 24, "componentType": 5126, "count": 4, "type": "VEC2"},
                          {"bufferView": v_idx, "byteOffset": 12, "componentType": 5125, "count": 6, "type": "SCALAR"},
                          {"bufferView": v_tri, "componentType": 5126, "count": 3, "type": "VEC3"}],
            "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.6, 0.7, 1], "baseColorTexture": {"index": 0}}}],
            "textures": [{"source": 1}], "images": [{"bufferView": v_png, "mimeType": "image/png"}, {"bufferView": v_png2, "mimeType": "image/png"}],
            "bufferViews": views, "buffers": [{"byteLength": 0}]}
    binary = b"".join(chunks)
    binary += b"\0" * ((-len(binary)) % 4)
    gltf["buffers"][0]["byteLength"] = len(binary)
    js = json.dumps(gltf).encode()
    js += b" " * ((-len(js)) % 4)
    glb = b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(binary)) + struct.pack("<I", len(js)) + b"JSON" + js + struct.pack("<I", len(binary)) + b"BIN\0" + binary
    path = tmp_path / "strict.glb"
    path.write_bytes(glb)
    return locals() if want_locals else path


def test_strict_loading_honours_what_the_reference_ignores(tmp_path):
    """A GLB built to break every assumption of Scene.cu:120-200: u32 indices behind an accessor byteOffset, interleaved
    vertices (byteStride), a parent node with translation + rotation + scale, a child with a matrix, an empty node, the
    same mesh used by two nodes, a primitive without NORMAL / TEXCOORD_0 / material / indices, texture -> image
    indirection.  Expected geometry is computed here with numpy in float64."""
    L = _write_assumption_breaking_glb(tmp_path, want_locals=True)
    path, q, quad_p, quad_n, quad_t, tri_p, idx32, child_matrix = (L[k] for k in ("path", "q", "quad_p", "quad_n", "quad_t", "tri_p", "idx32", "child_matrix"))

    sc = drt.Scene()
    sc.loadGLTFmodel(str(path), strict=True)
    tris = sc.m_PrimitivesBuffer
    # expected: DFS from the scene's roots: node0 (no mesh) -> child 1 (mesh 0) -> child 2 (empty) ; then node 3 (mesh 0)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M0 = np.eye(4)
    M0[:3, :3] = R * 2.0
    M0[:3, 3] = [1, 2, 3]
    W1 = M0 @ child_matrix

    def instance(Wm):
        out = []
        for tri in (idx32[:3], idx32[3:]):
            out.append(((Wm[:3, :3] @ quad_p[tri].astype(np.float64).T).T + Wm[:3, 3], 0))
        out.append(((Wm[:3, :3] @ tri_p.astype(np.float64).T).T + Wm[:3, 3], 1))
        return out
    want = instance(W1) + instance(np.eye(4))
    assert len(tris) == len(want) == 6
    for t, (p, m) in zip(tris, want):
        assert np.allclose(t["vertex"]["position"], p, rtol=1e-6, atol=1e-6)
        assert int(t["material"]) == m
    mats = sc.m_Material
    assert len(mats) == 2 and int(mats["albedo_tex"][0]) == 1 and int(mats["albedo_tex"][1]) == -1      # textures[0].source = image 1; default material
    assert np.allclose(mats["albedo"][1], 1.0)
    # the untransformed instance keeps the file's floats; the transformed quad's normals follow the rotation (z stays z here)
    assert np.array_equal(tris[3]["vertex"]["position"], quad_p[idx32[:3]]) and np.array_equal(tris[3]["vertex"]["normal"], quad_n[:3])
    assert np.allclose(tris[0]["vertex"]["normal"], [[0, 0, 1]] * 3, atol=1e-6) and np.array_equal(tris[0]["vertex"]["uv"], quad_t[idx32[:3]])
    # no NORMAL: geometric normal; no TEXCOORD_0: zeros
    assert np.allclose(np.abs(tris[5]["face_normal"]), [0, 0, 1]) and not tris[5]["vertex"]["uv"].any()
    assert [(int(m["primitives_offset"]), int(m["tris_count"])) for m in sc.m_Meshes] == [(0, 3), (3, 3)]
    # the reference's reading of the same file: node 0 has no mesh -> the error that stands for its crash
    with pytest.raises(drt.DrtError):
        drt.Scene().loadGLTFmodel(str(path))
    b = drt.BVHBuilder()
    b.buildIterative(sc)
    assert len(sc.m_BVHNodes) == 1


def test_strict_loading_of_the_reference_scenes_with_transforms():
    """suzanne_plane.gltf carries node translations the reference drops; room.glb's second mesh has no TEXCOORD_0 (the
    reference reads unrelated bytes as UVs); lightweightRTtest.glb nests nodes (the reference walks the flat node list)."""
    a, b = drt.Scene(), drt.Scene()
    a.loadGLTFmodel(scene_path("suzanne_plane_gltf"))
    b.loadGLTFmodel(scene_path("suzanne_plane_gltf"), strict=True)
    pa, pb = a.m_PrimitivesBuffer["vertex"]["position"], b.m_PrimitivesBuffer["vertex"]["position"]
    assert pa.shape == pb.shape and not np.allclose(pa, pb)
    a.loadGLTFmodel(scene_path("room"))
    b.loadGLTFmodel(scene_path("room"), strict=True)
    ta, tb = a.m_PrimitivesBuffer, b.m_PrimitivesBuffer
    assert np.array_equal(ta["vertex"]["position"], tb["vertex"]["position"])
    assert not np.array_equal(ta["vertex"]["uv"], tb["vertex"]["uv"])
    a.loadGLTFmodel(scene_path("lightweight_rt"))
    b.loadGLTFmodel(scene_path("lightweight_rt"), strict=True)
    ca = np.sort(a.m_PrimitivesBuffer["centroid"].view([("x", "<f4"), ("y", "<f4"), ("z", "<f4")]).reshape(-1), order=["x", "y", "z"])
    cb = np.sort(b.m_PrimitivesBuffer["centroid"].view([("x", "<f4"), ("y", "<f4"), ("z", "<f4")]).reshape(-1), order=["x", "y", "z"])
    assert len(ca) == len(cb) == 542


@pytest.mark.parametrize("name", ["cornell_box", "room", "suzanne_plane", "dense_monkey", "bvh_split_test", "cs16_dust"])
@pytest.mark.parametrize("leaf,bins", [(20, 8), (6, 8), (12, 4)])
def test_recursive_build_numbers_the_nodes_like_the_reference_recursion(name, leaf, bins):
    """BVHBuilder::build (BVHBuilder.cu:100-173) makes the partitions of buildIterative but appends a node's two children after
    BOTH of their subtrees.  The product renumbers its iterative build; the oracle restates the recursion itself: the node arrays
    must be equal bit for bit, the triangle order equal to the iterative build's, and the tree the same tree."""
    a = drt.Scene(); a.loadGLTFmodel(scene_path(name))
    b = drt.Scene(); b.loadGLTFmodel(scene_path(name))
    bld = drt.BVHBuilder(); bld.m_TargetLeafPrimitivesCount, bld.m_BinCount = leaf, bins
    bld.buildIterative(a); bld.build(b)
    osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(leaf, bins, recursive=True)
    na, nb = a.m_BVHNodes, b.m_BVHNodes
    assert len(nb) == len(na) == len(osc.nodes)
    assert np.array_equal(a.m_PrimitivesBuffer.view(np.uint8), b.m_PrimitivesBuffer.view(np.uint8))
    for f in ("is_leaf", "bmin", "bmax", "child1", "child2", "prim_count", "prim_start"):
        assert np.array_equal(np.asarray(nb[f]), np.asarray(osc.nodes[f])), f
    b.validate()
    if len(nb) > 1:
        assert not np.array_equal(np.asarray(na["child1"]), np.asarray(nb["child1"])) or len(nb) == 3      # a different order, really
        assert nb[-1]["prim_start"] == -1 and na[-1]["prim_start"] == 0

    def shape(nodes, i):                                # the tree as nested tuples of leaf ranges: independent of the numbering
        n = nodes[i]
        return (int(n["prim_start"]), int(n["prim_count"])) if n["is_leaf"] else (shape(nodes, int(n["child1"])), shape(nodes, int(n["child2"])))
    import sys
    sys.setrecursionlimit(10000)
    assert shape(na, len(na) - 1) == shape(nb, len(nb) - 1)


def _assert_same_records(sc, osc, what):
    tris = sc.m_PrimitivesBuffer
    assert len(tris) == len(osc.tris) > 0, what
    assert np.array_equal(bits(tris["vertex"]["position"]), bits(osc.tris["p"])), what + ": positions"
    assert np.array_equal(bits(tris["vertex"]["normal"]), bits(osc.tris["n"])), what + ": normals"
    assert np.array_equal(bits(tris["vertex"]["uv"]), bits(osc.tris["uv"])), what + ": uvs"
    assert np.array_equal(bits(tris["centroid"]), bits(osc.tris["centroid"])), what + ": centroids"
    assert np.array_equal(bits(tris["face_normal"]), bits(osc.tris["face_n"])), what + ": face normals"
    assert np.array_equal(tris["material"], osc.tris["material"]), what + ": materials"
    mats = sc.m_Material
    assert len(mats) == osc.n_mats and np.array_equal(mats["albedo_tex"], osc.mats["albedo_tex"][: osc.n_mats]), what + ": texture indices"
    assert np.array_equal(bits(mats["albedo"]), bits(osc.mats["albedo"][: osc.n_mats]))
    assert [(int(m["primitives_offset"]), int(m["tris_count"])) for m in sc.m_Meshes] == list(osc.meshes), what + ": mesh ranges"


@pytest.mark.parametrize("name", ["scene_hier_test", "suzanne_plane_gltf", "room", "lightweight_rt", "cornell_box", "uv_texture_gltf", "multi_material"])
def test_strict_loading_matches_the_independent_strict_restatement(name):
    """DRT_LOAD_STRICT (the opt-in departure from Scene.cu:120-200: scene graph, node transforms, accessor offsets / strides /
    component types, optional indices / normals / uvs / materials, texture -> image) against oracle/gltf_flatten.py's
    strict mode -- a separate numpy restatement of the glTF 2.0 rules: the 128-byte triangle records, materials and mesh
    ranges must agree bit for bit.  suzanne_plane.gltf has node translations, lightweightRTtest.glb nested nodes, room.glb a
    mesh without TEXCOORD_0, sceneHierTest.glb (the reference's own hierarchy test scene) shared index accessors."""
    sc = drt.Scene()
    sc.loadGLTFmodel(scene_path(name), strict=True)
    osc = oracle.Scene.load_glb(scene_path(name), strict=True)
    _assert_same_records(sc, osc, name)


def test_strict_loading_of_a_file_that_breaks_every_assumption_matches_the_restatement(tmp_path):
    """The synthetic GLB of test_strict_loading_honours_what_the_reference_ignores (u32 indices behind an accessor byteOffset,
    interleaved vertices, TRS parent + matrix child, an empty node, an instanced mesh, a primitive with positions only,
    texture -> image indirection) through both strict loaders."""
    path = _write_assumption_breaking_glb(tmp_path)
    sc = drt.Scene()
    sc.loadGLTFmodel(str(path), strict=True)
    osc = oracle.Scene.load_glb(str(path), strict=True)
    _assert_same_records(sc, osc, "synthetic")


@pytest.mark.parametrize("field,value", [("count", -3), ("byteOffset", -8), ("count", 2 ** 62), ("byteOffset", 2 ** 62), ("count", 2 ** 33)])
def test_strict_loader_refuses_accessors_that_would_wrap_the_range_check(tmp_path, field, value):
    """A crafted file: a negative or enormous accessor count / byteOffset must be an error, not a wrapped size_t that slips past the
    bufferView range check and reads outside the buffer."""
    import json
    import struct
    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]).tobytes()
    acc = {"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}
    acc[field] = value
    gltf = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(pos)}], "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(pos)}],
            "accessors": [acc], "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}]}
    js = json.dumps(gltf).encode()
    js += b" " * ((-len(js)) % 4)
    blob = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(pos)) + struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(pos), b"BIN\0") + pos
    path = tmp_path / "crafted.glb"
    path.write_bytes(blob)
    with pytest.raises(drt.DrtError) as e:
        drt.Scene().loadGLTFmodel(str(path), strict=True)
    assert e.value.code in (drt.ERR_PARSE, drt.ERR_INVALID, drt.ERR_UNSUPPORTED), e.value
