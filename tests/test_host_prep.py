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
