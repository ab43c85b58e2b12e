"""The GPU BVH build (drt_scene_build_bvh_device, kernel_bvh_build.hip) against the host restatement of
BVHBuilder.cu (drt_scene_build_bvh): same nodes in the same order, same triangle order, for every shipped scene and
for synthetic soups, over several (leaf, bins) settings -- and the same image afterwards.  The oracle's own builder
(oracle/drt_oracle.c) is the third voice on the small cases."""
import numpy as np
import pytest

import oracle
from tests.scenes import SCENES, scene_path

drt = pytest.importorskip("dustraytracer_amd")
pytestmark = pytest.mark.gpu

INT_FIELDS = ("is_leaf", "child1", "child2", "prim_count", "prim_start")


def _build(scene_factory, leaf, bins, device):
    sc = scene_factory()
    b = drt.BVHBuilder()
    b.m_TargetLeafPrimitivesCount, b.m_BinCount, b.m_BuildDevice = leaf, bins, device
    b.buildIterative(sc)
    return sc, b


def _assert_same_tree(host, dev, what):
    hn, dn = host.m_BVHNodes, dev.m_BVHNodes
    assert len(hn) == len(dn), what
    for f in INT_FIELDS:
        assert np.array_equal(hn[f], dn[f]), (what, f)
    # bounds: equal as floats (a zero bound may carry the other sign, see kernel_bvh_build.hip)
    assert np.array_equal(hn["bmin"], dn["bmin"]) and np.array_equal(hn["bmax"], dn["bmax"]), what
    assert host.m_PrimitivesBuffer.tobytes() == dev.m_PrimitivesBuffer.tobytes(), what       # triangle order, byte for byte
    return hn.tobytes() == dn.tobytes()


def _file_scene(name):
    def make():
        sc = drt.Scene()
        sc.loadGLTFmodel(scene_path(name))
        return sc
    return make


def _soup(n, seed, clustered=False):
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-10, 10, (n, 1, 3))
    size = 0.2
    if clustered:                                   # half of the triangles in a tight cluster: deep, unbalanced tree
        centre[: n // 2] = rng.normal(0, 0.5, (n // 2, 1, 3)) + 3.0
        size = 0.02
    pos = (centre + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)
    nrm = np.tile(np.array([0, 1, 0], np.float32), (n, 3, 1))
    uv = np.zeros((n, 3, 2), np.float32)
    mat = np.zeros(n, np.int32)

    def make():
        sc = drt.Scene()
        sc.addMaterial((0.8, 0.8, 0.8), -1)
        sc.setGeometry(pos, nrm, uv, mat)
        return sc
    return make


@pytest.mark.parametrize("leaf,bins", [(20, 8), (6, 8), (1, 4), (3, 16), (40, 2)])
def test_device_build_equals_host_build_on_every_scene(leaf, bins):
    identical_bytes = refused = 0
    for name in sorted(SCENES):
        make = _file_scene(name)
        try:
            host, _ = _build(make, leaf, bins, -1)
        except drt.DrtError as e:                      # coincident triangles and small leaves: the reference would never return
            assert e.code == drt.ERR_BVH
            with pytest.raises(drt.DrtError) as e2:
                _build(make, leaf, bins, 0)
            assert e2.value.code == drt.ERR_BVH, name
            refused += 1
            continue
        dev, _ = _build(make, leaf, bins, 0)
        identical_bytes += _assert_same_tree(host, dev, (name, leaf, bins))
    assert identical_bytes + refused >= len(SCENES) - 2          # the sign of a zero bound is the only freedom, and it is rare
    assert refused < len(SCENES)


@pytest.mark.parametrize("n,seed,clustered,leaf,bins", [(1000, 1, False, 4, 8), (70_000, 2, False, 20, 8), (70_000, 3, True, 12, 8),
                                                       (300_000, 4, False, 20, 8), (513, 5, False, 1, 3)])
def test_device_build_equals_host_build_on_soups(n, seed, clustered, leaf, bins):
    make = _soup(n, seed, clustered)
    host, _ = _build(make, leaf, bins, -1)
    dev, b = _build(make, leaf, bins, 0)
    _assert_same_tree(host, dev, (n, seed, leaf, bins))
    assert b.m_LastBuildDeviceMs > 0


def test_device_build_agrees_with_the_oracle_builder():
    """Not only with our own host code: the oracle's C restatement of BVHBuilder.cu gives the same nodes."""
    for name in ("cornell_box", "room", "suzanne_plane"):
        dev, _ = _build(_file_scene(name), 20, 8, 0)
        osc = oracle.Scene.load_glb(scene_path(name)).build_bvh(20, 8)
        on = osc.nodes
        dn = dev.m_BVHNodes
        assert len(on) == len(dn)
        for f_dev, f_or in (("child1", "child1"), ("child2", "child2"), ("prim_count", "prim_count"), ("prim_start", "prim_start")):
            assert np.array_equal(dn[f_dev], on[f_or]), (name, f_dev)
        assert np.array_equal(dn["bmin"], on["bmin"]) and np.array_equal(dn["bmax"], on["bmax"])


def test_one_leaf_and_degenerate_inputs():
    make = _soup(10, 7)
    host, _ = _build(make, 20, 8, -1)
    dev, _ = _build(make, 20, 8, 0)                       # a single leaf: nothing to split
    assert host.m_BVHNodes.tobytes() == dev.m_BVHNodes.tobytes() and len(dev.m_BVHNodes) == 1
    # 50 copies of one triangle: every plane leaves a side empty; the reference loops forever, both builders refuse
    pos = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (50, 1, 1))

    def same():
        sc = drt.Scene()
        sc.addMaterial((1, 1, 1), -1)
        sc.setGeometry(pos, np.zeros_like(pos), np.zeros((50, 3, 2), np.float32), np.zeros(50, np.int32))
        return sc
    for device in (-1, 0):
        with pytest.raises(drt.DrtError) as e:
            _build(same, 4, 8, device)
        assert e.value.code == drt.ERR_BVH
    with pytest.raises(drt.DrtError) as e:
        _build(make, 2, 1, 0)
    assert e.value.code == drt.ERR_INVALID


def test_image_after_a_device_build_is_the_same_image():
    cam = drt.Camera((3.6, 1.25, 0.0))
    cam.m_Forward_dir = np.array((-1, 0, 0), np.float32)
    images = []
    for device in (-1, 0):
        sc, _ = _build(_file_scene("cornell_box"), 20, 8, device)
        r = drt.Renderer(0)
        r.m_RendererSettings = drt.RendererSettings(ray_bounce_limit=4, max_samples=100)
        r.ResizeBuffer(160, 90)
        r.RenderBatch(cam, sc, 4)
        images.append(r.GetRenderTargetImage())
    assert np.array_equal(images[0], images[1])
