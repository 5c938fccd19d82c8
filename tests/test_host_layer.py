"""The host scene layer (libpthost.so) against the oracle: every derived quantity must agree in bits."""
import ctypes as C

import numpy as np
import pytest

from oracle.pyoracle import OracleScene
from pathtracer_amd import scenes
from pathtracer_amd.core import PT_NODE_LEAF, HostScene, PtError, host_lib
from pathtracer_amd.scene_src import CameraSrc, InstanceSrc, PrimSet, SceneSource, new_box


def _compare(src):
    o, h = OracleScene(src), HostScene(src)
    assert np.array_equal(o.camera_constants().view(np.uint32), h.camera_constants().view(np.uint32))
    for set_ in (0, 1):
        sd = h.set_desc(set_)
        for type_, buf in enumerate((sd.tris, sd.spheres, sd.quads, sd.instances)):
            assert o.buffer_len(set_, type_) == buf.prim_count
            assert np.array_equal(o.perm(set_, type_), h.perm(set_, type_))
            bb, meta = o.bvh(set_, type_)
            nodes = h.nodes(buf)
            assert len(nodes) == len(bb)
            if len(nodes):
                hb = np.concatenate([nodes["bmin"], nodes["bmax"]], axis=1)
                assert np.array_equal(hb.view(np.uint32), bb.view(np.uint32))
                assert np.array_equal(((nodes["b"] & PT_NODE_LEAF) != 0).astype(np.uint32), meta[:, 0])
                assert np.array_equal(nodes["a"], meta[:, 1])
                assert np.array_equal(nodes["b"] & 0x7FFFFFFF, meta[:, 2])
        assert np.array_equal(o.quads(set_).view(np.uint32), h.quads(sd.quads).view(np.uint32))
        assert np.array_equal(o.triangles(set_).view(np.uint32), h.triangles(sd.tris).view(np.uint32))
        assert np.array_equal(o.instances(set_).view(np.uint32), h.instances(sd.instances).view(np.uint32))
    return o, h


@pytest.mark.parametrize("name", ["cornell", "small_mixed", "knob", "next_week_like", "interior", "lucy"])
def test_builder_and_derived_quantities_match_oracle(name):
    src = {"cornell": lambda: scenes.cornell(64, 16), "small_mixed": lambda: scenes.small_mixed(32, 16),
           "knob": lambda: scenes.knob(32, 16), "next_week_like": lambda: scenes.next_week_like(32, 16),
           "interior": lambda: scenes.interior(32, 16, n_lights=12, columns=2), "lucy": lambda: scenes.lucy(32, 16, detail=0.5)}[name]()
    _compare(src)


def test_all_instance_transform_ops():
    s = SceneSource()
    m = s.lambertian((.5, .5, .5))
    inst = InstanceSrc(new_box((0, 0, 0), (1, 2, 3), m))
    inst.rotate(0.3, (0.2, 0.9, 0.1)).rotate_x(0.1).rotate_y(-0.2).rotate_z(0.7).translate((1, 2, 3)).translate_x(-4).translate_y(5)
    inst.translate_z(0.5).scale((1.5, 0.5, 2.0)).scale_x(1.1).scale_y(0.9).scale_z(3.0).scale_uniform(0.25)
    s.objects.add_instance(inst)
    s.objects.add_instance(InstanceSrc(new_box((0, 0, 0), (1, 1, 1), m)))  # identity
    s.objects.build_bvh = True
    s.camera = CameraSrc(16 / 9, 50, 33.0, (3, 4, 5), (0, 0.5, -1), 2.5, 1.5, 50, 7, (0.1, 0.2, 0.3))
    o, h = _compare(s)
    assert (h.width, h.height) == (50, 28) and h.sqrt_spp == 7


def test_degenerate_builder_inputs():
    s = SceneSource()
    m = s.lambertian((.5, .5, .5))
    for k in range(40):  # identical centroids: one leaf of 40 (src/bvh.rs:339-341)
        s.objects.add_sphere((1, 2, 3), 0.5 + 0.01 * k, m)
    for k in range(9):   # collinear quads with tiny extents
        s.objects.add_quad((k * 1e-5, 0, 0), (1e-5, 0, 0), (0, 1e-5, 0), m)
    tri = np.zeros((5, 3, 3), np.float32)  # degenerate (zero-area) triangles at distinct places
    tri[:, :, 0] = np.arange(5)[:, None]
    s.objects.add_triangles(tri, None, None, m)
    s.objects.build_bvh = True
    s.camera = CameraSrc(1.0, 8, 40.0, (0, 0, -5), (0, 0, 0), 1.0, 0.0, 4, 4, (0, 0, 0))
    o, h = _compare(s)
    _, meta = o.bvh(0, 1)
    assert meta.tolist() == [[1, 0, 40]]


def test_empty_scene_and_errors():
    s = SceneSource()
    s.camera = CameraSrc(1.0, 8, 40.0, (0, 0, -5), (0, 0, 0), 1.0, 0.0, 4, 4, (0, 0, 0))
    h = HostScene(s)
    assert h.desc.n_tris == 0 and h.desc.n_sets == 2 and h.desc.n_nodes == 0
    # bad material index -> error text, no crash
    s2 = SceneSource()
    s2.objects.add_sphere((0, 0, 0), 1.0, 3)
    s2.camera = s.camera
    with pytest.raises(PtError, match="material index"):
        HostScene(s2)
    err = C.create_string_buffer(64)
    assert host_lib().pth_scene_build(None, None, err, 64) is None and b"NULL" in err.value


def test_deep_tree_is_rejected_like_the_reference_panics():
    # centroids at 2^-k make every split peel one primitive off: depth grows linearly past 63
    s = SceneSource()
    m = s.lambertian((.5, .5, .5))
    for k in range(90):
        s.objects.add_sphere((float(2.0 ** (-k)) * 1e6, 0, 0), 1e-9, m)
    s.objects.build_bvh = True
    s.camera = CameraSrc(1.0, 8, 40.0, (0, 0, -5), (0, 0, 0), 1.0, 0.0, 4, 4, (0, 0, 0))
    try:
        o = OracleScene(s)
        oracle_ok = True
    except RuntimeError:
        oracle_ok = False
    if oracle_ok:
        _compare(s)  # the tree happened to stay within 63 levels: both must still agree
    else:
        with pytest.raises(PtError, match="63"):
            HostScene(s)


def test_builder_gives_the_same_tree_for_any_number_of_threads(monkeypatch):
    """Large subtrees are grown on their own threads (the small test scenes never get there): same nodes, same order."""
    import ctypes as C
    from pathtracer_amd.core import PT_PRIM_TRI, HostScene, host_lib
    src = scenes.knob(32, 4)  # 184 k triangles
    got = []
    for threads in ("1", "8"):
        monkeypatch.setenv("PTH_BUILD_THREADS", threads)
        h = HostScene(src)
        buf = h.set_desc(0).tris
        perm = np.zeros(buf.prim_count, np.uint32)
        host_lib().pth_perm(h.h, 0, PT_PRIM_TRI, perm.ctypes.data_as(C.c_void_p))
        got.append((h.nodes(buf).tobytes(), perm.tobytes(), h.triangles(buf).tobytes()))
    assert got[0] == got[1]
