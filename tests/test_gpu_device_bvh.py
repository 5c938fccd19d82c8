"""BVH built on the device (PT_OPT_DEVICE_BVH) against the host builder's tree and the oracle: the tree differs, so
what must agree is the closest distance of every ray, bit for bit, and the primitive wherever no other primitive is
hit at exactly that distance."""
import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.scene_src import CameraSrc, SceneSource

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo, hi, towards=None):
    rng = np.random.default_rng(seed)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = rng.uniform(lo, hi, size=(n, 3))
    if towards is None:
        r[:, 3:6] = rng.normal(size=(n, 3))
    else:
        r[:, 3:6] = np.asarray(towards, np.float32) + rng.normal(scale=1.5, size=(n, 3)) - r[:, :3]
    r[:, 6], r[:, 7] = 0.001, np.inf
    return r


def _pair(src, device_from=8):
    from pathtracer_amd.core import HostScene, Renderer
    host = HostScene(src)
    dev = HostScene(src, device_bvh_from=device_from)
    return host, Renderer(host, device=0), dev, Renderer(dev, device=0)


def _original_tri_ids(host, renderer, hits):
    """Insertion index (within the world's triangle buffer) of the triangle each hit names, -1 for other hits."""
    from pathtracer_amd.core import PT_PRIM_TRI, host_lib
    import ctypes as C
    buf = host.set_desc(0).tris
    perm = np.zeros(buf.prim_count, np.uint32)
    host_lib().pth_perm(host.h, 0, PT_PRIM_TRI, perm.ctypes.data_as(C.c_void_p))
    order = renderer.triangle_order()
    out = np.full(hits.shape[0], -1, np.int64)
    is_tri = (hits["prim_type"] == PT_PRIM_TRI) & (hits["inst_id"] < 0)
    g = hits["prim_id"][is_tri].astype(np.int64)
    assert np.all((g >= buf.prim_first) & (g < buf.prim_first + buf.prim_count))
    uploaded = order[g].astype(np.int64) - buf.prim_first  # position in the arrays the host layer handed over
    out[is_tri] = perm[uploaded]
    return out


def test_distances_bit_equal_and_primitives_equal_up_to_ties():
    from oracle.pyoracle import OracleScene
    src = scenes.knob(width=64, spp=16, detail=0.3)
    host, rh, dev, rd = _pair(src)
    assert dev.set_desc(0).tris.node_count == 0 and host.set_desc(0).tris.node_count > 0
    n_tris = host.set_desc(0).tris.prim_count
    assert n_tris > 5000
    rays = np.concatenate([_rays(60000, 1, -6, 6, towards=(0.0, 1.0, 0.0)), _rays(20000, 2, -3, 3)])
    a, b = rh.trace(rays), rd.trace(rays)
    ids, tuv = OracleScene(src).trace(rays)
    assert np.array_equal(a["t"].view(np.uint32), tuv[:, 0].view(np.uint32))
    assert np.array_equal(b["t"].view(np.uint32), a["t"].view(np.uint32))
    assert np.array_equal(a["prim_type"], b["prim_type"])
    hit = a["prim_type"] >= 0
    assert hit.mean() > 0.3
    ia, ib = _original_tri_ids(host, rh, a), _original_tri_ids(dev, rd, b)
    differ = ia != ib
    # a different winner needs two triangles at exactly the same distance (a shared edge or vertex)
    assert differ.sum() <= 1e-3 * hit.sum(), f"{differ.sum()} of {hit.sum()} hits name another triangle"
    order = rd.triangle_order()
    assert np.array_equal(np.sort(order), np.arange(order.size, dtype=np.uint32))  # a permutation
    assert not np.array_equal(order, np.arange(order.size, dtype=np.uint32))


def test_rendered_image_equal_up_to_ties():
    src = scenes.knob(width=48, spp=16, detail=0.3)
    _, rh, _, rd = _pair(src)
    rh.render(0, 16, seed=3)
    rd.render(0, 16, seed=3)
    a, b = rh.read_accum()[0], rd.read_accum()[0]
    same = np.all(a.view(np.uint32).reshape(-1, 3) == b.view(np.uint32).reshape(-1, 3), axis=1)
    assert same.mean() > 0.995, f"{(~same).sum()} of {same.size} pixels differ"
    assert abs(float(a.mean()) - float(b.mean())) <= 1e-3 * float(a.mean())


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 9, 64, 257])
def test_small_buffers_and_duplicate_centres(n):
    """Few triangles (leaf-only root, smallest trees) and many triangles with one and the same centre (equal keys)."""
    from oracle.pyoracle import OracleScene
    rng = np.random.default_rng(n)
    pos = rng.uniform(-1, 1, size=(n, 3, 3)).astype(np.float32)
    if n >= 9:  # a fan of copies: identical boxes and centres
        pos[: n // 2] = pos[0]
    s = SceneSource()
    s.name = f"tris{n}"
    m = s.lambertian((0.5, 0.5, 0.5))
    emit = s.diffuse_light((1.0, 1.0, 1.0))
    s.objects.add_triangles(pos, None, None, m)
    s.objects.add_sphere((0.0, 5.0, 0.0), 0.5, emit)
    s.lights.add_sphere((0.0, 5.0, 0.0), 0.5, emit)
    s.objects.build_bvh = True
    s.camera = CameraSrc(1.0, 16, 40.0, (0.0, 0.0, 4.0), (0.0, 0.0, 0.0), 1.0, 0.0, 4, 4, (0.2, 0.2, 0.2))
    host, rh, dev, rd = _pair(s, device_from=1)
    rays = _rays(4000, n, -2, 2, towards=(0.0, 0.0, 0.0))
    a, b = rh.trace(rays), rd.trace(rays)
    _, tuv = OracleScene(s).trace(rays)
    assert np.array_equal(a["t"].view(np.uint32), tuv[:, 0].view(np.uint32))
    assert np.array_equal(b["t"].view(np.uint32), a["t"].view(np.uint32))
    assert np.array_equal(a["prim_type"], b["prim_type"])
    if n < 9:  # no exact ties by construction
        assert np.array_equal(_original_tri_ids(host, rh, a), _original_tri_ids(dev, rd, b))


def _all_tris(h):
    from pathtracer_amd.core import pt_buffer_desc
    return pt_buffer_desc(0, h.desc.n_tris, 0, 0)


def _tri_first(h, hits):
    """First scene-wide triangle index of the buffer each hit lies in (world set or the instance's child set)."""
    first = np.full(hits.shape[0], h.set_desc(0).tris.prim_first, np.int64)
    insts = h.set_desc(0).instances
    for k in range(insts.prim_count):
        child = h.desc.instances[insts.prim_first + k].child_set
        first[hits["inst_id"] == k] = h.set_desc(child).tris.prim_first
    return first


def test_triangle_buffers_inside_instances_and_in_the_light_list():
    """Every triangle buffer above the threshold gets a device tree, the children of instances included; the light
    list is never traversed, so its triangles keep their order (LightBuffer::sample picks them by index)."""
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY
    src = scenes.small_mixed(48, 16)
    rng = np.random.default_rng(3)
    c = rng.uniform(-3, 3, (40, 1, 3)) + np.array([0.0, 7.0, 0.0])
    lights = (c + rng.normal(scale=0.2, size=(40, 3, 3))).astype(np.float32)   # 40 small emissive triangles near the ceiling
    emit = src.diffuse_light_texture(src.solid((4.0, 4.0, 4.0)))
    src.objects.add_triangles(lights, None, None, emit)
    src.lights.add_triangles(lights, None, None, emit)
    host, rh, dev, rd = _pair(src, device_from=8)
    n_sets = host.desc.n_sets
    assert n_sets > 2
    built = [k for k in range(n_sets) if host.set_desc(k).tris.node_count > 0]
    assert any(k >= 2 for k in built)                                            # a child of an instance holds a triangle tree
    for k in built:
        assert dev.set_desc(k).tris.node_count == 0                              # left to the device
    rays = np.concatenate([_rays(40000, 5, -6, 6, towards=(0.0, 2.0, 0.0)), _rays(10000, 6, -5, 5)])
    a, b = rh.trace(rays), rd.trace(rays)
    _, tuv = OracleScene(src).trace(rays)
    assert np.array_equal(a["t"].view(np.uint32), tuv[:, 0].view(np.uint32))
    assert np.array_equal(b["t"].view(np.uint32), a["t"].view(np.uint32))
    assert np.array_equal(a["prim_type"], b["prim_type"]) and np.array_equal(a["inst_id"], b["inst_id"])
    # pt_read_triangle_order is a permutation of the uploaded triangles, also for the buffers of instanced sets (whose
    # slices do not start at 0), and maps every hit back to the uploaded triangle the host-built tree names
    order = rd.triangle_order()
    assert np.array_equal(np.sort(order), np.arange(dev.desc.n_tris, dtype=np.uint32))
    assert np.array_equal(rh.triangle_order(), np.arange(host.desc.n_tris, dtype=np.uint32))
    tri = a["prim_type"] == 0
    assert (tri & (a["inst_id"] >= 0)).sum() > 100 and (tri & (a["inst_id"] < 0)).sum() > 100
    first_h, first_d = _tri_first(host, a), _tri_first(dev, b)
    up_h = (first_h + a["prim_id"])[tri]                       # host build: positions are the uploaded ones
    up_d = order[(first_d + b["prim_id"])[tri]]
    th = host.triangles(_all_tris(host))[up_h]
    td = dev.triangles(_all_tris(dev))[up_d]
    same_tri = np.all(th == td, axis=1)
    assert same_tri.mean() > 0.999, "hits inside device-built trees must map back to the triangles the host-built trees hit (exact ties aside)"
    # light sampling sees the same lights in the same order: same pdf for every ray, same image up to ties
    assert np.array_equal(rh.light_pdf(rays).view(np.uint32), rd.light_pdf(rays).view(np.uint32))
    for r in (rh, rd):
        r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, 0.5)
        r.render(0, 16, seed=3)
    x, y = rh.read_accum()[0], rd.read_accum()[0]
    same = np.all(np.nan_to_num(x).view(np.uint32).reshape(-1, 3) == np.nan_to_num(y).view(np.uint32).reshape(-1, 3), axis=1)
    assert same.mean() > 0.99, f"{(~same).sum()} of {same.size} pixels differ"


@pytest.mark.parametrize("bsdf_probability", [1.0, 0.5])
def test_device_built_scene_renders_the_same_bits_however_early_batches_are_finished(bsdf_probability):
    # PT_OPT_BATCH_TICKS on a scene whose triangle tree was built on the device: what is left of a batch is finished
    # by one lane per path, which must walk that tree in the order the ray-drawing search walks it (1: never finished
    # early, 2: after two launches, 0: learnt); with BSDF_SAMPLE_PROBABILITY < 1 paths outlive max_depth as well
    from pathtracer_amd.core import PT_OPT_BATCH_TICKS, PT_OPT_BSDF_SAMPLE_PROBABILITY, PT_OPT_ROLLING, HostScene, Renderer
    src = scenes.knob(width=40, spp=36, max_depth=8, detail=0.25)
    dev = HostScene(src, device_bvh_from=8)
    assert dev.set_desc(0).tris.node_count == 0 and dev.set_desc(0).tris.prim_count > 3000
    ref = None
    for rolling, ticks in ((0, 1), (1 << 22, 1), (1 << 22, 2), (1 << 22, 3), (1 << 22, 0), (3 * 40 * 40, 2)):
        r = Renderer(dev, device=0)
        r.set_option(PT_OPT_ROLLING, rolling)
        r.set_option(PT_OPT_BATCH_TICKS, ticks)
        r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, bsdf_probability)
        r.render(0, 20, seed=5)
        r.sync()
        r.render(20, 36, seed=5)
        acc, n = r.read_accum()
        assert n == 36
        if ref is None:
            ref = acc
        else:
            assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"rolling={rolling} ticks={ticks}"
        r.close()
