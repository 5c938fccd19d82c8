"""GPU parity tests proper: the HIP path through the C ABI against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import scenes
from tests.util import bits_equal, image_errors, samples_agree

pytestmark = pytest.mark.gpu


def _pair(src):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import HostScene, Renderer
    return OracleScene(src), Renderer(HostScene(src))


def _random_rays(rng, n, lo, hi, special=True):
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    if special:  # axis-aligned and zero-component directions exercise the inf / NaN arms of the slab test
        k = n // 8
        d[:k, 0] = 0.0
        d[k:2 * k, 1] = 0.0
        d[2 * k:3 * k, [0, 2]] = 0.0
        d[3 * k:4 * k] = np.sign(d[3 * k:4 * k]) * np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)]
        d[4 * k:5 * k, 2] = -0.0
    r = np.zeros((n, 8), np.float32)
    r[:, :3], r[:, 3:6], r[:, 6], r[:, 7] = o, d, 0.001, np.inf
    r[n // 2:, 7] = rng.uniform(0.5, 30.0, size=n - n // 2).astype(np.float32)  # bounded rays too
    return r


@pytest.mark.parametrize("name", ["small_mixed", "knob", "next_week_like", "interior", "lucy"])
def test_random_and_degenerate_rays_both_traversals(name):
    from pathtracer_amd.core import PT_OPT_TRAVERSAL
    src = {"small_mixed": lambda: scenes.small_mixed(32, 16), "knob": lambda: scenes.knob(32, 16),
           "next_week_like": lambda: scenes.next_week_like(32, 16), "interior": lambda: scenes.interior(32, 16, n_lights=16, columns=3),
           "lucy": lambda: scenes.lucy(32, 16, detail=0.5)}[name]()
    o, r = _pair(src)
    rng = np.random.default_rng(5)
    box = {"small_mixed": (-7, 9), "knob": (-3, 4), "next_week_like": (-300, 700), "interior": (0.5, 15), "lucy": (-8, 20)}[name]
    rays = _random_rays(rng, 60000, *box)
    # rays that start exactly on surfaces (what every bounce ray does)
    ids0, tuv0 = o.trace(rays)
    hit = ids0[:, 0] >= 0
    onsurf = rays[hit][:20000].copy()
    onsurf[:, :3] = onsurf[:, :3] + tuv0[hit][:20000, 0:1] * onsurf[:, 3:6]
    onsurf[:, 3:6] = rng.normal(size=(onsurf.shape[0], 3)).astype(np.float32)
    onsurf[:, 7] = np.inf
    rays = np.concatenate([rays, onsurf])
    ids, tuv = o.trace(rays)
    for mode in (0, 1, 2):  # 2: 64 consecutive rays searched together
        r.set_option(PT_OPT_TRAVERSAL, mode)
        h = r.trace(rays)
        assert np.array_equal(h["prim_type"], ids[:, 0]), mode
        assert np.array_equal(h["prim_id"], ids[:, 1]), mode
        assert np.array_equal(h["inst_id"], ids[:, 2]), mode
        assert bits_equal(h["t"], tuv[:, 0]), mode
        m = ids[:, 0] >= 0
        assert bits_equal(h["u"][m], tuv[m, 1]) and bits_equal(h["v"][m], tuv[m, 2])


def test_spheres_and_quads_tested_when_a_ray_is_handed_on_keep_the_reference_order():
    """A world without instances whose sphere and quad buffers are one leaf each: the ray-drawing kernel tests them after
    its loop, for all the rays that leave it together.  PrimitiveBuffer::hit (src/objects.rs:191-197) takes triangles,
    spheres, quads in that order and a later hit at the same distance replaces an earlier one: a quad lying in the plane
    of the triangles, a sphere touching it and one cutting it make such ties, and rays along the axes, rays that graze
    the buffers' boxes and rays that start on the surfaces must come out as the oracle's, id for id and bit for bit."""
    from pathtracer_amd.core import PT_OPT_TRAVERSAL
    from pathtracer_amd.scene_src import CameraSrc, SceneSource
    s = SceneSource()
    s.name = "tail"
    grey, red = s.lambertian((0.5, 0.5, 0.5)), s.lambertian((0.8, 0.2, 0.2))
    n = 24
    g = np.linspace(-3.0, 3.0, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(g, g, indexing="ij")
    P = np.stack([X, np.zeros_like(X), Z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    tris = np.concatenate([np.stack([a, b, c], axis=-2).reshape(-1, 3, 3), np.stack([a, c, d], axis=-2).reshape(-1, 3, 3)]).astype(np.float32)
    s.objects.add_triangles(tris, None, None, grey)
    s.objects.add_quad((-1.0, 0.0, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), red)      # in the triangles' plane
    s.objects.add_quad((-2.0, 0.5, -2.0), (0.0, 1.5, 0.0), (0.0, 0.0, 4.0), red)      # a wall
    s.objects.add_sphere((1.5, 0.5, 1.5), 0.5, red)                                   # touches the plane
    s.objects.add_sphere((-1.5, 0.25, 1.5), 0.5, red)                                 # cuts it
    s.objects.add_sphere((0.0, 2.0, 0.0), 0.25, red)
    s.lights.add_sphere((0.0, 8.0, 0.0), 1.0, s.diffuse_light((5.0, 5.0, 5.0)))
    s.objects.build_bvh = True
    s.camera = CameraSrc(1.0, 32, 40.0, (0.0, 4.0, 8.0), (0.0, 0.0, 0.0), 10.0, 0.0, 16, 8, (0.1, 0.1, 0.1))
    o, r = _pair(s)
    rng = np.random.default_rng(11)
    rays = _random_rays(rng, 40000, -3.5, 3.5)
    down = np.zeros((6000, 8), np.float32)  # straight down onto the plane, the quad in it, the touching sphere's pole
    down[:, 0], down[:, 2] = rng.uniform(-3.2, 3.2, 6000), rng.uniform(-3.2, 3.2, 6000)
    down[:2000, 0], down[:2000, 2] = rng.uniform(-1.0, 1.0, 2000), rng.uniform(-1.0, 1.0, 2000)
    down[2000:2200, 0], down[2000:2200, 2] = 1.5, 1.5
    down[:, 1], down[:, 4], down[:, 6], down[:, 7] = 3.0, -1.0, 0.001, np.inf
    rays = np.concatenate([rays, down])
    ids0, tuv0 = o.trace(rays)
    hit = ids0[:, 0] >= 0
    onsurf = rays[hit][:15000].copy()
    onsurf[:, :3] = onsurf[:, :3] + tuv0[hit][:15000, 0:1] * onsurf[:, 3:6]
    onsurf[:, 3:6] = rng.normal(size=(onsurf.shape[0], 3)).astype(np.float32)
    onsurf[:, 7] = np.inf
    rays = np.concatenate([rays, onsurf])
    ids, tuv = o.trace(rays)
    assert (ids[:, 0] == 0).sum() > 1000 and (ids[:, 0] == 1).sum() > 1000 and (ids[:, 0] == 2).sum() > 1000
    for mode in (0, 1, 2):
        r.set_option(PT_OPT_TRAVERSAL, mode)
        h = r.trace(rays)
        assert np.array_equal(h["prim_type"], ids[:, 0]), mode
        assert np.array_equal(h["prim_id"], ids[:, 1]), mode
        assert bits_equal(h["t"], tuv[:, 0]), mode
        m = ids[:, 0] >= 0
        assert bits_equal(h["u"][m], tuv[m, 1]) and bits_equal(h["v"][m], tuv[m, 2])
    # and rendered: the accumulator of the ray-drawing kernels equals that of the reference-order kernels bit for bit,
    # and per-path radiance agrees with the oracle's recursion
    accs = []
    for mode in (0, 1):
        r.set_option(PT_OPT_TRAVERSAL, mode)
        r.reset_accum()
        r.render(0, 9, seed=9)
        accs.append(r.read_accum()[0])
    assert bits_equal(accs[0], accs[1])
    ref = o.render(0, 4, seed=9, iterative=True, per_sample=True)["per_sample"]
    r.set_option(PT_OPT_TRAVERSAL, 0)
    samples_agree(np.nan_to_num(r.render_samples(0, 4, seed=9)), np.nan_to_num(ref), 0.999, "one-leaf spheres and quads")


def test_traversal_counters_equal_the_oracles_on_identical_rays():
    """With depth 1 only camera rays are traced, and those are bit-identical on both sides, so the
    device's reference-order counters must equal the oracle's exactly."""
    from pathtracer_amd.core import PT_OPT_COUNTERS
    for src in (scenes.knob(96, 16, max_depth=1), scenes.small_mixed(64, 16, max_depth=1), scenes.cornell(64, 16, max_depth=1)):
        o, r = _pair(src)
        r.set_option(PT_OPT_COUNTERS, 1)
        r.stats_reset()
        r.render(0, 3, seed=4)
        st = r.stats()
        ctr = np.zeros(5, np.uint64)
        for s in range(3):
            rays, _ = o.sample_rays(4, s)
            ctr += o.trace(rays, counters=True)[2]
        assert st["rays"] == int(ctr[4]) == 3 * r.npix
        assert (st["node_tests"], st["tri_tests"], st["sphere_tests"], st["quad_tests"]) == tuple(int(x) for x in ctr[:4]), src.name


@pytest.mark.parametrize("name,prob", [("cornell", 1.0), ("small_mixed", 1.0), ("small_mixed", 0.5), ("cornell", 0.25), ("lucy", 1.0), ("lucy", 0.5)])
def test_per_sample_radiance_with_and_without_light_sampling(name, prob):
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY
    src = {"cornell": lambda: scenes.cornell(48, 16), "small_mixed": lambda: scenes.small_mixed(48, 16),
           "lucy": lambda: scenes.lucy(64, 16, detail=0.5)}[name]()  # lucy.rs: 16:9, sky 1.5, sphere light of radiance 255
    o, r = _pair(src)
    o.set_bsdf_probability(prob)
    r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, prob)
    ref = np.nan_to_num(o.render(0, 6, seed=13, iterative=True, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 6, seed=13), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, "")
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)


@pytest.mark.parametrize("name", ["cornell", "small_mixed", "knob"])
def test_converged_image_within_twice_the_oracle_noise_floor(name):
    """Equal spp, different seeds, linear f32: RMSE and mean relative error of GPU-vs-oracle must stay within
    2x the oracle-vs-oracle (two seeds) figures."""
    src = {"cornell": lambda: scenes.cornell(64, 256), "small_mixed": lambda: scenes.small_mixed(48, 256),
           "knob": lambda: scenes.knob(64, 256)}[name]()
    o, r = _pair(src)
    n = 256
    a = o.render(0, n, seed=1)["accum"] / n
    b = o.render(0, n, seed=2)["accum"] / n
    r.render(0, n, seed=3)
    g, ns = r.read_accum()
    assert ns == n
    g = g / n
    floor_rmse, floor_mre = image_errors(a, b)
    rmse, mre = image_errors(g, a)
    assert rmse <= 2.0 * floor_rmse and mre <= 2.0 * floor_mre, (rmse, floor_rmse, mre, floor_mre)
    # and the means agree to within the Monte-Carlo error of the mean
    assert abs(g.mean() - a.mean()) <= 3.0 * max(abs(a.mean() - b.mean()), 0.01 * a.mean())


def test_accumulator_is_deterministic_additive_and_shardable():
    src = scenes.small_mixed(64, 64)
    _, r = _pair(src)
    r.render(0, 12, seed=8)
    a, n = r.read_accum()
    r.reset_accum()
    r.render(0, 12, seed=8)
    b, _ = r.read_accum()
    assert n == 12 and bits_equal(a, b)            # same seed -> same bits (fixed summation order)
    r.reset_accum()
    r.render(0, 5, seed=8)
    r.render(5, 12, seed=8)
    c, n2 = r.read_accum()
    assert n2 == 12 and bits_equal(a, c)           # passes accumulate in stratum order like `*pixel += c`
    from pathtracer_amd.core import PT_OPT_MAX_PATHS
    r.set_option(PT_OPT_MAX_PATHS, 3 * r.npix)     # different batching, same sums
    r.reset_accum()
    r.render(0, 12, seed=8)
    d, _ = r.read_accum()
    assert bits_equal(a, d)
    r.reset_accum()
    r.render(6, 12, seed=8)                        # a shard: differs from the first half
    e, n3 = r.read_accum()
    assert n3 == 6 and not bits_equal(a, e)


def test_tonemap_matches_reference_quantisation():
    from oracle import pyoracle as po
    src = scenes.small_mixed(64, 64)
    _, r = _pair(src)
    r.render(0, 16, seed=2)
    acc, n = r.read_accum()
    img = r.read_rgb8().reshape(-1, 3).astype(np.int32)
    ref = po.to_rgb8(acc, float(n)).astype(np.int32)
    # powf may differ in the last bit between libm and the device: a value on a quantisation edge can move by one
    assert np.abs(img - ref).max() <= 1 and (img != ref).mean() < 1e-3


def test_device_copy_round_trip_for_the_reduce():
    import torch
    src = scenes.cornell(32, 16)
    _, r = _pair(src)
    r.render(0, 4, seed=1)
    acc, n = r.read_accum()
    t = torch.empty(r.npix * 3, dtype=torch.float32, device="cuda")
    r.copy_accum_to_device(t.data_ptr())
    assert bits_equal(t.cpu().numpy().reshape(-1, 3), acc)
    t *= 2
    r.set_accum_from_device(t.data_ptr(), 2 * n)
    acc2, n2 = r.read_accum()
    assert n2 == 2 * n and bits_equal(acc2, acc * 2)


def test_error_codes_instead_of_crashes():
    from pathtracer_amd import core
    L = core.core_lib()
    h = C.c_void_p()
    assert L.pt_create(0, C.byref(h)) == 0
    assert L.pt_render(h, 0, 1, 0) == 5 and b"upload" in L.pt_last_error(h)            # PT_ERR_STATE
    hs = core.HostScene(scenes.cornell(16, 4))
    assert L.pt_upload_scene(h, C.byref(hs.desc)) == 0 and L.pt_set_camera(h, C.byref(hs.camera)) == 0
    assert L.pt_render(h, 0, 5, 0) == 1                                               # 4 strata only -> PT_ERR_INVALID
    assert L.pt_render(h, 3, 2, 0) == 1
    assert L.pt_set_option(h, 99, 1.0) == 1 and L.pt_set_option(h, core.PT_OPT_BSDF_SAMPLE_PROBABILITY, 1.5) == 1
    # a corrupted node array is refused at upload instead of hanging a kernel
    bad = core.pt_scene_desc.from_buffer_copy(hs.desc)
    nodes = (core.pt_node * hs.desc.n_nodes)(*[hs.desc.nodes[i] for i in range(hs.desc.n_nodes)])
    nodes[0].a = 0
    bad.nodes = C.cast(nodes, C.POINTER(core.pt_node))
    assert L.pt_upload_scene(h, C.byref(bad)) == 1
    assert L.pt_render(h, 0, 4, 0) == 0                                               # the earlier scene is still loaded
    assert L.pt_create(12345, C.byref(C.c_void_p())) == 1
    L.pt_destroy(h)


def test_full_size_properties_knob_768():
    """The benchmark configuration at its full size (768x768, depth 50): size-independent properties only."""
    src = scenes.knob(768, 10240, 50)
    from pathtracer_amd.core import HostScene, Renderer
    r = Renderer(HostScene(src))
    assert r.sqrt_spp == 101
    r.render(0, 4, seed=1)
    a, n = r.read_accum()
    r.reset_accum()
    r.render(0, 2, seed=1)
    r.render(2, 4, seed=1)
    b, _ = r.read_accum()
    assert n == 4 and bits_equal(a, b) and np.isfinite(a).all() and a.min() >= 0.0 and a.max() > 0.0
    st = r.stats()
    assert st["paths"] == 2 * 4 * 768 * 768 and st["rays"] >= st["paths"]
    img = r.read_rgb8()
    assert img.shape == (768, 768, 3) and img.mean() > 1.0


def test_vertices_inside_a_sphere_light_poison_their_paths_as_in_the_reference():
    """LightBuffer::pdf is NaN for an origin inside a sphere light (src/objects/sphere.rs:109-120) and the reference
    multiplies it by zero (src/camera.rs:339-340), which kills the path.  The device skips the sum over the lights
    when its weight is zero and must still reproduce that."""
    from pathtracer_amd.scene_src import CameraSrc, SceneSource
    s = SceneSource()
    s.name = "inside_a_light"
    emit = s.diffuse_light((3.0, 3.0, 3.0))
    grey = s.lambertian((0.7, 0.7, 0.7))
    o = s.objects
    o.add_sphere((0.0, 0.0, 0.0), 6.0, emit)          # encloses the two grey spheres: seen from inside it emits nothing
    s.lights.add_sphere((0.0, 0.0, 0.0), 6.0, emit)
    o.add_quad((-1, 3, -1), (2, 0, 0), (0, 0, 2), emit)
    s.lights.add_quad((-1, 3, -1), (2, 0, 0), (0, 0, 2), emit)
    o.add_sphere((-1.2, 0.0, 0.0), 1.0, grey)
    o.add_sphere((1.2, 0.0, 0.0), 1.0, s.disney_diffuse((0.3, 0.3, 0.8), 1.0, 0.5))
    o.add_sphere((12.0, 0.0, 0.0), 2.0, grey)         # outside the light sphere: ordinary shading
    o.add_quad((8, -2, -6), (10, 0, 0), (0, 0, 12), grey)
    o.build_bvh = True
    s.camera = CameraSrc(1.0, 48, 60.0, (4.0, 2.0, -16.0), (4.0, 0.0, 0.0), 10.0, 0.0, 16, 8, (0.2, 0.2, 0.2))
    o_, r = _pair(s)
    ref = o_.render(0, 6, seed=3, iterative=True, per_sample=True)["per_sample"]
    got = r.render_samples(0, 6, seed=3)
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    ref, got = np.nan_to_num(ref), np.nan_to_num(got)
    samples_agree(got, ref, 0.999, "")
    assert ref.sum() > 0


@pytest.mark.parametrize("name,depth,prob", [("cornell", 3, 0.5), ("small_mixed", 2, 0.25), ("cornell", 1, 0.5)])
def test_paths_that_outlive_max_depth_under_light_sampling(name, depth, prob):
    """A vertex that samples a light resets the depth left to 2 (src/camera.rs:332), so with a small max_depth and
    BSDF_SAMPLE_PROBABILITY < 1 many paths go on beyond max_depth bounces.  The recursion of the reference follows
    them; so must the device, in both schedules, with the same bits in both and from run to run."""
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY, PT_OPT_ROLLING
    src = {"cornell": lambda: scenes.cornell(48, 36, max_depth=depth), "small_mixed": lambda: scenes.small_mixed(48, 36, max_depth=depth)}[name]()
    o, r = _pair(src)
    o.set_bsdf_probability(prob)
    r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, prob)
    ref = np.nan_to_num(o.render(0, 6, seed=13, iterative=False, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 6, seed=13), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, "")
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)
    accs = []
    rays = []
    for rolling in (0, 48 * 48 * 5, 48 * 48 * 5, 1 << 22):
        r.set_option(PT_OPT_ROLLING, rolling)
        r.reset_accum()
        r.stats_reset()
        r.render(0, 20, seed=5)
        r.render(20, 36, seed=5)
        accs.append(r.read_accum()[0].copy())
        rays.append(r.stats()["rays"])
    assert rays[0] == rays[1] == rays[2] == rays[3], rays
    assert rays[0] > (depth + 0.5) * 36 * 48 * 48 * 0.2  # many paths are longer than max_depth
    for a in accs[1:]:
        assert np.array_equal(a.view(np.uint32), accs[0].view(np.uint32))
