"""Smallest and emptiest inputs through the C ABI: scenes without primitives, of one primitive type only, one-pixel
images, one stratum, depth 0 and 1, empty stratum ranges, empty ray lists.  Each against the oracle where there
is something to compare (the reference's tests hold no such cases; these are the edges of its own loops:
src/camera.rs:176-198 over strata and pixels, src/objects.rs:191-197 over the four buffers)."""
import numpy as np
import pytest

from pathtracer_amd.scene_src import CameraSrc, SceneSource
from tests.util import bits_equal, samples_agree

pytestmark = pytest.mark.gpu


def _camera(width=8, spp=4, depth=4, background=(0.25, 0.5, 0.75)):
    return CameraSrc(1.0, width, 40.0, (0.0, 1.0, -6.0), (0.0, 0.0, 0.0), 1.0, 0.0, spp, depth, background)


def _pair(src):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import HostScene, Renderer
    return OracleScene(src), Renderer(HostScene(src))


def _same_image(src, strata, seed=3):
    o, r = _pair(src)
    ref = np.nan_to_num(o.render(0, strata, seed=seed, iterative=True, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, strata, seed=seed), nan=0.0, posinf=0.0, neginf=0.0)
    assert got.shape == ref.shape
    samples_agree(got, ref, 0.999, "")
    # the accumulating entry point (launches shared between batches) adds up exactly these samples, in stratum
    # order, without the ones that are NaN (src/camera.rs:199, :212-217)
    raw = r.render_samples(0, strata, seed=seed)
    want = np.zeros((r.npix, 3), np.float32)
    for k in range(strata):
        keep = ~np.isnan(raw[k]).any(axis=-1)
        want[keep] = want[keep] + raw[k][keep]
    r.reset_accum()
    r.render(0, strata, seed=seed)
    acc, n = r.read_accum()
    assert n == strata and bits_equal(acc, want)
    r.reset_accum()
    return o, r


def test_a_scene_with_nothing_in_it_shows_the_background():
    s = SceneSource()
    s.name = "void"
    s.camera = _camera()
    o, r = _pair(s)
    r.render(0, 4, seed=1)
    acc, n = r.read_accum()
    assert n == 4 and np.allclose(acc, 4 * np.array([0.25, 0.5, 0.75], np.float32))
    assert bits_equal(acc, o.render(0, 4, seed=1)["accum"])
    rays = r.generate_rays(0, seed=1)
    assert (r.trace(rays)["prim_type"] == -1).all()
    assert r.stats()["rays"] == 4 * r.npix


@pytest.mark.parametrize("kind", ["tris", "spheres", "quads", "instances"])
@pytest.mark.parametrize("bvh", [False, True])
@pytest.mark.parametrize("lit", [False, True])
def test_one_primitive_type_alone(kind, bvh, lit):
    """Without a light the reference's LightBuffer::pdf is 0 / 0 (src/objects.rs:260, :277) and every diffuse vertex ends
    its path in black (src/camera.rs:355-357); with one, of the same primitive type, paths go on."""
    from pathtracer_amd.scene_src import InstanceSrc, new_box
    s = SceneSource()
    s.name = "only_" + kind
    m = s.lambertian((0.6, 0.5, 0.4))
    if lit:
        e = s.diffuse_light_texture(s.solid((4.0, 4.0, 4.0)))
        for target in (s.objects, s.lights):
            if kind == "tris":
                target.add_triangles(np.array([[[-3, 4, -1], [3, 4, -1], [0, 4, 3]]], np.float32), None, None, e)
            elif kind == "spheres":
                target.add_sphere((0.0, 5.0, 0.0), 1.5, e)
            elif kind == "quads":
                target.add_quad((-2, 4, -1), (4, 0, 0), (0, 0, 3), e)
        if kind == "instances":
            s.objects.add_instance(InstanceSrc(new_box((-1, 4, -1), (1, 4.5, 1), e)))
            s.lights.add_quad((-1, 4, -1), (2, 0, 0), (0, 0, 2), e)
    if kind == "tris":
        s.objects.add_triangles(np.array([[[-2, -1, 0], [2, -1, 0], [0, 2, 0]], [[-2, -1, 1], [2, -1, 1], [0, 2, 1]]], np.float32), None, None, m)
    elif kind == "spheres":
        s.objects.add_sphere((0.0, 0.0, 0.0), 1.0, m)
        s.objects.add_sphere((1.5, 0.5, 1.0), 0.7, m)
    elif kind == "quads":
        s.objects.add_quad((-2, -1, 0), (4, 0, 0), (0, 3, 0), m)
        s.objects.add_quad((-2, -1, 0), (4, 0, 0), (0, 0, 4), m)
    else:
        s.objects.add_instance(InstanceSrc(new_box((-1, -1, -1), (1, 1, 1), m)).rotate_y(0.5))
        s.objects.add_instance(InstanceSrc(new_box((-1, -1, -1), (1, 1, 1), m)).translate((2.5, 0.0, 1.0)))
    s.objects.build_bvh = bvh
    s.camera = _camera(width=12, spp=4)
    o, r = _same_image(s, 4)
    rays = r.generate_rays(1, seed=2)
    ho, hd = o.trace(rays), r.trace(rays)
    assert np.array_equal(hd["prim_type"], ho[0][:, 0]) and np.array_equal(hd["prim_id"], ho[0][:, 1])
    assert bits_equal(hd["t"][hd["prim_type"] >= 0], ho[1][:, 0][hd["prim_type"] >= 0])


@pytest.mark.parametrize("width,spp,depth", [(1, 1, 1), (1, 4, 3), (3, 1, 1), (5, 9, 0), (2, 4, 300)])
def test_smallest_images_sample_counts_and_depths(width, spp, depth):
    """Depth 0 returns black without tracing (src/camera.rs:271-273); depth 300 is beyond what the shared launches take."""
    from pathtracer_amd import scenes
    src = scenes.cornell(width, spp, max_depth=depth)
    strata = int(np.floor(np.sqrt(spp))) ** 2
    o, r = _same_image(src, strata)
    r.render(0, strata, seed=3)
    acc, n = r.read_accum()
    assert n == strata and acc.shape == (r.npix, 3)
    if depth == 0:
        assert not acc.any() and r.stats()["rays"] == 0


def test_empty_ranges_and_lists_do_nothing():
    from pathtracer_amd import scenes
    _, r = _pair(scenes.cornell(8, 4))
    r.render(2, 2, seed=1)
    acc, n = r.read_accum()
    assert n == 0 and not acc.any()
    assert r.trace(np.zeros((0, 8), np.float32)).shape == (0,)
    assert r.light_pdf(np.zeros((0, 8), np.float32)).shape == (0,)
    assert r.render_samples(1, 1, seed=1).shape == (0, r.npix, 3)
    r.render(0, 4, seed=1)
    assert r.read_accum()[1] == 4


def _odd_scene(which):
    from pathtracer_amd.scene_src import InstanceSrc, new_box
    s = SceneSource()
    s.name = "odd_" + which
    grey = s.lambertian((0.7, 0.7, 0.7))
    e = s.diffuse_light_texture(s.solid((5.0, 4.0, 3.0)))
    s.objects.add_quad((-4, -1, -4), (8, 0, 0), (0, 0, 8), grey)
    if which == "no_material":
        # Sphere::new(.., None): the path that reaches it returns nothing (src/camera.rs:286-287)
        s.objects.add_sphere((0.0, 0.0, 0.0), 1.0, -1)
        s.objects.add_quad((-2, 3, -2), (4, 0, 0), (0, 0, 4), e)
        s.lights.add_quad((-2, 3, -2), (4, 0, 0), (0, 0, 4), e)
    elif which == "light_seen_from_behind":
        # Diffuse::emitted is zero on the back face (src/materials/light.rs:42-47): this quad faces up, away from the floor
        s.objects.add_quad((-2, 3, -2), (0, 0, 4), (4, 0, 0), e)
        s.lights.add_quad((-2, 3, -2), (0, 0, 4), (4, 0, 0), e)
        s.objects.add_sphere((0.0, 0.0, 0.0), 1.0, s.metal((0.9, 0.9, 0.9), 0.0))
    elif which == "mirrors_and_glass":
        # specular vertices only between the camera and the light: no pdf, no roulette (src/camera.rs:298-313)
        s.objects.add_sphere((-1.2, 0.0, 0.0), 1.0, s.metal((0.9, 0.8, 0.7), 0.3))
        s.objects.add_sphere((1.2, 0.0, 0.0), 1.0, s.dielectric(1.5))
        s.objects.add_sphere((1.2, 0.0, 0.0), -0.8, s.dielectric(1.5))   # hollow, as in the book's scenes
        s.objects.add_quad((-3, 4, -3), (6, 0, 0), (0, 0, 6), e)
        s.lights.add_quad((-3, 4, -3), (6, 0, 0), (0, 0, 6), e)
    elif which == "instance_in_the_light_list":
        # Hittable::pdf and ::sample defaults: pdf 0 and a zero vector (src/objects.rs:97-102)
        box = InstanceSrc(new_box((-1, 3, -1), (1, 3.5, 1), e))
        s.objects.add_instance(box)
        s.lights.add_instance(InstanceSrc(new_box((-1, 3, -1), (1, 3.5, 1), e)))
        s.lights.add_quad((-3, 4, -3), (6, 0, 0), (0, 0, 6), e)
        s.objects.add_quad((-3, 4, -3), (6, 0, 0), (0, 0, 6), e)
        s.objects.add_sphere((0.0, 0.0, 0.0), 1.0, grey)
    s.camera = _camera(width=16, spp=16, depth=8, background=(0.02, 0.03, 0.05))
    return s


@pytest.mark.parametrize("which", ["no_material", "light_seen_from_behind", "mirrors_and_glass", "instance_in_the_light_list"])
@pytest.mark.parametrize("prob", [1.0, 0.5])
def test_corners_of_ray_color(which, prob):
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY
    src = _odd_scene(which)
    o, r = _pair(src)
    o.set_bsdf_probability(prob)
    r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, prob)
    ref = np.nan_to_num(o.render(0, 16, seed=9, iterative=True, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 16, seed=9), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, (which, prob))
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)
    r.stats_reset()
    r.render(0, 16, seed=9)
    assert abs(r.stats()["rays"] - o.render(0, 16, seed=9)["rays"]) <= 0.01 * r.stats()["rays"]


@pytest.mark.parametrize("which", ["no_material", "light_seen_from_behind", "mirrors_and_glass", "instance_in_the_light_list"])
def test_the_same_corners_under_next_event_estimation(which):
    from pathtracer_amd.core import PT_OPT_INTEGRATOR
    o, r = _pair(_odd_scene(which))
    r.set_option(PT_OPT_INTEGRATOR, 1)
    ref = np.nan_to_num(o.render(0, 16, seed=9, iterative=2, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 16, seed=9), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, (which,))
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)


def _bent_scene(which):
    from pathtracer_amd.scene_src import InstanceSrc, PrimSet
    s = SceneSource()
    s.name = "bent_" + which
    img = (np.arange(5 * 3 * 3, dtype=np.uint32).reshape(3, 5, 3) * 53 % 256).astype(np.uint8)   # 5 wide, 3 high: not square, not a power of two
    tex = s.lambertian_texture(s.checkered(s.image(img), s.checkered(s.solid((0.9, 0.2, 0.2)), s.image(None), 7.0), 3.0))
    e = s.diffuse_light_texture(s.image(img))   # a textured emitter
    s.objects.add_quad((-4, -1, -4), (8, 0, 0), (0, 0, 8), tex)
    s.objects.add_sphere((-1.5, 0.0, 0.5), 1.0, tex)
    s.objects.add_quad((-2, 4, -2), (4, 0, 0), (0, 0, 4), e)
    s.lights.add_quad((-2, 4, -2), (4, 0, 0), (0, 0, 4), e)
    child = PrimSet()
    child.add_sphere((0.0, 0.0, 0.0), 1.0, s.metal((0.8, 0.8, 0.9), 0.1))
    child.add_quad((-1, -1, 1.2), (2, 0, 0), (0, 2, 0), tex)
    # a squashed, turned and shifted copy: Instance::hit carries the normal with the matrix itself, not its inverse
    # transpose, and never renormalises the ray (src/objects.rs:385-396)
    s.objects.add_instance(InstanceSrc(child).scale((1.5, 0.5, 1.0)).rotate(0.7, (1.0, 2.0, 0.5)).translate((1.5, 0.3, 0.0)))
    s.objects.build_bvh = True
    if which == "wide_lens":
        s.camera = CameraSrc(1.0, 16, 35.0, (0.0, 1.5, -7.0), (0.0, 0.0, 0.0), 6.5, 4.0, 16, 6, (0.1, 0.1, 0.1))   # defocus angle 4 degrees
    elif which == "one_row":
        s.camera = CameraSrc(16.0 / 9.0 * 20.0, 24, 35.0, (0.0, 1.5, -7.0), (0.0, 0.0, 0.0), 1.0, 0.0, 16, 6, (0.1, 0.1, 0.1))   # height = max(1, 24 / 35.6) = 1
    else:
        s.camera = CameraSrc(0.5, 9, 60.0, (0.0, 1.5, -7.0), (0.0, 0.0, 0.0), 1.0, 0.0, 16, 6, (0.1, 0.1, 0.1))   # taller than wide, odd width
    return s


@pytest.mark.parametrize("which", ["wide_lens", "one_row", "tall"])
def test_bent_cameras_textures_and_a_squashed_instance(which):
    src = _bent_scene(which)
    o, r = _same_image(src, 16, seed=21)
    assert (o.width, o.height) == (r.width, r.height)
    if which == "one_row":
        assert r.height == 1
    for stratum in (0, 7):
        ro, rd = o.sample_rays(21, stratum)[0], r.generate_rays(stratum, seed=21)
        assert bits_equal(ro, rd)
        ho, hd = o.trace(rd), r.trace(rd)
        assert np.array_equal(hd["prim_type"], ho[0][:, 0]) and np.array_equal(hd["prim_id"], ho[0][:, 1]) and np.array_equal(hd["inst_id"], ho[0][:, 2])


def _haze_scene(which):
    """Corners of the two scene features the reference does not have: media and moving spheres."""
    s = SceneSource()
    s.name = "haze_" + which
    grey = s.lambertian((0.7, 0.7, 0.7))
    e = s.diffuse_light_texture(s.solid((6.0, 6.0, 6.0)))
    s.objects.add_quad((-6, -1, -6), (12, 0, 0), (0, 0, 12), grey)
    s.objects.add_quad((-2, 5, -2), (4, 0, 0), (0, 0, 4), e)
    s.lights.add_quad((-2, 5, -2), (4, 0, 0), (0, 0, 4), e)
    s.objects.add_sphere((0.0, 0.0, 0.0), 1.0, s.dielectric(1.5))
    if which == "camera_inside":
        s.add_medium((0.0, 1.0, -6.0), 3.0, 0.4, s.solid((0.9, 0.9, 0.9)))        # the camera sits at its centre
    elif which == "overlapping":
        s.add_medium((-0.5, 0.5, 0.0), 2.0, 0.7, s.solid((0.9, 0.3, 0.3)))
        s.add_medium((0.5, 0.5, 0.0), 2.0, 0.7, s.solid((0.3, 0.3, 0.9)))
        s.add_medium((0.0, 0.0, 0.0), 1.0, 5.0, s.noise((0.8, 0.8, 0.8), 3.0, seed=4))   # inside the glass ball, noise for albedo
    elif which == "around_the_light":
        s.add_medium((0.0, 5.0, 0.0), 3.0, 0.5, s.solid((0.8, 0.8, 0.8)))
        s.add_medium((0.0, 0.0, 0.0), 5000.0, 1e-4, s.solid((1.0, 1.0, 1.0)))           # the mist of the Next Week scene
    elif which == "fast_spheres":
        s.add_moving_sphere((-3.0, 0.0, 1.0), (3.0, 2.0, 1.0), 0.6, grey)                  # crosses the whole view
        s.add_moving_sphere((0.0, 3.0, 0.0), (0.0, 3.0, 0.0), 0.5, s.metal((0.9, 0.9, 0.9), 0.0))   # does not move
        s.add_moving_sphere((0.0, 0.0, 0.0), (0.0, 0.5, 0.0), 0.4, e)                      # an emitter, inside the glass ball
    s.camera = _camera(width=20, spp=16, depth=10, background=(0.05, 0.07, 0.1))
    return s


@pytest.mark.parametrize("which", ["camera_inside", "overlapping", "around_the_light", "fast_spheres"])
def test_corners_of_media_and_motion(which):
    src = _haze_scene(which)
    o, r = _same_image(src, 16, seed=4)
    r.stats_reset()
    r.render(0, 16, seed=4)
    want = o.render(0, 16, seed=4, iterative=True)["rays"]
    assert abs(r.stats()["rays"] - want) <= 0.02 * want


def test_one_context_many_scenes_and_two_contexts_at_once():
    """pt_upload_scene replaces the scene (other material types, other queues), pt_set_camera the image size; contexts
    do not share anything.  Each result must have the bits a fresh context gives."""
    from pathtracer_amd import scenes
    from pathtracer_amd.core import HostScene, Renderer
    srcs = [scenes.cornell(24, 16), scenes.small_mixed(40, 16), _haze_scene("overlapping"), scenes.cornell(8, 4), _odd_scene("mirrors_and_glass")]
    fresh = []
    for src in srcs:
        r = Renderer(HostScene(src))
        r.render(0, 4, seed=6)
        fresh.append(r.read_accum()[0])
    one = Renderer(HostScene(srcs[0]))
    for k, src in enumerate(srcs):
        if k:
            one.render(0, 2, seed=1)          # work under way when the scene is replaced
            one.load(HostScene(src))
        one.render(0, 4, seed=6)
        acc, n = one.read_accum()
        assert n == 4 and bits_equal(acc, fresh[k]), k
    a, b = Renderer(HostScene(srcs[1])), Renderer(HostScene(srcs[2]))
    for lo in range(4):                       # interleaved, neither waits for the other
        a.render(lo, lo + 1, seed=6)
        b.render(lo, lo + 1, seed=6)
    assert bits_equal(a.read_accum()[0], fresh[1]) and bits_equal(b.read_accum()[0], fresh[2])


@pytest.mark.parametrize("make,name", [(_odd_scene, "no_material"), (_odd_scene, "mirrors_and_glass"), (_odd_scene, "instance_in_the_light_list"),
                                       (_bent_scene, "wide_lens"), (_haze_scene, "camera_inside"), (_haze_scene, "overlapping"),
                                       (_haze_scene, "around_the_light")])
def test_the_reference_order_traversal_gives_the_same_samples(make, name):
    """PT_OPT_TRAVERSAL = 1 walks the 32-byte nodes as src/bvh.rs:725-797 does, PT_OPT_COUNTERS counts on the way; the
    closest hits are the same, so every sample must have the same bits as with the two-box nodes."""
    from pathtracer_amd.core import PT_OPT_COUNTERS, PT_OPT_TRAVERSAL, HostScene, Renderer
    src = make(name)
    want = Renderer(HostScene(src)).render_samples(0, 8, seed=12)
    for option in (PT_OPT_TRAVERSAL, PT_OPT_COUNTERS):
        r = Renderer(HostScene(src))
        r.set_option(option, 1)
        got = r.render_samples(0, 8, seed=12)
        assert bits_equal(np.nan_to_num(got), np.nan_to_num(want)) and np.array_equal(np.isnan(got), np.isnan(want)), option
        r.render(0, 8, seed=12)
        if option == PT_OPT_COUNTERS:
            st = r.stats()
            assert st["node_tests"] + st["quad_tests"] + st["sphere_tests"] + st["tri_tests"] > 0


@pytest.mark.parametrize("which", ["overlapping", "fast_spheres"])
def test_media_and_motion_with_rays_aimed_at_the_light(which):
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY
    src = _haze_scene(which)
    o, r = _pair(src)
    o.set_bsdf_probability(0.6)
    r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, 0.6)
    ref = np.nan_to_num(o.render(0, 16, seed=2, iterative=True, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 16, seed=2), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, (which,))
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)
