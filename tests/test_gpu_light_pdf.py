"""LightBuffer::pdf (src/objects.rs:259-278) on the device against the oracle, ray by ray and bit by bit.

The device skips the lights whose box a ray cannot reach before it runs the exact per-light tests; the sum must
still be the reference's sum over every light.  The rays here are chosen to sit where a wrong skip would show:
aimed at corners, edges and silhouettes of the lights, grazing along their planes, from far away and from inside."""
import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.scene_src import CameraSrc, SceneSource
from tests.util import samples_agree

pytestmark = pytest.mark.gpu


def _zoo(seed=5, n_tris=40, n_spheres=30, n_quads=60):
    """Lights of all three kinds (130 by default: three groups of 64 on the device), sizes over four decades."""
    rng = np.random.default_rng(seed)
    s = SceneSource()
    s.name = "light_zoo"
    emit = s.diffuse_light_texture(s.solid((3.0, 3.0, 3.0)))
    grey = s.lambertian((0.5, 0.5, 0.5))
    s.objects.add_quad((-50, -1, -50), (100, 0, 0), (0, 0, 100), grey)
    targets = []
    c = rng.uniform(-20, 20, (n_tris, 1, 3)).astype(np.float32)
    e = (rng.normal(size=(n_tris, 3, 3)) * 10.0 ** rng.uniform(-2, 1, (n_tris, 1, 1))).astype(np.float32)
    tris = (c + e).astype(np.float32)
    s.lights.add_triangles(tris, None, None, emit)
    s.objects.add_triangles(tris, None, None, emit)
    for t in tris:
        targets += [t[0], t[1], t[2], (t[0] + t[1]) / 2, (t[1] + t[2]) / 2, (t[0] + t[1] + t[2]) / 3]
    for k in range(n_spheres):
        cc = rng.uniform(-20, 20, 3).astype(np.float32)
        r = np.float32(10.0 ** rng.uniform(-3, 0.7))
        s.lights.add_sphere(tuple(cc), float(r), emit)
        s.objects.add_sphere(tuple(cc), float(r), emit)
        d = rng.normal(size=(4, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        targets += [cc, *(cc + r * d).astype(np.float32)]
    for k in range(n_quads):
        o = rng.uniform(-20, 20, 3).astype(np.float32)
        if k % 3 == 0:   # axis-aligned, as ceiling lights are
            u = np.array([10.0 ** rng.uniform(-1, 1), 0, 0], np.float32)
            v = np.array([0, 0, 10.0 ** rng.uniform(-1, 1)], np.float32)
        else:
            u = (rng.normal(size=3) * 10.0 ** rng.uniform(-2, 1)).astype(np.float32)
            v = (rng.normal(size=3) * 10.0 ** rng.uniform(-2, 1)).astype(np.float32)
        s.lights.add_quad(tuple(o), tuple(u), tuple(v), emit)
        s.objects.add_quad(tuple(o), tuple(u), tuple(v), emit)
        targets += [o, o + u, o + v, o + u + v, o + u / 2, o + (u + v) / 2]
    s.camera = CameraSrc(1.0, 16, 40.0, (0.0, 5.0, -60.0), (0.0, 0.0, 0.0), 10.0, 0.0, 4, 4, (0.0, 0.0, 0.0))
    return s, np.asarray(targets, np.float32)


def _rays(targets, seed, n_origins=24):
    rng = np.random.default_rng(seed)
    T = targets.astype(np.float64)
    out = []
    origins = np.concatenate([rng.uniform(-25, 25, (n_origins, 3)),               # among the lights
                              rng.normal(size=(6, 3)) * 3.0e3,                    # far outside
                              T[rng.integers(0, len(T), 6)] + rng.normal(size=(6, 3)) * 1e-3])  # next to (or inside) a light
    for o in origins:
        d = T - o
        out.append(np.concatenate([np.broadcast_to(o, d.shape), d], axis=1))                       # at the point itself
        out.append(np.concatenate([np.broadcast_to(o, d.shape), d * (1 + 1e-6 * rng.normal(size=d.shape))], axis=1))  # a hair beside it
        out.append(np.concatenate([np.broadcast_to(o, d.shape), -d * rng.uniform(0.1, 10)], axis=1))  # away from it
    # from a point of one light to a point of another: many of these run along a light's plane
    i, j = rng.integers(0, len(T), 4000), rng.integers(0, len(T), 4000)
    out.append(np.concatenate([T[i], T[j] - T[i]], axis=1))
    # axis-parallel directions (zero components) and fully random ones
    o = rng.uniform(-25, 25, (3000, 3))
    axis = np.eye(3)[rng.integers(0, 3, 3000)] * rng.choice([-1.0, 1.0], (3000, 1))
    out.append(np.concatenate([o, axis], axis=1))
    out.append(np.concatenate([T[rng.integers(0, len(T), 3000)] - axis * 7.0, axis], axis=1))
    out.append(np.concatenate([o, rng.normal(size=(3000, 3))], axis=1))
    r = np.concatenate(out).astype(np.float32)
    return np.concatenate([r, np.zeros((len(r), 2), np.float32)], axis=1)


def _same_bits(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("solid_angle_tris", [False, True])
@pytest.mark.parametrize("sizes", [(40, 30, 60), (300, 280, 400)], ids=["130_lights", "980_lights"])
def test_every_ray_gets_the_reference_sum_over_all_lights(sizes, solid_angle_tris):
    """With more than 256 lights of a kind the device tests boxes around groups of 32 lights before the lights' own."""
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import HostScene, Renderer
    src, targets = _zoo(5, *sizes)
    rays = _rays(targets, seed=11, n_origins=24 if sizes[0] == 40 else 4)
    ref = OracleScene(src).light_pdfs(rays, solid_angle_tris)
    got = Renderer(HostScene(src)).light_pdf(rays, solid_angle_tris)
    same = _same_bits(got, ref)
    assert same.all(), (int((~same).sum()), rays[~same][:4], got[~same][:4], ref[~same][:4])
    assert (ref > 0).mean() > 0.2 and (ref == 0).mean() > 0.2   # the set has both hits and misses in quantity


@pytest.mark.parametrize("name", ["interior", "small_mixed", "knob"])
def test_on_the_benchmark_scenes_with_their_own_camera_and_bounce_rays(name):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import HostScene, Renderer
    src = {"interior": lambda: scenes.interior(64, 16, n_lights=100, columns=3), "small_mixed": lambda: scenes.small_mixed(48, 16),
           "knob": lambda: scenes.knob(48, 16, detail=0.3)}[name]()
    o, r = OracleScene(src), Renderer(HostScene(src))
    rays = r.generate_rays(0, seed=3)
    hits = r.trace(rays)
    m = hits["prim_type"] >= 0
    p = rays[m, 0:3] + hits["t"][m, None] * rays[m, 3:6]       # first vertices of the paths
    rng = np.random.default_rng(2)
    d = rng.normal(size=p.shape).astype(np.float32)
    d[:, 1] = np.abs(d[:, 1])                                  # mostly upwards, where the lights are
    bounce = np.concatenate([p, d, np.zeros((len(p), 2), np.float32)], axis=1).astype(np.float32)
    for rr in (rays, bounce):
        for nee in (False, True):
            same = _same_bits(r.light_pdf(rr, nee), o.light_pdfs(rr, nee))
            assert same.all(), (name, nee, int((~same).sum()))


def test_a_scene_without_lights_is_refused():
    from pathtracer_amd.core import HostScene, PtError, Renderer
    s = SceneSource()
    s.name = "dark"
    s.objects.add_quad((-1, 0, -1), (2, 0, 0), (0, 0, 2), s.lambertian((0.5, 0.5, 0.5)))
    s.camera = CameraSrc(1.0, 8, 40.0, (0.0, 5.0, -6.0), (0.0, 0.0, 0.0), 10.0, 0.0, 4, 4, (0.0, 0.0, 0.0))
    r = Renderer(HostScene(s))
    with pytest.raises(PtError):
        r.light_pdf(np.zeros((1, 8), np.float32))
    # and so is sampling lights that do not exist: the reference panics in gen_range(0..0) (src/objects.rs:236-237)
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY, PT_OPT_INTEGRATOR
    r.render(0, 1, seed=1)
    r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, 0.5)
    with pytest.raises(PtError):
        r.render(0, 1, seed=1)
    r.set_option(PT_OPT_INTEGRATOR, 1)     # next-event estimation simply finds no light to aim at
    r.render(0, 1, seed=1)
    assert np.isfinite(r.read_accum()[0]).all()


@pytest.mark.parametrize("integrator", ["mixture", "nee"])
def test_rendering_with_more_lights_than_the_flat_scan_takes(integrator):
    """300 emissive quads: inside the shade kernels LightBuffer::pdf goes through the boxes around groups of lights."""
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY, PT_OPT_INTEGRATOR, HostScene, Renderer
    src = scenes.interior(40, 16, n_lights=300, columns=3)
    o, r = OracleScene(src), Renderer(HostScene(src))
    if integrator == "nee":
        r.set_option(PT_OPT_INTEGRATOR, 1)
        ref = o.render(0, 4, seed=8, iterative=2, per_sample=True)["per_sample"]
    else:
        r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, 0.5)
        o.set_bsdf_probability(0.5)
        ref = o.render(0, 4, seed=8, iterative=True, per_sample=True)["per_sample"]
    ref = np.nan_to_num(ref, nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 4, seed=8), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, "")
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)
