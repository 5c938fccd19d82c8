"""Participating media (pts_medium) are not in the reference; the oracle states them, the device library restates
them.  CPU part: what the oracle's statement must satisfy on its own.  GPU part: the device against the oracle."""
import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.scene_src import CameraSrc, SceneSource
from tests.util import samples_agree


def _fog_scene(density, albedo=(1.0, 1.0, 1.0), width=32, spp=64, background=(1.0, 1.0, 1.0), depth=1, solids=False):
    """Camera at z = -10 looking at a medium sphere of radius 2 at the origin, in front of a uniform background."""
    s = SceneSource()
    s.name = "fog"
    emit = s.diffuse_light((4.0, 4.0, 4.0))
    s.objects.add_quad((-1, 50, -1), (2, 0, 0), (0, 0, 2), emit)  # a light far out of sight, so that the light list is not empty
    s.lights.add_quad((-1, 50, -1), (2, 0, 0), (0, 0, 2), emit)
    if solids:
        grey = s.lambertian((0.6, 0.6, 0.6))
        s.objects.add_sphere((0.5, -0.3, 0.0), 0.7, grey)
        s.objects.add_quad((-6, -2, -6), (12, 0, 0), (0, 0, 12), s.lambertian_texture(s.noise((0.5, 0.5, 0.5), 1.0, 2)))
        s.objects.add_sphere((-1.0, 0.5, -0.5), 0.5, s.dielectric(1.5))
        s.objects.build_bvh = True
    s.add_medium((0.0, 0.0, 0.0), 2.0, density, s.solid(albedo))
    s.camera = CameraSrc(1.0, width, 40.0, (0.0, 0.0, -10.0), (0.0, 0.0, 0.0), 10.0, 0.0, spp, depth, background)
    return s


def test_transmittance_follows_beer_lambert():
    """max_depth = 1: a path that scatters is cut off (radiance 0), one that passes sees the background (1), so the
    centre pixel's mean estimates exp(-density * chord) with chord = 4 through the middle of the sphere."""
    from oracle.pyoracle import OracleScene
    for density in (0.1, 0.4):
        src = _fog_scene(density, width=65, spp=1024)
        img = OracleScene(src).render(0, 1024, seed=3)["accum"].reshape(65, 65, 3) / 1024
        centre = float(img[32, 32].mean())
        want = np.exp(-density * 4.0)
        assert abs(centre - want) < 4.0 * np.sqrt(want * (1 - want) / 1024) + 0.01, (density, centre, want)
        assert float(img[0, 0].mean()) == 1.0  # the corner rays pass beside the sphere


def test_scattered_light_carries_the_albedo():
    """White background all around and an absorbing-free medium: every path ends on the background whatever it does
    inside, so radiance is albedo^(number of scattering events): 1 for white, and below 1 only where events happen."""
    from oracle.pyoracle import OracleScene
    white = OracleScene(_fog_scene(1.0, depth=64, spp=16)).render(0, 16, seed=1, per_sample=True)["per_sample"]
    assert np.mean(np.abs(white - 1.0) <= 1e-5) > 0.999  # the rest found the small light
    tinted = OracleScene(_fog_scene(1.0, albedo=(0.5, 0.5, 0.5), depth=64, spp=16)).render(0, 16, seed=1, per_sample=True)["per_sample"]
    powers = np.array([0.5 ** k for k in range(0, 64)], np.float32)
    is_power = np.min(np.abs(tinted[..., 0, None] - powers), axis=-1) <= 1e-6
    assert is_power.mean() > 0.999 and len(np.unique(tinted[..., 0][is_power])) > 2


def test_recursive_and_unrolled_forms_agree_with_media():
    from oracle.pyoracle import OracleScene
    o = OracleScene(_fog_scene(0.8, albedo=(0.7, 0.8, 0.9), depth=12, spp=16, solids=True, background=(0.4, 0.5, 0.7)))
    a = o.render(0, 4, seed=9, iterative=False, per_sample=True)["per_sample"]
    b = o.render(0, 4, seed=9, iterative=True, per_sample=True)["per_sample"]
    assert np.allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["fog", "next_week"])
def test_device_restatement_against_the_oracle_per_sample(which):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_PIPES, PT_OPT_ROLLING, HostScene, Renderer
    src = (_fog_scene(0.8, albedo=(0.7, 0.8, 0.9), depth=12, spp=36, width=40, solids=True, background=(0.4, 0.5, 0.7)) if which == "fog"
           else scenes.next_week_like(40, 36, max_depth=12, extensions=True))
    o, r = OracleScene(src), Renderer(HostScene(src))
    ref = np.nan_to_num(o.render(0, 4, seed=5, iterative=True, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 4, seed=5), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, "")
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)
    # both schedules, same bits; ray counts as the oracle's
    accs = []
    for pipes, rolling in ((1, 0), (2, 3 * 40 * 40)):
        r2 = Renderer(HostScene(src))
        r2.set_option(PT_OPT_PIPES, pipes)
        r2.set_option(PT_OPT_ROLLING, rolling)
        r2.stats_reset()
        r2.render(0, 20, seed=7)
        accs.append(r2.read_accum()[0])
        rays = r2.stats()["rays"]
    assert np.array_equal(accs[0].view(np.uint32), accs[1].view(np.uint32))
    want = o.render(0, 20, seed=7, iterative=True)["rays"]
    assert abs(rays - want) <= 1e-2 * want, (rays, want)  # paths part ways where the two math libraries differ in the last bit


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["fog", "next_week"])
def test_next_event_estimation_with_media_against_the_oracles_statement(which):
    """Next-event estimation in a scene with participating media: the path's own rays scatter as under the reference's
    estimator, shadow rays are dimmed by the transmittance up to the emitter they reach.  Per sample against the
    oracle's statement of the same estimator, both schedules bit for bit, and the converged image against the
    reference estimator's (they estimate the same integral)."""
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_INTEGRATOR, PT_OPT_PIPES, PT_OPT_ROLLING, HostScene, Renderer
    src = (_fog_scene(0.8, albedo=(0.7, 0.8, 0.9), depth=12, spp=256, width=40, solids=True, background=(0.4, 0.5, 0.7)) if which == "fog"
           else scenes.next_week_like(40, 256, max_depth=12, extensions=True))
    o, r = OracleScene(src), Renderer(HostScene(src))
    r.set_option(PT_OPT_INTEGRATOR, 1)
    ref = np.nan_to_num(o.render(0, 4, seed=5, iterative=2, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 4, seed=5), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, which)
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)
    accs = []
    for pipes, rolling in ((1, 0), (2, 3 * 40 * 40)):
        r2 = Renderer(HostScene(src))
        r2.set_option(PT_OPT_INTEGRATOR, 1)
        r2.set_option(PT_OPT_PIPES, pipes)
        r2.set_option(PT_OPT_ROLLING, rolling)
        r2.render(0, 256, seed=7)
        accs.append(r2.read_accum()[0])
    assert np.array_equal(accs[0].view(np.uint32), accs[1].view(np.uint32))
    # the same integral as the reference's estimator: image means agree within the Monte-Carlo error of 256 passes
    r3 = Renderer(HostScene(src))
    r3.render(0, 256, seed=8)
    plain = np.nan_to_num(r3.read_accum()[0], posinf=0.0)
    nee = np.nan_to_num(accs[0], posinf=0.0)
    assert abs(nee.mean() - plain.mean()) <= 0.03 * plain.mean(), (nee.mean(), plain.mean())
