"""The headline workload at its full size (knob scene, 768x768, depth 50): first hits of every camera ray
against the oracle, and the size-independent properties of the accumulator."""
import numpy as np
import pytest

from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def knob():
    from pathtracer_amd.core import HostScene
    src = scenes.knob(width=768)
    return src, HostScene(src)


def test_every_camera_ray_of_a_stratum_hits_what_the_oracle_hits(knob):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import Renderer
    src, host = knob
    o = OracleScene(src)
    r = Renderer(host)
    rays, _ = o.sample_rays(3, 5000)  # seed 3, stratum 5000
    assert np.array_equal(r.generate_rays(5000, seed=3).view(np.uint32), rays.view(np.uint32))
    ids, tuv = o.trace(rays)
    h = r.trace(rays)
    assert h.shape[0] == 768 * 768
    assert np.array_equal(h["prim_type"], ids[:, 0]) and np.array_equal(h["prim_id"], ids[:, 1]) and np.array_equal(h["inst_id"], ids[:, 2])
    assert np.array_equal(h["t"].view(np.uint32), tuv[:, 0].view(np.uint32))
    assert (ids[:, 0] >= 0).mean() > 0.5  # the object and its backdrop fill most of the frame


def test_accumulator_is_the_ordered_sum_of_the_samples_however_the_strata_are_batched(knob):
    from pathtracer_amd.core import PT_OPT_MAX_PATHS, PT_OPT_PIPES, Renderer
    src, host = knob
    npix = 768 * 768
    a = Renderer(host)
    a.stats_reset()
    a.render(100, 108, seed=11)  # one batch of eight strata
    acc_a, n_a = a.read_accum()
    st = a.stats()
    assert n_a == 8 and st["paths"] == 8 * npix and st["paths"] <= st["rays"] <= 50 * st["paths"]

    b = Renderer(host)
    b.set_option(PT_OPT_PIPES, 1)
    b.set_option(PT_OPT_MAX_PATHS, 3 * npix)  # batches of at most three strata, one at a time
    b.render(100, 103, seed=11)
    b.render(103, 108, seed=11)
    acc_b, n_b = b.read_accum()
    assert n_b == 8 and np.array_equal(acc_a.view(np.uint32), acc_b.view(np.uint32))

    samples = Renderer(host).render_samples(100, 108, seed=11)  # (8, npix, 3)
    total = np.zeros((npix, 3), np.float32)
    for k in range(8):  # `*pixel += c` per pass, NaN samples skipped (camera.rs:212-217)
        s = samples[k]
        bad = np.isnan(s).any(axis=1, keepdims=True)
        total = np.where(bad, total, total + s)
    assert np.array_equal(total.view(np.uint32), acc_a.reshape(npix, 3).view(np.uint32))
    finite = samples[np.isfinite(samples)]
    assert finite.min() >= 0.0  # radiance is never negative


def test_image_is_reproducible_and_lit(knob):
    from pathtracer_amd.core import Renderer
    src, host = knob
    imgs = []
    for _ in range(2):
        r = Renderer(host)
        r.render(0, 32, seed=1)
        imgs.append(r.read_rgb8())
    assert np.array_equal(imgs[0], imgs[1])
    assert imgs[0].shape == (768, 768, 3) and 5.0 < imgs[0].mean() < 200.0
