"""The headline workload at its full size (knob scene, 768x768, depth 50): first hits of every camera ray
against the oracle, and the size-independent properties of the accumulator."""
import numpy as np
import pytest

from pathtracer_amd import scenes
from tests.util import samples_agree

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


# ---------------------------------------------------------------- the 1 M-triangle mesh in the Cornell box, 1024 x 1024
@pytest.fixture(scope="module")
def dragon():
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import HostScene
    src = scenes.dragon_box()  # 1024 x 1024, 10^6 triangles, depth 16
    return src, HostScene(src), OracleScene(src)


def test_million_triangle_tree_is_the_oracles_tree(dragon):
    """BVHBuilder::build (src/bvh.rs:295-516) on 10^6 triangles: node arrays, permutation and triangles bit for bit."""
    src, host, o = dragon
    assert host.desc.n_tris == 1_000_000 and host.width == 1024 and host.height == 1024
    buf = host.set_desc(0).tris
    bb, meta = o.bvh(0, 0)
    nodes = host.nodes(buf)
    assert len(nodes) == len(bb) > 1_000_000
    assert np.array_equal(np.concatenate([nodes["bmin"], nodes["bmax"]], axis=1).view(np.uint32), bb.view(np.uint32))
    assert np.array_equal(((nodes["b"] & 0x80000000) != 0).astype(np.uint32), meta[:, 0])
    assert np.array_equal(nodes["a"], meta[:, 1]) and np.array_equal(nodes["b"] & 0x7FFFFFFF, meta[:, 2])
    assert np.array_equal(host.perm(0, 0), o.perm(0, 0))
    assert np.array_equal(host.triangles(buf).view(np.uint32), o.triangles(0).view(np.uint32))


def test_million_triangle_first_hits_of_every_camera_ray(dragon):
    from pathtracer_amd.core import PT_OPT_TRAVERSAL, Renderer
    src, host, o = dragon
    r = Renderer(host)
    rays, _ = o.sample_rays(7, 600)  # seed 7, stratum 600
    assert np.array_equal(r.generate_rays(600, seed=7).view(np.uint32), rays.view(np.uint32))
    ids, tuv = o.trace(rays)
    for reference_order in (0, 1):  # the two-box traversal that renders, and the reference's loop
        r.set_option(PT_OPT_TRAVERSAL, reference_order)
        h = r.trace(rays)
        assert h.shape[0] == 1024 * 1024
        assert np.array_equal(h["prim_type"], ids[:, 0]) and np.array_equal(h["prim_id"], ids[:, 1]) and np.array_equal(h["inst_id"], ids[:, 2])
        assert np.array_equal(h["t"].view(np.uint32), tuv[:, 0].view(np.uint32))
        assert np.array_equal(h["u"].view(np.uint32), tuv[:, 1].view(np.uint32)) and np.array_equal(h["v"].view(np.uint32), tuv[:, 2].view(np.uint32))
    assert (ids[:, 0] == 0).mean() > 0.1 and (ids[:, 0] == 2).mean() > 0.2  # the mesh and the box both fill a good part of the frame
    # incoherent rays: from points on the first hits into random directions (what bounces look like)
    rng = np.random.default_rng(5)
    hit = ids[:, 0] >= 0
    p = rays[hit, :3] + tuv[hit, :1] * rays[hit, 3:6]
    d = rng.normal(size=p.shape).astype(np.float32)
    sel = rng.choice(p.shape[0], 300_000, replace=False)
    bounce = np.concatenate([p[sel], d[sel], np.full((sel.size, 1), 0.001, np.float32), np.full((sel.size, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    ids2, tuv2 = o.trace(bounce)
    r.set_option(PT_OPT_TRAVERSAL, 0)
    h2 = r.trace(bounce)
    assert np.array_equal(h2["prim_type"], ids2[:, 0]) and np.array_equal(h2["prim_id"], ids2[:, 1])
    assert np.array_equal(h2["t"].view(np.uint32), tuv2[:, 0].view(np.uint32))


def test_million_triangle_accumulator_is_the_ordered_sum_of_the_samples(dragon):
    from pathtracer_amd.core import PT_OPT_ROLLING, Renderer
    src, host, o = dragon
    npix = 1024 * 1024
    a = Renderer(host)
    a.render(40, 44, seed=2)  # the rolling schedule, as benchmarked
    acc_a, n_a = a.read_accum()
    b = Renderer(host)
    b.set_option(PT_OPT_ROLLING, 0)  # one batch at a time
    b.render(40, 42, seed=2)
    b.render(42, 44, seed=2)
    acc_b, n_b = b.read_accum()
    assert n_a == n_b == 4 and np.array_equal(acc_a.view(np.uint32), acc_b.view(np.uint32))
    samples = b.render_samples(40, 44, seed=2)
    total = np.zeros((npix, 3), np.float32)
    for k in range(4):
        s = samples[k]
        bad = np.isnan(s).any(axis=1, keepdims=True)
        total = np.where(bad, total, total + s)
    assert np.array_equal(total.view(np.uint32), acc_a.reshape(npix, 3).view(np.uint32))
    # per-sample radiance of one stratum at full size against the oracle
    want = o.render(40, 41, seed=2, iterative=True, per_sample=True)["per_sample"][0]
    g = np.nan_to_num(samples[0], nan=0.0, posinf=0.0, neginf=0.0)
    w = np.nan_to_num(want, nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(g, w, 0.999, "")
    assert 0.0 < w.mean() < 10.0
