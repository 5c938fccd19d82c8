"""Instances inside instances (src/objects.rs:281-286: an Instance holds a whole PrimitiveBuffer, which has an
instance buffer of its own).  The device library searches two levels; the host layer lists deeper instances at the
top level and keeps the levels of their nesting as chain records, which the device applies one after the other.
The oracle evaluates the nesting as the reference does, level by level."""
import numpy as np
import pytest

from pathtracer_amd.core import HostScene
from pathtracer_amd.scene_src import CameraSrc, InstanceSrc, PrimSet, SceneSource, new_box
from tests.util import samples_agree


def nested_scene(width=48, spp=16, top_bvh=True) -> SceneSource:
    s = SceneSource()
    s.name = "nested"
    red, green, blue, grey = s.lambertian((0.8, 0.2, 0.2)), s.lambertian((0.2, 0.8, 0.2)), s.lambertian((0.2, 0.3, 0.9)), s.lambertian((0.6, 0.6, 0.6))
    mirror = s.metal((0.9, 0.9, 0.9), 0.0)
    emit = s.diffuse_light((6.0, 6.0, 6.0))
    # level 3: a few spheres
    beads = PrimSet()
    for k in range(5):
        beads.add_sphere((0.35 * k, 0.0, 0.0), 0.15, blue if k % 2 else mirror)
    beads.build_bvh = True
    # level 2: a box with a string of beads on top, and a second string turned by 90 degrees
    post = new_box((0.0, 0.0, 0.0), (0.4, 1.2, 0.4), green)
    post.add_instance(InstanceSrc(beads).translate((-0.5, 1.35, 0.2)))
    post.add_instance(InstanceSrc(beads).rotate_y(1.5707964).scale_uniform(0.8).translate((0.2, 1.7, 0.9)))
    # level 1: two posts on a slab, and the slab's own geometry
    slab = new_box((0.0, 0.0, 0.0), (4.0, 0.2, 2.0), red)
    slab.add_sphere((3.5, 0.5, 1.0), 0.3, mirror)
    slab.add_instance(InstanceSrc(post).translate((0.5, 0.2, 0.5)))
    slab.add_instance(InstanceSrc(post).rotate_y(0.6).translate((2.4, 0.2, 1.2)))
    only_nested = PrimSet()  # an instance whose set holds nothing but another instance
    only_nested.add_instance(InstanceSrc(beads).scale((1.5, 1.0, 1.0)))
    o = s.objects
    o.add_quad((-8, 0, -8), (16, 0, 0), (0, 0, 16), grey)
    o.add_quad((-2, 6, -2), (4, 0, 0), (0, 0, 4), emit)
    s.lights.add_quad((-2, 6, -2), (4, 0, 0), (0, 0, 4), emit)
    o.add_instance(InstanceSrc(slab).rotate_y(-0.3).translate((-2.0, 0.0, -1.0)))
    o.add_instance(InstanceSrc(slab).scale_uniform(0.5).rotate_z(0.2).translate((-3.0, 1.5, 2.0)))
    o.add_instance(InstanceSrc(only_nested).translate((1.0, 2.5, -2.0)))
    o.add_instance(InstanceSrc(new_box((0, 0, 0), (0.8, 0.8, 0.8), blue)).translate((3.0, 0.0, 2.5)))  # an ordinary one
    o.build_bvh = top_bvh
    s.camera = CameraSrc(1.0, width, 45.0, (0.5, 4.0, -9.0), (0.0, 1.2, 0.0), 9.0, 0.0, spp, 8, (0.3, 0.35, 0.5))
    return s


def test_host_layer_lifts_nested_instances_and_keeps_their_levels():
    from oracle.pyoracle import OracleScene
    src = nested_scene()
    h = HostScene(src)
    # 2 slabs x (slab itself + 2 posts x (post itself + 2 bead strings)) + (only_nested: just its bead string) + 1 box
    world = h.set_desc(0)
    n_top = 2 * (1 + 2 * 3) + 1 + 1
    assert world.instances.prim_count == n_top and world.instances.node_count > 0
    for k in range(2, h.desc.n_sets):
        assert h.set_desc(k).instances.prim_count == 0  # no set below the top level holds instances
    # levels further in: one per post (below a slab), two per bead string below a post, one for the string of only_nested
    depth = sorted(h.desc.instances[world.instances.prim_first + k].chain_count for k in range(n_top))
    assert depth == [0] * 3 + [1] * 5 + [2] * 8
    assert h.desc.n_instances == n_top + sum(depth)
    for k in range(n_top):
        rec = h.desc.instances[world.instances.prim_first + k]
        assert rec.chain_count == 0 or rec.chain_first >= n_top  # chain records belong to no set
        for j in range(rec.chain_count):
            step = h.desc.instances[rec.chain_first + j]
            assert step.chain_count == 0 and step.child_set == rec.child_set
    for k in range(h.desc.n_instances):  # transform and inverse are inverses of each other at every level, and affine
        rec = h.desc.instances[k]
        t, i = np.array(rec.transform[:], np.float64).reshape(4, 4), np.array(rec.inverse[:], np.float64).reshape(4, 4)
        assert np.allclose(t @ i, np.eye(4), atol=2e-5)
        assert np.array_equal(t[3], [0, 0, 0, 1]) and np.array_equal(i[3], [0, 0, 0, 1])
    # a scene without nesting is described exactly as before: same instances in the builder's order as the oracle's
    from pathtracer_amd import scenes
    plain = scenes.cornell(32, 4)
    assert np.array_equal(HostScene(plain).instances(HostScene(plain).set_desc(0).instances).view(np.uint32), OracleScene(plain).instances(0).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("top_bvh", [True, False])
def test_nested_instances_first_hits_and_radiance_against_the_oracle(top_bvh):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import Renderer
    src = nested_scene(top_bvh=top_bvh)
    o = OracleScene(src)
    r = Renderer(HostScene(src), device=0)
    rng = np.random.default_rng(4)
    n = 40_000
    org = rng.uniform(-6, 6, (n, 3)).astype(np.float32) + np.array([0, 5, 0], np.float32)
    tgt = rng.uniform(-3.5, 3.5, (n, 3)).astype(np.float32) * np.array([1, 0.4, 1], np.float32) + np.array([0, 1.2, 0], np.float32)
    rays = np.concatenate([org, tgt - org, np.full((n, 1), 0.001, np.float32), np.full((n, 1), np.inf, np.float32)], axis=1)
    rays = np.concatenate([rays, o.primary_rays()])
    ids, tuv = o.trace(rays)
    h = r.trace(rays)
    nested_hits = (ids[:, 2] >= 0).sum()
    assert nested_hits > 3000
    # the device takes the ray through the levels one at a time as the reference's nested Instance::hit does, so the
    # distance of every hit is the oracle's to the bit; the instance numbering differs (the oracle numbers the top
    # level's instances, the device the lifted list), so ids are compared through type and distance.  Two hits at
    # exactly the same distance in two different lifted instances (a post stands on its slab: the two faces coincide)
    # may be resolved the other way round, which shows in (u, v) only: a handful of the rays.
    assert np.array_equal(h["prim_type"], ids[:, 0])
    assert np.array_equal(h["t"].view(np.uint32), tuv[:, 0].view(np.uint32))
    m = ids[:, 0] >= 0
    other = m & ((h["u"].view(np.uint32) != tuv[:, 1].view(np.uint32)) | (h["v"].view(np.uint32) != tuv[:, 2].view(np.uint32)))
    assert other.sum() <= 4 and (ids[other, 0] == 2).all() and (ids[other, 2] >= 0).all(), (int(other.sum()), ids[other])
    assert np.array_equal(h["inst_id"] >= 0, ids[:, 2] >= 0)
    got = np.nan_to_num(r.render_samples(0, 16, seed=8), nan=0.0, posinf=0.0, neginf=0.0)
    want = np.nan_to_num(o.render(0, 16, seed=8, iterative=True, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, want, 0.999, f"nested top_bvh={top_bvh}")
    assert abs(got.mean() - want.mean()) < 0.02 * want.mean()
    r.close()


@pytest.mark.gpu
def test_nested_instances_in_the_reference_order_traversal_and_under_next_event_estimation():
    """The per-lane reference-order loop (PT_OPT_TRAVERSAL = 1, also what finishes the last paths of a batch) and the
    shadow rays of the next-event estimator go through the levels of a nested instance like the ray-drawing kernel."""
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_INTEGRATOR, PT_OPT_TRAVERSAL, Renderer
    src = nested_scene(width=40, spp=16)
    o = OracleScene(src)
    host = HostScene(src)
    rays = o.primary_rays()
    ids, tuv = o.trace(rays)
    for mode in (0, 1, 2):  # 2 asks for the wave-at-a-time search, which this scene cannot use: the per-lane kernel answers
        r = Renderer(host, device=0)
        r.set_option(PT_OPT_TRAVERSAL, mode)
        h = r.trace(rays)
        assert np.array_equal(h["prim_type"], ids[:, 0]), mode
        assert np.array_equal(h["t"].view(np.uint32), tuv[:, 0].view(np.uint32)), mode
        r.close()
    r = Renderer(host, device=0)
    r.set_option(PT_OPT_TRAVERSAL, 1)
    got = np.nan_to_num(r.render_samples(0, 4, seed=8), nan=0.0, posinf=0.0, neginf=0.0)
    want = np.nan_to_num(o.render(0, 4, seed=8, iterative=True, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, want, 0.999, "nested, reference-order traversal")
    r.close()
    r = Renderer(host, device=0)
    r.set_option(PT_OPT_INTEGRATOR, 1)
    got = np.nan_to_num(r.render_samples(0, 4, seed=8), nan=0.0, posinf=0.0, neginf=0.0)
    want = np.nan_to_num(o.render(0, 4, seed=8, iterative=2, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, want, 0.999, "nested, next-event estimation")
    r.close()
