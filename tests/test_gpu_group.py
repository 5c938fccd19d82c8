"""pt_group on the one GPU of the test box: a group of one member (merged through RCCL, one rank) equals a plain
context bit for bit; groups that list device 0 several times run the stratum split and the merge of the HIP
renderer itself (two and three contexts side by side on one GPU) and agree with one context to rounding."""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.core import GroupRenderer, HostScene, Renderer, group_split

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    return HostScene(scenes.small_mixed(width=96, spp=64, max_depth=10))


def _member_accum(g: GroupRenderer, i: int):
    ctx = g.L.pt_group_member(g.g, i)
    acc = np.zeros((g.npix, 3), np.float32)
    n = C.c_uint32(0)
    assert g.L.pt_read_accum(ctx, acc.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
    return acc, int(n.value)


def test_group_of_one_equals_a_context_bit_for_bit(host, monkeypatch):
    monkeypatch.setenv("PT_GROUP_RCCL_ALWAYS", "1")  # a group of one needs no collective: ask for the one-rank reduce all the same
    r = Renderer(host, device=0)
    r.render(0, 40, seed=5)
    want, n_want = r.read_accum()
    img_want = r.read_rgb8()
    g = GroupRenderer(host, [0])
    assert g.size() == 1
    g.render(0, 40, seed=5)
    got, n = g.read_accum()  # ncclReduce with one rank
    assert n == n_want == 40
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(g.read_rgb8(), img_want)
    st = g.stats()
    assert st["paths"] == 40 * g.npix and st["rays"] >= st["paths"]
    g.close()
    monkeypatch.delenv("PT_GROUP_RCCL_ALWAYS")
    g = GroupRenderer(host, [0])  # and without it: the merge is a copy
    g.render(0, 40, seed=5)
    got, n = g.read_accum()
    assert n == 40 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    g.close()
    r.close()


@pytest.mark.parametrize("n_members", [2, 3])
def test_members_on_one_gpu_split_the_strata_and_merge_to_the_single_context_image(host, n_members):
    r = Renderer(host, device=0)
    r.render(0, 64, seed=9)
    want, _ = r.read_accum()
    img_want = r.read_rgb8()
    g = GroupRenderer(host, [0] * n_members)
    g.render(0, 64, seed=9)
    got, n = g.read_accum()
    assert n == 64
    # the merge is the f32 sum of the members' accumulators in member order, exactly
    parts = [_member_accum(g, i) for i in range(n_members)]
    assert [p[1] for p in parts] == [e - b for b, e in (group_split(0, 64, n_members, i) for i in range(n_members))]
    total = parts[0][0].copy()
    for acc, _ in parts[1:]:
        total = total + acc
    assert np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(total).view(np.uint32))
    # each member's accumulator is what a context renders for that piece alone
    b, e = group_split(0, 64, n_members, 1)
    solo = Renderer(host, device=0)
    solo.render(b, e, seed=9)
    assert np.array_equal(solo.read_accum()[0].view(np.uint32), parts[1][0].view(np.uint32))
    solo.close()
    # against one context: the same samples summed in another association
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= 2e-5 * scale
    diff = np.abs(g.read_rgb8().astype(np.int32) - img_want.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    g.close()
    r.close()


def test_group_renders_progressively_and_resets(host):
    g = GroupRenderer(host, [0, 0])
    g.render(0, 16, seed=2)
    first, n1 = g.read_accum()  # reading leaves the members' accumulators alone
    g.render(16, 48, seed=2)
    both, n2 = g.read_accum()
    assert (n1, n2) == (16, 48)
    h = GroupRenderer(host, [0, 0])
    h.render(0, 16, seed=2)
    h.render(16, 48, seed=2)
    again, _ = h.read_accum()
    assert np.array_equal(both.view(np.uint32), again.view(np.uint32))
    assert not np.array_equal(first, both)
    g.reset_accum()
    zero, n0 = g.read_accum()
    assert n0 == 0 and not zero.any()
    g.close()
    h.close()


def test_members_are_created_side_by_side(host):
    """pt_group_create makes its contexts on one host thread each: four members on one GPU take about as long as one
    context, not four times as long (slack for a loaded box)."""
    import time
    Renderer(host, device=0).close()  # the runtime and the code object are loaded by now
    t0 = time.perf_counter()
    r = Renderer(host, device=0)
    one = time.perf_counter() - t0
    r.close()
    t0 = time.perf_counter()
    g = GroupRenderer(host, [0, 0, 0, 0])
    four = time.perf_counter() - t0
    assert g.size() == 4
    g.close()
    assert four <= 2.0 * one + 0.25, f"one context {one:.3f} s, a group of four {four:.3f} s"
