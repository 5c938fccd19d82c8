"""Scheduling options of the device library must not change any result: batches in flight, ray refill
threshold, leaf batching, queue sizes at the chunk boundaries."""
import numpy as np
import pytest

from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def small_mixed_scene():
    return scenes.small_mixed(32, 16)


def cornell_small():
    return scenes.cornell(48, 16)


def _rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = rng.uniform(lo, hi, size=(n, 3))
    r[:, 3:6] = rng.normal(size=(n, 3))
    r[:, 6], r[:, 7] = 0.001, np.inf
    return r


def _renderer(src):
    from pathtracer_amd.core import HostScene, Renderer
    return Renderer(HostScene(src), device=0)


def test_accumulator_identical_for_any_number_of_batches_in_flight():
    from pathtracer_amd.core import PT_OPT_MAX_PATHS, PT_OPT_PIPES
    src = small_mixed_scene()
    ref = None
    for pipes, max_paths in ((1, 1 << 24), (2, 1 << 24), (4, 1 << 24), (8, 1 << 24), (4, 3 * 32 * 32)):
        r = _renderer(src)
        r.set_option(PT_OPT_PIPES, pipes)
        r.set_option(PT_OPT_MAX_PATHS, max_paths)
        r.render(0, 13, seed=9)
        acc, n = r.read_accum()
        assert n == 13
        if ref is None:
            ref = acc
        else:
            assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"pipes={pipes} max_paths={max_paths}"


def test_pipes_chosen_by_launch_size_give_the_sums_of_one_pipe():
    """Unless PT_OPT_PIPES says how many, the rolling schedule runs two pipes where a launch starts at most 12 M paths and
    one above that.  A small render (two pipes by that rule), the same with one pipe and with three set by hand, and a
    context whose first request is small and whose second is large enough to cross the threshold (the second pipe is given
    up at a synchronisation point) must all produce the same bits."""
    from pathtracer_amd.core import PT_OPT_PIPES
    src = scenes.cornell(64, 36)
    accs = {}
    for pipes in (None, 1, 3):
        r = _renderer(src)
        if pipes is not None:
            r.set_option(PT_OPT_PIPES, pipes)
        r.render(0, 7, seed=5)
        r.render(7, 36, seed=5)
        accs[pipes], n = r.read_accum()
        assert n == 36
    assert np.array_equal(accs[None].view(np.uint32), accs[1].view(np.uint32))
    assert np.array_equal(accs[None].view(np.uint32), accs[3].view(np.uint32))
    # across the threshold: 2 strata of 2048 x 1536 are 6.3 M paths per launch, 5 are 15.7 M
    big = scenes.cornell(2048, 9)
    big.camera.aspect_ratio = 2048.0 / 1536.0
    outs = []
    for pipes in (None, 1):
        r = _renderer(big)
        if pipes is not None:
            r.set_option(PT_OPT_PIPES, pipes)
        r.render(0, 2, seed=3)
        r.render(2, 4, seed=3)
        r.render(4, 9, seed=3)
        acc, n = r.read_accum()
        assert n == 9
        outs.append(acc)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_first_hits_identical_for_any_refill_and_leaf_batch():
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_LEAF_BATCH, PT_OPT_REFILL_BELOW
    src = small_mixed_scene()
    oracle = OracleScene(src)
    rays = _rays(20000, 4, -7, 9)
    ids, tuv = oracle.trace(rays)
    for refill, leaf in ((1, 1), (64, 1), (24, 8), (40, 64), (64, 64)):
        r = _renderer(src)
        r.set_option(PT_OPT_REFILL_BELOW, refill)
        r.set_option(PT_OPT_LEAF_BATCH, leaf)
        hits = r.trace(rays)
        assert np.array_equal(hits["prim_type"], ids[:, 0]) and np.array_equal(hits["prim_id"], ids[:, 1]), (refill, leaf)
        assert np.array_equal(hits["inst_id"], ids[:, 2]), (refill, leaf)
        assert np.array_equal(hits["t"].view(np.uint32), tuv[:, 0].view(np.uint32)), (refill, leaf)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1023, 4097])
def test_trace_sizes_around_the_queue_chunk(n):
    from oracle.pyoracle import OracleScene
    src = cornell_small()
    oracle = OracleScene(src)
    rays = _rays(n, n, 0, 555)
    ids, tuv = oracle.trace(rays)
    hits = _renderer(src).trace(rays)
    assert hits.shape[0] == n
    assert np.array_equal(hits["prim_type"], ids[:, 0]) and np.array_equal(hits["prim_id"], ids[:, 1]) and np.array_equal(hits["inst_id"], ids[:, 2])
    assert np.array_equal(hits["t"].view(np.uint32), tuv[:, 0].view(np.uint32))


def test_option_ranges_are_checked():
    from pathtracer_amd.core import PT_OPT_LEAF_BATCH, PT_OPT_PIPES, PT_OPT_REFILL_BELOW, PtError
    r = _renderer(cornell_small())
    for key, bad in ((PT_OPT_PIPES, 0), (PT_OPT_PIPES, 9), (PT_OPT_REFILL_BELOW, 0), (PT_OPT_REFILL_BELOW, 65), (PT_OPT_LEAF_BATCH, 0)):
        with pytest.raises(PtError):
            r.set_option(key, bad)


def test_per_sample_radiance_identical_across_options():
    from pathtracer_amd.core import PT_OPT_LEAF_BATCH, PT_OPT_REFILL_BELOW
    src = small_mixed_scene()
    a = _renderer(src).render_samples(0, 3, seed=2)
    r = _renderer(src)
    r.set_option(PT_OPT_REFILL_BELOW, 1)
    r.set_option(PT_OPT_LEAF_BATCH, 1)
    b = r.render_samples(0, 3, seed=2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_tail_option_is_accepted_and_changes_nothing():
    # PT_OPT_TAIL_BELOW belonged to an earlier schedule; callers that still set it get the same sums
    from pathtracer_amd.core import PT_OPT_TAIL_BELOW
    src = scenes.small_mixed(32, 36)
    accs = []
    for tail in (0, 600, 1_000_000_000):
        r = _renderer(src)
        r.set_option(PT_OPT_TAIL_BELOW, tail)
        r.render(0, 31, seed=5)
        accs.append(r.read_accum()[0])
    assert np.array_equal(accs[0].view(np.uint32), accs[1].view(np.uint32))
    assert np.array_equal(accs[0].view(np.uint32), accs[2].view(np.uint32))


@pytest.mark.parametrize("make,width", [(scenes.cornell, 48), (scenes.interior, 64)])
def test_accumulator_identical_however_few_rays_may_be_in_flight(make, width):
    # PT_OPT_LIVE_FACTOR sizes the travelling path state as a multiple of the paths started per launch.
    # These scenes keep 2.5 to 4 rays alive per started path, so a factor of 2 makes the host hold
    # batches back until earlier ones have thinned out; the sums must not notice, and no queue may
    # run out of room (sync reports that as an error).
    from pathtracer_amd.core import PT_OPT_LIVE_FACTOR, PT_OPT_PIPES, PT_OPT_ROLLING
    src = make(width, 36)
    npix = src.camera.image_width * max(1, int(src.camera.image_width / src.camera.aspect_ratio))
    ref = None
    for pipes, rolling, factor in ((1, 0, 4), (1, npix, 2), (1, 3 * npix, 2), (2, 2 * npix, 2), (1, npix, 3), (2, npix, 50), (1, 5 * npix + 7, 2.5)):
        r = _renderer(src)
        r.set_option(PT_OPT_PIPES, pipes)
        r.set_option(PT_OPT_ROLLING, rolling)
        r.set_option(PT_OPT_LIVE_FACTOR, factor)
        r.render(0, 20, seed=5)
        r.render(20, 36, seed=5)
        acc, n = r.read_accum()
        assert n == 36
        if ref is None:
            ref = acc
        else:
            assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"pipes={pipes} rolling={rolling} factor={factor}"


@pytest.mark.parametrize("make,width", [(scenes.small_mixed, 32), (scenes.cornell, 48)])
def test_accumulator_identical_with_batches_sharing_launches(make, width):
    # PT_OPT_ROLLING = paths started per launch; 0 = every batch has its launches to itself
    from pathtracer_amd.core import PT_OPT_PIPES, PT_OPT_ROLLING
    src = make(width, 36)
    npix = width * width
    ref = None
    for pipes, rolling in ((1, 0), (1, npix), (1, 3 * npix), (2, 2 * npix), (3, npix), (4, 1 << 22)):
        r = _renderer(src)
        r.set_option(PT_OPT_PIPES, pipes)
        r.set_option(PT_OPT_ROLLING, rolling)
        r.render(0, 16, seed=5)
        r.render(16, 17, seed=5)
        r.render(17, 31, seed=5)
        acc, n = r.read_accum()
        assert n == 31
        if ref is None:
            ref = acc
        else:
            assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"pipes={pipes} rolling={rolling}"


def test_rolling_schedule_survives_setting_changes_between_calls():
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY, PT_OPT_ROLLING
    src = scenes.small_mixed(32, 36)
    accs = []
    for rolling in (0, 2 * 32 * 32):
        r = _renderer(src)
        r.set_option(PT_OPT_ROLLING, rolling)
        r.render(0, 9, seed=1)
        r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, 0.5)  # applies to the batches started after it
        r.render(9, 20, seed=2)
        accs.append(r.read_accum()[0])
    assert np.array_equal(accs[0].view(np.uint32), accs[1].view(np.uint32))


def test_deep_paths_give_the_same_sums_under_both_schedules():
    # 300 bounces allowed: 300 regions of per-slot sums under the rolling schedule
    from pathtracer_amd.core import PT_OPT_ROLLING
    src = scenes.cornell(32, 16, max_depth=300)
    accs = []
    for rolling in (1 << 20, 0):
        r = _renderer(src)
        r.set_option(PT_OPT_ROLLING, rolling)
        r.render(0, 9, seed=4)
        accs.append(r.read_accum()[0])
    assert np.array_equal(accs[0].view(np.uint32), accs[1].view(np.uint32))


def test_new_option_ranges_are_checked():
    from pathtracer_amd.core import PT_OPT_DEVICE_BVH, PT_OPT_INTEGRATOR, PT_OPT_LIVE_FACTOR, PT_OPT_ROLLING, PT_OPT_TAIL_BELOW, PtError
    r = _renderer(cornell_small())
    for key, bad in ((PT_OPT_ROLLING, -1), (PT_OPT_TAIL_BELOW, -1), (PT_OPT_LIVE_FACTOR, 1.5), (PT_OPT_INTEGRATOR, 2), (PT_OPT_INTEGRATOR, 0.5), (PT_OPT_DEVICE_BVH, -3)):
        with pytest.raises(PtError):
            r.set_option(key, bad)


def _packet_scenes():
    from tests.test_motion import _scene as moving
    # worlds without an instance buffer: triangles in a BVH, spheres and quads with and without one, a lens,
    # an image narrower than a wave, moving spheres
    yield "knob", scenes.knob(width=80, spp=36, max_depth=12, detail=0.2)
    yield "interior", scenes.interior(width=96, spp=36, max_depth=8, n_lights=9, columns=3)
    yield "lucy", scenes.lucy(width=72, spp=36, max_depth=8, detail=0.1)
    yield "moving", moving((-2.0, 0.0, 0.0), (2.0, 0.5, 0.0), width=40, spp=36, depth=6, extras=True)
    yield "one_row", scenes.knob(width=7, spp=36, max_depth=5, detail=0.1)
    yield "cornell", scenes.cornell(56, 36)  # two instances
    yield "next_week", scenes.next_week_like(48, 36, max_depth=12, extensions=True)  # 401 instances, media, a moving sphere
    yield "small_mixed", scenes.small_mixed(40, 36)


@pytest.mark.parametrize("name", ["knob", "interior", "lucy", "moving", "one_row", "cornell", "next_week", "small_mixed"])
def test_camera_rays_searched_a_wave_at_a_time_give_the_same_bits(name):
    # PT_OPT_PACKET: per-sample radiance (one batch at a time) and the accumulator (rolling schedule) with and without
    from pathtracer_amd.core import PT_OPT_PACKET, PT_OPT_ROLLING
    src = dict(_packet_scenes())[name]
    got = []
    for packet in (0, 1):
        r = _renderer(src)
        r.set_option(PT_OPT_PACKET, packet)
        samples = r.render_samples(0, 5, seed=3)
        r.render(0, 20, seed=7)
        r.render(20, 36, seed=7)
        acc, n = r.read_accum()
        assert n == 36
        r.set_option(PT_OPT_ROLLING, 0)
        r.reset_accum()
        r.render(0, 36, seed=7)
        got.append((samples, acc, r.read_accum()[0]))
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(got[1][1].view(np.uint32), got[1][2].view(np.uint32))  # both schedules, packets on


@pytest.mark.parametrize("make,width", [(scenes.small_mixed, 32), (scenes.cornell, 48), (scenes.extensions_demo, 40)])
def test_accumulator_identical_however_early_the_last_paths_of_a_batch_are_finished(make, width):
    # PT_OPT_BATCH_TICKS: 1 = a batch keeps its slots for max_depth launches, n = what is left after n launches is
    # finished in one go, 0 = learnt from the first drain (so the second half of the render runs with what was learnt)
    from pathtracer_amd.core import PT_OPT_BATCH_TICKS, PT_OPT_ROLLING
    src = make(width, 36)
    ref = None
    for rolling, ticks in ((0, 1), (1 << 22, 1), (1 << 22, 2), (1 << 22, 3), (1 << 22, 7), (1 << 22, 0), (3 * width * width, 0), (3 * width * width, 2)):
        r = _renderer(src)
        r.set_option(PT_OPT_ROLLING, rolling)
        r.set_option(PT_OPT_BATCH_TICKS, ticks)
        r.render(0, 20, seed=5)
        r.sync()
        r.render(20, 29, seed=5)
        r.render(29, 36, seed=5)
        acc, n = r.read_accum()
        assert n == 36
        if ref is None:
            ref = acc
        else:
            assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"rolling={rolling} ticks={ticks}"


def test_leftover_paths_beyond_the_finishing_list_are_found_by_the_full_scan(monkeypatch):
    """What is left of a batch when it is due is listed by a scan and finished from the list; a list that is too short
    (PT_FINISH_LIST_CAP, read when the path state is allocated) leaves the rest in the queue for the scanning form of
    the finishing kernel.  With two launches per batch nearly every path is left over: same bits as without a limit
    and as the schedule that never finishes early."""
    from pathtracer_amd.core import PT_OPT_BATCH_TICKS, PT_OPT_ROLLING
    src = scenes.small_mixed(40, 36)
    accs = []
    for cap, ticks in ((None, 1), (None, 2), ("8", 2), ("1", 3), ("100000", 2)):
        if cap is None:
            monkeypatch.delenv("PT_FINISH_LIST_CAP", raising=False)
        else:
            monkeypatch.setenv("PT_FINISH_LIST_CAP", cap)
        r = _renderer(src)
        r.set_option(PT_OPT_ROLLING, 3 * 40 * 40)
        r.set_option(PT_OPT_BATCH_TICKS, ticks)
        r.render(0, 20, seed=5)
        r.render(20, 36, seed=5)
        acc, n = r.read_accum()
        assert n == 36
        accs.append(acc)
        r.close()
    for a in accs[1:]:
        assert np.array_equal(a.view(np.uint32), accs[0].view(np.uint32))


def test_batch_length_is_learnt_from_the_first_drain():
    # A scene whose paths end quickly at a depth limit of 50: the first drain shows that nearly nothing is left
    # after a dozen launches, and the per-path sums shrink from 50 regions to that many (PT_OPT_BATCH_TICKS = 0,
    # the default); with 1 they keep max_depth regions.  Same bits either way.
    from pathtracer_amd.core import PT_OPT_BATCH_TICKS, PT_OPT_ROLLING
    src = scenes.knob(width=128, spp=64, max_depth=50, detail=0.2)
    out = {}
    for ticks in (1, 0):
        r = _renderer(src)
        r.set_option(PT_OPT_ROLLING, 4 * 128 * 128)
        r.set_option(PT_OPT_BATCH_TICKS, ticks)
        r.render(0, 24, seed=2)
        r.sync()
        before = r.stats()["device_bytes"]
        r.render(24, 64, seed=2)
        acc, n = r.read_accum()
        out[ticks] = (acc, before, r.stats()["device_bytes"])
        assert n == 64
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    regions_bytes = 16 * 4 * 128 * 128  # one region of sums
    assert out[1][1] - out[0][1] >= 20 * regions_bytes, (out[1][1], out[0][1])  # at least 20 of the 50 regions were given back
    assert out[1][2] == out[1][1]  # nothing changes where max_depth is asked for


@pytest.mark.parametrize("make,width", [(scenes.small_mixed, 32), (scenes.cornell, 48), (scenes.extensions_demo, 40)])
def test_accumulator_identical_however_the_shading_kernels_share_out_their_queues(make, width):
    # PT_OPT_SHADE_CLAIM: pieces dealt out in advance (0), drawn in blocks of n chunks from a counter per XCD (n),
    # or chosen step by step (-1, the default); both schedules, both estimators
    from pathtracer_amd.core import PT_OPT_INTEGRATOR, PT_OPT_ROLLING, PT_OPT_SHADE_CLAIM
    src = make(width, 36)
    for integrator in (0, 1):
        ref = None
        for rolling, claim in ((1 << 22, 0), (1 << 22, -1), (1 << 22, 1), (1 << 22, 4), (3 * width * width, 2), (3 * width * width, 64), (0, 0), (0, 3)):
            r = _renderer(src)
            r.set_option(PT_OPT_INTEGRATOR, integrator)
            r.set_option(PT_OPT_ROLLING, rolling)
            r.set_option(PT_OPT_SHADE_CLAIM, claim)
            r.render(0, 20, seed=5)
            r.render(20, 36, seed=5)
            acc, n = r.read_accum()
            assert n == 36
            if ref is None:
                ref = acc
            else:
                assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"integrator={integrator} rolling={rolling} claim={claim}"
