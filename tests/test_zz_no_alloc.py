"""pt_trace_first_hit_static (the first-hit query that allocates nothing) against pt_trace_first_hit.

Both run the same kernels; the static one keeps its path state in a fixed device array that every call on a
GPU shares and that keeps the last call's contents, so the test makes two calls in a row, the second with a
smaller input, and asks for bit-identical hit records each time.  Named so that pytest collects it last.
"""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.core import HIT_DTYPE, HostScene, PtError, Renderer

pytestmark = pytest.mark.gpu

MAX_RAYS = 262144  # PT_STATIC_TRACE_MAX_RAYS


class DeviceBuffer:
    """Device memory from the HIP runtime that libptcore is linked against (the test owns it, not the library)."""

    def __init__(self, nbytes: int):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), nbytes) == 0
        self.nbytes = nbytes

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        assert self.hip.hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0

    def download(self, shape, dtype) -> np.ndarray:
        out = np.zeros(shape, dtype)
        assert out.nbytes <= self.nbytes
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2) == 0
        return out

    def fill(self, byte: int):
        assert self.hip.hipMemset(self.ptr, byte, self.nbytes) == 0

    def free(self):
        if self.ptr:
            self.hip.hipFree(self.ptr)
            self.ptr = C.c_void_p()


def _rays(src_rays: np.ndarray, n: int, seed: int) -> np.ndarray:
    """n rays: camera rays of the scene, then rays between random points of its bounding region."""
    rng = np.random.default_rng(seed)
    cam = src_rays[rng.integers(0, src_rays.shape[0], n // 2)]
    lo, hi = -50.0, 600.0
    o = rng.uniform(lo, hi, (n - n // 2, 3)).astype(np.float32)
    d = (rng.uniform(lo, hi, (n - n // 2, 3)).astype(np.float32) - o)
    extra = np.concatenate([o, d, np.full((n - n // 2, 1), 0.001, np.float32), np.full((n - n // 2, 1), np.inf, np.float32)], axis=1)
    return np.concatenate([cam, extra]).astype(np.float32)


@pytest.mark.parametrize("scene", ["cornell", "knob", "small_mixed"])
def test_static_first_hits_equal_the_allocating_entry_point(scene):
    src = {"cornell": lambda: scenes.cornell(width=96, spp=4, max_depth=8),
           "knob": lambda: scenes.knob(width=96, spp=4, detail=0.25),
           "small_mixed": lambda: scenes.small_mixed(width=96, spp=4)}[scene]()
    r = Renderer(HostScene(src), device=0)
    primary = r.generate_rays(0, seed=3)
    d_rays = DeviceBuffer(MAX_RAYS * 32)
    d_out = DeviceBuffer(MAX_RAYS * HIT_DTYPE.itemsize)
    try:
        for n, seed in ((200_000, 1), (5_000, 2)):  # the second call finds the first one's leftovers in the array
            rays = _rays(primary, n, seed)
            if scene != "cornell":  # the random rays are scaled to the scene
                rays[n // 2:, :6] *= 0.05
            want = r.trace(rays)
            d_rays.upload(rays)
            d_out.fill(0xAB)
            r.trace_static(d_rays.ptr.value, n, d_out.ptr.value)
            got = d_out.download(n, HIT_DTYPE)
            assert (want["prim_type"] >= 0).sum() > n // 10, "the ray set should hit the scene"
            for f in ("prim_type", "prim_id", "inst_id"):
                assert np.array_equal(got[f], want[f]), f
            for f in ("t", "u", "v"):
                assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
        # the largest size it takes, and one beyond
        rays = _rays(primary, MAX_RAYS, 3)
        if scene != "cornell":
            rays[MAX_RAYS // 2:, :6] *= 0.05
        d_rays.upload(rays)
        r.trace_static(d_rays.ptr.value, MAX_RAYS, d_out.ptr.value)
        got = d_out.download(MAX_RAYS, HIT_DTYPE)
        want = r.trace(rays)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        with pytest.raises(PtError, match="at most"):
            r.trace_static(d_rays.ptr.value, MAX_RAYS + 1, d_out.ptr.value)
    finally:
        d_rays.free()
        d_out.free()
        r.close()


def test_static_first_hits_while_a_render_is_in_flight():
    """The query uses its own array, so rendering enqueued before it is not disturbed and does not disturb it."""
    src = scenes.cornell(width=128, spp=16, max_depth=8)
    host = HostScene(src)
    r = Renderer(host, device=0)
    rays = r.generate_rays(1, seed=9)
    n = rays.shape[0]
    want = r.trace(rays)
    r.render(0, 16, seed=4)
    ref_accum, _ = r.read_accum()
    r.reset_accum()
    d_rays = DeviceBuffer(n * 32)
    d_out = DeviceBuffer(n * HIT_DTYPE.itemsize)
    try:
        d_rays.upload(rays)
        r.render(0, 16, seed=4)  # enqueued, not waited for
        r.trace_static(d_rays.ptr.value, n, d_out.ptr.value)
        got = d_out.download(n, HIT_DTYPE)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        accum, _ = r.read_accum()
        assert np.array_equal(accum.view(np.uint32), ref_accum.view(np.uint32))
    finally:
        d_rays.free()
        d_out.free()
        r.close()


def test_two_contexts_on_two_threads_take_turns_on_the_shared_array():
    """Every context of a GPU uses the same fixed array; the library lets one call at a time through, so two host
    threads that call at once each get the hits they get alone."""
    import threading
    src_a, src_b = scenes.cornell(width=96, spp=4, max_depth=8), scenes.knob(width=96, spp=4, detail=0.25)
    ra, rb = Renderer(HostScene(src_a), device=0), Renderer(HostScene(src_b), device=0)
    n = 150_000
    rays_a = _rays(ra.generate_rays(0, seed=3), n, 5)
    rays_b = _rays(rb.generate_rays(0, seed=3), n, 6)
    rays_b[n // 2:, :6] *= 0.05
    want_a, want_b = ra.trace(rays_a), rb.trace(rays_b)
    bufs = [DeviceBuffer(n * 32), DeviceBuffer(n * HIT_DTYPE.itemsize), DeviceBuffer(n * 32), DeviceBuffer(n * HIT_DTYPE.itemsize)]
    try:
        bufs[0].upload(rays_a)
        bufs[2].upload(rays_b)
        errors = []

        def work(r, d_rays, d_out, want, rounds=6):
            try:
                for _ in range(rounds):
                    d_out.fill(0xAB)
                    r.trace_static(d_rays.ptr.value, n, d_out.ptr.value)
                    got = d_out.download(n, HIT_DTYPE)
                    if not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
                        errors.append("hits differ")
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        ta = threading.Thread(target=work, args=(ra, bufs[0], bufs[1], want_a))
        tb = threading.Thread(target=work, args=(rb, bufs[2], bufs[3], want_b))
        ta.start()
        tb.start()
        ta.join()
        tb.join()
        assert not errors, errors
    finally:
        for b in bufs:
            b.free()
        ra.close()
        rb.close()
