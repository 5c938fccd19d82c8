"""The multi-GPU group of the C ABI, the parts that need no GPU: the stratum split behind pt_group_render and the
error path of pt_group_create."""
import ctypes as C

import pytest

from pathtracer_amd import core
from pathtracer_amd.sharding import stratum_ranges


def test_group_split_is_the_contiguous_split_of_the_process_per_gpu_path():
    for total, world in [(10201, 8), (10201, 1), (64, 3), (5, 8), (0, 2), (1024, 7), (51076, 8)]:
        want = stratum_ranges(total, world)
        got = [core.group_split(0, total, world, i) for i in range(world)]
        assert got == want
    # a sub-range of the stratum list, as a progressive render passes it
    pieces = [core.group_split(100, 1000, 4, i) for i in range(4)]
    assert pieces == [(100, 325), (325, 550), (550, 775), (775, 1000)]
    # out-of-range member numbers and a reversed range give empty pieces
    assert core.group_split(0, 10, 4, 4) == (0, 0) and core.group_split(0, 10, 0, 0) == (0, 0)
    assert all(b == e for b, e in (core.group_split(10, 5, 2, i) for i in range(2)))


def test_group_create_reports_errors_instead_of_crashing():
    L = core.core_lib()
    h = C.c_void_p()
    assert L.pt_group_create(None, 1, C.byref(h)) != 0 and not h.value
    devs = (C.c_int * 1)(0)
    assert L.pt_group_create(devs, 0, C.byref(h)) != 0 and not h.value
    assert L.pt_group_create(devs, 1, None) != 0
    import torch
    if not torch.cuda.is_available():  # without a HIP device there is no CPU fallback: the call fails and says why
        rc = L.pt_group_create(devs, 1, C.byref(h))
        assert rc != 0 and not h.value
        assert b"device" in L.pt_group_last_error(None).lower()
    L.pt_group_destroy(None)
    assert L.pt_group_size(None) == 0 and not L.pt_group_member(None, 0)
