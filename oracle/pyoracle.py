"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (``oracle/liboracle.so``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
import this module.  The shipped renderer never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "liboracle.so")


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.pto_load.restype = C.c_void_p
        L.pto_load.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.pto_free.argtypes = [C.c_void_p]
        L.pto_set_bsdf_probability.argtypes = [C.c_void_p, C.c_float]
        L.pto_camera_constants.argtypes = [C.c_void_p, C.c_void_p]
        L.pto_buffer_len.restype = C.c_uint32
        L.pto_buffer_len.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pto_bvh_node_count.restype = C.c_uint32
        L.pto_bvh_node_count.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pto_bvh_nodes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.pto_perm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pto_quads.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.pto_triangles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.pto_instances.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.pto_primary_rays.argtypes = [C.c_void_p, C.c_void_p]
        L.pto_sample_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pto_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.pto_trace_at.restype = None
        L.pto_trace_at.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.pto_trace_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_render_counters.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pto_to_rgb8.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]
        L.pto_refract.restype = C.c_int
        L.pto_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.pto_fresnel_o.restype = C.c_float
        L.pto_fresnel_o.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.pto_onb.argtypes = [C.c_void_p, C.c_void_p]
        L.pto_material_eval.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6
        L.pto_material_scatter.restype = C.c_uint32
        L.pto_material_scatter.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                           C.c_uint32, C.c_void_p]
        L.pto_material_probe.restype = None
        L.pto_material_probe.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        L.pto_texture_value.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p]
        L.pto_light_pdf.restype = C.c_float
        L.pto_light_pdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_light_pdfs.restype = None
        L.pto_light_pdfs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.pto_rng_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f3(x):
    return np.ascontiguousarray(x, dtype=np.float32)


class OracleScene:
    """The reference's data structures built from a SceneSource by the restatement."""

    def __init__(self, src):
        self.src = src
        sc, cam = src.to_c()
        err = C.create_string_buffer(256)
        self.h = lib().pto_load(C.byref(sc), C.byref(cam), err, 256)
        if not self.h:
            raise RuntimeError("oracle: " + err.value.decode())
        k = self.camera_constants()
        self.width, self.height = int(k[0]), int(k[1])
        self.sqrt_spp = int(k[21])

    def __del__(self):
        if getattr(self, "h", None):
            lib().pto_free(self.h)
            self.h = None

    def set_bsdf_probability(self, p: float):
        lib().pto_set_bsdf_probability(self.h, p)

    def camera_constants(self) -> np.ndarray:
        out = np.zeros(27, np.float32)
        lib().pto_camera_constants(self.h, _p(out))
        return out

    def buffer_len(self, set_: int, type_: int) -> int:
        return lib().pto_buffer_len(self.h, set_, type_)

    def bvh(self, set_: int, type_: int):
        n = lib().pto_bvh_node_count(self.h, set_, type_)
        bbox = np.zeros((n, 6), np.float32)
        meta = np.zeros((n, 3), np.uint32)
        if n:
            lib().pto_bvh_nodes(self.h, set_, type_, _p(bbox), _p(meta))
        return bbox, meta

    def perm(self, set_: int, type_: int) -> np.ndarray:
        out = np.zeros(self.buffer_len(set_, type_), np.uint32)
        if out.size:
            lib().pto_perm(self.h, set_, type_, _p(out))
        return out

    def quads(self, set_: int) -> np.ndarray:
        out = np.zeros((self.buffer_len(set_, 2), 17), np.float32)
        if out.size:
            lib().pto_quads(self.h, set_, _p(out))
        return out

    def triangles(self, set_: int) -> np.ndarray:
        out = np.zeros((self.buffer_len(set_, 0), 24), np.float32)
        if out.size:
            lib().pto_triangles(self.h, set_, _p(out))
        return out

    def instances(self, set_: int) -> np.ndarray:
        out = np.zeros((self.buffer_len(set_, 3), 32), np.float32)
        if out.size:
            lib().pto_instances(self.h, set_, _p(out))
        return out

    def primary_rays(self) -> np.ndarray:
        out = np.zeros((self.width * self.height, 8), np.float32)
        lib().pto_primary_rays(self.h, _p(out))
        return out

    def sample_rays(self, seed: int, stratum: int):
        out = np.zeros((self.width * self.height, 8), np.float32)
        ctr = np.zeros(self.width * self.height, np.uint32)
        lib().pto_sample_rays(self.h, seed, stratum, _p(out), _p(ctr))
        return out, ctr

    def trace(self, rays: np.ndarray, counters: bool = False, threads: int = 0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        ids = np.zeros((n, 3), np.int32)
        tuv = np.zeros((n, 3), np.float32)
        ctr = np.zeros(5, np.uint64) if counters else None
        lib().pto_trace(self.h, _p(rays), n, _p(ids), _p(tuv), _p(ctr), threads)
        return (ids, tuv, ctr) if counters else (ids, tuv)

    def trace_at(self, rays: np.ndarray, times: Optional[np.ndarray] = None):
        """Closest hits including the moving spheres at the given times (prim_type 4, prim_id = index of the moving sphere)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        times = None if times is None else np.ascontiguousarray(times, np.float32)
        ids = np.zeros((n, 3), np.int32)
        tuv = np.zeros((n, 3), np.float32)
        lib().pto_trace_at(self.h, _p(rays), _p(times), n, _p(ids), _p(tuv))
        return ids, tuv

    def trace_counts(self, rays: np.ndarray):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        ids = np.zeros((n, 3), np.int32)
        tuv = np.zeros((n, 3), np.float32)
        per_ray = np.zeros((n, 2), np.uint32)
        hist = np.zeros(65, np.uint64)
        lib().pto_trace_counts(self.h, _p(rays), n, _p(ids), _p(tuv), _p(per_ray), _p(hist))
        return ids, tuv, per_ray, hist

    def render(self, stratum_begin: int, stratum_end: int, seed: int = 0, threads: int = 0, iterative: bool = False,
               tonemap_each_pass: bool = False, per_sample: bool = False, accum: Optional[np.ndarray] = None,
               want_rgb8: bool = False):
        npix = self.width * self.height
        if accum is None:
            accum = np.zeros((npix, 3), np.float32)
        rgb8 = np.zeros((npix, 3), np.uint8) if want_rgb8 else None
        ps = np.zeros((stratum_end - stratum_begin, npix, 3), np.float32) if per_sample else None
        rays = np.zeros(1, np.uint64)
        samples = np.zeros(1, np.uint32)
        lib().pto_render(self.h, stratum_begin, stratum_end, seed, threads, int(iterative), int(tonemap_each_pass),
                         _p(accum), _p(rgb8), _p(ps), _p(rays), _p(samples))
        return {"accum": accum, "rgb8": rgb8, "per_sample": ps, "rays": int(rays[0]), "samples": int(samples[0])}

    def render_counters(self, stratum_begin: int, stratum_end: int, seed: int = 0, pixel_stride: int = 1) -> np.ndarray:
        ctr = np.zeros(5, np.uint64)
        self.by_bounce = np.zeros((3, 64), np.uint64)
        lib().pto_render_counters(self.h, stratum_begin, stratum_end, seed, pixel_stride, _p(ctr), _p(self.by_bounce))
        return ctr

    def material_eval(self, mat: int, w_out, w_in, u=0.0, v=0.0, p=(0, 0, 0), front_face=True):
        rec = np.array([u, v, p[0], p[1], p[2], 1.0 if front_face else 0.0, 0.0], np.float32)
        bsdf = np.zeros(3, np.float32); pdf = np.zeros(1, np.float32); em = np.zeros(3, np.float32)
        lib().pto_material_eval(self.h, mat, _p(_f3(w_out)), _p(_f3(w_in)), _p(rec), _p(bsdf), _p(pdf), _p(em))
        return bsdf, float(pdf[0]), em

    def material_scatter(self, mat: int, w_out, seed=0, pixel=0, stratum=0, u=0.0, v=0.0, p=(0, 0, 0), front_face=True):
        rec = np.array([u, v, p[0], p[1], p[2], 1.0 if front_face else 0.0, 0.0], np.float32)
        w = np.zeros(3, np.float32)
        n = lib().pto_material_scatter(self.h, mat, _p(_f3(w_out)), _p(rec), seed, pixel, stratum, _p(w))
        return w, n

    def material_probe(self, mat: int, probes: np.ndarray, seed: int = 0) -> np.ndarray:
        """The oracle's side of pt_material_probe: probes[n] and the result use the structs of include/ptcore.h."""
        from pathtracer_amd.core import PROBE_IN_DTYPE, PROBE_OUT_DTYPE  # layouts only
        probes = np.ascontiguousarray(probes, PROBE_IN_DTYPE)
        out = np.zeros(probes.shape[0], PROBE_OUT_DTYPE)
        lib().pto_material_probe(self.h, mat, seed, _p(probes), probes.shape[0], _p(out))
        return out

    def texture_value(self, tex: int, u: float, v: float) -> np.ndarray:
        out = np.zeros(3, np.float32)
        lib().pto_texture_value(self.h, tex, u, v, _p(out))
        return out

    def light_pdf(self, origin, direction) -> float:
        return lib().pto_light_pdf(self.h, _p(_f3(origin)), _p(_f3(direction)))

    def light_pdfs(self, rays: np.ndarray, solid_angle_tris: bool = False) -> np.ndarray:
        """LightBuffer::pdf for rays[n][8] (origin, direction, two unused floats)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], np.float32)
        lib().pto_light_pdfs(self.h, _p(rays), rays.shape[0], int(solid_angle_tris), _p(out))
        return out


def refract(w_o, n, eta):
    out = np.zeros(3, np.float32)
    ok = lib().pto_refract(_p(_f3(w_o)), _p(_f3(n)), eta, _p(out))
    return out if ok else None


def fresnel_o(w_o, n, eta) -> float:
    return lib().pto_fresnel_o(_p(_f3(w_o)), _p(_f3(n)), eta)


def onb(n) -> np.ndarray:
    out = np.zeros(9, np.float32)
    lib().pto_onb(_p(_f3(n)), _p(out))
    return out.reshape(3, 3)  # rows = columns u, v, w of the basis matrix


def to_rgb8(rgb: np.ndarray, divisor: float = 1.0) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    out = np.zeros((rgb.shape[0], 3), np.uint8)
    lib().pto_to_rgb8(_p(rgb), rgb.shape[0], divisor, _p(out))
    return out


def rng_draws(seed: int, pixel: int, stratum: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint64)
    lib().pto_rng_draws(seed, pixel, stratum, n, _p(out))
    return out
