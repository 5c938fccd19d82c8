"""ctypes bindings of ``libpthost.so`` (host scene layer) and ``libptcore.so`` (HIP kernels).

This is plumbing only: scene construction, BVH building and rendering all happen in the two
native libraries.  There is no CPU fallback — creating a :class:`Renderer` without a usable
HIP device raises :class:`PtError`, and a missing library raises ``OSError`` at import of
the first symbol.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .scene_src import SceneSource, pts_image, pts_medium, pts_texture

_HERE = os.path.dirname(os.path.abspath(__file__))
_core: Optional[C.CDLL] = None
_host: Optional[C.CDLL] = None

PT_NODE_LEAF = 0x80000000
(PT_OPT_BSDF_SAMPLE_PROBABILITY, PT_OPT_MAX_PATHS, PT_OPT_PROFILE, PT_OPT_COUNTERS, PT_OPT_TRAVERSAL, PT_OPT_PIPES,
 PT_OPT_REFILL_BELOW, PT_OPT_LEAF_BATCH, PT_OPT_BLOCKS_PER_CU, PT_OPT_TAIL_BELOW, PT_OPT_ROLLING, PT_OPT_DEVICE_BVH, PT_OPT_INTEGRATOR,
 PT_OPT_LIVE_FACTOR, PT_OPT_PACKET, PT_OPT_BATCH_TICKS, PT_OPT_SHADE_CLAIM) = range(17)


class PtError(RuntimeError):
    pass


class pt_node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("a", C.c_uint32), ("b", C.c_uint32)]


class pt_buffer_desc(C.Structure):
    _fields_ = [("prim_first", C.c_uint32), ("prim_count", C.c_uint32), ("node_first", C.c_uint32), ("node_count", C.c_uint32)]


class pt_primset_desc(C.Structure):
    _fields_ = [("tris", pt_buffer_desc), ("spheres", pt_buffer_desc), ("quads", pt_buffer_desc), ("instances", pt_buffer_desc)]


class pt_instance_desc(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("inverse", C.c_float * 16), ("child_set", C.c_uint32), ("chain_first", C.c_uint32), ("chain_count", C.c_uint32), ("reserved", C.c_uint32)]


class pt_material_desc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("texture", C.c_int32), ("p", C.c_float * 14), ("derived", C.c_float * 8)]


class pt_scene_desc(C.Structure):
    _fields_ = [
        ("n_tris", C.c_uint32), ("tri_data", C.POINTER(C.c_float)), ("tri_mat", C.POINTER(C.c_int32)),
        ("n_spheres", C.c_uint32), ("spheres", C.POINTER(C.c_float)), ("sphere_mat", C.POINTER(C.c_int32)),
        ("n_quads", C.c_uint32), ("quads", C.POINTER(C.c_float)), ("quad_mat", C.POINTER(C.c_int32)),
        ("n_instances", C.c_uint32), ("instances", C.POINTER(pt_instance_desc)),
        ("n_nodes", C.c_uint32), ("nodes", C.POINTER(pt_node)),
        ("n_sets", C.c_uint32), ("sets", C.POINTER(pt_primset_desc)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(pt_material_desc)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(pts_texture)),
        ("n_images", C.c_uint32), ("images", C.POINTER(pts_image)),
        ("n_texel_bytes", C.c_uint64), ("texels", C.POINTER(C.c_uint8)),
        ("n_media", C.c_uint32), ("media", C.POINTER(pts_medium)),
        ("n_moving", C.c_uint32), ("moving", C.POINTER(C.c_float)),
    ]


class pt_camera_desc(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("center", C.c_float * 3), ("pixel00", C.c_float * 3),
                ("pixel_du", C.c_float * 3), ("pixel_dv", C.c_float * 3), ("defocus_disk_u", C.c_float * 3),
                ("defocus_disk_v", C.c_float * 3), ("defocus_angle", C.c_float), ("sqrt_spp", C.c_uint32),
                ("inv_sqrt_spp", C.c_float), ("max_depth", C.c_uint32), ("background", C.c_float * 3)]


class pt_hit(C.Structure):
    _fields_ = [("prim_type", C.c_int32), ("prim_id", C.c_int32), ("inst_id", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float)]


class pt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("extend_launches", C.c_uint64), ("extend_ms", C.c_double),
                ("extend_busy_ms", C.c_double), ("shade_ms", C.c_double), ("other_ms", C.c_double), ("render_ms", C.c_double), ("host_wait_ms", C.c_double), ("node_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("quad_tests", C.c_uint64), ("device_bytes", C.c_uint64),
                ("bad_refs", C.c_uint64), ("walk_ms", C.c_double), ("packet_ms", C.c_double), ("walk_launches", C.c_uint64), ("packet_launches", C.c_uint64), ("shadow_rays", C.c_uint64)]


PT_PRIM_TRI, PT_PRIM_SPHERE, PT_PRIM_QUAD = 0, 1, 2  # pt_hit::prim_type
HIT_DTYPE = np.dtype([("prim_type", np.int32), ("prim_id", np.int32), ("inst_id", np.int32), ("t", np.float32), ("u", np.float32), ("v", np.float32)])

PROBE_IN_DTYPE = np.dtype([("w_out", np.float32, 3), ("u", np.float32), ("v", np.float32), ("p", np.float32, 3), ("front_face", np.uint32),
                           ("pixel", np.uint32), ("stratum", np.uint32), ("w_in", np.float32, 3)])  # pt_material_probe_in
PROBE_OUT_DTYPE = np.dtype([("w_in", np.float32, 3), ("draws", np.uint32), ("bsdf", np.float32, 3), ("pdf", np.float32), ("bsdf_at", np.float32, 3),
                            ("pdf_at", np.float32), ("emitted", np.float32, 3), ("reserved", np.uint32)])  # pt_material_probe_out

# every symbol include/ptcore.h and include/pthost.h declare
CORE_SYMBOLS = ["pt_create", "pt_destroy", "pt_last_error", "pt_set_option", "pt_upload_scene", "pt_set_camera", "pt_render", "pt_sync",
                "pt_render_samples", "pt_generate_rays", "pt_trace_first_hit", "pt_trace_first_hit_at", "pt_trace_first_hit_static", "pt_light_pdf", "pt_material_probe", "pt_read_accum", "pt_read_rgb8", "pt_reset_accum", "pt_read_triangle_order",
                "pt_copy_accum_to_device", "pt_set_accum_from_device", "pt_stats_get", "pt_stats_reset",
                "pt_group_create", "pt_group_destroy", "pt_group_last_error", "pt_group_size", "pt_group_member", "pt_group_split", "pt_group_set_option",
                "pt_group_upload_scene", "pt_group_set_camera", "pt_group_render", "pt_group_sync", "pt_group_read_accum", "pt_group_read_rgb8",
                "pt_group_reset_accum", "pt_group_stats_get", "pt_group_stats_reset"]
HOST_SYMBOLS = ["pth_scene_build", "pth_scene_build_opts", "pth_scene_free", "pth_scene_desc", "pth_camera_desc", "pth_perm", "pth_camera_render",
                "pth_obj_load", "pth_obj_free", "pth_obj_mesh_count", "pth_obj_mesh_get", "pth_obj_triangles", "pth_image_load", "pth_image_save_png"]


def core_lib() -> C.CDLL:
    global _core
    if _core is None:
        # PTCORE_LIB: another build of the same library (the bounds-checking one, for debugging)
        L = C.CDLL(os.environ.get("PTCORE_LIB") or os.path.join(_HERE, "libptcore.so"), mode=C.RTLD_GLOBAL)
        vp = C.c_void_p
        L.pt_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.pt_destroy.argtypes = [vp]
        L.pt_destroy.restype = None
        L.pt_last_error.argtypes = [vp]
        L.pt_last_error.restype = C.c_char_p
        L.pt_set_option.argtypes = [vp, C.c_int, C.c_double]
        L.pt_sync.argtypes = [vp]
        L.pt_upload_scene.argtypes = [vp, vp]
        L.pt_set_camera.argtypes = [vp, vp]
        L.pt_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64]
        L.pt_render_samples.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, vp]
        L.pt_generate_rays.argtypes = [vp, C.c_uint32, C.c_uint64, vp]
        L.pt_trace_first_hit.argtypes = [vp, vp, C.c_uint64, vp]
        L.pt_trace_first_hit_at.argtypes = [vp, vp, vp, C.c_uint64, vp]
        L.pt_trace_first_hit_static.argtypes = [vp, vp, C.c_uint64, vp]
        L.pt_light_pdf.argtypes = [vp, vp, C.c_uint64, C.c_int, vp]
        L.pt_material_probe.argtypes = [vp, C.c_int32, C.c_uint64, vp, C.c_uint64, vp]
        L.pt_read_accum.argtypes = [vp, vp, vp]
        L.pt_read_rgb8.argtypes = [vp, vp]
        L.pt_read_triangle_order.argtypes = [vp, vp, C.c_uint32]
        L.pt_reset_accum.argtypes = [vp]
        L.pt_copy_accum_to_device.argtypes = [vp, vp]
        L.pt_set_accum_from_device.argtypes = [vp, vp, C.c_uint32]
        L.pt_stats_get.argtypes = [vp, vp]
        L.pt_stats_reset.argtypes = [vp]
        L.pt_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.pt_group_destroy.argtypes = [vp]
        L.pt_group_destroy.restype = None
        L.pt_group_last_error.argtypes = [vp]
        L.pt_group_last_error.restype = C.c_char_p
        L.pt_group_size.argtypes = [vp]
        L.pt_group_member.argtypes = [vp, C.c_int]
        L.pt_group_member.restype = vp
        L.pt_group_split.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_group_split.restype = None
        L.pt_group_set_option.argtypes = [vp, C.c_int, C.c_double]
        L.pt_group_upload_scene.argtypes = [vp, vp]
        L.pt_group_set_camera.argtypes = [vp, vp]
        L.pt_group_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64]
        L.pt_group_sync.argtypes = [vp]
        L.pt_group_read_accum.argtypes = [vp, vp, vp]
        L.pt_group_read_rgb8.argtypes = [vp, vp]
        L.pt_group_reset_accum.argtypes = [vp]
        L.pt_group_stats_get.argtypes = [vp, vp]
        L.pt_group_stats_reset.argtypes = [vp]
        _core = L
    return _core


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        core_lib()
        L = C.CDLL(os.path.join(_HERE, "libpthost.so"))
        vp = C.c_void_p
        L.pth_scene_build.argtypes = [vp, vp, C.c_char_p, C.c_size_t]
        L.pth_scene_build.restype = vp
        L.pth_scene_build_opts.argtypes = [vp, vp, C.c_uint32, C.c_char_p, C.c_size_t]
        L.pth_scene_build_opts.restype = vp
        L.pth_scene_free.argtypes = [vp]
        L.pth_scene_free.restype = None
        L.pth_scene_desc.argtypes = [vp]
        L.pth_scene_desc.restype = C.POINTER(pt_scene_desc)
        L.pth_camera_desc.argtypes = [vp]
        L.pth_camera_desc.restype = C.POINTER(pt_camera_desc)
        L.pth_perm.argtypes = [vp, C.c_int, C.c_int, vp]
        L.pth_perm.restype = C.c_uint32
        L.pth_camera_render.argtypes = [vp, vp, C.c_uint64, vp]
        L.pth_obj_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.pth_obj_load.restype = vp
        L.pth_obj_free.argtypes = [vp]
        L.pth_obj_free.restype = None
        L.pth_obj_mesh_count.argtypes = [vp]
        L.pth_obj_mesh_count.restype = C.c_uint32
        L.pth_obj_mesh_get.argtypes = [vp, C.c_uint32, vp]
        L.pth_obj_triangles.argtypes = [vp, C.c_uint32, vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pth_obj_triangles.restype = C.c_int64
        L.pth_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp, C.c_size_t, C.c_char_p, C.c_size_t]
        L.pth_image_save_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
        _host = L
    return _host


def load_obj(path: str):
    """TriMesh::load_as_vec + to_triangles (src/objects/tri_mesh.rs:299-415) through the host library:
    one ``(pos (n,3,3), nrm (n,3,3) or None, uv (n,3,2) or None)`` per model of the file, in file order.
    Raises PtError where the reference would log the failure and continue with no meshes."""
    L = host_lib()
    err = C.create_string_buffer(256)
    h = L.pth_obj_load(os.fsencode(path), err, len(err))
    if not h:
        raise PtError(f"cannot load {path}: {err.value.decode()}")
    try:
        out = []
        for i in range(L.pth_obj_mesh_count(h)):
            hn, hu = C.c_int(0), C.c_int(0)
            n = L.pth_obj_triangles(h, i, None, None, None, C.byref(hn), C.byref(hu))
            if n < 0:
                raise PtError(f"{path}: model {i} is not triangulated")
            pos = np.zeros((n, 3, 3), np.float32)
            nrm = np.zeros((n, 3, 3), np.float32) if hn.value else None
            uv = np.zeros((n, 3, 2), np.float32) if hu.value else None
            L.pth_obj_triangles(h, i, pos.ctypes.data_as(C.c_void_p), None if nrm is None else nrm.ctypes.data_as(C.c_void_p),
                                None if uv is None else uv.ctypes.data_as(C.c_void_p), None, None)
            out.append((pos, nrm, uv))
        return out
    finally:
        L.pth_obj_free(h)


def load_image(path: str) -> np.ndarray:
    """image::open(path)?.to_rgb8() of Image::load (src/textures/image.rs:17-26) through the host library: a PNG or PPM
    file as (height, width, 3) uint8, the array SceneSource.image takes.  Raises PtError where the reference would log
    the failure and fall back to its default texture (SceneSource.image(None) gives that texture)."""
    L = host_lib()
    err = C.create_string_buffer(256)
    w, h = C.c_uint32(0), C.c_uint32(0)
    if L.pth_image_load(os.fsencode(path), C.byref(w), C.byref(h), None, 0, err, len(err)):
        raise PtError(f"cannot load {err.value.decode()}")
    out = np.zeros((h.value, w.value, 3), np.uint8)
    if L.pth_image_load(os.fsencode(path), None, None, out.ctypes.data_as(C.c_void_p), out.nbytes, err, len(err)):
        raise PtError(f"cannot load {err.value.decode()}")
    return out


def save_png(path: str, rgb8: np.ndarray) -> None:
    """RgbImage::save("*.png") (src/main.rs:93): (height, width, 3) uint8 to an 8-bit RGB PNG file."""
    a = np.ascontiguousarray(rgb8, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected (height, width, 3) uint8")
    if host_lib().pth_image_save_png(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p)):
        raise PtError(f"cannot write {path}")


def _arr(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


NODE_DTYPE = np.dtype([("bmin", np.float32, 3), ("bmax", np.float32, 3), ("a", np.uint32), ("b", np.uint32)])


class HostScene:
    """A scene after the host layer built it: primitives in post-build order, flat BVHs, camera constants."""

    def __init__(self, src: SceneSource, device_bvh_from: int = 0):
        """device_bvh_from: triangle buffers of at least this many triangles are left without a BVH here and get
        one built on the device when a Renderer uploads the scene (0: the host builds every BVH)."""
        self.src = src
        self.device_bvh_from = int(device_bvh_from)
        sc, cam = src.to_c()
        err = C.create_string_buffer(512)
        self.h = host_lib().pth_scene_build_opts(C.byref(sc), C.byref(cam), self.device_bvh_from, err, 512)
        if not self.h:
            raise PtError("pth_scene_build: " + err.value.decode())
        self.desc = host_lib().pth_scene_desc(self.h).contents
        self.camera = host_lib().pth_camera_desc(self.h).contents
        self.width, self.height = self.camera.image_width, self.camera.image_height
        self.sqrt_spp = self.camera.sqrt_spp

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().pth_scene_free(self.h)
            self.h = None

    # --- views of the flat description (read-only numpy copies) -------------------------
    def set_desc(self, k: int) -> pt_primset_desc:
        return self.desc.sets[k]

    def nodes(self, buf: pt_buffer_desc) -> np.ndarray:
        allnodes = _arr(self.desc.nodes, (self.desc.n_nodes,), NODE_DTYPE)
        return allnodes[buf.node_first:buf.node_first + buf.node_count].copy()

    def quads(self, buf: pt_buffer_desc) -> np.ndarray:
        q = _arr(self.desc.quads, (self.desc.n_quads, 17), np.float32)
        return q[buf.prim_first:buf.prim_first + buf.prim_count].copy()

    def triangles(self, buf: pt_buffer_desc) -> np.ndarray:
        t = _arr(self.desc.tri_data, (self.desc.n_tris, 24), np.float32)
        return t[buf.prim_first:buf.prim_first + buf.prim_count].copy()

    def instances(self, buf: pt_buffer_desc) -> np.ndarray:
        out = np.zeros((buf.prim_count, 32), np.float32)
        for k in range(buf.prim_count):
            inst = self.desc.instances[buf.prim_first + k]
            out[k, :16] = inst.transform[:]
            out[k, 16:] = inst.inverse[:]
        return out

    def perm(self, set_: int, type_: int) -> np.ndarray:
        n = host_lib().pth_perm(self.h, set_, type_, None)
        out = np.zeros(n, np.uint32)
        if n:
            host_lib().pth_perm(self.h, set_, type_, out.ctypes.data_as(C.c_void_p))
        return out

    def camera_constants(self) -> np.ndarray:
        c = self.camera
        return np.array([c.image_width, c.image_height, *c.center, *c.pixel00, *c.pixel_du, *c.pixel_dv, *c.defocus_disk_u,
                         *c.defocus_disk_v, c.defocus_angle, c.sqrt_spp, c.inv_sqrt_spp, c.max_depth, *c.background], np.float32)


class Renderer:
    """One ``pt_context`` on one GPU with a scene and a camera loaded."""

    def __init__(self, scene: HostScene, device: int = 0):
        self.scene = scene
        self.L = core_lib()
        h = C.c_void_p()
        rc = self.L.pt_create(device, C.byref(h))
        if rc:
            raise PtError("pt_create: " + self.L.pt_last_error(None).decode())
        self.ctx = h
        if getattr(scene, "device_bvh_from", 0):
            self.set_option(PT_OPT_DEVICE_BVH, scene.device_bvh_from)
        self._check(self.L.pt_upload_scene(self.ctx, C.byref(scene.desc)), "pt_upload_scene")
        self._check(self.L.pt_set_camera(self.ctx, C.byref(scene.camera)), "pt_set_camera")
        self.width, self.height, self.sqrt_spp = scene.width, scene.height, scene.sqrt_spp
        self.npix = self.width * self.height

    def load(self, scene: HostScene):
        """Another scene and camera into the same context (pt_upload_scene replaces the earlier scene)."""
        self._check(self.L.pt_upload_scene(self.ctx, C.byref(scene.desc)), "pt_upload_scene")
        self._check(self.L.pt_set_camera(self.ctx, C.byref(scene.camera)), "pt_set_camera")
        self.scene = scene
        self.width, self.height, self.sqrt_spp = scene.width, scene.height, scene.sqrt_spp
        self.npix = self.width * self.height

    def _check(self, rc: int, what: str):
        if rc:
            raise PtError(f"{what} failed ({rc}): " + self.L.pt_last_error(self.ctx).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.L.pt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def set_option(self, key: int, value: float):
        self._check(self.L.pt_set_option(self.ctx, key, float(value)), "pt_set_option")

    def render(self, stratum_begin: int, stratum_end: int, seed: int = 0):
        """Enqueues the strata; returns without waiting for the device (see ``sync``)."""
        self._check(self.L.pt_render(self.ctx, stratum_begin, stratum_end, seed), "pt_render")

    def sync(self):
        self._check(self.L.pt_sync(self.ctx), "pt_sync")

    def triangle_order(self) -> np.ndarray:
        """order[i] = index in the uploaded triangle arrays of the triangle that hit records call i."""
        out = np.zeros(self.scene.desc.n_tris, np.uint32)
        self._check(self.L.pt_read_triangle_order(self.ctx, out.ctypes.data_as(C.c_void_p), out.size), "pt_read_triangle_order")
        return out

    def render_samples(self, stratum_begin: int, stratum_end: int, seed: int = 0) -> np.ndarray:
        out = np.zeros((stratum_end - stratum_begin, self.npix, 3), np.float32)
        self._check(self.L.pt_render_samples(self.ctx, stratum_begin, stratum_end, seed, out.ctypes.data_as(C.c_void_p)), "pt_render_samples")
        return out

    def generate_rays(self, stratum: int, seed: int = 0) -> np.ndarray:
        out = np.zeros((self.npix, 8), np.float32)
        self._check(self.L.pt_generate_rays(self.ctx, stratum, seed, out.ctypes.data_as(C.c_void_p)), "pt_generate_rays")
        return out

    def trace(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], HIT_DTYPE)
        self._check(self.L.pt_trace_first_hit(self.ctx, rays.ctypes.data_as(C.c_void_p), rays.shape[0], out.ctypes.data_as(C.c_void_p)), "pt_trace_first_hit")
        return out

    def trace_at(self, rays: np.ndarray, times: Optional[np.ndarray] = None) -> np.ndarray:
        """pt_trace_first_hit_at: closest hits with the moving spheres where they are at times[i] (None: time 0)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        t = None if times is None else np.ascontiguousarray(times, np.float32)
        out = np.zeros(rays.shape[0], HIT_DTYPE)
        self._check(self.L.pt_trace_first_hit_at(self.ctx, rays.ctypes.data_as(C.c_void_p), None if t is None else t.ctypes.data_as(C.c_void_p), rays.shape[0],
                                                 out.ctypes.data_as(C.c_void_p)), "pt_trace_first_hit_at")
        return out

    def trace_static(self, rays_device: int, n: int, out_device: int):
        """pt_trace_first_hit_static: rays and hits are device addresses; nothing is allocated."""
        self._check(self.L.pt_trace_first_hit_static(self.ctx, C.c_void_p(rays_device), n, C.c_void_p(out_device)), "pt_trace_first_hit_static")

    def light_pdf(self, rays: np.ndarray, tris_by_solid_angle: bool = False) -> np.ndarray:
        """LightBuffer::pdf of the reference (src/objects.rs:259-278) for rays[n][8]; only origin and direction count."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], np.float32)
        self._check(self.L.pt_light_pdf(self.ctx, rays.ctypes.data_as(C.c_void_p), rays.shape[0], int(tris_by_solid_angle), out.ctypes.data_as(C.c_void_p)), "pt_light_pdf")
        return out

    def material_probe(self, material: int, probes: np.ndarray, seed: int = 0) -> np.ndarray:
        """pt_material_probe: scatter / bsdf_evaluate / scattering_pdf / emitted of one material at probes[n] (PROBE_IN_DTYPE)."""
        probes = np.ascontiguousarray(probes, PROBE_IN_DTYPE)
        out = np.zeros(probes.shape[0], PROBE_OUT_DTYPE)
        self._check(self.L.pt_material_probe(self.ctx, material, seed, probes.ctypes.data_as(C.c_void_p), probes.shape[0], out.ctypes.data_as(C.c_void_p)), "pt_material_probe")
        return out

    def read_accum(self):
        acc = np.zeros((self.npix, 3), np.float32)
        n = C.c_uint32(0)
        self._check(self.L.pt_read_accum(self.ctx, acc.ctypes.data_as(C.c_void_p), C.byref(n)), "pt_read_accum")
        return acc, int(n.value)

    def read_rgb8(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 3), np.uint8)
        self._check(self.L.pt_read_rgb8(self.ctx, out.ctypes.data_as(C.c_void_p)), "pt_read_rgb8")
        return out

    def reset_accum(self):
        self._check(self.L.pt_reset_accum(self.ctx), "pt_reset_accum")

    def copy_accum_to_device(self, device_ptr: int):
        self._check(self.L.pt_copy_accum_to_device(self.ctx, C.c_void_p(device_ptr)), "pt_copy_accum_to_device")

    def set_accum_from_device(self, device_ptr: int, n_samples: int):
        self._check(self.L.pt_set_accum_from_device(self.ctx, C.c_void_p(device_ptr), n_samples), "pt_set_accum_from_device")

    def stats(self) -> dict:
        s = pt_stats()
        self._check(self.L.pt_stats_get(self.ctx, C.byref(s)), "pt_stats_get")
        return {k: getattr(s, k) for k, _ in pt_stats._fields_}

    def stats_reset(self):
        self._check(self.L.pt_stats_reset(self.ctx), "pt_stats_reset")


def group_split(stratum_begin: int, stratum_end: int, n: int, i: int):
    """pt_group_split: the piece of [stratum_begin, stratum_end) that member i of n renders (no GPU needed)."""
    b, e = C.c_uint32(0), C.c_uint32(0)
    core_lib().pt_group_split(stratum_begin, stratum_end, n, i, C.byref(b), C.byref(e))
    return int(b.value), int(e.value)


class GroupRenderer:
    """One ``pt_group``: the same scene on several GPUs of this node, strata split between them, one reduce on reading."""

    def __init__(self, scene: HostScene, devices):
        self.scene = scene
        self.L = core_lib()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        if self.L.pt_group_create(devs, len(devices), C.byref(h)):
            raise PtError("pt_group_create: " + self.L.pt_group_last_error(None).decode())
        self.g = h
        if getattr(scene, "device_bvh_from", 0):
            self.set_option(PT_OPT_DEVICE_BVH, scene.device_bvh_from)
        self._check(self.L.pt_group_upload_scene(self.g, C.byref(scene.desc)), "pt_group_upload_scene")
        self._check(self.L.pt_group_set_camera(self.g, C.byref(scene.camera)), "pt_group_set_camera")
        self.width, self.height, self.sqrt_spp = scene.width, scene.height, scene.sqrt_spp
        self.npix = self.width * self.height

    def _check(self, rc: int, what: str):
        if rc:
            raise PtError(f"{what} failed ({rc}): " + self.L.pt_group_last_error(self.g).decode())

    def close(self):
        if getattr(self, "g", None):
            self.L.pt_group_destroy(self.g)
            self.g = None

    def __del__(self):
        self.close()

    def size(self) -> int:
        return self.L.pt_group_size(self.g)

    def set_option(self, key: int, value: float):
        self._check(self.L.pt_group_set_option(self.g, key, float(value)), "pt_group_set_option")

    def render(self, stratum_begin: int, stratum_end: int, seed: int = 0):
        self._check(self.L.pt_group_render(self.g, stratum_begin, stratum_end, seed), "pt_group_render")

    def sync(self):
        self._check(self.L.pt_group_sync(self.g), "pt_group_sync")

    def read_accum(self):
        acc = np.zeros((self.npix, 3), np.float32)
        n = C.c_uint32(0)
        self._check(self.L.pt_group_read_accum(self.g, acc.ctypes.data_as(C.c_void_p), C.byref(n)), "pt_group_read_accum")
        return acc, int(n.value)

    def read_rgb8(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 3), np.uint8)
        self._check(self.L.pt_group_read_rgb8(self.g, out.ctypes.data_as(C.c_void_p)), "pt_group_read_rgb8")
        return out

    def reset_accum(self):
        self._check(self.L.pt_group_reset_accum(self.g), "pt_group_reset_accum")

    def stats(self) -> dict:
        s = pt_stats()
        self._check(self.L.pt_group_stats_get(self.g, C.byref(s)), "pt_group_stats_get")
        return {k: getattr(s, k) for k, _ in pt_stats._fields_}

    def stats_reset(self):
        self._check(self.L.pt_group_stats_reset(self.g), "pt_group_stats_reset")
