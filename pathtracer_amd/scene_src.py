"""Authoring-level scene description, mirrored from ``include/ptscene_src.h``.

A :class:`SceneSource` collects what a scene file of the reference renderer states
(``src/scenes/cornell.rs``, ``object.rs``, ``lucy.rs``): primitives in insertion
order, materials and textures by constructor arguments, instances as lists of
transform operations, and the arguments of ``Camera::new`` (``src/camera.rs:79-90``).
``to_c()`` lays it out as the ``pts_scene`` / ``pts_camera`` C structs.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# ---- enums (keep in step with ptscene_src.h) -------------------------------
TEX_SOLID, TEX_CHECKERED, TEX_IMAGE, TEX_NOISE = 0, 1, 2, 3
(MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_DISNEY, MAT_DISNEY_DIFFUSE,
 MAT_DISNEY_METAL, MAT_DISNEY_GLASS, MAT_DISNEY_CLEARCOAT, MAT_DISNEY_SHEEN) = range(10)
(OP_ROTATE, OP_ROTATE_X, OP_ROTATE_Y, OP_ROTATE_Z, OP_TRANSLATE, OP_TRANSLATE_X, OP_TRANSLATE_Y,
 OP_TRANSLATE_Z, OP_SCALE, OP_SCALE_X, OP_SCALE_Y, OP_SCALE_Z, OP_SCALE_UNIFORM) = range(13)
TRI_HAS_NORMALS, TRI_HAS_UVS = 1, 2
PRIM_TRI, PRIM_SPHERE, PRIM_QUAD, PRIM_INSTANCE = 0, 1, 2, 3


class pts_texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", C.c_float * 3), ("odd", C.c_int32), ("even", C.c_int32),
                ("scale", C.c_float), ("image", C.c_int32)]


class pts_image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class pts_material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("texture", C.c_int32), ("p", C.c_float * 14)]


class pts_range(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32)]


class pts_primset(C.Structure):
    _fields_ = [("tris", pts_range), ("spheres", pts_range), ("quads", pts_range), ("instances", pts_range),
                ("build_bvh", C.c_uint32), ("reserved", C.c_uint32)]


class pts_xform_op(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", C.c_float * 4)]


class pts_instance(C.Structure):
    _fields_ = [("child_set", C.c_uint32), ("op_first", C.c_uint32), ("op_count", C.c_uint32),
                ("reserved", C.c_uint32)]


class pts_camera(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_uint32), ("vfov", C.c_float),
                ("look_from", C.c_float * 3), ("look_at", C.c_float * 3), ("focal_length", C.c_float),
                ("defocus_angle", C.c_float), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("background", C.c_float * 3)]


class pts_medium(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("density", C.c_float), ("texture", C.c_int32)]


class pts_moving_sphere(C.Structure):
    _fields_ = [("center0", C.c_float * 3), ("center1", C.c_float * 3), ("radius", C.c_float), ("material", C.c_int32)]


class pts_scene(C.Structure):
    _fields_ = [
        ("n_textures", C.c_uint32), ("textures", C.POINTER(pts_texture)),
        ("n_images", C.c_uint32), ("images", C.POINTER(pts_image)),
        ("n_texel_bytes", C.c_uint64), ("texels", C.POINTER(C.c_uint8)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(pts_material)),
        ("n_tris", C.c_uint32), ("tri_pos", C.POINTER(C.c_float)), ("tri_nrm", C.POINTER(C.c_float)),
        ("tri_uv", C.POINTER(C.c_float)), ("tri_flags", C.POINTER(C.c_uint8)), ("tri_mat", C.POINTER(C.c_int32)),
        ("n_spheres", C.c_uint32), ("spheres", C.POINTER(C.c_float)), ("sphere_mat", C.POINTER(C.c_int32)),
        ("n_quads", C.c_uint32), ("quads", C.POINTER(C.c_float)), ("quad_mat", C.POINTER(C.c_int32)),
        ("n_instances", C.c_uint32), ("instances", C.POINTER(pts_instance)),
        ("n_ops", C.c_uint32), ("ops", C.POINTER(pts_xform_op)),
        ("n_sets", C.c_uint32), ("sets", C.POINTER(pts_primset)),
        ("n_media", C.c_uint32), ("media", C.POINTER(pts_medium)),
        ("n_moving", C.c_uint32), ("moving", C.POINTER(pts_moving_sphere)),
    ]


def _ptr(arr: Optional[np.ndarray], ctype):
    if arr is None or arr.size == 0:
        return C.cast(None, C.POINTER(ctype))
    return arr.ctypes.data_as(C.POINTER(ctype))


@dataclass
class PrimSet:
    """One ``PrimitiveBuffer`` (``src/objects.rs:130-206``) while it is being filled."""
    tri_pos: List[np.ndarray] = field(default_factory=list)   # each (k,3,3)
    tri_nrm: List[Optional[np.ndarray]] = field(default_factory=list)
    tri_uv: List[Optional[np.ndarray]] = field(default_factory=list)
    tri_mat: List[np.ndarray] = field(default_factory=list)
    spheres: List[Sequence[float]] = field(default_factory=list)
    sphere_mat: List[int] = field(default_factory=list)
    quads: List[Sequence[float]] = field(default_factory=list)
    quad_mat: List[int] = field(default_factory=list)
    instances: List["InstanceSrc"] = field(default_factory=list)
    build_bvh: bool = False

    # -- PrimitiveBuffer::add_* ------------------------------------------------
    def add_triangles(self, pos, nrm=None, uv=None, material=-1):
        """pos (k,3,3) float32; nrm (k,3,3) or None; uv (k,3,2) or None; material int or (k,) ints."""
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3, 3)
        k = pos.shape[0]
        self.tri_pos.append(pos)
        self.tri_nrm.append(None if nrm is None else np.ascontiguousarray(nrm, dtype=np.float32).reshape(k, 3, 3))
        self.tri_uv.append(None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(k, 3, 2))
        m = np.asarray(material, dtype=np.int32)
        self.tri_mat.append(np.full(k, int(m), dtype=np.int32) if m.ndim == 0 else m.reshape(k).copy())

    def add_sphere(self, center, radius, material=-1):
        self.spheres.append([float(center[0]), float(center[1]), float(center[2]), float(radius)])
        self.sphere_mat.append(int(material))

    def add_quad(self, origin, u, v, material=-1):
        self.quads.append([float(x) for x in (*origin, *u, *v)])
        self.quad_mat.append(int(material))

    def add_instance(self, inst: "InstanceSrc"):
        self.instances.append(inst)

    def n_tris(self) -> int:
        return sum(p.shape[0] for p in self.tri_pos)


def new_box(a, b, material=-1) -> PrimSet:
    """``Quad::new_box`` (``src/objects/quad_mesh.rs:54-102``): six quads, no BVH."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    mn, mx = np.minimum(a, b), np.maximum(a, b)
    z = np.float32(0.0)
    dx = np.array([mx[0] - mn[0], z, z], dtype=np.float32)
    dy = np.array([z, mx[1] - mn[1], z], dtype=np.float32)
    dz = np.array([z, z, mx[2] - mn[2]], dtype=np.float32)
    s = PrimSet()
    s.add_quad((mn[0], mn[1], mx[2]), dx, dy, material)
    s.add_quad((mx[0], mn[1], mx[2]), -dz, dy, material)
    s.add_quad((mx[0], mn[1], mn[2]), -dx, dy, material)
    s.add_quad((mn[0], mn[1], mn[2]), dz, dy, material)
    s.add_quad((mn[0], mx[1], mx[2]), dx, -dz, material)
    s.add_quad((mn[0], mn[1], mn[2]), dx, dz, material)
    return s


@dataclass
class InstanceSrc:
    """``Instance::from_obj(obj)`` followed by transform calls (``src/objects.rs:298-380``)."""
    child: PrimSet
    ops: List[tuple] = field(default_factory=list)

    def rotate(self, angle, axis):
        self.ops.append((OP_ROTATE, float(angle), *[float(x) for x in axis])); return self

    def rotate_x(self, a): self.ops.append((OP_ROTATE_X, float(a))); return self
    def rotate_y(self, a): self.ops.append((OP_ROTATE_Y, float(a))); return self
    def rotate_z(self, a): self.ops.append((OP_ROTATE_Z, float(a))); return self

    def translate(self, s):
        self.ops.append((OP_TRANSLATE, *[float(x) for x in s])); return self

    def translate_x(self, x): self.ops.append((OP_TRANSLATE_X, float(x))); return self
    def translate_y(self, y): self.ops.append((OP_TRANSLATE_Y, float(y))); return self
    def translate_z(self, z): self.ops.append((OP_TRANSLATE_Z, float(z))); return self

    def scale(self, s):
        self.ops.append((OP_SCALE, *[float(x) for x in s])); return self

    def scale_x(self, x): self.ops.append((OP_SCALE_X, float(x))); return self
    def scale_y(self, y): self.ops.append((OP_SCALE_Y, float(y))); return self
    def scale_z(self, z): self.ops.append((OP_SCALE_Z, float(z))); return self
    def scale_uniform(self, s): self.ops.append((OP_SCALE_UNIFORM, float(s))); return self


@dataclass
class CameraSrc:
    aspect_ratio: float
    image_width: int
    vfov: float
    look_from: Sequence[float]
    look_at: Sequence[float]
    focal_length: float
    defocus_angle: float
    spp: int
    max_depth: int
    background: Sequence[float]

    def to_c(self) -> pts_camera:
        c = pts_camera()
        c.aspect_ratio = self.aspect_ratio
        c.image_width = self.image_width
        c.vfov = self.vfov
        c.look_from[:] = [float(x) for x in self.look_from]
        c.look_at[:] = [float(x) for x in self.look_at]
        c.focal_length = self.focal_length
        c.defocus_angle = self.defocus_angle
        c.spp = self.spp
        c.max_depth = self.max_depth
        c.background[:] = [float(x) for x in self.background]
        return c

    @property
    def image_height(self) -> int:
        # `max(1, (image_width as f32 / aspect_ratio) as u32)`, src/camera.rs:91
        h = int(np.float32(self.image_width) / np.float32(self.aspect_ratio))
        return max(1, h)

    @property
    def sqrt_spp(self) -> int:
        return int(np.sqrt(np.float32(self.spp)))  # src/camera.rs:119


class SceneSource:
    """objects + lights + materials + textures, ready to be laid out as ``pts_scene``."""

    def __init__(self):
        self.textures: List[tuple] = []
        self.images: List[np.ndarray] = []
        self.materials: List[tuple] = []
        self.objects = PrimSet()
        self.lights = PrimSet()
        self.camera: Optional[CameraSrc] = None
        self.name = ""
        self.media: List[tuple] = []
        self.moving: List[tuple] = []
        self._keep = None

    def add_moving_sphere(self, center0, center1, radius: float, material: int):
        """A sphere moving from ``center0`` (time 0) to ``center1`` (time 1) (``pts_moving_sphere``; not in the reference)."""
        self.moving.append((tuple(float(c) for c in center0), tuple(float(c) for c in center1), float(radius), int(material)))

    def add_medium(self, center, radius: float, density: float, albedo_texture: int):
        """A homogeneous medium filling a sphere (``pts_medium``; not in the reference)."""
        self.media.append((tuple(float(c) for c in center), float(radius), float(density), int(albedo_texture)))

    # -- textures (src/textures.rs) -------------------------------------------
    def solid(self, color) -> int:
        self.textures.append((TEX_SOLID, tuple(float(c) for c in color), -1, -1, 0.0, -1))
        return len(self.textures) - 1

    def checkered(self, odd: int, even: int, scale: float) -> int:
        """``Checkered::new(odd, even, inv_scale)`` — argument order as in the reference."""
        self.textures.append((TEX_CHECKERED, (0.0, 0.0, 0.0), int(odd), int(even), float(scale), -1))
        return len(self.textures) - 1

    def noise(self, color, scale: float, seed: int = 1) -> int:
        """Marble-like gradient noise (not in the reference): ``color * (1 + sin(scale * p.z + 10 * turbulence(p)))``."""
        self.textures.append((TEX_NOISE, tuple(float(c) for c in color), int(seed), -1, float(scale), -1))
        return len(self.textures) - 1

    def image(self, rgb8: Optional[np.ndarray]) -> int:
        """``Image::new(Some(img))``; ``None`` is the failed-load fallback (procedural checker)."""
        idx = -1
        if rgb8 is not None:
            self.images.append(np.ascontiguousarray(rgb8, dtype=np.uint8))
            idx = len(self.images) - 1
        self.textures.append((TEX_IMAGE, (0.0, 0.0, 0.0), -1, -1, 0.0, idx))
        return len(self.textures) - 1

    def image_file(self, path: str) -> int:
        """``Image::load(path)`` (src/textures/image.rs:17-26): the file's texels, or the default texture (and the
        reference's log line) when the file cannot be read."""
        from .core import PtError, load_image
        try:
            return self.image(load_image(path))
        except PtError as e:
            import sys
            print(f"Failed to load image: '{path}'... Using default texture instead... ({e})", file=sys.stderr)
            return self.image(None)

    # -- materials (src/materials/*.rs) ---------------------------------------
    def _mat(self, type_, tex, params) -> int:
        p = [0.0] * 14
        p[:len(params)] = [float(x) for x in params]
        self.materials.append((type_, int(tex), p))
        return len(self.materials) - 1

    def lambertian(self, albedo) -> int: return self._mat(MAT_LAMBERTIAN, self.solid(albedo), [])
    def lambertian_texture(self, tex: int) -> int: return self._mat(MAT_LAMBERTIAN, tex, [])
    def metal(self, albedo, fuzz) -> int: return self._mat(MAT_METAL, -1, [*albedo, fuzz])
    def dielectric(self, ref_idx) -> int: return self._mat(MAT_DIELECTRIC, -1, [ref_idx])
    def diffuse_light(self, albedo) -> int: return self._mat(MAT_DIFFUSE_LIGHT, self.solid(albedo), [])
    def diffuse_light_texture(self, tex: int) -> int: return self._mat(MAT_DIFFUSE_LIGHT, tex, [])

    def disney(self, tex, subsurface, metallic, specular_transmission, specular, specular_tint, roughness,
               anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss, ior) -> int:
        return self._mat(MAT_DISNEY, tex, [subsurface, metallic, specular_transmission, specular, specular_tint,
                                          roughness, anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss, ior])

    def disney_diffuse(self, color, roughness, subsurface) -> int:
        return self._mat(MAT_DISNEY_DIFFUSE, self.solid(color), [roughness, subsurface])

    def disney_metal(self, color, roughness, anisotropic) -> int:
        return self._mat(MAT_DISNEY_METAL, self.solid(color), [roughness, anisotropic])

    def disney_glass(self, color, roughness, anisotropic, eta) -> int:
        return self._mat(MAT_DISNEY_GLASS, self.solid(color), [roughness, anisotropic, eta])

    def disney_clearcoat(self, clearcoat, clearcoat_gloss) -> int:
        return self._mat(MAT_DISNEY_CLEARCOAT, -1, [clearcoat, clearcoat_gloss])

    def disney_sheen(self, color, sheen, sheen_tint) -> int:
        return self._mat(MAT_DISNEY_SHEEN, self.solid(color), [sheen, sheen_tint])

    # -- layout ------------------------------------------------------------------
    def to_c(self):
        """Returns (pts_scene, pts_camera). Backing arrays stay alive as long as ``self`` does."""
        sets: List[PrimSet] = [self.objects, self.lights]
        i = 0
        while i < len(sets):  # breadth-first: children of instances get the next set indices
            for inst in sets[i].instances:
                sets.append(inst.child)
            i += 1
        set_index = {id(s): k for k, s in enumerate(sets)}

        tri_pos, tri_nrm, tri_uv, tri_flags, tri_mat = [], [], [], [], []
        spheres, sphere_mat, quads, quad_mat = [], [], [], []
        instances, ops = [], []
        csets = (pts_primset * len(sets))()
        nt = ns = nq = ni = 0
        for k, s in enumerate(sets):
            t0 = nt
            for pos, nrm, uv, mat in zip(s.tri_pos, s.tri_nrm, s.tri_uv, s.tri_mat):
                n = pos.shape[0]
                tri_pos.append(pos)
                tri_nrm.append(nrm if nrm is not None else np.zeros((n, 3, 3), np.float32))
                tri_uv.append(uv if uv is not None else np.zeros((n, 3, 2), np.float32))
                tri_flags.append(np.full(n, (TRI_HAS_NORMALS if nrm is not None else 0) | (TRI_HAS_UVS if uv is not None else 0), np.uint8))
                tri_mat.append(mat)
                nt += n
            csets[k].tris = pts_range(t0, nt - t0)
            csets[k].spheres = pts_range(ns, len(s.spheres))
            spheres += s.spheres; sphere_mat += s.sphere_mat; ns += len(s.spheres)
            csets[k].quads = pts_range(nq, len(s.quads))
            quads += s.quads; quad_mat += s.quad_mat; nq += len(s.quads)
            csets[k].instances = pts_range(ni, len(s.instances))
            for inst in s.instances:
                instances.append((set_index[id(inst.child)], len(ops), len(inst.ops)))
                ops += inst.ops
                ni += 1
            csets[k].build_bvh = 1 if s.build_bvh else 0

        def cat(lst, shape, dtype):
            return np.ascontiguousarray(np.concatenate(lst, axis=0), dtype=dtype) if lst else np.zeros(shape, dtype)

        a_tri_pos = cat(tri_pos, (0, 3, 3), np.float32)
        a_tri_nrm = cat(tri_nrm, (0, 3, 3), np.float32)
        a_tri_uv = cat(tri_uv, (0, 3, 2), np.float32)
        a_tri_flags = cat(tri_flags, (0,), np.uint8)
        a_tri_mat = cat(tri_mat, (0,), np.int32)
        a_spheres = np.asarray(spheres, dtype=np.float32).reshape(-1, 4)
        a_sphere_mat = np.asarray(sphere_mat, dtype=np.int32)
        a_quads = np.asarray(quads, dtype=np.float32).reshape(-1, 9)
        a_quad_mat = np.asarray(quad_mat, dtype=np.int32)

        ctex = (pts_texture * max(1, len(self.textures)))()
        for k, (ty, col, odd, even, scale, img) in enumerate(self.textures):
            ctex[k].type = ty; ctex[k].color[:] = col; ctex[k].odd = odd; ctex[k].even = even
            ctex[k].scale = scale; ctex[k].image = img
        cimg = (pts_image * max(1, len(self.images)))()
        blobs, off = [], 0
        for k, im in enumerate(self.images):
            cimg[k].height, cimg[k].width = im.shape[0], im.shape[1]
            cimg[k].offset = off
            blobs.append(im.reshape(-1)); off += im.size
        a_texels = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
        cmat = (pts_material * max(1, len(self.materials)))()
        for k, (ty, tex, p) in enumerate(self.materials):
            cmat[k].type = ty; cmat[k].texture = tex; cmat[k].p[:] = p
        cinst = (pts_instance * max(1, len(instances)))()
        for k, (cs, of, oc) in enumerate(instances):
            cinst[k].child_set = cs; cinst[k].op_first = of; cinst[k].op_count = oc
        cops = (pts_xform_op * max(1, len(ops)))()
        for k, op in enumerate(ops):
            cops[k].op = op[0]
            args = list(op[1:]) + [0.0] * (4 - len(op[1:]))
            cops[k].a[:] = args

        sc = pts_scene()
        sc.n_textures = len(self.textures); sc.textures = C.cast(ctex, C.POINTER(pts_texture))
        sc.n_images = len(self.images); sc.images = C.cast(cimg, C.POINTER(pts_image))
        sc.n_texel_bytes = a_texels.size; sc.texels = _ptr(a_texels, C.c_uint8)
        sc.n_materials = len(self.materials); sc.materials = C.cast(cmat, C.POINTER(pts_material))
        sc.n_tris = nt
        sc.tri_pos = _ptr(a_tri_pos, C.c_float); sc.tri_nrm = _ptr(a_tri_nrm, C.c_float)
        sc.tri_uv = _ptr(a_tri_uv, C.c_float); sc.tri_flags = _ptr(a_tri_flags, C.c_uint8)
        sc.tri_mat = _ptr(a_tri_mat, C.c_int32)
        sc.n_spheres = ns; sc.spheres = _ptr(a_spheres, C.c_float); sc.sphere_mat = _ptr(a_sphere_mat, C.c_int32)
        sc.n_quads = nq; sc.quads = _ptr(a_quads, C.c_float); sc.quad_mat = _ptr(a_quad_mat, C.c_int32)
        sc.n_instances = ni; sc.instances = C.cast(cinst, C.POINTER(pts_instance))
        sc.n_ops = len(ops); sc.ops = C.cast(cops, C.POINTER(pts_xform_op))
        sc.n_sets = len(sets); sc.sets = C.cast(csets, C.POINTER(pts_primset))
        cmed = (pts_medium * max(1, len(self.media)))()
        for k, (center, radius, density, tex) in enumerate(self.media):
            cmed[k].center[:] = center; cmed[k].radius = radius; cmed[k].density = density; cmed[k].texture = tex
        sc.n_media = len(self.media); sc.media = C.cast(cmed, C.POINTER(pts_medium))
        cmov = (pts_moving_sphere * max(1, len(self.moving)))()
        for k, (c0, c1, radius, mat) in enumerate(self.moving):
            cmov[k].center0[:] = c0; cmov[k].center1[:] = c1; cmov[k].radius = radius; cmov[k].material = mat
        sc.n_moving = len(self.moving); sc.moving = C.cast(cmov, C.POINTER(pts_moving_sphere))
        cam = self.camera.to_c()
        self._keep = (cmed, cmov, a_tri_pos, a_tri_nrm, a_tri_uv, a_tri_flags, a_tri_mat, a_spheres, a_sphere_mat, a_quads,
                      a_quad_mat, ctex, cimg, a_texels, cmat, cinst, cops, csets, sc, cam)
        return sc, cam
