"""Scenes, authored once and consumed by every renderer in this repository.

``cornell`` restates ``src/scenes/cornell.rs``; ``knob`` restates ``src/scenes/object.rs``
with a procedural stand-in for ``mitsuba_smooth.obj`` (the reference ignores ``*.obj`` in
git, so the asset is not available); ``lucy`` restates ``src/scenes/lucy.rs`` (camera, light,
materials) around a procedural stand-in for the author's local ``sponza_tri.obj``; ``dragon_box``,
``interior`` and ``next_week_like`` are procedural scenes for the remaining benchmark configurations.  Procedural geometry uses
only a hash-based noise and float64 trigonometry rounded to float32.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

from .scene_src import CameraSrc, InstanceSrc, PrimSet, SceneSource, new_box


def _radians_f32(deg: float) -> float:
    # f32::to_radians: x * (PI_f32 / 180_f32), all in f32
    return float(np.float32(deg) * (np.float32(math.pi) / np.float32(180.0)))


# --------------------------------------------------------------------------- C1
def cornell(width: int = 768, spp: int = 1024, max_depth: int = 16) -> SceneSource:
    """``Cornell::build_scene`` / ``build_camera`` (``src/scenes/cornell.rs:18-133``)."""
    s = SceneSource()
    s.name = "cornell"
    red = s.lambertian((0.65, 0.05, 0.05))
    mirror = s.metal((0.8, 0.85, 0.88), 0.05)
    white = s.lambertian((0.73, 0.73, 0.73))
    green = s.lambertian((0.12, 0.45, 0.15))
    light = s.diffuse_light((15.0, 15.0, 15.0))
    glass = s.disney_glass((1.0, 1.0, 1.0), 0.3, 0.0, 1.5)

    o = s.objects
    o.add_quad((555.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), green)          # q1
    o.add_quad((0.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), red)              # q2
    s.lights.add_quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), light)
    o.add_quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), light)    # q3
    o.add_quad((0.0, 0.0, 0.0), (555.0, 0.0, 0.0), (0.0, 0.0, 555.0), white)            # q4
    o.add_quad((555.0, 555.0, 555.0), (-555.0, 0.0, 0.0), (0.0, 0.0, -555.0), white)    # q5
    o.add_quad((0.0, 0.0, 555.5), (555.0, 0.0, 0.0), (0.0, 555.0, 0.0), white)          # q6

    box1 = InstanceSrc(new_box((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), mirror))
    box1.rotate_y(_radians_f32(15.0)).translate((265.0, 0.0, 295.0))
    sph = PrimSet()
    sph.add_sphere((190.0, 90.0, 190.0), 90.0, glass)
    o.add_instance(box1)
    o.add_instance(InstanceSrc(sph))  # box2 is built by the reference but never added (:119-130)
    o.build_bvh = True                # main.rs:72

    s.camera = CameraSrc(1.0, width, 40.0, (278.0, 278.0, -800.0), (278.0, 278.0, 0.0), 1.0, 0.0, spp, max_depth,
                         (0.0, 0.0, 0.0))
    return s


# ------------------------------------------------------------ procedural meshes
def _hash01(ix: np.ndarray, seed: int) -> np.ndarray:
    """SplitMix64 finaliser of an integer lattice -> float64 in [0,1)."""
    with np.errstate(over="ignore"):
        z = ix.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _value_noise(theta: np.ndarray, phi: np.ndarray, freq: int, seed: int) -> np.ndarray:
    """Smooth periodic value noise on a (theta, phi) lattice with `freq` cells per turn."""
    x = theta / (2 * np.pi) * freq
    y = phi / np.pi * freq
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    fx, fy = x - x0, y - y0
    fx = fx * fx * (3 - 2 * fx)
    fy = fy * fy * (3 - 2 * fy)

    def h(ix, iy):
        return _hash01((ix % freq) * 7919 + iy * 104729 + 17, seed)

    return ((h(x0, y0) * (1 - fx) + h(x0 + 1, y0) * fx) * (1 - fy)
            + (h(x0, y0 + 1) * (1 - fx) + h(x0 + 1, y0 + 1) * fx) * fy)


def _grid_to_tris(P: np.ndarray, N: np.ndarray, UV: np.ndarray, keep: np.ndarray | None = None, flip: bool = False):
    """(rows, cols, 3) vertex grid -> per-triangle expanded arrays, two triangles per cell."""
    r, c = P.shape[0] - 1, P.shape[1] - 1
    i00 = (slice(0, r), slice(0, c)); i10 = (slice(1, r + 1), slice(0, c))
    i01 = (slice(0, r), slice(1, c + 1)); i11 = (slice(1, r + 1), slice(1, c + 1))

    def tri(A, a, b, cc):
        return np.stack([A[a], A[b], A[cc]], axis=2)  # (r, c, 3 verts, k)

    order1 = (i00, i10, i11) if not flip else (i00, i11, i10)
    order2 = (i00, i11, i01) if not flip else (i00, i01, i11)
    out = []
    for A in (P, N, UV):
        t = np.stack([tri(A, *order1), tri(A, *order2)], axis=2)  # (r, c, 2, 3, k)
        if keep is not None:
            t = t[keep]
        out.append(np.ascontiguousarray(t.reshape(-1, 3, A.shape[-1]), dtype=np.float32))
    return out


def _sphere_grid(center, radius_fn, n_theta: int, n_phi: int):
    """Lat-long grid; radius_fn(theta, phi) -> radius.  Normals by central differences (smooth)."""
    th = np.linspace(0.0, 2 * np.pi, n_theta + 1)
    ph = np.linspace(0.0, np.pi, n_phi + 1)
    PH, TH = np.meshgrid(ph, th, indexing="ij")
    R = radius_fn(TH, PH)
    d = np.stack([np.sin(PH) * np.cos(TH), np.cos(PH), np.sin(PH) * np.sin(TH)], axis=-1)
    P = np.asarray(center, dtype=np.float64) + R[..., None] * d
    # finite-difference normals of the displaced surface
    e = 1e-4
    Pt = np.asarray(center) + radius_fn(TH + e, PH)[..., None] * np.stack(
        [np.sin(PH) * np.cos(TH + e), np.cos(PH), np.sin(PH) * np.sin(TH + e)], axis=-1)
    Pp = np.asarray(center) + radius_fn(TH, PH + e)[..., None] * np.stack(
        [np.sin(PH + e) * np.cos(TH), np.cos(PH + e), np.sin(PH + e) * np.sin(TH)], axis=-1)
    N = np.cross(Pp - P, Pt - P)
    nn = np.linalg.norm(N, axis=-1, keepdims=True)
    N = np.where(nn > 1e-12, N / np.maximum(nn, 1e-30), d)
    # orient outward
    N = np.where((N * d).sum(-1, keepdims=True) < 0, -N, N)
    UV = np.stack([TH / (2 * np.pi), 1.0 - PH / np.pi], axis=-1)
    return P, N, UV, TH, PH


def _scaled(n: int, detail: float) -> int:
    return max(4, int(round(n * detail)))


def knob_meshes(seed: int = 1, detail: float = 1) -> Tuple[tuple, tuple, tuple]:
    """Three meshes standing in for the three models of ``mitsuba_smooth.obj`` (``object.rs:60-67``):
    backdrop (uv-mapped), inner core, outer shell with cut-outs.  ~150 k triangles at detail=1."""
    # 0: backdrop sheet, flat in front, curving up behind the object
    n = _scaled(96, detail)
    xs = np.linspace(-14.0, 14.0, n + 1)
    zs = np.linspace(-14.0, 14.0, n + 1)
    Z, X = np.meshgrid(zs, xs, indexing="ij")
    back = np.clip((-Z - 5.0) / 9.0, 0.0, 1.0)
    Y = 6.0 * back * back
    P0 = np.stack([X, Y, Z], axis=-1)
    dYdz = np.where((-Z - 5.0 > 0) & (-Z - 5.0 < 9.0), -12.0 * back / 9.0, 0.0)
    N0 = np.stack([np.zeros_like(X), np.ones_like(X), -dYdz], axis=-1)
    N0 /= np.linalg.norm(N0, axis=-1, keepdims=True)
    UV0 = np.stack([(X + 14.0) / 28.0, (Z + 14.0) / 28.0], axis=-1)
    m0 = _grid_to_tris(P0, N0, UV0, flip=True)

    # 1: inner core
    c = (0.0, 1.05, 0.0)
    P1, N1, UV1, _, _ = _sphere_grid(c, lambda t, p: 0.78 + 0.0 * t, _scaled(192, detail), _scaled(96, detail))
    m1 = _grid_to_tris(P1, N1, UV1)

    # 2: outer shell — outer and inner surface of a bumpy sphere with an equatorial band of windows
    # removed, plus a pedestal
    nt, npx = _scaled(288, detail), _scaled(144, detail)

    def r_out(t, p):
        return 1.22 + 0.035 * (_value_noise(np.mod(t, 2 * np.pi), np.clip(p, 0, np.pi), 24, seed) - 0.5)

    Po, No, UVo, TH, PH = _sphere_grid(c, r_out, nt, npx)
    Pi, Ni, UVi, _, _ = _sphere_grid(c, lambda t, p: r_out(t, p) - 0.12, nt, npx)
    thc = 0.5 * (TH[:-1, :-1] + TH[1:, 1:])
    phc = 0.5 * (PH[:-1, :-1] + PH[1:, 1:])
    window = (np.abs(phc - 0.5 * np.pi) < 0.42) & (np.mod(thc * 6 / (2 * np.pi), 1.0) < 0.55)
    cap = phc < 0.30  # open top
    keep = ~(window | cap)
    keep2 = np.repeat(keep[:, :, None], 2, axis=2)
    mo = _grid_to_tris(Po, No, UVo, keep=keep2)
    mi = _grid_to_tris(Pi, -Ni, UVi, keep=keep2, flip=True)
    # pedestal: a cone frustum ring under the shell
    k = _scaled(192, detail)
    tt = np.linspace(0, 2 * np.pi, k + 1)
    hh = np.linspace(0.0, 1.0, 9)
    Hh, Tt = np.meshgrid(hh, tt, indexing="ij")
    rad = 1.1 - 0.45 * Hh
    Pp_ = np.stack([rad * np.cos(Tt), 0.0 + 0.35 * Hh, rad * np.sin(Tt)], axis=-1)
    Np_ = np.stack([np.cos(Tt) * 0.35, np.full_like(Tt, 0.45), np.sin(Tt) * 0.35], axis=-1)
    Np_ /= np.linalg.norm(Np_, axis=-1, keepdims=True)
    UVp = np.stack([Tt / (2 * np.pi), Hh], axis=-1)
    mp = _grid_to_tris(Pp_, Np_, UVp, flip=True)
    m2 = [np.concatenate([a, b, d], axis=0) for a, b, d in zip(mo, mi, mp)]
    return tuple(m0), tuple(m1), tuple(m2)


# --------------------------------------------------------------------------- C3
def knob(width: int = 768, spp: int = 10240, max_depth: int = 50, seed: int = 1, detail: float = 1, asset: Optional[str] = None) -> SceneSource:
    """``Object::build_scene`` / ``build_camera`` (``src/scenes/object.rs:22-96``).  The meshes come from
    ``asset`` (an .obj file, loaded as ``TriMesh::load_as_vec`` does, materials assigned by model order,
    ``object.rs:60-67``) or, without one, from the procedural stand-in."""
    s = SceneSource()
    s.name = "knob"
    lite = s.solid((0.6, 0.6, 0.6))
    dark = s.solid((0.2, 0.2, 0.2))
    blue = s.solid((0.8, 0.6, 0.2))
    check = s.checkered(lite, dark, 13.0)  # Checkered::new(lite, dark, 13.0): odd=lite, even=dark
    s.metal((0.45, 0.45, 0.45), 0.5)                       # "ground" (unused by any primitive)
    emit = s.diffuse_light((20.0, 20.0, 20.0))
    m_check = s.lambertian_texture(check)
    m_mat1 = s.lambertian((0.2, 0.2, 0.2))
    s.lambertian((0.8, 0.8, 0.3))                          # "mat2" (unused)
    m_test = s.disney(blue, 0.5, 0.1, 0.0, 0.01, 0.2, 0.1, 0.8, 0.3, 0.1, 0.2, 0.5, 1.1)

    s.lights.add_sphere((-10.0, 15.0, 5.0), 2.0, emit)
    s.objects.add_sphere((-10.0, 15.0, 5.0), 2.0, emit)
    if asset is not None:
        from .core import load_obj
        meshes = load_obj(asset)
        if len(meshes) > 3:
            raise ValueError(f"{asset}: {len(meshes)} models, the scene has materials for three (object.rs:60)")
    else:
        meshes = knob_meshes(seed, detail)
    for (pos, nrm, uv), mat in zip(meshes, (m_check, m_mat1, m_test)):
        if pos.shape[0]:
            s.objects.add_triangles(pos, nrm, uv, mat)
    s.objects.build_bvh = True
    s.camera = CameraSrc(1.0, width, 15.0, (2.0, 10.0, 16.0), (0.0, 1.0, 0.0), 10.0, 0.0, spp, max_depth,
                         (0.0, 0.0, 0.0))
    return s


# --------------------------------------------------------------------------- src/scenes/lucy.rs
def lucy_meshes(seed: int = 1, detail: float = 1):
    """Stand-in for ``sponza_tri.obj`` (``lucy.rs:21`` names a path on the author's machine): an open courtyard in
    front of the camera of ``lucy.rs:51-72`` — a tessellated floor, two arcades of columns carrying a lintel, and a
    back wall — as separate meshes, the way ``TriMesh::load_as_vec`` returns one mesh per model of the file."""
    meshes = []
    nx, nz = _scaled(160, detail), _scaled(64, detail)
    xs, zs = np.linspace(-6.0, 34.0, nx + 1), np.linspace(-9.0, 9.0, nz + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = 0.03 * _value_noise(X * 0.4, Z * 0.4, 8, seed)  # worn flagstones
    P = np.stack([X, Y, Z], axis=-1)
    N = np.stack([np.zeros_like(X), np.ones_like(X), np.zeros_like(X)], axis=-1)
    UV = np.stack([(X + 6.0) / 40.0, (Z + 9.0) / 18.0], axis=-1)
    meshes.append(_grid_to_tris(P, N, UV))
    nt, nh = _scaled(48, detail), _scaled(24, detail)
    for side in (-1.0, 1.0):
        for k in range(9):
            cx, cz = -2.0 + 4.0 * k, side * 6.0
            tt, hh = np.linspace(0, 2 * np.pi, nt + 1), np.linspace(0.0, 5.0, nh + 1)
            Hh, Tt = np.meshgrid(hh, tt, indexing="ij")
            rad = 0.45 + 0.03 * np.cos(10 * Tt) + 0.1 * np.exp(-((Hh - 0.2) ** 2) * 10) + 0.1 * np.exp(-((Hh - 4.8) ** 2) * 10)
            P = np.stack([cx + rad * np.cos(Tt), Hh, cz + rad * np.sin(Tt)], axis=-1)
            N = np.stack([np.cos(Tt), np.zeros_like(Tt), np.sin(Tt)], axis=-1)
            UV = np.stack([Tt / (2 * np.pi), Hh / 5.0], axis=-1)
            meshes.append(_grid_to_tris(P, N, UV, flip=True))
        # lintel over the arcade: a long box as a closed grid of four faces
        z0, z1 = side * 6.0 - 0.6, side * 6.0 + 0.6
        corners = np.array([[z0, 5.0], [z1, 5.0], [z1, 5.8], [z0, 5.8], [z0, 5.0]])
        xs2 = np.linspace(-3.0, 31.0, _scaled(34, detail) + 1)
        P = np.stack([np.broadcast_to(xs2[:, None], (xs2.size, 5)), np.broadcast_to(corners[None, :, 1], (xs2.size, 5)),
                      np.broadcast_to(corners[None, :, 0], (xs2.size, 5))], axis=-1).astype(np.float64)
        nrm_ring = np.array([[0, -1, 0], [0, 0, 1], [0, 1, 0], [0, 0, -1], [0, -1, 0]], np.float64)
        N = np.broadcast_to(nrm_ring[None, :, :], P.shape)
        UV = np.stack([np.broadcast_to(((xs2 + 3.0) / 34.0)[:, None], (xs2.size, 5)), np.broadcast_to(np.linspace(0, 1, 5)[None, :], (xs2.size, 5))], axis=-1)
        meshes.append(_grid_to_tris(P, N, UV))
    ny = _scaled(40, detail)
    ys, zs = np.linspace(0.0, 9.0, ny + 1), np.linspace(-9.0, 9.0, nz + 1)
    Yy, Zz = np.meshgrid(ys, zs, indexing="ij")
    Xx = 34.0 + 0.05 * _value_noise(Yy * 0.5, Zz * 0.5, 6, seed + 1)
    P = np.stack([Xx, Yy, Zz], axis=-1)
    N = np.stack([-np.ones_like(Xx), np.zeros_like(Xx), np.zeros_like(Xx)], axis=-1)
    UV = np.stack([(Zz + 9.0) / 18.0, Yy / 9.0], axis=-1)
    meshes.append(_grid_to_tris(P, N, UV))
    return meshes


def lucy(width: int = 1200, spp: int = 51200, max_depth: int = 16, seed: int = 1, detail: float = 1, asset: Optional[str] = None) -> SceneSource:
    """``Lucy::build_scene`` / ``build_camera`` (``src/scenes/lucy.rs:16-75``): every mesh of the .obj file with the
    grey Lambertian ``mat2`` (``:39-44``), a small sphere light of radiance 255 (``:24, :33-36``), a sky of 1.5
    (``:70``), 16:9 at 1200 pixels, 51200 -> 226^2 samples, depth 16 (``:51-60``).  ``asset``: the .obj file; without
    one, the procedural courtyard of ``lucy_meshes``."""
    s = SceneSource()
    s.name = "lucy"
    s.metal((0.3, 0.3, 0.8), 0.1)                        # "ground" (its sphere is commented out, lucy.rs:27-31, :37)
    emit = s.diffuse_light((255.0, 255.0, 255.0))
    s.dielectric(1.5)                                    # "mat1" (unused)
    mat2 = s.lambertian((0.5, 0.5, 0.5))
    s.lights.add_sphere((0.7, 2.3, 1.0), 0.5, emit)
    s.objects.add_sphere((0.7, 2.3, 1.0), 0.5, emit)
    if asset is not None:
        from .core import load_obj
        meshes = load_obj(asset)
    else:
        meshes = lucy_meshes(seed, detail)
    for pos, nrm, uv in meshes:
        if pos.shape[0]:
            s.objects.add_triangles(pos, nrm, uv, mat2)
    s.objects.build_bvh = True  # main.rs:72 builds the BVHs of whichever scene it renders
    s.camera = CameraSrc(16.0 / 9.0, width, 20.0, (-12.0, 18.0, 0.0), (20.0, 0.4, 0.0), 10.0, 0.0, spp, max_depth, (1.5, 1.5, 1.5))
    return s


# --------------------------------------------------------------------------- C4
def dragon_box(width: int = 1024, spp: int = 1024, max_depth: int = 16, seed: int = 1, n_theta: int = 1000) -> SceneSource:
    """~1 M-triangle displaced closed surface inside the Cornell walls (traversal-bound)."""
    s = SceneSource()
    s.name = "dragon_box"
    red = s.lambertian((0.65, 0.05, 0.05))
    white = s.lambertian((0.73, 0.73, 0.73))
    green = s.lambertian((0.12, 0.45, 0.15))
    light = s.diffuse_light((15.0, 15.0, 15.0))
    body = s.lambertian((0.55, 0.6, 0.5))
    o = s.objects
    o.add_quad((555.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), green)
    o.add_quad((0.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), red)
    s.lights.add_quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), light)
    o.add_quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), light)
    o.add_quad((0.0, 0.0, 0.0), (555.0, 0.0, 0.0), (0.0, 0.0, 555.0), white)
    o.add_quad((555.0, 555.0, 555.0), (-555.0, 0.0, 0.0), (0.0, 0.0, -555.0), white)
    o.add_quad((0.0, 0.0, 555.5), (555.0, 0.0, 0.0), (0.0, 555.0, 0.0), white)

    def r(t, p):
        tm, pc = np.mod(t, 2 * np.pi), np.clip(p, 0, np.pi)
        return (150.0 + 40.0 * (_value_noise(tm, pc, 6, seed) - 0.5) + 18.0 * (_value_noise(tm, pc, 24, seed + 1) - 0.5)
                + 5.0 * (_value_noise(tm, pc, 96, seed + 2) - 0.5))

    P, N, UV, _, _ = _sphere_grid((278.0, 190.0, 278.0), r, n_theta, n_theta // 2)
    pos, nrm, uv = _grid_to_tris(P, N, UV)
    o.add_triangles(pos, nrm, uv, body)
    o.build_bvh = True
    s.camera = CameraSrc(1.0, width, 40.0, (278.0, 278.0, -800.0), (278.0, 278.0, 0.0), 1.0, 0.0, spp, max_depth,
                         (0.0, 0.0, 0.0))
    return s


# --------------------------------------------------------------------------- C5
def interior(width: int = 1920, spp: int = 1024, max_depth: int = 16, seed: int = 1, n_lights: int = 64,
             columns: int = 10) -> SceneSource:
    """Procedural multi-mesh hall: floor, walls, ceiling, rows of tessellated columns, and a grid of
    emissive quads under the ceiling which are all registered as lights."""
    s = SceneSource()
    s.name = "interior"
    wall = s.lambertian((0.7, 0.68, 0.62))
    floor_tex = s.checkered(s.solid((0.55, 0.5, 0.45)), s.solid((0.25, 0.25, 0.28)), 24.0)
    floor = s.lambertian_texture(floor_tex)
    stone = s.lambertian((0.6, 0.58, 0.55))
    brass = s.disney(s.solid((0.9, 0.7, 0.3)), 0.0, 0.9, 0.0, 0.5, 0.0, 0.25, 0.0, 0.0, 0.0, 0.0, 0.5, 1.5)
    emit = s.diffuse_light((12.0, 11.0, 9.5))
    o = s.objects
    L, W, H = 40.0, 16.0, 9.0
    o.add_quad((0, 0, 0), (L, 0, 0), (0, 0, W), floor)
    o.add_quad((0, H, 0), (0, 0, W), (L, 0, 0), wall)
    o.add_quad((0, 0, 0), (0, H, 0), (L, 0, 0), wall)
    o.add_quad((0, 0, W), (L, 0, 0), (0, H, 0), wall)
    o.add_quad((L, 0, 0), (0, H, 0), (0, 0, W), wall)
    o.add_quad((0, 0, 0), (0, 0, W), (0, H, 0), wall)
    # lights: n_lights emissive quads facing down
    gx = int(math.ceil(math.sqrt(n_lights * L / W)))
    gz = int(math.ceil(n_lights / gx))
    k = 0
    for iz in range(gz):
        for ix in range(gx):
            if k >= n_lights:
                break
            cx = (ix + 0.5) * L / gx
            cz = (iz + 0.5) * W / gz
            q = ((cx - 0.45, H - 0.05, cz - 0.3), (0.9, 0.0, 0.0), (0.0, 0.0, 0.6))
            s.lights.add_quad(*q, emit)
            o.add_quad(*q, emit)
            k += 1
    # columns: fluted cylinders with a brass ring each
    for ci in range(columns):
        for side in (0, 1):
            cx = (ci + 0.5) * L / columns
            cz = 3.0 if side == 0 else W - 3.0
            nt, nh = 96, 48
            tt = np.linspace(0, 2 * np.pi, nt + 1)
            hh = np.linspace(0, H, nh + 1)
            Hh, Tt = np.meshgrid(hh, tt, indexing="ij")
            rad = 0.55 + 0.04 * np.cos(12 * Tt) + 0.08 * np.exp(-((Hh - 0.3) ** 2) * 8) + 0.08 * np.exp(-((Hh - H + 0.3) ** 2) * 8)
            P = np.stack([cx + rad * np.cos(Tt), Hh, cz + rad * np.sin(Tt)], axis=-1)
            N = np.stack([np.cos(Tt), np.zeros_like(Tt), np.sin(Tt)], axis=-1)
            UV = np.stack([Tt / (2 * np.pi), Hh / H], axis=-1)
            pos, nrm, uv = _grid_to_tris(P, N, UV, flip=True)
            o.add_triangles(pos, nrm, uv, stone)
            # ring
            P2, N2, UV2, _, _ = _sphere_grid((cx, 4.5, cz), lambda t, p: 0.8 + 0.0 * t, 48, 24)
            P2[..., 1] = 4.5 + (P2[..., 1] - 4.5) * 0.25
            pos, nrm, uv = _grid_to_tris(P2, N2, UV2)
            o.add_triangles(pos, nrm, uv, brass)
    o.build_bvh = True
    s.camera = CameraSrc(16.0 / 9.0, width, 55.0, (2.0, 4.0, W / 2 + 1.0), (L, 3.0, W / 2 - 1.0), 10.0, 0.0, spp,
                         max_depth, (0.0, 0.0, 0.0))
    return s


# --------------------------------------------------------------------------- C2
def next_week_like(width: int = 800, spp: int = 1024, max_depth: int = 50, seed: int = 1, extensions: bool = False) -> SceneSource:
    """The primitives-only part of the "Next Week" final scene: a floor of boxes (instances), an emissive
    quad, glass / metal / textured spheres and a rotated, translated cluster of ~1000 small spheres.
    Volumes, Perlin noise and motion blur do not exist in the reference; ``extensions`` adds the ones this
    build has of its own: the marble sphere's noise texture, the blue medium inside the second glass sphere,
    the thin mist around everything and the motion of the orange sphere."""
    s = SceneSource()
    s.name = "next_week_like"
    ground = s.lambertian((0.48, 0.83, 0.53))
    light = s.diffuse_light((7.0, 7.0, 7.0))
    white = s.lambertian((0.73, 0.73, 0.73))
    glass = s.dielectric(1.5)
    metal = s.metal((0.8, 0.8, 0.9), 1.0)
    orange = s.lambertian((0.7, 0.3, 0.1))
    earth = s.lambertian_texture(s.image(None))
    checker = s.lambertian_texture(s.checkered(s.solid((0.9, 0.9, 0.9)), s.solid((0.2, 0.3, 0.1)), 16.0))
    marble = s.lambertian_texture(s.noise((0.5, 0.5, 0.5), 0.2, seed)) if extensions else checker
    o = s.objects
    boxes_per_side = 20
    h = _hash01(np.arange(boxes_per_side * boxes_per_side, dtype=np.int64), seed)
    for i in range(boxes_per_side):
        for j in range(boxes_per_side):
            w = 100.0
            x0, z0 = -1000.0 + i * w, -1000.0 + j * w
            y1 = 1.0 + 100.0 * float(h[i * boxes_per_side + j])
            inst = InstanceSrc(new_box((0.0, 0.0, 0.0), (w, y1, w), ground)).translate((x0, 0.0, z0))
            o.add_instance(inst)
    q = ((123.0, 554.0, 147.0), (300.0, 0.0, 0.0), (0.0, 0.0, 265.0))
    o.add_quad(*q, light)
    s.lights.add_quad(*q, light)
    if extensions:
        s.add_moving_sphere((400.0, 400.0, 200.0), (430.0, 400.0, 200.0), 50.0, orange)
    else:
        o.add_sphere((400.0, 400.0, 200.0), 50.0, orange)
    o.add_sphere((260.0, 150.0, 45.0), 50.0, glass)
    o.add_sphere((0.0, 150.0, 145.0), 50.0, metal)
    o.add_sphere((360.0, 150.0, 145.0), 70.0, glass)
    o.add_sphere((400.0, 200.0, 400.0), 100.0, earth)
    o.add_sphere((220.0, 280.0, 300.0), 80.0, marble)
    cluster = PrimSet()
    r = _hash01(np.arange(3000, dtype=np.int64) + 77777, seed).reshape(1000, 3) * 165.0
    for k in range(1000):
        cluster.add_sphere((float(r[k, 0]), float(r[k, 1]), float(r[k, 2])), 10.0, white)
    cluster.build_bvh = True
    ci = InstanceSrc(cluster).rotate_y(_radians_f32(15.0)).translate((-100.0, 270.0, 395.0))
    o.add_instance(ci)
    o.build_bvh = True
    if extensions:
        s.add_medium((360.0, 150.0, 145.0), 70.0, 0.2, s.solid((0.2, 0.4, 0.9)))
        s.add_medium((0.0, 0.0, 0.0), 5000.0, 0.0001, s.solid((1.0, 1.0, 1.0)))
    s.camera = CameraSrc(1.0, width, 40.0, (478.0, 278.0, -600.0), (278.0, 278.0, 0.0), 1.0, 0.0, spp, max_depth,
                         (0.0, 0.0, 0.0))
    return s


# ------------------------------------------------------------------ small scenes
def small_mixed(width: int = 64, spp: int = 16, max_depth: int = 12, seed: int = 3) -> SceneSource:
    """A small scene that touches every primitive type, every material type, all three texture
    types, an instance with a BVH-less child and an instance with a BVH'd child."""
    s = SceneSource()
    s.name = "small_mixed"
    img = (np.arange(8 * 8 * 3, dtype=np.uint32).reshape(8, 8, 3) * 37 % 256).astype(np.uint8)
    t_img = s.image(img)
    t_chk = s.checkered(s.solid((0.9, 0.1, 0.1)), s.checkered(s.solid((0.1, 0.9, 0.1)), s.image(None), 3.0), 5.0)
    mats = [
        s.lambertian((0.7, 0.7, 0.7)),
        s.lambertian_texture(t_img),
        s.lambertian_texture(t_chk),
        s.metal((0.8, 0.85, 0.88), 0.0),
        s.metal((0.9, 0.6, 0.3), 0.4),
        s.dielectric(1.5),
        s.disney(s.solid((0.8, 0.6, 0.2)), 0.5, 0.1, 0.0, 0.01, 0.2, 0.1, 0.8, 0.3, 0.1, 0.2, 0.5, 1.1),
        s.disney(s.solid((0.3, 0.5, 0.9)), 0.2, 0.0, 0.8, 0.5, 0.0, 0.3, 0.2, 0.0, 0.0, 1.0, 0.9, 1.45),
        s.disney_diffuse((0.3, 0.3, 0.8), 1.0, 0.5),
        s.disney_metal((0.3, 0.3, 0.8), 0.5, 0.3),
        s.disney_glass((0.5, 0.5, 0.8), 0.1, 0.0, 1.2),
        s.disney_clearcoat(1.0, 0.8),
        s.disney_sheen((0.3, 0.3, 0.8), 1.0, 0.2),
    ]
    emit = s.diffuse_light_texture(s.solid((6.0, 6.0, 5.0)))
    o = s.objects
    o.add_quad((-6, 0, -6), (12, 0, 0), (0, 0, 12), mats[2])       # floor (checker)
    o.add_quad((-6, 0, 6), (12, 0, 0), (0, 8, 0), mats[1])         # back wall (image)
    o.add_quad((-6, 0, -6), (0, 0, 12), (0, 8, 0), mats[0])
    o.add_quad((-2, 7.9, -2), (4, 0, 0), (0, 0, 4), emit)
    s.lights.add_quad((-2, 7.9, -2), (4, 0, 0), (0, 0, 4), emit)
    s.lights.add_sphere((4.5, 6.0, 0.0), 0.6, emit)
    o.add_sphere((4.5, 6.0, 0.0), 0.6, emit)
    h = _hash01(np.arange(64, dtype=np.int64), seed)
    for k in range(len(mats)):
        cx = -4.5 + 0.75 * k
        cz = -2.0 + 3.0 * float(h[k])
        o.add_sphere((cx, 0.6 + 0.5 * float(h[k + 20]), cz), 0.45 + 0.2 * float(h[k + 40]), mats[k])
    # a small tessellated blob with normals + uvs, and a flat-shaded one without
    P, N, UV, _, _ = _sphere_grid((0.0, 2.5, 2.0), lambda t, p: 1.2 + 0.15 * np.sin(3 * t) * np.sin(2 * p), 24, 12)
    pos, nrm, uv = _grid_to_tris(P, N, UV)
    o.add_triangles(pos, nrm, uv, mats[6])
    P, N, UV, _, _ = _sphere_grid((-3.0, 3.0, 3.0), lambda t, p: 0.9 + 0.0 * t, 12, 6)
    pos, _, _ = _grid_to_tris(P, N, UV)
    o.add_triangles(pos, None, None, mats[7])
    tri_light = np.array([[[2.0, 7.5, 3.0], [3.5, 7.5, 3.0], [2.0, 7.5, 4.5]]], dtype=np.float32)
    o.add_triangles(tri_light, None, None, emit)
    s.lights.add_triangles(tri_light, None, None, emit)
    box = InstanceSrc(new_box((0, 0, 0), (1.5, 3.0, 1.5), mats[3])).rotate_y(_radians_f32(20.0)).translate((2.0, 0.0, -3.0))
    o.add_instance(box)
    cl = PrimSet()
    for k in range(24):
        cl.add_sphere((float(h[k]) * 2, float(h[k + 8]) * 2, float(h[k + 16]) * 2), 0.18, mats[k % len(mats)])
    P, N, UV, _, _ = _sphere_grid((1.0, 1.0, 1.0), lambda t, p: 0.5 + 0.0 * t, 10, 5)
    pos, nrm, uv = _grid_to_tris(P, N, UV)
    cl.add_triangles(pos, nrm, uv, mats[9])
    cl.build_bvh = True
    o.add_instance(InstanceSrc(cl).rotate(0.4, (1.0, 1.0, 0.0)).scale_uniform(1.2).translate((-5.0, 0.5, -4.0)))
    o.build_bvh = True
    s.camera = CameraSrc(1.0, width, 50.0, (0.0, 4.0, -11.0), (0.0, 3.0, 0.0), 10.0, 0.6, spp, max_depth, (0.1, 0.12, 0.2))
    return s


def extensions_demo(width: int = 64, spp: int = 16, max_depth: int = 10) -> SceneSource:
    """A small scene with the three things this build adds to the reference's vocabulary: a noise texture, a
    participating medium and a moving sphere, next to ordinary geometry."""
    s = SceneSource()
    s.name = "extensions_demo"
    marble = s.lambertian_texture(s.noise((0.5, 0.5, 0.5), 0.8, 5))
    veined = s.lambertian_texture(s.checkered(s.noise((0.8, 0.3, 0.2), 1.5, 3), s.solid((0.25, 0.25, 0.25)), 4.0))
    emit = s.diffuse_light((5.0, 5.0, 4.5))
    o = s.objects
    o.add_quad((-10.0, 0.0, -10.0), (20.0, 0.0, 0.0), (0.0, 0.0, 20.0), veined)
    o.add_sphere((-2.2, 1.0, 0.5), 1.0, marble)
    o.add_sphere((0.0, 1.0, -1.0), 1.0, s.dielectric(1.5))
    o.add_sphere((2.4, 0.8, 2.0), 0.8, s.metal((0.8, 0.8, 0.7), 0.2))
    o.add_quad((-2.0, 6.0, -2.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), emit)
    s.lights.add_quad((-2.0, 6.0, -2.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), emit)
    o.build_bvh = True
    s.add_medium((0.0, 1.0, -1.0), 1.0, 1.5, s.solid((0.2, 0.4, 0.9)))      # inside the glass sphere
    s.add_medium((0.0, 0.0, 0.0), 30.0, 0.02, s.solid((1.0, 1.0, 1.0)))     # mist
    s.add_moving_sphere((2.0, 2.6, -1.5), (2.8, 2.9, -1.5), 0.6, s.lambertian((0.7, 0.3, 0.1)))
    s.camera = CameraSrc(1.0, width, 35.0, (0.0, 3.0, -12.0), (0.0, 1.2, 0.0), 12.0, 0.0, spp, max_depth, (0.35, 0.45, 0.7))
    return s
