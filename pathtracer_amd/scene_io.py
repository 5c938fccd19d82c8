"""Round trip of a :class:`SceneSource` through plain numpy arrays (``.npz`` friendly)."""
from __future__ import annotations

import json

import numpy as np

from .scene_src import CameraSrc, InstanceSrc, PrimSet, SceneSource


def _set_to_dict(s: PrimSet, sets: list) -> int:
    idx = len(sets)
    d = {"build_bvh": bool(s.build_bvh), "spheres": [list(map(float, x)) for x in s.spheres], "sphere_mat": list(map(int, s.sphere_mat)),
         "quads": [list(map(float, x)) for x in s.quads], "quad_mat": list(map(int, s.quad_mat)), "tri_chunks": [], "instances": []}
    sets.append(d)
    for k in range(len(s.tri_pos)):
        d["tri_chunks"].append({"has_nrm": s.tri_nrm[k] is not None, "has_uv": s.tri_uv[k] is not None, "n": int(s.tri_pos[k].shape[0])})
    for inst in s.instances:
        child = _set_to_dict(inst.child, sets)
        d["instances"].append({"child": child, "ops": [list(op) for op in inst.ops]})
    return idx


def _collect_tris(s: PrimSet, out: dict):
    for k in range(len(s.tri_pos)):
        out["pos"].append(s.tri_pos[k])
        out["nrm"].append(s.tri_nrm[k] if s.tri_nrm[k] is not None else np.zeros_like(s.tri_pos[k]))
        out["uv"].append(s.tri_uv[k] if s.tri_uv[k] is not None else np.zeros((s.tri_pos[k].shape[0], 3, 2), np.float32))
        out["mat"].append(s.tri_mat[k])
    for inst in s.instances:
        _collect_tris(inst.child, out)


def scene_to_arrays(src: SceneSource) -> dict:
    """Everything needed to rebuild `src`: a JSON structure plus the triangle arrays (float32 kept bit for bit)."""
    sets: list = []
    _set_to_dict(src.objects, sets)
    lights_index = _set_to_dict(src.lights, sets)
    tri = {"pos": [], "nrm": [], "uv": [], "mat": []}
    _collect_tris(src.objects, tri)
    _collect_tris(src.lights, tri)
    cam = src.camera
    meta = {"name": src.name, "sets": sets, "lights_index": lights_index,
            "textures": [[t[0], list(t[1]), t[2], t[3], t[4], t[5]] for t in src.textures],
            "media": [[list(m[0]), m[1], m[2], m[3]] for m in src.media],
            "moving": [[list(m[0]), list(m[1]), m[2], m[3]] for m in src.moving],
            "materials": [[m[0], m[1], list(m[2])] for m in src.materials],
            "camera": [cam.aspect_ratio, cam.image_width, cam.vfov, list(cam.look_from), list(cam.look_at), cam.focal_length,
                       cam.defocus_angle, cam.spp, cam.max_depth, list(cam.background)]}

    def cat(lst, shape):
        return np.concatenate(lst, axis=0).astype(np.float32) if lst else np.zeros(shape, np.float32)

    out = {"meta_json": np.frombuffer(json.dumps(meta).encode(), np.uint8), "tri_pos": cat(tri["pos"], (0, 3, 3)),
           "tri_nrm": cat(tri["nrm"], (0, 3, 3)), "tri_uv": cat(tri["uv"], (0, 3, 2)),
           "tri_mat": np.concatenate(tri["mat"]).astype(np.int32) if tri["mat"] else np.zeros(0, np.int32),
           # exact float32 payloads of the JSON numbers that must survive bit for bit
           "f32_spheres": np.asarray([x for s in _all_sets(src) for x in s.spheres], np.float32).reshape(-1, 4),
           "f32_quads": np.asarray([x for s in _all_sets(src) for x in s.quads], np.float32).reshape(-1, 9)}
    for k, im in enumerate(src.images):
        out[f"image_{k}"] = im
    return out


def _all_sets(src: SceneSource):
    order = []

    def walk(s):
        order.append(s)
        for inst in s.instances:
            walk(inst.child)

    walk(src.objects)
    walk(src.lights)
    return order


def scene_from_arrays(a) -> SceneSource:
    meta = json.loads(bytes(a["meta_json"]).decode())
    src = SceneSource()
    src.name = meta["name"]
    src.textures = [(t[0], tuple(t[1]), t[2], t[3], t[4], t[5]) for t in meta["textures"]]
    src.media = [(tuple(m[0]), m[1], m[2], m[3]) for m in meta.get("media", [])]
    src.moving = [(tuple(m[0]), tuple(m[1]), m[2], m[3]) for m in meta.get("moving", [])]
    src.materials = [(m[0], m[1], list(m[2])) for m in meta["materials"]]
    k = 0
    while f"image_{k}" in a:
        src.images.append(np.asarray(a[f"image_{k}"], np.uint8))
        k += 1
    c = meta["camera"]
    src.camera = CameraSrc(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9])
    sets = meta["sets"]
    cursor = {"tri": 0, "sph": 0, "quad": 0}

    def build(idx: int) -> PrimSet:
        d = sets[idx]
        s = PrimSet()
        s.build_bvh = d["build_bvh"]
        for ch in d["tri_chunks"]:
            n = ch["n"]
            b = cursor["tri"]
            s.add_triangles(a["tri_pos"][b:b + n], a["tri_nrm"][b:b + n] if ch["has_nrm"] else None,
                            a["tri_uv"][b:b + n] if ch["has_uv"] else None, a["tri_mat"][b:b + n])
            cursor["tri"] += n
        for m in d["sphere_mat"]:
            row = a["f32_spheres"][cursor["sph"]]
            s.spheres.append([float(x) for x in row])
            s.sphere_mat.append(m)
            cursor["sph"] += 1
        for m in d["quad_mat"]:
            row = a["f32_quads"][cursor["quad"]]
            s.quads.append([float(x) for x in row])
            s.quad_mat.append(m)
            cursor["quad"] += 1
        for inst in d["instances"]:
            i = InstanceSrc(build(inst["child"]))
            i.ops = [tuple(op) for op in inst["ops"]]
            s.instances.append(i)
        return s

    # triangles were collected depth-first per set, spheres/quads in _all_sets order: rebuild in the same order
    src.objects = build(0)
    src.lights = build(meta["lights_index"])
    return src
