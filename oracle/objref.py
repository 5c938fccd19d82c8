"""TEST INFRASTRUCTURE — a second, independent restatement (pure Python) of how the reference turns a
Wavefront .obj file into triangles: tobj 4.0.2 with ``LoadOptions::default()`` (call site
src/objects/tri_mesh.rs:301) followed by ``TriMesh::to_triangles`` (tri_mesh.rs:383-415).

Parity unpinned: tobj is a crates.io dependency (Cargo.toml:16) whose source is not in the reference
tree and the reference holds no .obj fixture with expected output, so this file and
pathtracer_amd/host/objload.cpp are two readings of the crate's published behaviour that the tests hold
against each other.  Only ``tests/`` may import it.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Optional, Tuple

import numpy as np

_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")
_SPECIAL = re.compile(r"^[+-]?(inf|infinity|nan)$", re.I)
_INT = re.compile(r"^[+-]?\d+$")


class ObjError(Exception):
    pass


def _floats(words: List[str], n: int, what: str) -> List[float]:
    out = []
    for w in words[:n]:
        if not (_FLOAT.match(w) or _SPECIAL.match(w)):
            raise ObjError(f"{what} parse error")
        out.append(float(np.float32(float(w))))
    if len(out) != n:
        raise ObjError(f"{what} parse error")
    return out


def _vertex(word: str, counts: Tuple[int, int, int]):
    idx = [None, None, None]
    for field, part in enumerate(word.split("/")):
        if part == "":
            continue
        if field > 2 or not _INT.match(part):
            raise ObjError("face parse error")
        x = int(part)
        idx[field] = counts[field] + x if x < 0 else x - 1
    return tuple(idx)


def _export(pos, tex, nrm, faces):
    """One model: (triangle positions (n,3,3), normals (n,3,3) or None, uvs (n,3,2) or None)."""
    verts = [v for f in faces for v in f]
    with_vt = sum(v[1] is not None for v in verts)
    with_vn = sum(v[2] is not None for v in verts)
    use_vt = bool(tex) and with_vt > 0
    use_vn = bool(nrm) and with_vn > 0
    if (use_vt and with_vt != len(verts)) or (use_vn and with_vn != len(verts)):
        raise ObjError("mixed attribute indices")
    for v in verts:
        if v[0] is None or not (0 <= v[0] < len(pos)):
            raise ObjError("face vertex index out of bounds")
        if use_vt and not (0 <= v[1] < len(tex)):
            raise ObjError("face texcoord index out of bounds")
        if use_vn and not (0 <= v[2] < len(nrm)):
            raise ObjError("face normal index out of bounds")
    return {"verts": verts, "use_vt": use_vt, "use_vn": use_vn}


def load(path: str):
    """List of models; each a dict with pos (k,3) float32 per index, nrm/uv likewise or None, and the
    flat index count.  Raises ObjError where the reference would end up with no meshes."""
    if not os.path.exists(path):
        raise ObjError("cannot open")
    pos, tex, nrm = [], [], []
    faces: list = []
    models = []
    mats: Dict[str, int] = {}
    mat_id: Optional[int] = None
    base = os.path.dirname(path)

    def flush():
        e = _export(pos, tex, nrm, faces)
        v = e["verts"]
        models.append({
            "pos": np.array([pos[a[0]] for a in v], np.float32).reshape(-1, 3),
            "nrm": np.array([nrm[a[2]] for a in v], np.float32).reshape(-1, 3) if e["use_vn"] else None,
            "uv": np.array([tex[a[1]] for a in v], np.float32).reshape(-1, 2) if e["use_vt"] else None,
        })

    with open(path, "r", newline="") as f:
        for raw in f.read().split("\n"):
            line = raw[:-1] if raw.endswith("\r") else raw
            words = line.split()
            if not words or words[0] == "#":
                continue
            kw = words[0]
            if kw == "v":
                pos.append(_floats(words[1:], 3, "position"))
            elif kw == "vt":
                tex.append(_floats(words[1:], 2, "texcoord"))
            elif kw == "vn":
                nrm.append(_floats(words[1:], 3, "normal"))
            elif kw in ("f", "l"):
                if len(words) < 2:
                    raise ObjError("face parse error")
                faces.append([_vertex(w, (len(pos), len(tex), len(nrm))) for w in words[1:]])
            elif kw in ("o", "g"):
                if faces:
                    flush()
                    faces = []
            elif kw == "mtllib":
                name = line[line.index("mtllib") + 6:].strip()
                try:
                    with open(os.path.join(base, name)) as mf:
                        for ml in mf:
                            mw = ml.split()
                            if mw and mw[0] == "newmtl":
                                nm = ml[ml.index("newmtl") + 6:].strip()
                                if nm and nm not in mats:
                                    mats[nm] = len(mats)
                except OSError:
                    pass
            elif kw == "usemtl":
                name = line[line.index("usemtl") + 6:].strip()
                if not name:
                    raise ObjError("material parse error")
                new = mats.get(name)
                if new != mat_id and faces:
                    flush()
                    faces = []
                mat_id = new
    flush()
    return models


def triangles(model):
    """TriMesh::to_triangles: the per-index attribute streams taken three at a time."""
    k = model["pos"].shape[0]
    if k % 3:
        raise ObjError("index stream is not a multiple of three")
    n = k // 3
    return (model["pos"].reshape(n, 3, 3),
            None if model["nrm"] is None else model["nrm"].reshape(n, 3, 3),
            None if model["uv"] is None else model["uv"].reshape(n, 3, 2))
