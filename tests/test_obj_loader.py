"""The .obj loader of the host layer against an independent restatement of tobj's default options
(oracle/objref.py) and against hand-written expectations.  No GPU needed."""
import os

import numpy as np
import pytest

from oracle import objref
from pathtracer_amd import core

CUBE = """# a cube in two objects, with a material library
mtllib cube.mtl
o bottom
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1 0.5 0.5 0.5
v 1 0 1
v 1 1 1
v 0 1 1
vn 0 0 -1
vn 0 0 1
vt 0 0
vt 1 0 0
vt 1 1
vt 0 1
usemtl red
f 1/1/1 3/3/1 2/2/1
f 1/1/1 4/4/1 3/3/1
o top
usemtl blue
f 5/1/2 6/2/2 7/3/2
f -4/1/-1 -2/3/2 -1/4/2
usemtl red
f 1//1 2//1 6//2
"""
MTL = "newmtl red\nKd 1 0 0\nnewmtl blue\nKd 0 0 1\n"


def _write(tmp_path, name, text, mtl=None):
    p = tmp_path / name
    p.write_bytes(text.encode())
    if mtl is not None:
        (tmp_path / "cube.mtl").write_text(mtl)
    return str(p)


def _same(a, b):
    assert len(a) == len(b)
    for (pa, na, ua), mb in zip(a, b):
        pb, nb, ub = objref.triangles(mb)
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
        assert (na is None) == (nb is None) and (ua is None) == (ub is None)
        if na is not None:
            assert np.array_equal(na.view(np.uint32), nb.view(np.uint32))
        if ua is not None:
            assert np.array_equal(ua.view(np.uint32), ub.view(np.uint32))


def test_models_split_at_objects_and_material_changes(tmp_path):
    path = _write(tmp_path, "cube.obj", CUBE, MTL)
    meshes = core.load_obj(path)
    # bottom (red) | top (blue) | top again after `usemtl red`
    assert [m[0].shape[0] for m in meshes] == [2, 2, 1]
    assert np.array_equal(meshes[0][0][0], np.array([[0, 0, 0], [1, 1, 0], [1, 0, 0]], np.float32))
    assert np.array_equal(meshes[0][2][1], np.array([[0, 0], [0, 1], [1, 1]], np.float32))  # vt keeps two components
    # relative indices: -4 is vertex 5, -2 vertex 7, -1 vertex 8; normal -1 is the second normal
    assert np.array_equal(meshes[1][0][1], np.array([[0, 0, 1], [1, 1, 1], [0, 1, 1]], np.float32))
    assert np.array_equal(meshes[1][1][1][0], np.array([0, 0, 1], np.float32))
    # the last model uses v//vn: normals, no texcoords
    assert meshes[2][1] is not None and meshes[2][2] is None
    _same(meshes, objref.load(path))


def test_without_a_readable_mtl_material_changes_do_not_split(tmp_path):
    # cube.mtl missing: the material table stays empty, `usemtl` never starts a new model
    path = _write(tmp_path, "cube.obj", CUBE.replace("f 1//1 2//1 6//2", "f 1/1/1 2/2/1 6/2/2"))
    meshes = core.load_obj(path)
    assert [m[0].shape[0] for m in meshes] == [2, 3]
    _same(meshes, objref.load(path))


def test_crlf_comments_groups_and_positions_only(tmp_path):
    text = "# header\r\ng first\r\nv 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n\r\ng second\r\nv 0 0 1e0\r\nv 1 0 1.\r\nv 0 1 +1\r\nf 4 5 6\r\nf 6 5 4\r\n"
    path = _write(tmp_path, "a.obj", text)
    meshes = core.load_obj(path)
    assert [m[0].shape[0] for m in meshes] == [1, 2]
    assert all(m[1] is None and m[2] is None for m in meshes)
    assert np.array_equal(meshes[1][0][1, :, 2], np.ones(3, np.float32))
    _same(meshes, objref.load(path))


def test_random_meshes_match_the_restatement(tmp_path):
    rng = np.random.default_rng(7)
    for case in range(6):
        nv, nf = int(rng.integers(4, 60)), int(rng.integers(1, 80))
        lines = ["o m0"]
        for v in rng.normal(size=(nv, 3)):
            lines.append("v " + " ".join(repr(float(np.float32(x))) for x in v))
        style = case % 3
        if style >= 1:
            for v in rng.normal(size=(nv, 3)):
                lines.append("vn " + " ".join(f"{x:.7g}" for x in v))
        if style == 2:
            for v in rng.uniform(size=(nv, 2)):
                lines.append("vt " + " ".join(f"{x:.6f}" for x in v))
        for k in range(nf):
            if k and k % 17 == 0:
                lines.append(f"g part{k}")
            idx = rng.integers(1, nv + 1, size=3)
            if rng.uniform() < 0.3:
                idx = idx - nv - 1  # relative
            word = {0: lambda i: f"{i}", 1: lambda i: f"{i}//{i}", 2: lambda i: f"{i}/{i}/{i}"}[style]
            lines.append("f " + " ".join(word(int(i)) for i in idx))
        path = _write(tmp_path, f"r{case}.obj", "\n".join(lines) + "\n")
        _same(core.load_obj(path), objref.load(path))


@pytest.mark.parametrize("text", [
    "v 0 0\nf 1 1 1\n",                  # short position
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n",  # index out of bounds
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 x\n",  # not an integer
    "v 0 0 0x10\n",                      # not a decimal float
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3\n",  # some vertices with a normal, some without
    "usemtl\n",
])
def test_malformed_files_fail_on_both_sides(tmp_path, text):
    path = _write(tmp_path, "bad.obj", text)
    with pytest.raises(core.PtError):
        core.load_obj(path)
    with pytest.raises(objref.ObjError):
        objref.load(path)


def test_missing_file_and_untriangulated_faces(tmp_path):
    with pytest.raises(core.PtError):
        core.load_obj(str(tmp_path / "nope.obj"))
    quad = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n"  # four indices: not a whole number of triangles
    path = _write(tmp_path, "quad.obj", quad)
    with pytest.raises(core.PtError):
        core.load_obj(path)
    with pytest.raises(objref.ObjError):
        objref.triangles(objref.load(path)[0])


def test_loaded_mesh_renders_like_the_same_triangles_given_directly(tmp_path):
    """The scene built from a loaded .obj and the one built from the same arrays have identical BVHs."""
    from pathtracer_amd import scenes
    from pathtracer_amd.core import HostScene
    src = scenes.knob(32, 16, detail=0.25)
    tri_sets = [(p, n, u) for p, n, u in zip(src.objects.tri_pos, src.objects.tri_nrm, src.objects.tri_uv)]
    lines = []
    vbase = 0
    for k, (p, n, u) in enumerate(tri_sets):
        lines.append(f"o mesh{k}")
        flat = p.reshape(-1, 3)
        for v in flat:
            lines.append("v " + " ".join(repr(float(x)) for x in v))
        if n is not None:
            for v in n.reshape(-1, 3):
                lines.append("vn " + " ".join(repr(float(x)) for x in v))
        if u is not None:
            for v in u.reshape(-1, 2):
                lines.append("vt " + " ".join(repr(float(x)) for x in v))
        for t in range(p.shape[0]):
            i = [vbase + 3 * t + c + 1 for c in range(3)]
            if n is not None and u is not None:
                lines.append("f " + " ".join(f"{a}/{a}/{a}" for a in i))
            elif n is not None:
                lines.append("f " + " ".join(f"{a}//{a}" for a in i))
            else:
                lines.append("f " + " ".join(str(a) for a in i))
        vbase += flat.shape[0]
    # vn/vt lists are file-global and 1-based like v: when a mesh lacks them the indices above would
    # drift, so this test needs every mesh to carry the same attributes
    assert len({(n is None, u is None) for _, n, u in tri_sets}) == 1
    path = _write(tmp_path, "knob.obj", "\n".join(lines) + "\n")
    meshes = core.load_obj(path)
    assert len(meshes) == len(tri_sets)
    for (p, n, u), (lp, ln, lu) in zip(tri_sets, meshes):
        assert np.array_equal(p.view(np.uint32), lp.view(np.uint32))
        if n is not None:
            assert np.array_equal(n.view(np.uint32), ln.view(np.uint32))
        if u is not None:
            assert np.array_equal(u.view(np.uint32), lu.view(np.uint32))
    src2 = scenes.knob(32, 16, detail=0.25, asset=path)
    a, b = HostScene(src), HostScene(src2)
    ta, tb = a.desc.sets[0].tris, b.desc.sets[0].tris
    assert np.array_equal(a.nodes(ta).view(np.uint8), b.nodes(tb).view(np.uint8))
    assert np.array_equal(a.triangles(ta).view(np.uint8), b.triangles(tb).view(np.uint8))
