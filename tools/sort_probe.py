#!/usr/bin/env python3
"""How much would the closest-hit search of bounce rays gain from an ordered ray queue?

Builds realistic bounce rays on the host (first hits of camera rays through the C ABI, a cosine-weighted
direction about the geometric normal at each hit; a second generation the same way from the hits of the first),
then times pt_trace_first_hit's search kernel (PT_OPT_PROFILE: events around it) on the same rays in several
orders: as the shading kernels would queue them (path order), shuffled, and sorted by direction octant and / or
the cell of the origin in a grid over the rays' bounding box.

    python tools/sort_probe.py knob 768 [strata]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pathtracer_amd import scenes  # noqa: E402
from pathtracer_amd.core import PT_OPT_PROFILE, PT_OPT_TRAVERSAL, HostScene, Renderer  # noqa: E402


def normals_of(host, hits, rays):
    """Geometric normal at every hit, facing the ray; NaN rows for misses and hits inside instances."""
    n = np.full((hits.size, 3), np.nan, np.float32)
    w = host.set_desc(0)
    ok = (hits["prim_type"] >= 0) & (hits["inst_id"] < 0)
    tri = ok & (hits["prim_type"] == 0)
    if tri.any():
        t = host.triangles(w.tris)[hits["prim_id"][tri]]
        v0, v1, v2 = t[:, [0, 3, 6]], t[:, [1, 4, 7]], t[:, [2, 5, 8]]
        n[tri] = np.cross(v1 - v0, v2 - v0)
    q = ok & (hits["prim_type"] == 2)
    if q.any():
        n[q] = host.quads(w.quads)[hits["prim_id"][q]][:, 9:12]
    s = ok & (hits["prim_type"] == 1)
    if s.any():
        sph = np.ctypeslib.as_array(host.desc.spheres, shape=(host.desc.n_spheres * 4,)).reshape(-1, 4)[hits["prim_id"][s]]
        p = rays[s, 0:3] + hits["t"][s, None] * rays[s, 3:6]
        n[s] = p - sph[:, :3]
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    flip = np.einsum("ij,ij->i", n, rays[:, 3:6]) > 0
    n[flip] = -n[flip]
    return n


def bounce(host, r, rays, rng):
    hits = r.trace(rays)
    nrm = normals_of(host, hits, rays)
    keep = np.isfinite(nrm).all(axis=1)
    p = (rays[:, 0:3] + hits["t"][:, None] * rays[:, 3:6])[keep]
    nrm = nrm[keep]
    a = np.where(np.abs(nrm[:, 0:1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    t1 = np.cross(nrm, a)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    r1, r2 = rng.random(p.shape[0], np.float32), rng.random(p.shape[0], np.float32)
    phi = 2 * np.pi * r1
    d = (np.cos(phi) * np.sqrt(r2))[:, None] * t1 + np.sqrt(1 - r2)[:, None] * nrm + (np.sin(phi) * np.sqrt(r2))[:, None] * t2
    out = np.zeros((p.shape[0], 8), np.float32)
    out[:, 0:3], out[:, 3:6], out[:, 6], out[:, 7] = p, d, 0.001, np.inf
    return out


def part1by2(x):
    x = x.astype(np.uint32) & 0x3FF
    x = (x | (x << 16)) & 0x030000FF
    x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3
    x = (x | (x << 2)) & 0x09249249
    return x


def keys(rays, bits):
    o, d = rays[:, 0:3], rays[:, 3:6]
    octant = ((d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2))
    lo, hi = np.percentile(o, 1, axis=0), np.percentile(o, 99, axis=0)
    g = np.clip((o - lo) / np.maximum(hi - lo, 1e-9), 0, 0.999999)
    c = (g * (1 << bits)).astype(np.uint32)
    morton = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    return octant, morton


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "knob"
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    strata = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    src = {"knob": lambda: scenes.knob(width=width), "dragon_box": lambda: scenes.dragon_box(width=width),
           "interior": lambda: scenes.interior(width=width)}[name]()
    host = HostScene(src)
    r = Renderer(host, device=0)
    rng = np.random.default_rng(7)
    cam = np.concatenate([r.generate_rays(s, seed=3) for s in range(strata)])
    # path order of a batch: pixel-major (the strata of a pixel next to each other)
    cam = cam.reshape(strata, -1, 8).transpose(1, 0, 2).reshape(-1, 8)
    b1 = bounce(host, r, cam, rng)
    b2 = bounce(host, r, b1, rng)
    sets = {"bounce1": b1, "bounce2": b2, "tick (bounce1 then bounce2)": np.concatenate([b1, b2])}
    r.set_option(PT_OPT_PROFILE, 1)
    res = {"scene": name, "width": width, "strata": strata}
    for label, rays in sets.items():
        octant, m3 = keys(rays, 3)
        _, m4 = keys(rays, 4)
        _, m5 = keys(rays, 5)
        orders = {
            "path order": np.arange(rays.shape[0]),
            "shuffled": rng.permutation(rays.shape[0]),
            "octant": np.argsort(octant, kind="stable"),
            "octant, cell 8^3": np.argsort((octant << 9) | m3, kind="stable"),
            "cell 8^3, octant": np.argsort((m3 << 3) | octant, kind="stable"),
            "cell 16^3, octant": np.argsort((m4 << 3) | octant, kind="stable"),
            "octant, cell 32^3": np.argsort((octant << 15) | m5, kind="stable"),
        }
        row = {"rays": int(rays.shape[0])}
        ref = None
        for oname, perm in orders.items():
            rr = np.ascontiguousarray(rays[perm])
            best = None
            for _ in range(2):
                r.stats_reset()
                h = r.trace(rr)
                ms = r.stats()["extend_ms"]
                best = ms if best is None else min(best, ms)
            if ref is None:
                ref = h
            else:  # same hits whatever the order
                back = np.empty_like(h)
                back[perm] = h
                assert np.array_equal(back["prim_id"], ref["prim_id"]) and np.array_equal(back["t"].view(np.uint32), ref["t"].view(np.uint32))
            row[oname] = {"ms": round(best, 3), "Grays_s": round(rays.shape[0] / best / 1e6, 3)}
            print(label, oname, row[oname], flush=True)
        res[label] = row
    print(json.dumps(res))


if __name__ == "__main__":
    main()
