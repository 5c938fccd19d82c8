#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run once, from the repository root):

    python tests/golden/make_golden.py

The reference renderer is a Rust crate that cannot be built or run here and ships no test
vectors, so these are not outputs of the reference: they pin the oracle (and, through the
parity tests, the device path) against accidental change.  Each file holds a small scene in
the authoring-level format (so that it does not depend on how this machine's numpy rounds
procedural geometry) together with what the oracle computes for it.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyoracle import OracleScene  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402
from pathtracer_amd.scene_io import scene_to_arrays  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def bvh_digest(o: OracleScene) -> str:
    h = hashlib.sha256()
    for set_ in (0, 1):
        for type_ in range(4):
            bb, meta = o.bvh(set_, type_)
            h.update(bb.tobytes()); h.update(meta.tobytes()); h.update(o.perm(set_, type_).tobytes())
    return h.hexdigest()


def make(name, src, n_strata=2, seed=11, nee=False):
    o = OracleScene(src)
    rays = o.primary_rays()
    ids, tuv = o.trace(rays)
    jit, _ = o.sample_rays(seed=seed, stratum=1)
    rec = o.render(0, n_strata, seed=seed, iterative=False, per_sample=True)
    it = o.render(0, n_strata, seed=seed, iterative=True, per_sample=True)
    out = scene_to_arrays(src)
    out.update(
        camera_constants=o.camera_constants(), first_hit_ids=ids, first_hit_tuv=tuv, jittered_rays_stratum1=jit,
        per_sample_recursive=rec["per_sample"], per_sample_iterative=it["per_sample"], rays_traced=np.array([rec["rays"]], np.uint64),
        bvh_sha256=np.frombuffer(bvh_digest(o).encode(), np.uint8), seed=np.array([seed], np.uint64), n_strata=np.array([n_strata], np.uint32))
    if nee:  # the next-event estimator of this build (not in the reference): ray_color_nee
        out["per_sample_nee"] = o.render(0, n_strata, seed=seed, iterative=2, per_sample=True)["per_sample"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "rays", rec["rays"], "hits", int((ids[:, 0] >= 0).sum()), "of", ids.shape[0])


if __name__ == "__main__":
    which = sys.argv[1:]  # names to (re)write; none = all.  Existing fixtures are only rewritten on purpose.
    todo = {"cornell_48": lambda: make("cornell_48", scenes.cornell(48, 16, 16)),
            "small_mixed_40": lambda: make("small_mixed_40", scenes.small_mixed(40, 16, 12)),
            "small_mixed_nee_32": lambda: make("small_mixed_nee_32", scenes.small_mixed(32, 16, 12), nee=True),
            "extensions_demo_40": lambda: make("extensions_demo_40", scenes.extensions_demo(40, 16, 10))}
    for name, fn in todo.items():
        if not which or name in which:
            fn()
