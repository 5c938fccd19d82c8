#!/usr/bin/env python3
"""Distribution of device-vs-oracle differences of pt_material_probe, per material and field (run on the GPU box).
The thresholds of tests/test_gpu_materials.py come from this table."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleScene  # noqa: E402
from pathtracer_amd.core import HostScene, Renderer  # noqa: E402
from tests.test_gpu_materials import gallery, make_probes  # noqa: E402
from tests.util import ulp_diff  # noqa: E402

src, mats = gallery()
oracle = OracleScene(src)
r = Renderer(HostScene(src), device=0)
rows = {}
for name, mat in mats.items():
    probes = make_probes(48000, seed=1000 + mat)
    got = r.material_probe(mat, probes, seed=77)
    want = oracle.material_probe(mat, probes, seed=77)
    row = {"draws_differ": float(np.mean(got["draws"] != want["draws"]))}
    for f in ("w_in", "bsdf", "pdf", "bsdf_at", "pdf_at", "emitted"):
        g = got[f].reshape(len(probes), -1)
        w = want[f].reshape(len(probes), -1)
        fin = np.isfinite(g) & np.isfinite(w)
        cls = float(np.mean((np.isnan(g) != np.isnan(w)) | (np.isinf(g) != np.isinf(w))))
        u = ulp_diff(np.where(fin, g, 0).astype(np.float32), np.where(fin, w, 0).astype(np.float32))
        a = np.abs(np.where(fin, g, 0).astype(np.float64) - np.where(fin, w, 0))
        small = a <= 1e-6
        row[f] = {"class_mismatch": cls, "bit_equal": float(np.mean(u == 0)), "le4": float(np.mean((u <= 4) | small)), "le64": float(np.mean((u <= 64) | small)),
                  "le4096": float(np.mean((u <= 4096) | small)), "max_ulp": int(u.max()), "nan_frac": float(np.mean(np.isnan(w)))}
    rows[name] = row
    print(name, json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "material_probe_stats.json"), "w") as f:
    json.dump(rows, f, indent=1)
