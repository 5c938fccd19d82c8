"""Traces the rays of the degenerate-ray parity test on one scene and prints where device and oracle differ."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import scenes
from oracle.pyoracle import OracleScene
from pathtracer_amd.core import HostScene, Renderer
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _random_rays

name = sys.argv[1] if len(sys.argv) > 1 else "next_week_like"
src = {"small_mixed": lambda: scenes.small_mixed(32, 16), "knob": lambda: scenes.knob(32, 16),
       "next_week_like": lambda: scenes.next_week_like(32, 16), "interior": lambda: scenes.interior(32, 16, n_lights=16, columns=3)}[name]()
o, r = OracleScene(src), Renderer(HostScene(src))
rng = np.random.default_rng(5)
box = {"small_mixed": (-7, 9), "knob": (-3, 4), "next_week_like": (-300, 700), "interior": (0.5, 15)}[name]
rays = _random_rays(rng, 60000, *box)
ids, tuv = o.trace(rays)
h = r.trace(rays)
bad = np.nonzero((h["prim_type"] != ids[:, 0]) | (h["prim_id"] != ids[:, 1]) | (h["inst_id"] != ids[:, 2]) | (h["t"].view(np.uint32) != tuv[:, 0].view(np.uint32)))[0]
print("rays", rays.shape[0], "mismatches", bad.size)
for i in bad[:12]:
    print(i, "ray", rays[i], "device", (h["prim_type"][i], h["prim_id"][i], h["inst_id"][i], h["t"][i]), "oracle", tuple(ids[i]), tuv[i, 0])
if bad.size:
    print("device inst ids among mismatches", np.unique(h["inst_id"][bad])[:20], "oracle", np.unique(ids[bad, 2])[:20])
    print("index classes (i // (n//8)):", np.unique(bad // (60000 // 8), return_counts=True))
