"""Throughput of the knob workload as a function of max_depth: what the thin late bounces cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import scenes
from pathtracer_amd.core import HostScene, Renderer

for depth in (4, 8, 16, 50):
    src = scenes.knob(width=768, max_depth=depth)
    r = Renderer(HostScene(src))
    r.render(0, 128, seed=1)
    r.sync()
    r.stats_reset()
    t0 = time.perf_counter()
    for k in range(4):
        r.render(128 * (k + 1), 128 * (k + 2), seed=1)
    r.sync()
    dt = time.perf_counter() - t0
    st = r.stats()
    print(f"depth {depth:2d}: {st['rays'] / dt / 1e6:8.1f} Mrays/s  {512 / dt:7.1f} spp/s  mean path {st['rays'] / st['paths']:.3f}")
