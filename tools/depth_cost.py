"""Throughput of the knob workload as a function of max_depth: what the thin late bounces cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import scenes
from pathtracer_amd.core import HostScene, Renderer, PT_OPT_PROFILE, PT_OPT_ROLLING, PT_OPT_TAIL_BELOW

import sys as _s
for depth in ([int(a) for a in _s.argv[1:]] or (4, 8, 16, 50)):
    src = scenes.knob(width=768, max_depth=depth)
    r = Renderer(HostScene(src))
    if os.environ.get("TAIL_BELOW"):
        r.set_option(PT_OPT_TAIL_BELOW, int(os.environ["TAIL_BELOW"]))
    if os.environ.get("ROLLING"):
        r.set_option(PT_OPT_ROLLING, int(os.environ["ROLLING"]))
    if os.environ.get("PT_DEBUG_BOUNCES"):
        r.set_option(PT_OPT_PROFILE, 1)
    r.render(0, 128, seed=1)
    r.sync()
    r.stats_reset()
    t0 = time.perf_counter()
    for k in range(4):
        r.render(128 * (k + 1), 128 * (k + 2), seed=1)
    r.sync()
    dt = time.perf_counter() - t0
    st = r.stats()
    print(f"depth {depth:2d}: {st['rays'] / dt / 1e6:8.1f} Mrays/s  {512 / dt:7.1f} spp/s  mean path {st['rays'] / st['paths']:.3f}  host enqueue {st['render_ms'] - st['host_wait_ms']:.1f} ms + wait {st['host_wait_ms']:.1f} ms of {dt * 1e3:.1f} ms")
