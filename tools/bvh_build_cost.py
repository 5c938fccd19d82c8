"""Host binned-SAH build against the device build (PT_OPT_DEVICE_BVH): time until the scene is resident, and
what the different tree costs in rendering throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import scenes
from pathtracer_amd.core import HostScene, Renderer

names = sys.argv[1:] or ["knob", "dragon_box"]
warm = Renderer(HostScene(scenes.cornell(64, 16)))  # library load, context creation
warm.render(0, 1, seed=1)
warm.sync()
for name in names:
    src = getattr(scenes, name)()
    for device_from in (0, 1024):
        t0 = time.perf_counter()
        host = HostScene(src, device_bvh_from=device_from)
        t1 = time.perf_counter()
        r = Renderer(host)
        r.sync()
        t2 = time.perf_counter()
        S = 32
        r.render(0, S, seed=1)
        r.sync()
        r.stats_reset()
        t3 = time.perf_counter()
        for k in range(1, 5):
            r.render(k * S, (k + 1) * S, seed=1)
        r.sync()
        dt = time.perf_counter() - t3
        st = r.stats()
        print(f"{name:12s} {'device' if device_from else 'host  '} build: host layer {t1 - t0:6.2f} s, upload{' + device build' if device_from else ''} "
              f"{(t2 - t1) * 1e3:7.1f} ms, {host.desc.n_tris} triangles; render {st['rays'] / dt / 1e6:7.1f} Mrays/s", flush=True)
        r.close()
