"""Renders the same strata with one and with four batches in flight and compares the accumulators bit for bit."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(pipes, out):
    env = dict(os.environ)
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import numpy as np
from pathtracer_amd import scenes
from pathtracer_amd.core import HostScene, Renderer, PT_OPT_PIPES
src = scenes.knob(width=256)
r = Renderer(HostScene(src), device=0)
r.set_option(PT_OPT_PIPES, {pipes})
r.render(0, 64, seed=3)
acc, n = r.read_accum()
img = r.read_rgb8()
print('pipes', {pipes}, 'n', n, 'mean', img.mean(), 'nan', np.isnan(acc).sum())
r.reset_accum()
r.render(0, 16, seed=3)
print('16 strata mean', r.read_rgb8().mean())
np.save({out!r}, acc)
"""
    subprocess.check_call([sys.executable, "-c", code], env=env)

run(1, "/tmp/acc1.npy")
run(4, "/tmp/acc4.npy")
a, b = np.load("/tmp/acc1.npy"), np.load("/tmp/acc4.npy")
print("bit-identical:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), "max abs diff", np.abs(a - b).max())
