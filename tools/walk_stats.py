#!/usr/bin/env python3
"""Lane occupancy of the walking kernel, pass by pass (a build with -DPT_WALK_STATS): python tools/walk_stats.py scene width [bench-like args]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import scenes
from pathtracer_amd.core import HostScene, Renderer, PT_OPT_BSDF_SAMPLE_PROBABILITY
import pathtracer_amd.core as core
scene, width = sys.argv[1], int(sys.argv[2])
strata = int(sys.argv[3]) if len(sys.argv) > 3 else 256
src = {"knob": lambda: scenes.knob(width=width), "dragon_box": lambda: scenes.dragon_box(width=width), "interior": lambda: scenes.interior(width=width),
       "cornell": lambda: scenes.cornell(width=width)}[scene]()
r = Renderer(HostScene(src))
if scene == "interior":
    r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, 0.5)
L = C.CDLL(os.environ["PTCORE_LIB"], mode=C.RTLD_GLOBAL)
out = (C.c_ulonglong * 16)()
r.render(0, strata, seed=1); r.sync()
L.ptc_debug_walk_stats(out, 1)
r.stats_reset()
r.render(strata, 2 * strata, seed=1); r.sync()
st = r.stats()
L.ptc_debug_walk_stats(out, 1)
p, busy, a, bp, b, pp, rounds, leafwait, tp, tb = [out[k] for k in range(10)]
done = 0
print(f"{scene}: rays {st['rays']:.3e} paths {st['paths']:.3e}")
print(f"passes {p:.3e}  busy lanes/pass {busy/p:.1f}  at-node lanes/pass (A) {a/p:.1f}  lanes with a leaf pending/pass {leafwait/p:.1f}  done-waiting lanes/pass {done/p:.1f}")
print(f"B passes {bp:.3e} ({bp/p:.2f} of passes) lanes per B pass {b/max(1,bp):.1f}   popping lanes/pass {pp/p:.1f}   refill rounds {rounds:.3e} passes/round {p/rounds:.1f}")
print(f"tail (queue used up): {tp/p:.2%} of passes, busy lanes/pass there {tb/max(1,tp):.1f}")
