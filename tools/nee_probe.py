import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pathtracer_amd import scenes
from pathtracer_amd.scene_src import CameraSrc, SceneSource
from pathtracer_amd.core import HostScene, Renderer, PT_OPT_INTEGRATOR

def lambert_room():
    s = SceneSource(); s.name = "room"
    white = s.lambertian((0.73, 0.73, 0.73)); red = s.lambertian((0.65, 0.05, 0.05)); emit = s.diffuse_light((15.0, 15.0, 15.0))
    o = s.objects
    o.add_quad((555, 0, 0), (0, 555, 0), (0, 0, 555), red)
    o.add_quad((0, 0, 0), (0, 555, 0), (0, 0, 555), white)
    o.add_quad((0, 0, 0), (555, 0, 0), (0, 0, 555), white)
    o.add_quad((555, 555, 555), (-555, 0, 0), (0, 0, -555), white)
    o.add_quad((0, 0, 555), (555, 0, 0), (0, 555, 0), white)
    o.add_quad((343, 554, 332), (-130, 0, 0), (0, 0, -105), emit)
    o.add_sphere((190, 90, 190), 90, white)
    s.lights.add_quad((343, 554, 332), (-130, 0, 0), (0, 0, -105), emit)
    o.build_bvh = True
    s.camera = CameraSrc(1.0, 32, 40.0, (278, 278, -800), (278, 278, 0), 1.0, 0.0, 1024, 16, (0, 0, 0))
    return s

for name, src in (("room", lambert_room()), ("cornell", scenes.cornell(32, 1024)), ("small_mixed", scenes.small_mixed(32, 1024)), ("knob", scenes.knob(32, 1024, detail=0.3))):
    imgs = []
    for integ in (0, 1):
        r = Renderer(HostScene(src)); r.set_option(PT_OPT_INTEGRATOR, integ)
        r.render(0, 1024, seed=1 + integ)
        imgs.append(r.read_accum()[0].reshape(32, 32, 3) / 1024)
    a, g = imgs
    print(name, "ref mean %.4f nee mean %.4f ratio %.3f" % (a.mean(), g.mean(), g.mean() / a.mean()))
    A = a.mean(axis=2).reshape(4, 8, 4, 8).mean(axis=(1, 3)); G = g.mean(axis=2).reshape(4, 8, 4, 8).mean(axis=(1, 3))
    print(np.round(G / np.maximum(A, 1e-6), 2))
