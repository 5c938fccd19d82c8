"""Material parity, call by call: pt_material_probe (the device's mat_scatter / mat_eval / mat_pdf, the functions
the shading kernels call) against the oracle's Material trait on the same probes.

Reference: src/materials/{lambertian,metal,dielectric,light,disney}.rs.  Covered per material: >= 24 000 probes,
both faces, outgoing directions from the normal down to grazing (and, for robustness, on and just under the
horizon), incoming directions above and below the surface, mirror and anti-parallel pairs, texture coordinates
inside and outside [0, 1].  The combined Disney material is probed with parameter sets that leave one lobe at a
time with all the sampling weight (src/materials/disney.rs:232-271), with the knob scene's own parameters
(src/scenes/object.rs:44-46) and with random parameters.

Tolerances, from what the two sides can differ in (tools/material_probe_stats.py prints the distributions):
- bsdf_evaluate and scattering_pdf at a direction both sides are GIVEN: scattering_pdf involves no math-library
  call (sqrt and divide are correctly rounded on both sides) and must agree bit for bit; bsdf_evaluate calls
  powf (Schlick weights) and may differ by `EVAL_ULPS` ulps.  NaN and infinity must sit in the same places.
- scatter: draw counts equal (the count names the lobe that was picked); the sampled direction goes through sinf /
  cosf / powf, whose last bits differ, and through cancellations that amplify them (the VNDF's
  sqrt(1 - t1^2 - t2^2)), so `DIR_CLOSE` of the directions must agree to `DIR_ULPS` ulps or 1e-6 and all of them
  to `DIR_ABS`, except for at most `FLIPS` of the probes: samplers branch on computed values (Fresnel against a
  draw), where a last-bit difference can pick the other branch.
- bsdf_evaluate / scattering_pdf at the SCATTERED direction: specular lobes turn a last-bit difference of the
  direction into a large difference of the value, so the device's values are held against the oracle evaluated at
  the device's own scattered direction (same inputs again, same bounds as the first item).
"""
import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.core import PROBE_IN_DTYPE, HostScene, Renderer
from pathtracer_amd.scene_src import CameraSrc, SceneSource
from tests.util import ulp_diff, samples_agree

pytestmark = pytest.mark.gpu

N_PROBES = 24_000
EVAL_ULPS = 8
DIR_ULPS, DIR_CLOSE, DIR_ABS = 4, 0.995, 5e-3
FLIPS = 5e-4


def gallery() -> tuple[SceneSource, dict]:
    """One quad per material; the names say which lobe of the combined Disney material a parameter set isolates."""
    s = SceneSource()
    s.name = "material_gallery"
    img = (np.arange(16 * 8 * 3, dtype=np.uint32).reshape(8, 16, 3) * 53 % 256).astype(np.uint8)
    chk = s.checkered(s.solid((0.9, 0.1, 0.1)), s.checkered(s.solid((0.1, 0.9, 0.1)), s.image(None), 3.0), 5.0)
    tex = s.solid((0.8, 0.6, 0.2))
    m = {
        "lambertian": s.lambertian((0.7, 0.4, 0.2)),
        "lambertian_image": s.lambertian_texture(s.image(img)),
        "lambertian_checker": s.lambertian_texture(chk),
        "metal_mirror": s.metal((0.8, 0.85, 0.88), 0.0),
        "metal_fuzz": s.metal((0.9, 0.6, 0.3), 0.4),
        "dielectric_1.5": s.dielectric(1.5),
        "dielectric_1.1": s.dielectric(1.1),
        "emitter": s.diffuse_light_texture(chk),
        # src/scenes/object.rs:44-46, the knob's `test` material
        "disney_knob": s.disney(tex, 0.5, 0.1, 0.0, 0.01, 0.2, 0.1, 0.8, 0.3, 0.1, 0.2, 0.5, 1.1),
        # metallic = 1: diffuse and glass weights are 0, clearcoat 0 -> the metal lobe alone
        "disney_only_metal": s.disney(tex, 0.3, 1.0, 0.0, 0.5, 0.5, 0.4, 0.3, 0.0, 0.0, 0.0, 0.5, 1.5),
        # metallic = 0, transmission = 1: diffuse and metal weights are 0 -> the glass lobe alone
        "disney_only_glass": s.disney(tex, 0.3, 0.0, 1.0, 0.5, 0.5, 0.25, 0.2, 0.0, 0.0, 0.0, 0.5, 1.45),
        # the same with clearcoat 1: glass and clearcoat share the weight 1 : 0.25
        "disney_glass_clearcoat": s.disney(tex, 0.3, 0.0, 1.0, 0.5, 0.5, 0.25, 0.2, 0.0, 0.0, 1.0, 0.9, 1.45),
        # metallic = 0, transmission at its floor of 1e-4: diffuse and metal 1 : 1, sheen on
        "disney_diffuse_metal_sheen": s.disney(s.image(img), 0.8, 0.0, 0.0, 0.8, 0.7, 0.6, 0.0, 1.0, 0.7, 0.0, 0.5, 1.3),
        "disney_all": s.disney(chk, 0.2, 0.3, 0.5, 0.5, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0, 0.2, 1.6),
        "disney_diffuse": s.disney_diffuse((0.3, 0.3, 0.8), 1.0, 0.5),
        "disney_diffuse_no_subsurface": s.disney_diffuse((0.3, 0.7, 0.8), 0.3, 0.0),
        "disney_metal": s.disney_metal((0.3, 0.3, 0.8), 0.5, 0.3),
        "disney_metal_smooth_aniso": s.disney_metal((0.9, 0.8, 0.3), 0.05, 0.9),
        "disney_glass": s.disney_glass((0.5, 0.5, 0.8), 0.1, 0.0, 1.2),
        "disney_glass_rough": s.disney_glass((0.9, 0.5, 0.4), 0.6, 0.5, 1.7),
        "disney_clearcoat": s.disney_clearcoat(1.0, 0.8),
        "disney_clearcoat_satin": s.disney_clearcoat(0.4, 0.0),
        "disney_sheen": s.disney_sheen((0.3, 0.3, 0.8), 1.0, 0.2),
    }
    rng = np.random.default_rng(11)
    for k in range(6):  # random parameter sets of the combined material
        p = rng.uniform(0.0, 1.0, 11)
        m[f"disney_random_{k}"] = s.disney(tex if k % 2 else chk, *[float(x) for x in p], float(rng.uniform(1.05, 2.0)))
    for k, mat in enumerate(m.values()):
        s.objects.add_quad((2.0 * k, 0.0, 0.0), (1.5, 0.0, 0.0), (0.0, 1.5, 0.0), mat)
    s.objects.build_bvh = True
    s.camera = CameraSrc(1.0, 32, 40.0, (10.0, 1.0, -30.0), (10.0, 1.0, 0.0), 10.0, 0.0, 4, 4, (0.5, 0.5, 0.5))
    return s, m


def _unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def make_probes(n: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    p = np.zeros(n, PROBE_IN_DTYPE)

    def hemisphere(k, y_lo=0.0, y_hi=1.0, sign=1.0):
        y = rng.uniform(y_lo, y_hi, k)
        phi = rng.uniform(0.0, 2.0 * np.pi, k)
        r = np.sqrt(np.maximum(0.0, 1.0 - y * y))
        return _unit(np.stack([r * np.cos(phi), sign * y, r * np.sin(phi)], axis=1))

    w_out = hemisphere(n)
    k = n // 10
    w_out[:k] = hemisphere(k, 1e-4, 0.02)           # grazing
    w_out[k:k + 64] = (0.0, 1.0, 0.0)                # along the normal
    w_out[k + 64:k + 128, 1] = 0.0                   # on the horizon (not reached by ray_color; results must still match)
    w_out[k + 64:k + 128] = _unit(w_out[k + 64:k + 128] + np.array([1e-3, 0.0, 0.0], np.float32))
    w_out[k + 128:k + 192] = hemisphere(64, 1e-4, 1e-3, -1.0)  # just under it
    w_in = hemisphere(n)
    a = n * 4 // 10
    w_in[a:a + n // 4] = hemisphere(n // 4, sign=-1.0)                 # below the surface
    b = a + n // 4
    w_in[b:b + k] = hemisphere(k, 1e-4, 0.02, 1.0)                     # grazing, above and below
    w_in[b + k // 2:b + k, 1] *= -1.0
    c = b + k
    w_in[c:c + k // 2] = w_out[c:c + k // 2] * np.array([-1.0, 1.0, -1.0], np.float32)  # the mirror direction
    w_in[c + k // 2:c + k] = -w_out[c + k // 2:c + k]                                     # anti-parallel: the half vector is 0 / 0
    d = c + k
    w_in[d:d + 32] = (0.0, 1.0, 0.0)
    w_in[d + 32:d + 64] = (0.0, -1.0, 0.0)
    e = d + 64
    k = min(k, n - e)
    w_in[e:e + k] = _unit(w_out[e:e + k] * np.array([-1.0, 1.0, -1.0], np.float32) + rng.normal(0.0, 0.02, (k, 3)).astype(np.float32))  # near the mirror direction
    p["w_out"] = w_out
    p["w_in"] = w_in
    p["u"] = rng.uniform(0.0, 1.0, n)
    p["v"] = rng.uniform(0.0, 1.0, n)
    p["u"][:n // 8] = rng.uniform(-1.5, 2.5, n // 8)
    p["v"][:n // 8] = rng.uniform(-1.5, 2.5, n // 8)
    p["p"] = rng.uniform(-5.0, 5.0, (n, 3))
    p["front_face"] = rng.integers(0, 2, n)
    p["pixel"] = rng.integers(0, 1 << 20, n)
    p["stratum"] = rng.integers(0, 10201, n)
    return p


def same_class(g, w):
    return (np.isnan(g) == np.isnan(w)) & (np.isposinf(g) == np.isposinf(w)) & (np.isneginf(g) == np.isneginf(w))


def ulps(g, w):
    """ulp distance where both are finite, 0 elsewhere (classes are compared separately)."""
    fin = np.isfinite(g) & np.isfinite(w)
    return ulp_diff(np.where(fin, g, 0.0).astype(np.float32), np.where(fin, w, 0.0).astype(np.float32)).reshape(g.shape)


def check_same_input_eval(name, bsdf_dev, pdf_dev, bsdf_ref, pdf_ref, what):
    assert np.all(same_class(bsdf_dev, bsdf_ref)), f"{name}: NaN / inf pattern of bsdf_evaluate differs ({what})"
    assert np.all(same_class(pdf_dev, pdf_ref)), f"{name}: NaN / inf pattern of scattering_pdf differs ({what})"
    u = ulps(bsdf_dev, bsdf_ref)
    assert u.max() <= EVAL_ULPS, f"{name}: bsdf_evaluate {what} differs by {u.max()} ulps at probe {int(np.argmax(u.max(axis=1)))}"
    fin = np.isfinite(pdf_ref)
    assert np.array_equal(pdf_dev[fin].view(np.uint32), pdf_ref[fin].view(np.uint32)), f"{name}: scattering_pdf {what} differs by up to {ulps(pdf_dev, pdf_ref).max()} ulps"


@pytest.fixture(scope="module")
def rig():
    from oracle.pyoracle import OracleScene
    src, mats = gallery()
    oracle = OracleScene(src)
    r = Renderer(HostScene(src), device=0)
    yield r, oracle, mats
    r.close()


@pytest.mark.parametrize("name", list(gallery()[1].keys()))
def test_material_functions_match_the_oracle_probe_by_probe(rig, name):
    r, oracle, mats = rig
    probes = make_probes(N_PROBES, seed=1000 + mats[name])
    got = r.material_probe(mats[name], probes, seed=77)
    want = oracle.material_probe(mats[name], probes, seed=77)
    # 1. the same direction given to both sides
    check_same_input_eval(name, got["bsdf_at"], got["pdf_at"], want["bsdf_at"], want["pdf_at"], "at the given direction")
    assert np.array_equal(got["emitted"].view(np.uint32), want["emitted"].view(np.uint32))
    # 2. the sampler
    assert np.array_equal(got["draws"], want["draws"]), f"{name}: scatter consumed different numbers of draws"
    assert np.all(same_class(got["w_in"], want["w_in"]))
    d = np.abs(got["w_in"].astype(np.float64) - want["w_in"])
    d = np.where(np.isfinite(d), d, 0.0)
    close = np.all((ulps(got["w_in"], want["w_in"]) <= DIR_ULPS) | (d <= 1e-6), axis=1)
    flipped = np.any(d > DIR_ABS, axis=1)
    assert close.mean() >= DIR_CLOSE, f"{name}: only {close.mean():.5f} of the sampled directions agree to {DIR_ULPS} ulps"
    assert flipped.mean() <= FLIPS, f"{name}: {flipped.mean():.5f} of the sampled directions differ by more than {DIR_ABS}"
    if name in ("metal_mirror", "metal_fuzz", "dielectric_1.5", "dielectric_1.1", "emitter"):  # no sinf / cosf / powf on the way
        assert np.array_equal(got["w_in"].view(np.uint32), want["w_in"].view(np.uint32)), f"{name}: scatter is pure arithmetic and must agree in bits"
    # 3. the device's evaluation at its own scattered direction, against the oracle at that same direction
    again = probes.copy()
    again["w_in"] = got["w_in"]
    ref = oracle.material_probe(mats[name], again, seed=77)
    check_same_input_eval(name, got["bsdf"], got["pdf"], ref["bsdf_at"], ref["pdf_at"], "at the device's scattered direction")
    if name != "emitter":
        assert np.any(want["w_in"] != 0.0)
    else:
        assert np.any(want["emitted"] != 0.0) and np.all(want["emitted"][probes["front_face"] == 0] == 0.0)


def test_disney_parameter_sets_isolate_the_lobes_they_claim(rig):
    """The draw count tells which lobe scatter() picked: 1 + 2 for diffuse, metal and clearcoat, 1 + 3 for glass."""
    r, oracle, mats = rig
    probes = make_probes(8000, seed=5)
    probes["front_face"] = 1
    draws = {k: r.material_probe(mats[k], probes, seed=3)["draws"] for k in ("disney_only_metal", "disney_only_glass", "disney_glass_clearcoat", "disney_knob")}
    assert np.all(draws["disney_only_metal"] == 3)
    assert np.all(draws["disney_only_glass"] == 4)
    share = np.mean(draws["disney_glass_clearcoat"] == 3)  # clearcoat: 0.25 of 1.25
    assert abs(share - 0.2) < 0.02, share
    assert np.mean(draws["disney_knob"] == 4) < 1e-3  # transmission at its floor: the glass lobe is picked once in ~20 000
    probes["front_face"] = 0  # from inside only the glass lobe is sampled (disney.rs:237-252)
    assert np.all(r.material_probe(mats["disney_knob"], probes, seed=3)["draws"] == 4)


@pytest.mark.parametrize("material", ["lambertian", "metal_fuzz", "dielectric", "disney_knob", "disney_glassy", "disney_diffuse", "disney_metal",
                                      "disney_glass", "disney_clearcoat", "disney_sheen"])
def test_frame_filling_sphere_per_sample_radiance(material):
    """One sphere of one material fills a 32-pixel frame under a bright sky, so every sample shades that material
    (most of them several times): per-sample radiance against the oracle, 99.9 % within 1e-3 relative."""
    from oracle.pyoracle import OracleScene
    s = SceneSource()
    s.name = f"one_{material}"
    tex = s.solid((0.8, 0.6, 0.2))
    mat = {
        "lambertian": lambda: s.lambertian((0.7, 0.4, 0.2)),
        "metal_fuzz": lambda: s.metal((0.9, 0.6, 0.3), 0.4),
        "dielectric": lambda: s.dielectric(1.5),
        "disney_knob": lambda: s.disney(tex, 0.5, 0.1, 0.0, 0.01, 0.2, 0.1, 0.8, 0.3, 0.1, 0.2, 0.5, 1.1),
        "disney_glassy": lambda: s.disney(tex, 0.2, 0.0, 0.9, 0.5, 0.0, 0.3, 0.2, 0.0, 0.0, 1.0, 0.9, 1.45),
        "disney_diffuse": lambda: s.disney_diffuse((0.3, 0.3, 0.8), 1.0, 0.5),
        "disney_metal": lambda: s.disney_metal((0.3, 0.3, 0.8), 0.5, 0.3),
        "disney_glass": lambda: s.disney_glass((0.5, 0.5, 0.8), 0.1, 0.0, 1.2),
        "disney_clearcoat": lambda: s.disney_clearcoat(1.0, 0.8),
        "disney_sheen": lambda: s.disney_sheen((0.3, 0.3, 0.8), 1.0, 0.2),
    }[material]()
    s.objects.add_sphere((0.0, 0.0, 0.0), 1.0, mat)
    emit = s.diffuse_light((4.0, 4.0, 4.0))
    s.objects.add_sphere((3.0, 4.0, -2.0), 1.0, emit)
    s.lights.add_sphere((3.0, 4.0, -2.0), 1.0, emit)
    s.objects.build_bvh = True
    s.camera = CameraSrc(1.0, 32, 18.0, (0.0, 0.0, -5.0), (0.0, 0.0, 0.0), 5.0, 0.0, 16, 12, (0.6, 0.7, 0.9))
    oracle = OracleScene(s)
    r = Renderer(HostScene(s), device=0)
    try:
        first = r.trace(oracle.primary_rays())
        assert np.mean((first["prim_type"] == 1) & (first["prim_id"] >= 0)) > 0.85, "the sphere should fill the frame"
        got = r.render_samples(0, 16, seed=21)
        want = oracle.render(0, 16, seed=21, iterative=True, per_sample=True)["per_sample"]
        assert np.array_equal(np.isnan(got), np.isnan(want)) or np.mean(np.isnan(got) != np.isnan(want)) < 1e-3
        g = np.nan_to_num(got, nan=0.0, posinf=0.0, neginf=0.0)
        w = np.nan_to_num(want, nan=0.0, posinf=0.0, neginf=0.0)
        samples_agree(g, w, 0.999, material)
        assert np.abs(w).sum() > 0.0
    finally:
        r.close()
