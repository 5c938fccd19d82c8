"""Which of the reference's materials have a scattering_pdf that is the density of their scatter?  (Oracle only.)

For a fixed outgoing direction, E[f |cos| / pdf] over scatter samples must equal the integral of f |cos| over the
directions where pdf > 0, and pdf must integrate to 1.  That holds for Lambertian, DisneyDiffuse and DisneySheen and
fails for the other Disney lobes as the reference wrote them (src/materials/disney.rs; its README lists the Disney
BSDF among the open problems).  The next-event estimator (PT_OPT_INTEGRATOR = 1) combines light sampling with BSDF
sampling only where this holds; this test pins that list."""
import numpy as np

from oracle.pyoracle import OracleScene
from pathtracer_amd import scenes

NAMES = ["lambertian", "metal", "dielectric", "diffuse_light", "disney", "disney_diffuse", "disney_metal", "disney_glass",
         "disney_clearcoat", "disney_sheen"]
CONSISTENT = {"lambertian", "disney_diffuse", "disney_sheen"}
INCONSISTENT = {"disney", "disney_metal", "disney_glass", "disney_clearcoat"}


def _measure(o, m, w_out, n_samples, n_quad, rng):
    acc = 0.0
    for k in range(n_samples):
        w, _ = o.material_scatter(m, w_out, seed=7, pixel=k, stratum=0)
        f, pdf, _ = o.material_eval(m, w_out, w)
        if pdf > 0 and np.isfinite(pdf) and np.isfinite(f).all():
            acc += float(f.mean()) * abs(float(w[1])) / pdf
    # midpoint rule on an equal-area grid of the sphere (uniform in cos(theta) and phi)
    n_c = int(np.sqrt(n_quad / 2))
    n_p = 2 * n_c
    quad = mass = 0.0
    for i in range(n_c):
        c = -1.0 + (i + 0.5) * 2.0 / n_c
        sn = np.sqrt(max(0.0, 1.0 - c * c))
        for j in range(n_p):
            phi = (j + 0.5) * 2.0 * np.pi / n_p
            d = np.array([sn * np.cos(phi), c, sn * np.sin(phi)], np.float32)
            f, pdf, _ = o.material_eval(m, w_out, d)
            if pdf > 0 and np.isfinite(pdf) and np.isfinite(f).all():
                quad += float(f.mean()) * abs(float(d[1]))
                mass += pdf
    return acc / n_samples, quad / (n_c * n_p) * 4 * np.pi, mass / (n_c * n_p) * 4 * np.pi


def test_which_materials_sample_with_the_density_they_report():
    src = scenes.small_mixed(16, 4)
    o = OracleScene(src)
    sc, _ = src.to_c()
    rng = np.random.default_rng(1)
    worst = {}
    for wo in ([0.3, 0.8, 0.2], [0.7, 0.25, -0.4]):
        w_out = np.asarray(wo, np.float32)
        w_out /= np.linalg.norm(w_out)
        for m in range(sc.n_materials):
            name = NAMES[sc.materials[m].type]
            if name not in CONSISTENT | INCONSISTENT:
                continue
            est, quad, mass = _measure(o, m, w_out, 8000 if name == "disney_sheen" else 2000, 12000, rng)
            err = max(abs(est - quad) / max(quad, est, 1e-3), abs(mass - 1.0))
            worst[name] = max(worst.get(name, 0.0), err)
    assert CONSISTENT <= set(worst) and INCONSISTENT <= set(worst), worst
    print(worst)
    for name in CONSISTENT:
        assert worst[name] < 0.06, (name, worst)
    for name in INCONSISTENT:
        assert worst[name] > 0.15, (name, worst)
