"""Next-event estimation (PT_OPT_INTEGRATOR = 1).  The reference has no such integrator, so there is nothing of
its own to compare with: the device path is held against the oracle's statement of the same estimator, sample by
sample, and both against the reference estimator's converged images (they estimate the same integral)."""
import numpy as np
import pytest

from pathtracer_amd import scenes
from tests.util import image_errors, samples_agree

pytestmark = pytest.mark.gpu

NEE = 2  # oracle: `iterative` value that selects ray_color_nee


def _pair(src):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_INTEGRATOR, HostScene, Renderer
    r = Renderer(HostScene(src))
    r.set_option(PT_OPT_INTEGRATOR, 1)
    return OracleScene(src), r


SCENES = {"cornell": lambda spp: scenes.cornell(48, spp), "small_mixed": lambda spp: scenes.small_mixed(48, spp),
          "interior": lambda spp: scenes.interior(48, spp, n_lights=16, columns=3), "knob": lambda spp: scenes.knob(48, spp, detail=0.3)}


@pytest.mark.parametrize("name", ["cornell", "small_mixed", "interior"])
def test_per_sample_radiance_against_the_oracles_statement(name):
    o, r = _pair(SCENES[name](16))
    ref = np.nan_to_num(o.render(0, 6, seed=13, iterative=NEE, per_sample=True)["per_sample"], nan=0.0, posinf=0.0, neginf=0.0)
    got = np.nan_to_num(r.render_samples(0, 6, seed=13), nan=0.0, posinf=0.0, neginf=0.0)
    samples_agree(got, ref, 0.999, "")
    assert abs(got.sum() - ref.sum()) <= 0.02 * max(ref.sum(), 1e-6)


@pytest.mark.parametrize("name", ["cornell", "small_mixed"])
def test_same_bits_from_both_schedules_and_ray_counts_match_the_oracle(name):
    from pathtracer_amd.core import PT_OPT_PIPES, PT_OPT_ROLLING
    src = SCENES[name](36)
    o, _ = _pair(src)
    accs = []
    for pipes, rolling in ((1, 0), (2, 0), (1, 48 * 48), (2, 3 * 48 * 48)):
        _, r = _pair(src)
        r.set_option(PT_OPT_PIPES, pipes)
        r.set_option(PT_OPT_ROLLING, rolling)
        r.stats_reset()
        r.render(0, 7, seed=5)
        r.render(7, 20, seed=5)
        accs.append(r.read_accum()[0])
        rays = r.stats()["rays"]
    for a in accs[1:]:
        assert np.array_equal(a.view(np.uint32), accs[0].view(np.uint32))
    ref = o.render(0, 20, seed=5, iterative=NEE)
    assert abs(rays - ref["rays"]) <= 2e-3 * ref["rays"], (rays, ref["rays"])  # shadow rays included on both sides


@pytest.mark.parametrize("name", ["cornell", "small_mixed", "knob"])
def test_converges_to_the_reference_estimators_image(name):
    """Equal spp: the two estimators on the device against each other, with the oracle's two-seed difference of the
    reference estimator as the yardstick.  Next-event estimation has the lower variance in these scenes, so it has
    to land within the reference estimator's own noise floor (x2)."""
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import HostScene, Renderer
    n = 256
    src = SCENES[name](n)
    o, r = _pair(src)
    base = Renderer(HostScene(src))
    base.render(0, n, seed=1)
    r.render(0, n, seed=2)
    a = base.read_accum()[0] / n
    g = r.read_accum()[0] / n
    b = OracleScene(src).render(0, n, seed=3)["accum"] / n
    floor_rmse, floor_mre = image_errors(a, b)
    rmse, mre = image_errors(g, a)
    assert rmse <= 2.0 * floor_rmse and mre <= 2.0 * floor_mre, (rmse, floor_rmse, mre, floor_mre)
    assert abs(g.mean() - a.mean()) <= 3.0 * max(abs(a.mean() - b.mean()), 0.01 * a.mean())
    # and the oracle's statement of the estimator agrees with the device's at a few strata of the same seed
    on = o.render(0, 4, seed=2, iterative=NEE)["accum"]
    r.reset_accum()
    r.render(0, 4, seed=2)
    gn = r.read_accum()[0]
    assert abs(float(gn.sum()) - float(on.sum())) <= 0.02 * float(on.sum())


def test_needs_the_ray_drawing_traversal():
    from pathtracer_amd.core import PT_OPT_TRAVERSAL, PtError
    _, r = _pair(SCENES["cornell"](16))
    r.set_option(PT_OPT_TRAVERSAL, 1)
    with pytest.raises(PtError):
        r.render(0, 1, seed=1)
