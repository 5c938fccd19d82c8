"""Converged images: the device's render (one seed) against the oracle's (two other seeds) at thousands of
samples per pixel, where the Monte-Carlo noise is small enough for a systematic difference of a per cent to show.

The image rule: RMSE and mean relative error of device-vs-oracle on the linear f32 images must stay within 1.25 x
the oracle's own two-seed figures (two independent renders of the same estimator: the floor), and the mean of every
tile of an 8 x 8 grid over the image must agree with the oracle's within 2 % of the image's mean tile value, or within
five standard deviations of the difference of two tile means where that is more (the variance of a pixel is estimated
from the oracle's two seeds, (a - b)^2 / 2, and summed over the tile).  RMSE is computed on values clamped to the
99.9th percentile of the oracle's image, so that a handful of rare very bright samples (a path that finds the small
light through a glossy bounce) do not decide the figure; the mean relative error is not clamped.  Pixels that are not
finite in one of the three images (the reference filters NaN samples but adds infinite ones, src/camera.rs:212-217)
are left out of the figures, and must be as rare on the device as in the oracle.

Corresponds to the progressive average of Camera::render (src/camera.rs:212-226).
"""
import os

import numpy as np
import pytest

from pathtracer_amd import scenes
from tests.util import _record_margin, image_errors

pytestmark = pytest.mark.gpu

THREADS = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)


def _tiles(img, width, height, grid=8, reduce="mean"):
    """Mean (or sum) of each tile of a grid x grid partition of the image, (grid*grid, 3), and the pixels per tile."""
    a = img.reshape(height, width, 3).astype(np.float64)
    ys = np.linspace(0, height, grid + 1).astype(int)
    xs = np.linspace(0, width, grid + 1).astype(int)
    f = np.mean if reduce == "mean" else np.sum
    vals = np.array([f(a[ys[j]:ys[j + 1], xs[i]:xs[i + 1]], axis=(0, 1)) for j in range(grid) for i in range(grid)])
    count = np.array([(ys[j + 1] - ys[j]) * (xs[i + 1] - xs[i]) for j in range(grid) for i in range(grid)], np.float64)
    return vals, count


def _check(name, src, n, prob=1.0):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd.core import PT_OPT_BSDF_SAMPLE_PROBABILITY, HostScene, Renderer
    o = OracleScene(src)
    r = Renderer(HostScene(src))
    if prob != 1.0:
        o.set_bsdf_probability(prob)
        r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, prob)
    assert n <= r.sqrt_spp * r.sqrt_spp
    a = o.render(0, n, seed=101, threads=THREADS)["accum"] / n
    b = o.render(0, n, seed=202, threads=THREADS)["accum"] / n
    r.render(0, n, seed=303)
    g, ns = r.read_accum()
    assert ns == n
    g = g / n
    w, h = r.width, r.height
    bad = [~np.isfinite(x).all(axis=1) for x in (a, b, g)]
    any_bad = bad[0] | bad[1] | bad[2]
    a, b, g = (np.where(any_bad[:, None], 0.0, x) for x in (a, b, g))
    cap = float(np.percentile(a, 99.9))
    clamp = lambda x: np.minimum(x, cap)
    floor_rmse, _ = image_errors(clamp(a), clamp(b))
    rmse, _ = image_errors(clamp(g), clamp(a))
    _, floor_mre = image_errors(a, b)
    _, mre = image_errors(g, a)
    (ta, cnt), (tb, _), (tg, _) = _tiles(a, w, h), _tiles(b, w, h), _tiles(g, w, h)
    scale = float(ta.mean())
    sigma_diff = np.sqrt(_tiles((a - b) ** 2, w, h, reduce="sum")[0]) / cnt[:, None]  # of the difference of two tile means
    tol = np.maximum(0.02 * scale, 5.0 * sigma_diff)
    tile_excess = float((np.abs(tg - ta) / tol).max())
    mean_ratio = float(g.mean() / a.mean())
    _record_margin(f"converged [{name}]", {"passes": n, "rmse_clamped": rmse, "rmse_floor": floor_rmse, "rmse_over_floor": rmse / floor_rmse, "mre": mre, "mre_floor": floor_mre,
                                           "mre_over_floor": mre / floor_mre, "worst_tile_difference_over_tolerance": tile_excess,
                                           "worst_tile_relative_difference": float((np.abs(tg - ta) / scale).max()), "mean_device_over_oracle": mean_ratio,
                                           "mean_oracle_two_seeds": float(b.mean() / a.mean()),
                                           "pixels_not_finite_oracle_oracle_device": [int(x.sum()) for x in bad], "pixels": int(any_bad.size)})
    assert abs(int(bad[2].sum()) - int(bad[0].sum())) <= max(3, 4 * int(np.sqrt(bad[0].sum() + bad[1].sum() + 1))), [int(x.sum()) for x in bad]
    assert any_bad.mean() < 0.01
    assert rmse <= 1.25 * floor_rmse, f"{name}: RMSE device-vs-oracle {rmse:.5g} against the oracle's two-seed floor {floor_rmse:.5g} ({rmse / floor_rmse:.3f} x, limit 1.25)"
    assert mre <= 1.25 * floor_mre, f"{name}: mean relative error {mre:.5g} against the floor {floor_mre:.5g} ({mre / floor_mre:.3f} x, limit 1.25)"
    assert tile_excess <= 1.0, f"{name}: a tile mean differs by {tile_excess:.2f} x its tolerance (2 % of {scale:.4g} or five standard deviations)"
    r.close()


def test_knob_at_4096_passes():
    _check("knob 128x128, 4096 passes", scenes.knob(width=128, spp=4096), 4096)


def test_cornell_at_its_own_64_passes():
    # configs[0]: the Cornell box at 256 x 256 and 64 passes
    _check("cornell 256x256, 64 passes", scenes.cornell(width=256, spp=64), 64)


def test_cornell_at_4096_passes():
    _check("cornell 256x256, 4096 passes", scenes.cornell(width=256, spp=4096), 4096)


def test_interior_with_light_sampling_at_2304_passes():
    # configs[4] reduced: BSDF_SAMPLE_PROBABILITY 0.5, 64 emissive quads
    _check("interior 160x90, p = 0.5, 2304 passes", scenes.interior(width=160, spp=2304, n_lights=64), 2304, prob=0.5)
