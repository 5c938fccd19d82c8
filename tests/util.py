"""Shared helpers for the parity tests."""
import numpy as np


def ulp_diff(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance in units of the last place between two float32 arrays (inf/NaN-free)."""
    ai = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


def image_errors(a: np.ndarray, b: np.ndarray):
    """RMSE and mean relative error between two linear float images of shape (npix, 3)."""
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    rmse = float(np.sqrt(np.mean((a - b) ** 2)))
    mre = float(np.mean(np.abs(a - b)) / max(1e-12, np.mean(np.abs(b))))
    return rmse, mre


def _record_margin(key: str, value: dict):
    """Observed agreement figures go to gpurun_out/parity_margins.json (what the thresholds are set from)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_margins.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data[key] = value
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass


def samples_agree(got: np.ndarray, ref: np.ndarray, min_frac: float, note=""):
    """Per-sample radiance of the device against the oracle's, both (..., 3) with NaN / infinity already zeroed:
    at least `min_frac` of the samples within 1e-3 (relative, absolute below 1), and the samples that differ must
    not lean one way: the signed sum of their differences stays within 0.2 % of the total radiance."""
    import os
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    close = np.all(np.abs(got - ref) <= 1e-3 * np.maximum(1.0, np.abs(ref)), axis=-1)
    frac = float(close.mean())
    total = float(np.abs(ref).sum())
    bias = float((got - ref)[~close].sum()) / max(total, 1e-30)
    key = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" (")[0] + (f" [{note}]" if note != "" else "")
    _record_margin(key, {"fraction_within_1e-3": frac, "signed_difference_of_the_others_over_total": bias, "samples": int(close.size), "threshold": min_frac})
    assert frac >= min_frac, f"{note}: only {frac:.5f} of the samples agree with the oracle (want {min_frac})"
    assert abs(bias) <= 2e-3, f"{note}: the {int((~close).sum())} samples that differ lean one way: signed sum {bias:+.5f} of the total"
    return frac
