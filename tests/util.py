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
