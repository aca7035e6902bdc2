"""Deterministic pseudo-random arrays shared by make_golden.py and the tests, so
large gradient inputs need not be stored in the fixtures."""
import numpy as np


def hashed_f32(shape, salt=0):
    """Reproducible fp32 values in [-0.5, 0.5): integer hash of the flat index."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64) + np.uint64(salt) * np.uint64(7919)
    h = (i * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(1000003)
    return (h.astype(np.float32) / np.float32(1000003.0) - np.float32(0.5)).reshape(shape)
