"""Seeded inputs shared by the GPU sweeps (tests/test_fit_gpu.py) and the CPU pins of the oracle modes they compare with
(tests/test_oracle.py): the same columns are fitted by the kernel, by the oracle's mode 2 (the kernel's difference quotients)
and by mode 0 (true lmdif = what scipy runs, pinned to the fixtures)."""
import numpy as np

SHAPES_AND_DTYPES = [(2, np.float64), (3, np.float32), (4, np.int16), (5, np.float32), (7, np.uint16), (8, np.float64),
                     (12, np.float32), (16, np.float32), (24, np.float32), (32, np.float64)]


def shape_sweep_case(E, dtype):
    """x (E,), y (E, N): tissue-like decays + noise, ragged N, every 17th column all-zero (the skip rule)."""
    rng = np.random.default_rng(E)
    N = 5000 + E  # ragged: not a multiple of the 256-voxel tile
    x = np.sort(rng.uniform(2, 90, E))
    y = rng.uniform(300, 1500, N) * np.exp(-x[:, None] / rng.uniform(15, 80, N))
    y = y + 8 * rng.standard_normal((E, N))
    if np.issubdtype(dtype, np.integer):
        y = np.clip(np.rint(y), 0 if dtype == np.uint16 else -32768, 32767)
    y = y.astype(dtype)
    y[:, ::17] = 0
    return x, y
