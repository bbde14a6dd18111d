"""Orientation tuples from an affine, and the transpose/flip needed to change orientation.

Shim for the one thing the fit / segmentation path needs from the reference's
``dosma/core/orientation.py`` (constants :78-80, ``get_transpose_inds`` :111-146,
``get_flip_inds`` :149-187, nibabel-code mapping :190-237): every ``fit`` call reformats its inputs to
``y[0].orientation`` (dosma/core/fitting.py:189-190, 692-699) and ``generate_mask`` reformats to
``SAGITTAL`` (dosma/models/oaiunet2d.py:295).  Own implementation, no nibabel: an orientation is
stored as ``(world_axis, sign)`` per array axis, where world axes are RAS+ (0 = L->R, 1 = P->A,
2 = I->S).  The string form is DOSMA's: ``"LR"`` means the array axis runs from Left to Right.
"""
import numpy as np

SAGITTAL = ("SI", "AP", "LR")
CORONAL = ("SI", "LR", "AP")
AXIAL = ("AP", "LR", "SI")

# string  ->  (RAS world axis, +1 if the axis runs toward R / A / S)
_CODE = {"LR": (0, 1), "RL": (0, -1), "PA": (1, 1), "AP": (1, -1), "IS": (2, 1), "SI": (2, -1)}
_NAME = {v: k for k, v in _CODE.items()}


def _parse(orientation):
    orientation = tuple(orientation)
    if len(orientation) != 3 or not all(isinstance(o, str) and o in _CODE for o in orientation):
        raise ValueError("Orientation format mismatch: Orientations must be tuple of strings of length 3")
    axes = [_CODE[o] for o in orientation]
    if len({a for a, _ in axes}) != 3:
        raise ValueError("Orientation format mismatch: Orientations must be tuple of strings of length 3")
    return axes


def orientation_from_affine(affine):
    """DOSMA orientation tuple of the first three array axes of a RAS+ affine.

    Each array axis is assigned the world axis its direction cosine is closest to (after
    orthogonalising the 3x3 part, so that sheared / anisotropic affines still get a permutation).
    """
    rzs = np.asarray(affine, dtype=np.float64)[:3, :3]
    norms = np.sqrt((rzs ** 2).sum(axis=0))
    norms[norms == 0] = 1.0
    u, s, vt = np.linalg.svd(rzs / norms)
    keep = s > s.max() * 3 * np.finfo(np.float64).eps
    r = u[:, keep] @ vt[keep, :]
    out = []
    for col in range(3):
        c = r[:, col].copy()
        ax = int(np.argmax(np.abs(c)))
        out.append(_NAME[(ax, 1 if c[ax] >= 0 else -1)])
        r[ax, :] = 0  # each world axis is used once
    return tuple(out)


def get_transpose_inds(curr_orientation, new_orientation):
    """Axes permutation so that the world axes appear in the order of ``new_orientation``."""
    cur = [a for a, _ in _parse(curr_orientation)]
    new = [a for a, _ in _parse(new_orientation)]
    return tuple(cur.index(a) for a in new)


def get_flip_inds(curr_orientation, new_orientation):
    """Axes whose direction differs (world axes must already be in the same order)."""
    cur, new = _parse(curr_orientation), _parse(new_orientation)
    if [a for a, _ in cur] != [a for a, _ in new]:
        raise ValueError("All axis orientations (S/I, L/R, A/P) must be ordered. "
                         "Use `get_transpose_inds` to reorder axes.")
    return [i for i in range(3) if cur[i][1] != new[i][1]]


def to_affine(orientation, spacing=None, origin=None):
    """RAS+ affine with the given orientation, voxel spacing and origin (reference :239-311)."""
    axes = _parse(orientation)
    spacing = np.ones(3) if spacing is None else np.broadcast_to(np.asarray(spacing, float), (3,))
    affine = np.zeros((4, 4))
    for col, (ax, sign) in enumerate(axes):
        affine[ax, col] = sign * spacing[col]
    affine[:3, 3] = 0.0 if origin is None else np.asarray(origin, float)
    affine[3, 3] = 1.0
    return affine
