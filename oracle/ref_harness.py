"""Import harness for the *reference* (ad12/DOSMA) -- build-side tool, THIS CONTAINER ONLY.

TEST INFRASTRUCTURE.  Nothing under ``dosma_amd/`` may import this file.

The reference's fit path (``dosma/core/fitting.py``) is pure Python over scipy,
so it can be imported here to (a) validate the restatement in ``oracle/`` and
(b) generate the golden vectors committed under ``tests/golden/``.  It cannot be
imported as-is:

* ``/root/reference`` is read-only and the package writes ``preferences.yml`` and a
  log file at import  -> we import from a throw-away copy under ``$TMPDIR``;
* it pre-dates NumPy 2 (``np.round_``, ``np.int``, ``np.bool``)  -> aliases injected;
* nibabel / pydicom / termcolor / nested_lookup / natsort / h5py / nipype / skimage /
  seaborn / Pmw / openpyxl are not installed  -> ``sys.modules`` stand-ins.  Only the
  nibabel stand-in has behaviour (axis codes from an affine, spatial slicing of an
  affine); everything else is an inert attribute bag, because the fit path never calls
  into those packages.

Nothing from the reference is copied into this repository: the throw-away copy lives
outside the repo and is deleted at interpreter exit.  ``/root/reference`` does not exist
on the GPU box, so this module is only ever used by ``oracle/make_golden.py`` and by
the ``not gpu`` tests that are explicitly skipped when the reference is absent.
"""
import atexit
import importlib
import importlib.util
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("DOSMA_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "dosma"))


# ----------------------------------------------------------------------------- nibabel stand-in
def _io_orientation(affine):
    """(axis, flip) per voxel axis: which RAS axis each array axis runs along."""
    rzs = np.asarray(affine, dtype=float)[:3, :3]
    zooms = np.sqrt((rzs * rzs).sum(axis=0))
    zooms[zooms == 0] = 1.0
    rs = rzs / zooms
    u, s, vt = np.linalg.svd(rs)
    keep = s > (s.max() * 3 * np.finfo(float).eps)
    r = u[:, keep] @ vt[keep, :]
    ornt = np.full((3, 2), np.nan)
    for col in range(3):
        c = r[:, col]
        if np.allclose(c, 0):
            continue
        out_ax = int(np.argmax(np.abs(c)))
        ornt[col, 0] = out_ax
        ornt[col, 1] = -1.0 if c[out_ax] < 0 else 1.0
        r[out_ax, :] = 0  # an output axis is used once
    return ornt


def _aff2axcodes(affine, labels=(("L", "R"), ("P", "A"), ("I", "S"))):
    codes = []
    for ax, flip in _io_orientation(affine):
        if np.isnan(ax):
            codes.append(None)
        else:
            codes.append(labels[int(ax)][0 if flip < 0 else 1])
    return tuple(codes)


class _SpatialFirstSlicer:
    """Slicing rules of a spatial image: first three axes are spatial and cannot be dropped."""

    def __init__(self, img):
        self.img = img

    def check_slicing(self, slicer, return_spatial=False):
        if not isinstance(slicer, tuple):
            slicer = (slicer,)
        ndim = len(self.img.shape)
        n_real = sum(1 for s in slicer if s is not None and s is not Ellipsis)
        out = []
        for s in slicer:
            if s is Ellipsis:
                out.extend([slice(None)] * (ndim - n_real))
            else:
                out.append(s)
        if len([s for s in out if s is not None]) > ndim:
            raise ValueError("too many indices")
        spatial = []
        seen = 0
        for s in out:
            if s is None:
                if seen < 3:
                    raise ValueError("Cannot add a new axis among the spatial axes")
                continue
            if seen < 3:
                if isinstance(s, (int, np.integer)):
                    raise ValueError("Cannot drop a spatial axis with an integer index")
                if not isinstance(s, slice):
                    if isinstance(s, np.ndarray) and s.dtype == bool:
                        raise ValueError("boolean spatial index not supported")
                    raise ValueError("fancy spatial index not supported")
                spatial.append(s)
            seen += 1
        out = tuple(out)
        return spatial if return_spatial else out

    def slice_affine(self, slicer):
        spatial = self.check_slicing(slicer, return_spatial=True)
        shape = self.img.shape[:3]
        t = np.eye(4)
        for i, s in enumerate(spatial):
            start, stop, step = s.indices(shape[i])
            t[i, i] = step
            t[i, 3] = start
        return np.asarray(self.img.affine) @ t


def _make_nibabel():
    nib = types.ModuleType("nibabel")
    ornt = types.ModuleType("nibabel.orientations")
    spat = types.ModuleType("nibabel.spatialimages")
    ornt.io_orientation = _io_orientation
    ornt.aff2axcodes = _aff2axcodes
    spat.SpatialFirstSlicer = _SpatialFirstSlicer
    nib.orientations = ornt
    nib.spatialimages = spat
    nib.aff2axcodes = _aff2axcodes
    nib.io_orientation = _io_orientation
    nib.__version__ = "0.0-stub"
    return {"nibabel": nib, "nibabel.orientations": ornt, "nibabel.spatialimages": spat}


# ----------------------------------------------------------------------------- inert stand-ins
class _Inert(types.ModuleType):
    """Module whose every attribute is an inert class (never exercised by the fit path)."""

    __path__ = []  # behave as a package so that submodule imports resolve

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _InertFinder:
    ROOTS = (
        "pydicom", "termcolor", "nested_lookup", "natsort", "h5py", "nipype", "skimage",
        "seaborn", "Pmw", "openpyxl", "SimpleITK", "sigpy", "cupy",
    )

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in self.ROOTS:
            from importlib.machinery import ModuleSpec

            return ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        m = _Inert(spec.name)
        if spec.name == "termcolor":
            m.colored = lambda s, *a, **k: s
        if spec.name == "nested_lookup":
            def _occ(d, key):
                n = 0
                if isinstance(d, dict):
                    for k, v in d.items():
                        n += (k == key) + _occ(v, key)
                return n

            def _lookup(key, d):
                found = []
                if isinstance(d, dict):
                    for k, v in d.items():
                        if k == key:
                            found.append(v)
                        found.extend(_lookup(key, v))
                return found

            m.get_occurrence_of_key = _occ
            m.nested_lookup = _lookup
        return m

    def exec_module(self, module):
        pass


_LOADED = {}
# stand-ins the product itself probes for (`import h5py` in dosma_amd/models/weights.py): removed from sys.modules once
# the reference is imported (its own modules keep their reference to the stub; dosma/utils/io_utils.py:7 imports h5py
# unconditionally, so the import itself needs one)
_DROP_AFTER_IMPORT = ("h5py",)
_SUBMODULES = ("dosma.scan_sequences", "dosma.scan_sequences.mri.qdess", "dosma.scan_sequences.mri.cube_quant",
               "dosma.scan_sequences.mri.cones", "dosma.scan_sequences.mri.mapss", "dosma.tissues", "dosma.models",
               "dosma.models.oaiunet2d", "dosma.models.seg_model", "dosma.models.stanford_qdess", "dosma.models.util",
               "dosma.core.quant_vals")


def load_reference():
    """Return the reference's ``dosma`` package imported from a throw-away copy."""
    if "dosma" in _LOADED:
        return _LOADED["dosma"]
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")

    # only packages that are really missing get a stand-in
    finder = _InertFinder()
    finder.ROOTS = tuple(
        r for r in finder.ROOTS if importlib.util.find_spec(r) is None
    )
    # cupy/sigpy/SimpleITK are probed by the reference via importlib.util.find_spec and must
    # stay "absent"; they are therefore NOT served by the inert finder.
    finder.ROOTS = tuple(r for r in finder.ROOTS if r not in ("cupy", "sigpy", "SimpleITK"))
    sys.meta_path.append(finder)
    if importlib.util.find_spec("nibabel") is None:
        sys.modules.update(_make_nibabel())

    for alias, target in (("round_", np.round), ("int", int), ("bool", bool),
                          ("float", float), ("complex", complex)):
        if not hasattr(np, alias):
            setattr(np, alias, target)

    tmp = tempfile.mkdtemp(prefix="dosma_ref_")
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    shutil.copytree(os.path.join(REFERENCE_ROOT, "dosma"), os.path.join(tmp, "dosma"))
    sys.path.insert(0, tmp)
    import matplotlib

    matplotlib.use("Agg")
    try:
        dosma = importlib.import_module("dosma")
        # the sub-packages `import dosma` does not pull in but the fixture generators import later
        # (oracle/make_golden.py g6: dosma.scan_sequences.mri.qdess -> dosma.tissues -> dosma.utils.img_utils ->
        # seaborn; g9: dosma.models.*): imported HERE, while the stand-ins are still resolvable
        for sub in _SUBMODULES:
            importlib.import_module(sub)
    finally:
        # the stand-ins are only for the reference's own import: leaving the finder on sys.meta_path would hand
        # an inert stub to any later `import h5py` / `import skimage` in the same process (e.g. the product's
        # weights loader choosing between h5py and its own HDF5 reader)
        sys.meta_path.remove(finder)
        for name in [m for m in sys.modules if m.split(".")[0] in _DROP_AFTER_IMPORT and isinstance(sys.modules[m], _Inert)]:
            del sys.modules[name]
    _LOADED["dosma"] = dosma
    return dosma


if __name__ == "__main__":
    d = load_reference()
    print("reference dosma", d.__version__, "imported from", os.path.dirname(d.__file__))
