"""dosma_amd/io/_hdf5_lite.py -- the dependency-free HDF5 reader behind ``load_keras_h5`` -- against files written by
the REAL h5py 3.3 / libhdf5 1.10.6 (tests/golden/make_h5_fixture.py, run once with an interpreter that has h5py; the
expected arrays are stored beside the files as .npz).  The reference loads its weights with
``keras_model.load_weights(path.h5)`` (dosma/models/seg_model.py:87-92)."""
import os

import numpy as np
import pytest

from dosma_amd.io import _hdf5_lite as h5
from dosma_amd.models import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _p(name):
    return os.path.join(GOLDEN, name)


def _s(v):
    return v.decode() if isinstance(v, bytes) else str(v)


@pytest.mark.parametrize("name", ["keras_unet_small.h5", "keras_unet_small_model.h5", "keras_unet_small_chunked.h5"])
def test_keras_weight_files_read_bit_exact(name):
    """Old-style groups + fixed-length string attributes (what Keras writes); the model.save layout with
    variable-length string attributes; chunked + deflate datasets."""
    ref = np.load(_p("keras_unet_small.npz"))
    with h5.File(_p(name)) as f:
        g = f["model_weights"] if "model_weights" in f else f
        names = [_s(n) for n in g.attrs["layer_names"]]
        assert names == [str(n) for n in ref["layer_names"]]
        assert _s(g.attrs["backend"]) == "tensorflow" and _s(g.attrs["keras_version"]).startswith("2.")
        n = 0
        for ln in names:
            wn = [_s(x) for x in g[ln].attrs["weight_names"]]
            assert sorted(g[ln].keys()) == ([ln] if wn else [])
            for w in wn:
                a = np.asarray(g[ln][w])
                assert a.dtype == np.float32 and np.array_equal(a, ref[w]), w
                n += 1
        assert n == len(ref.files) - 1 == 46
        with pytest.raises(KeyError):
            g["conv2d_1/nope"]


@pytest.mark.parametrize("name", ["keras_unet_small.h5", "keras_unet_small_model.h5"])
def test_load_keras_h5_maps_layers_to_the_network(name):
    w = W.load_keras_h5(_p(name), depth=3)
    nf = (4, 8, 16)
    W.validate(w, nf=nf, n_classes=4)
    ref = np.load(_p("keras_unet_small.npz"))
    assert np.array_equal(w["down0_conv1_kernel"], ref["conv2d_1/kernel:0"])
    assert np.array_equal(w["down2_bn_var"], ref["batch_normalization_3/moving_variance:0"])
    assert np.array_equal(w["up1_deconv_kernel"], ref["conv2d_transpose_1/kernel:0"]) and w["up1_deconv_kernel"].shape == (3, 3, 8, 16)
    assert np.array_equal(w["up0_conv1_kernel"], ref["conv2d_9/kernel:0"]) and w["up0_conv1_kernel"].shape == (3, 3, 8, 4)
    assert np.array_equal(w["head_kernel"], ref["conv2d_11/kernel:0"]) and w["head_kernel"].shape == (1, 1, 4, 4)
    with pytest.raises(ValueError):
        W.load_keras_h5(_p(name), depth=6)  # the file holds a 3-level network


def test_other_hdf5_features():
    with h5.File(_p("h5_misc_earliest.h5")) as f:  # 41 links: a group B-tree with several symbol-table nodes
        assert _s(f.attrs["title"]) == "earliest format"
        keys = f["grp"].keys()
        assert len(keys) == 41 and {f"d{i:02d}" for i in range(40)} <= set(keys)
        for i in (0, 17, 39):
            assert np.array_equal(f[f"grp/d{i:02d}"].read(), np.full(2, i, np.float32))
        assert f["grp/empty"].read().shape == (0, 3)
        assert np.array_equal(f["compact"].read(), np.arange(6, dtype=np.int32))
    with h5.File(_p("h5_misc_latest.h5")) as f:  # version-2 object headers, compact links, layout version 4
        assert f.attrs["title"] == "latest format" and f.attrs["n"] == 7
        assert np.array_equal(f.attrs["vec"], np.arange(5.0))
        a = f["a/b/f64_be"].read()
        assert a.dtype == np.float64 and a.shape == (3, 4) and np.isfinite(a).all() and abs(a).max() < 10
        assert np.array_equal(f["a/b/i16"].read(), np.arange(-5, 7, dtype=np.int16).reshape(3, 4))
        assert np.array_equal(f["a/b/u8"].read(), np.arange(10, dtype=np.uint8))
        assert f["a/scalar"].read() == np.float32(3.5)
        assert list(f["a/names"].read()) == [b"alpha", b"be", b"gamma"]
        with pytest.raises(NotImplementedError):
            f["a/shuf"].read()  # version-4 chunk index: refused, not guessed


def test_not_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 10)
    with pytest.raises(ValueError):
        h5.File(str(p))
