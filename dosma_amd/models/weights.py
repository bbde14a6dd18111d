"""Weights of the 2D U-Net in Keras layouts: naming, ordering for the C ABI, (de)serialisation.

The reference loads Keras ``.h5`` files (``keras_model.load_weights``, dosma/models/seg_model.py:87-92)
that are not distributed with the repository.  Here weights live in a dict ``name -> float32 array``
with Keras layouts (Conv2D kernel ``(kh, kw, Cin, Cout)``, Conv2DTranspose kernel
``(kh, kw, Cout, Cin)``, BatchNormalization ``gamma / beta / moving_mean / moving_variance``), can be
saved / loaded as ``.npz``, and are read from a Keras ``.h5`` by :func:`load_keras_h5` -- through h5py where it is
installed, otherwise through the dependency-free reader ``dosma_amd/io/_hdf5_lite.py`` (h5py is not in the MI355X
image; the reader is tested against files written by the real h5py, tests/test_hdf5_lite.py).
"""
import numpy as np

NF = (32, 64, 128, 256, 512, 1024)


def tensor_names(depth=6):
    """Tensor order expected by ``qmri_unet2d_create`` (= Keras layer-creation order)."""
    names = []
    for d in range(depth):
        names += [f"down{d}_conv1_kernel", f"down{d}_conv1_bias", f"down{d}_conv2_kernel", f"down{d}_conv2_bias",
                  f"down{d}_bn_gamma", f"down{d}_bn_beta", f"down{d}_bn_mean", f"down{d}_bn_var"]
    for d in range(depth - 2, -1, -1):
        names += [f"up{d}_deconv_kernel", f"up{d}_deconv_bias", f"up{d}_conv1_kernel", f"up{d}_conv1_bias",
                  f"up{d}_conv2_kernel", f"up{d}_conv2_bias",
                  f"up{d}_bn_gamma", f"up{d}_bn_beta", f"up{d}_bn_mean", f"up{d}_bn_var"]
    names += ["head_kernel", "head_bias"]
    return names


def expected_shapes(nf=NF, n_classes=4, in_channels=1):
    shapes = {}
    cin = in_channels
    for d, c in enumerate(nf):
        shapes[f"down{d}_conv1_kernel"] = (3, 3, cin, c)
        shapes[f"down{d}_conv2_kernel"] = (3, 3, c, c)
        for s in ("conv1_bias", "conv2_bias", "bn_gamma", "bn_beta", "bn_mean", "bn_var"):
            shapes[f"down{d}_{s}"] = (c,)
        cin = c
    for d in range(len(nf) - 2, -1, -1):
        c = nf[d]
        shapes[f"up{d}_deconv_kernel"] = (3, 3, c, nf[d + 1])
        shapes[f"up{d}_conv1_kernel"] = (3, 3, 2 * c, c)
        shapes[f"up{d}_conv2_kernel"] = (3, 3, c, c)
        for s in ("deconv_bias", "conv1_bias", "conv2_bias", "bn_gamma", "bn_beta", "bn_mean", "bn_var"):
            shapes[f"up{d}_{s}"] = (c,)
    shapes["head_kernel"] = (1, 1, nf[0], n_classes)
    shapes["head_bias"] = (n_classes,)
    return shapes


def validate(w, nf=NF, n_classes=4):
    shapes = expected_shapes(nf, n_classes)
    missing = [k for k in shapes if k not in w]
    if missing:
        raise ValueError(f"weights are missing tensors: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    for k, shp in shapes.items():
        if tuple(w[k].shape) != shp:
            raise ValueError(f"weight {k} has shape {tuple(w[k].shape)}, expected {shp}")


def to_abi_order(w, depth=6):
    return [np.ascontiguousarray(w[n], dtype=np.float32) for n in tensor_names(depth)]


def random_weights(seed=0, nf=NF, n_classes=4):
    """Seeded He-normal weights of the reference architecture (benchmarks / smoke tests: the trained
    .h5 files are not distributed, and throughput does not depend on the weight values)."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shp in expected_shapes(nf, n_classes).items():
        if name.endswith("_kernel"):
            fan_in = shp[0] * shp[1] * (shp[3] if "deconv" in name else shp[2])
            gain = 1.0 if ("deconv" in name or name.startswith("head")) else 2.0
            if "deconv" in name:
                fan_in = fan_in / 4.0 * 2.25 / 2.25  # ~2.25 taps contribute per output pixel
            w[name] = (rng.standard_normal(shp) * np.sqrt(gain / fan_in)).astype(np.float32)
        elif name.endswith("_gamma"):
            w[name] = rng.uniform(0.7, 1.3, shp).astype(np.float32)
        elif name.endswith("_var"):
            w[name] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        elif name.endswith("_mean"):
            w[name] = rng.uniform(0.2, 0.6, shp).astype(np.float32)
        else:
            w[name] = (0.05 * rng.standard_normal(shp)).astype(np.float32)
    return w


def save_npz(path, w):
    np.savez(path, **{k: np.asarray(v, dtype=np.float32) for k, v in w.items()})


def load_npz(path):
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


def load_keras_h5(path, depth=6):
    """Read a Keras ``.h5`` weights file of this architecture (layer groups in creation order, each
    with ``weight_names`` ``kernel:0 / bias:0 / gamma:0 / beta:0 / moving_mean:0 / moving_variance:0``)."""
    opener = None
    try:
        import h5py
        if isinstance(getattr(h5py, "File", None), type) and hasattr(h5py.File, "__enter__"):
            opener = lambda p: h5py.File(p, "r")  # noqa: E731
    except ImportError:
        pass
    if opener is None:  # no (real) h5py in this environment: the package's own reader of the HDF5 subset Keras writes
        from ..io import _hdf5_lite
        opener = _hdf5_lite.File
    with opener(path) as f:
        g = f["model_weights"] if "model_weights" in f else f
        layer_names = [n.decode() if isinstance(n, bytes) else str(n) for n in g.attrs["layer_names"]]
        per_layer = []
        for ln in layer_names:
            wn = [n.decode() if isinstance(n, bytes) else str(n) for n in g[ln].attrs["weight_names"]]
            if wn:
                per_layer.append({n.split("/")[-1].split(":")[0]: np.asarray(g[ln][n], dtype=np.float32) for n in wn})
    order = []
    for d in range(depth):
        order += [f"down{d}_conv1", f"down{d}_conv2", f"down{d}_bn"]
    for d in range(depth - 2, -1, -1):
        order += [f"up{d}_deconv", f"up{d}_conv1", f"up{d}_conv2", f"up{d}_bn"]
    order.append("head")
    if len(order) != len(per_layer):
        raise ValueError(f"expected {len(order)} weighted layers, file has {len(per_layer)}")
    w = {}
    for name, tensors in zip(order, per_layer):
        if name.endswith("_bn"):
            w[f"{name}_gamma"], w[f"{name}_beta"] = tensors["gamma"], tensors["beta"]
            w[f"{name}_mean"], w[f"{name}_var"] = tensors["moving_mean"], tensors["moving_variance"]
        else:
            w[f"{name}_kernel"], w[f"{name}_bias"] = tensors["kernel"], tensors["bias"]
    return w


def find_weights(weights_dir, tissue_id, extensions=(".h5", ".npz")):
    """The weights file of a tissue: the one regular file in ``weights_dir`` whose name contains ``tissue_id`` (the
    reference's ``Tissue.STR_ID``: "fc", "tc", "pc", "men") and ends in a weights extension -- the discovery rule of
    ``Tissue.find_weights`` (dosma/tissues/tissue.py:128-160; the reference looks for ``.h5`` only, this package also
    stores weights as ``.npz``).  ``ValueError`` when no file or more than one matches."""
    import os

    hits = sorted(os.path.join(weights_dir, name) for name in os.listdir(weights_dir)
                  if tissue_id in name and name.endswith(tuple(extensions))
                  and os.path.isfile(os.path.join(weights_dir, name)))
    if len(hits) > 1:
        raise ValueError("There are multiple weights files, please remove duplicates")
    if not hits:
        raise ValueError("No file found that contains '{}' and ends in '{}'".format(tissue_id, extensions))
    return hits[0]
