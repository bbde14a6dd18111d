"""Writes tests/golden/keras_unet_small*.h5 + .npz: Keras-layout weight files of a small instance of the reference's
U-Net (3 levels, nf = 4, 8, 16), produced with the REAL h5py / libhdf5 so that dosma_amd/io/_hdf5_lite.py (the
dependency-free reader used where h5py is not installed) is tested against genuine HDF5 bytes.

Keras / TensorFlow are not available in this container, so the group / attribute layout is restated from
keras.engine.saving.save_weights_to_hdf5_group (Keras 2.x, what `model.save_weights("x.h5")` and
dosma/models/seg_model.py:87-92 `load_weights` use):
    root attrs: layer_names (fixed-length byte strings), backend, keras_version
    per layer : group <layer>, attr weight_names = [b"<layer>/kernel:0", ...], datasets at <layer>/<layer>/kernel:0
Variants: `_model.h5` wraps everything in a "model_weights" group (model.save format) and stores the string attributes
as variable-length UTF-8 (what newer h5py does for str lists); `_chunked.h5` stores the datasets chunked + gzip
(not what Keras writes; the reader must refuse it with a clear message or read it).

Run with an interpreter that has h5py (here: /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py).
"""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NF = (4, 8, 16)
NCLS = 4
rng = np.random.default_rng(20260928)


def layers():
    """(keras layer name, {weight name: array}) in the reference's creation order (oaiunet2d.py:201-287)."""
    out, n = [], {"conv2d": 0, "batch_normalization": 0, "conv2d_transpose": 0}

    def name(kind):
        n[kind] += 1
        return f"{kind}_{n[kind]}"

    def conv(cin, cout, k=3):
        return {"kernel:0": rng.standard_normal((k, k, cin, cout)).astype(np.float32) * 0.1,
                "bias:0": rng.standard_normal(cout).astype(np.float32) * 0.1}

    def bn(c):
        return {"gamma:0": rng.uniform(0.5, 1.5, c).astype(np.float32), "beta:0": rng.standard_normal(c).astype(np.float32),
                "moving_mean:0": rng.standard_normal(c).astype(np.float32),
                "moving_variance:0": rng.uniform(0.5, 2.0, c).astype(np.float32)}

    out.append(("input_1", {}))
    cin = 1
    for d, c in enumerate(NF):
        out.append((name("conv2d"), conv(cin, c)))
        out.append((name("conv2d"), conv(c, c)))
        out.append((name("batch_normalization"), bn(c)))
        out.append((f"dropout_{d + 1}", {}))
        if d < len(NF) - 1:
            out.append((f"max_pooling2d_{d + 1}", {}))
        cin = c
    for d in range(len(NF) - 2, -1, -1):
        c = NF[d]
        w = {"kernel:0": rng.standard_normal((3, 3, c, NF[d + 1])).astype(np.float32) * 0.1,
             "bias:0": rng.standard_normal(c).astype(np.float32) * 0.1}
        out.append((name("conv2d_transpose"), w))
        out.append((f"concatenate_{len(NF) - 1 - d}", {}))
        out.append((name("conv2d"), conv(2 * c, c)))
        out.append((name("conv2d"), conv(c, c)))
        out.append((name("batch_normalization"), bn(c)))
        out.append((f"dropout_{len(NF) + len(NF) - 1 - d}", {}))
    out.append((name("conv2d"), conv(NF[0], NCLS, k=1)))
    return out


def write(path, ls, wrap=False, vlen=False, chunked=False):
    with h5py.File(path, "w") as f:
        g = f.create_group("model_weights") if wrap else f
        names = [n for n, _ in ls]
        if vlen:
            g.attrs["layer_names"] = names  # list of str -> variable-length UTF-8 strings
            g.attrs["backend"] = "tensorflow"
            g.attrs["keras_version"] = "2.4.0"
        else:
            g.attrs["layer_names"] = np.array([n.encode("utf8") for n in names])
            g.attrs["backend"] = np.bytes_("tensorflow")
            g.attrs["keras_version"] = np.bytes_("2.1.6")
        for n, ws in ls:
            lg = g.create_group(n)
            wn = [f"{n}/{k}" for k in ws]
            if vlen:
                lg.attrs["weight_names"] = wn if wn else np.array([], dtype=h5py.string_dtype())
            else:
                lg.attrs["weight_names"] = np.array([w.encode("utf8") for w in wn]) if wn else np.array([], dtype="S1")
            for k, v in ws.items():
                if chunked:
                    lg.create_dataset(f"{n}/{k}", data=v, chunks=True, compression="gzip")
                else:
                    lg.create_dataset(f"{n}/{k}", data=v)


def write_misc():
    """Other corners of the format the reader claims: a group large enough for a multi-node B-tree, empty / compact /
    scalar datasets, big-endian and integer types, and a libver="latest" file (version-2 object headers, compact
    links, layout version 4, whose chunk index the reader must refuse)."""
    r = np.random.default_rng(1)
    with h5py.File(os.path.join(HERE, "h5_misc_latest.h5"), "w", libver="latest") as f:
        f.attrs["title"] = "latest format"
        f.attrs["n"] = np.int32(7)
        f.attrs["vec"] = np.arange(5, dtype=np.float64)
        g = f.create_group("a")
        g2 = g.create_group("b")
        g2.create_dataset("f64_be", data=r.standard_normal((3, 4)).astype(">f8"))
        g2.create_dataset("i16", data=np.arange(-5, 7, dtype=np.int16).reshape(3, 4))
        g2.create_dataset("u8", data=np.arange(10, dtype=np.uint8))
        g.create_dataset("scalar", data=np.float32(3.5))
        g.create_dataset("shuf", data=r.standard_normal((40, 30)).astype(np.float32), chunks=(16, 8),
                         compression="gzip", shuffle=True)
        g.create_dataset("names", data=np.array([b"alpha", b"be", b"gamma"]))
    with h5py.File(os.path.join(HERE, "h5_misc_earliest.h5"), "w", libver="earliest") as f:
        f.attrs["title"] = np.bytes_("earliest format")
        g = f.create_group("grp")
        for i in range(40):
            g.create_dataset(f"d{i:02d}", data=np.full((2,), i, dtype=np.float32))
        g.create_dataset("empty", data=np.zeros((0, 3), np.float32))
        f.create_dataset("compact", data=np.arange(6, dtype=np.int32))


if __name__ == "__main__":
    write_misc()
    ls = layers()
    write(os.path.join(HERE, "keras_unet_small.h5"), ls)
    write(os.path.join(HERE, "keras_unet_small_model.h5"), ls, wrap=True, vlen=True)
    write(os.path.join(HERE, "keras_unet_small_chunked.h5"), ls, chunked=True)
    flat = {f"{n}/{k}": v for n, ws in ls for k, v in ws.items()}
    np.savez(os.path.join(HERE, "keras_unet_small.npz"), layer_names=np.array([n for n, _ in ls]), **flat)
    print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version, "->", sorted(os.listdir(HERE))[-6:], file=sys.stderr)
