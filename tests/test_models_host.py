"""Host logic of the segmentation path that needs no GPU: weights container, registry, validation,
whitening helper, and the torch restatement's layer semantics (pinned against naive definitions)."""
import numpy as np
import pytest

from dosma_amd.models import SUPPORTED_MODELS, weights as W, whiten_volume
from dosma_amd.models.oaiunet2d import IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, OAIUnet2D
from dosma_amd.models.stanford_qdess import StanfordQDessUNet2D
from oracle import unet_oracle as uo


def test_aliases_nonempty_and_disjoint():
    """reference tests/models/test_util.py:6-34."""
    models = [OAIUnet2D, IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, StanfordQDessUNet2D]
    aliases = [set(m.ALIASES) for m in models]
    assert all(a and "" not in a for a in aliases)
    for i in range(len(aliases)):
        for j in range(i + 1, len(aliases)):
            assert not (aliases[i] & aliases[j])
    # the reference registry (models/util.py:18) + the SKM-TEA template (stanford_qdess.py:60), which the
    # reference only exposes by class
    assert SUPPORTED_MODELS == ["oai-unet2d", "iwoai-2019-t6", "iwoai-2019-t6-normalized",
                                "stanford-qdess-2021-unet2d"]
    assert StanfordQDessUNet2D.CATEGORIES == ("pc", "fc", "tc", "men")


def test_weights_container(tmp_path):
    w = W.random_weights(seed=1, nf=(32, 64), n_classes=4)
    assert len(W.tensor_names(depth=2)) == 2 * 8 + 10 + 2
    W.validate(w, nf=(32, 64))
    assert w["up0_deconv_kernel"].shape == (3, 3, 32, 64) and w["up0_conv1_kernel"].shape == (3, 3, 64, 32)
    W.save_npz(tmp_path / "w.npz", w)
    w2 = W.load_npz(tmp_path / "w.npz")
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    bad = dict(w)
    bad["down1_conv2_kernel"] = bad["down1_conv2_kernel"][..., :3]
    with pytest.raises(ValueError):
        W.validate(bad, nf=(32, 64))
    del bad["head_bias"]
    with pytest.raises(ValueError):
        W.validate(bad, nf=(32, 64))
    full = W.random_weights(seed=0)
    assert sum(v.size for v in full.values()) == 34_597_892  # 34.6 M parameters
    assert [n for n in W.tensor_names()] == [n for n in W.tensor_names(6)] and len(W.tensor_names()) == 100


def test_find_weights(tmp_path):
    """Tissue.find_weights (tissue.py:128-160): exactly one file containing the tissue id with a weights extension."""
    for name in ("iwoai_fc_weights.h5", "tc_weights.npz", "notes_fc.txt", "men_a.h5", "men_b.h5"):
        (tmp_path / name).write_bytes(b"x")
    (tmp_path / "pc_dir.h5").mkdir()
    assert W.find_weights(str(tmp_path), "fc").endswith("iwoai_fc_weights.h5")
    assert W.find_weights(str(tmp_path), "tc").endswith("tc_weights.npz")
    with pytest.raises(ValueError):
        W.find_weights(str(tmp_path), "men")  # two candidates
    with pytest.raises(ValueError):
        W.find_weights(str(tmp_path), "pc")   # a directory does not count


def test_whiten_volume_matches_reference_formula():
    x = np.random.default_rng(0).uniform(0, 500, (5, 6, 7)).astype(np.float32)
    y = whiten_volume(x)
    assert abs(float(y.mean())) < 1e-5 and abs(float(y.std()) - 1) < 1e-5
    assert np.array_equal(whiten_volume(x, eps=1e-8), (x - np.mean(x)) / (np.std(x) + 1e-8))
    with pytest.raises(ValueError):
        whiten_volume(x[0])


def test_restatement_layer_semantics():
    """Conv2DTranspose alignment vs the scatter definition, and BN-after-ReLU / concat order through a
    2-level network evaluated by hand."""
    rng = np.random.default_rng(2)
    nf = (32, 64)
    w = uo.make_weights(seed=5, nf=nf)
    x = rng.standard_normal((1, 4, 4)).astype(np.float32)
    logits, feats = uo.forward(w, x, nf=nf, return_features=True, dtype="float64")
    assert logits.shape == (1, 4, 4, 4)
    d1 = feats["down1"]  # (1, 2, 2, 64) post-BN
    up = uo.deconv_naive(d1, w["up0_deconv_kernel"], w["up0_deconv_bias"])
    assert np.allclose(up, feats["up0_deconv"], atol=1e-10)
    # first block by hand at one pixel: conv -> relu -> conv -> relu -> BN
    k1, b1 = w["down0_conv1_kernel"].astype(np.float64), w["down0_conv1_bias"].astype(np.float64)
    xp = np.pad(x[0].astype(np.float64), 1)
    c1 = np.zeros((4, 4, 32))
    for i in range(4):
        for j in range(4):
            c1[i, j] = np.maximum(np.einsum("hw,hwc->c", xp[i:i + 3, j:j + 3], k1[:, :, 0, :]) + b1, 0)
    k2, b2 = w["down0_conv2_kernel"].astype(np.float64), w["down0_conv2_bias"].astype(np.float64)
    c1p = np.pad(c1, ((1, 1), (1, 1), (0, 0)))
    c2 = np.maximum(np.einsum("hwc,hwco->o", c1p[1:4, 2:5], k2) + b2, 0)  # pixel (1, 2)
    g, bt, mu, var = (w[f"down0_bn_{n}"].astype(np.float64) for n in ("gamma", "beta", "mean", "var"))
    want = g * (c2 - mu) / np.sqrt(var + 1e-3) + bt
    assert np.allclose(feats["down0"][0, 1, 2], want, atol=1e-10)


# ---- a14: generate_mask's pre / post logic against the reference's OWN code (golden g9, oracle/make_golden.py g9) ----
G9_TEMPLATES = {
    "iwoai": (IWOAIOAIUnet2D, None),
    "iwoai_norm": (IWOAIOAIUnet2DNormalized, None),
    "oai": (OAIUnet2D, None),
    "stanford": (StanfordQDessUNet2D, None),
    "stanford_thr": (StanfordQDessUNet2D, 0.7),
}


def _g9_probabilities(z, tname, net_in):
    """The stand-in ``predict`` of the fixture: sigmoid(a_c * v + b_c) per class (float64 -> float32)."""
    a4, b4 = z["a4"], z["b4"]
    a, b = {"iwoai": (a4 / 300.0, b4 - 0.8), "oai": (a4[:1], b4[:1])}.get(tname, (a4, b4))
    logits = net_in.astype(np.float64) * a + b
    return (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)


def test_generate_mask_pre_post_vs_reference(golden):
    """``_to_network_input`` hands the network exactly the array the reference's ``generate_mask`` hands ``model.predict``
    and ``_from_network_output`` turns the same probabilities into the same MedicalVolumes (keys and their order, dtype,
    voxels, affine, orientation) -- four input orientations with an anisotropic, offset affine, every template
    (oaiunet2d.py:140-175, 291-320, 344-345; stanford_qdess.py:158-205)."""
    from dosma_amd import MedicalVolume

    z = golden("g9_generate_mask.npz")
    for oname in z["orient_names"]:
        vol = MedicalVolume(z[f"{oname}_vol"], z[f"{oname}_affine"])
        for tname, (cls, thr) in G9_TEMPLATES.items():
            tag = f"{oname}_{tname}"
            model = object.__new__(cls)          # no engine: only the host halves are under test
            if thr is not None:
                model.sigmoid_threshold = thr
            vol_sag, v = model._to_network_input(vol)
            ref_in = z[f"{tag}_net_in"]
            assert v.shape == ref_in.shape and v.dtype == ref_in.dtype, tag
            assert np.array_equal(v, ref_in), tag        # same numpy expression on the same voxels: bit-equal
            out = model._from_network_output(_g9_probabilities(z, tname, ref_in), vol_sag, vol.orientation)
            keys = [str(k) for k in z[f"{tag}_keys"]]
            if keys:
                assert list(out.keys()) == keys, tag
                items = list(out.items())
            else:
                assert isinstance(out, MedicalVolume), tag
                items = [("", out)]
            for k, m in items:
                assert m.volume.dtype == np.uint8 and m.orientation == vol.orientation, (tag, k)
                assert np.array_equal(m.volume, z[f"{tag}_mask_{k}"]), (tag, k)
                assert np.allclose(m.affine, z[f"{tag}_affine_{k}"], rtol=0, atol=1e-12), (tag, k)


def test_whiten_volume_bit_exact_vs_reference(golden):
    """seg_model.py:114-127 on float32 and float64 volumes (the reference pins this bit-exactly on its own data,
    tests/models/test_oaiunet2d.py:61-81)."""
    z = golden("g9_generate_mask.npz")
    base = z["sag_vol"]
    for dt in (np.float32, np.float64):
        for eps in (0.0, 1e-8):
            ref = z[f"whiten_{np.dtype(dt).name}_{eps:g}"]
            got = whiten_volume(base.astype(dt), eps=eps)
            assert got.dtype == ref.dtype and np.array_equal(got, ref)
