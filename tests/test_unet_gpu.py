"""2D-UNet kernels on the GPU vs the torch-CPU restatement (oracle/unet_oracle.py).

PARITY WITH THE REFERENCE IS UNPINNED for this path (no Keras/TF, weights or test data in this image --
see the oracle's header); what is pinned here is self-consistency: same seeded Keras-layout weights and
inputs through the HIP kernels and through the line-by-line restatement of oaiunet2d.py:197-289.
Tolerance: logits within 1e-3 abs (north_star) in the parity mode "fp16x3" (fp16 hi + lo operand parts, three MFMAs
per product); the plain bf16 mode is checked at bf16 accuracy and on mask agreement.  Full-size (384 x 384, 512 x 512)
parity and the kernel dispatch are in tests/test_unet_fullsize_gpu.py."""
import numpy as np
import pytest

from dosma_amd import _lib as L
from oracle import unet_oracle as uo

pytestmark = pytest.mark.gpu


def torch_conv(x, k, b, relu, transposed):
    import torch
    import torch.nn.functional as F

    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    if transposed:
        kt = torch.from_numpy(k).double().permute(3, 2, 0, 1)
        y = F.conv_transpose2d(xt, kt, torch.from_numpy(b).double(), stride=2)[:, :, : 2 * x.shape[1], : 2 * x.shape[2]]
    else:
        kt = torch.from_numpy(k).double().permute(3, 2, 0, 1)
        y = F.conv2d(xt, kt, torch.from_numpy(b).double(), padding=1)
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("cin,cout,hw,transposed", [(32, 32, (20, 12), False), (64, 32, (9, 7), False),
                                                    (32, 64, (16, 16), False), (128, 128, (8, 24), False),
                                                    (64, 256, (5, 6), False), (128, 128, (12, 12), False),
                                                    (64, 256, (24, 24), False), (64, 32, (6, 5), True),
                                                    (128, 64, (8, 8), True), (256, 128, (3, 4), True),
                                                    # W % 32 == 0: the 8 x 32 tiles of conv_s3_kernel (partial last row of
                                                    # tiles: H = 12, 20), one / several channel blocks, 1 .. 4 K chunks
                                                    (32, 32, (12, 32), False), (64, 64, (20, 64), False),
                                                    (128, 128, (16, 96), False), (96, 256, (8, 32), False),
                                                    # widths the flattened tiling takes (W <= 48) with several tiles per launch
                                                    (64, 128, (48, 48), False), (32, 64, (30, 46), False),
                                                    # a width neither tiling takes (not a multiple of 32, > 48): general kernel
                                                    (32, 64, (8, 56), False)])
def test_conv_layer_vs_torch(cin, cout, hw, transposed):
    rng = np.random.default_rng(cin * 7 + cout)
    B, (H, W) = 3, hw
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((3, 3, cout, cin) if transposed else (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = torch_conv(x, k, b, relu=not transposed, transposed=transposed)
    y3 = L.conv2d_nhwc_host(x, k, b, relu=not transposed, transposed=transposed, precision="fp16x3")
    assert y3.shape == ref.shape
    assert np.abs(y3 - ref).max() < 2e-5, np.abs(y3 - ref).max()  # fp16 hi + lo parts: ~2^-21 per operand
    if not transposed:  # the same layer forced onto the general kernel (what the engine uses where conv_s3 does not tile)
        yg = L.conv2d_nhwc_host(x, k, b, relu=True, precision="fp16x3-general")
        assert np.abs(yg - ref).max() < 2e-5, np.abs(yg - ref).max()
    y1 = L.conv2d_nhwc_host(x, k, b, relu=not transposed, transposed=transposed, precision="bf16")
    assert np.abs(y1 - ref).max() < 6e-2  # bf16 operands: ~2^-9 relative per product
    if cin % 64 == 0 and (hw[1] % 32 == 0 or hw[1] + 2 <= 50):
        # the same layer in plain bf16 on conv_s3_kernel (64-channel chunks in the two planes of its LDS image): same
        # operand rounding as the round-1 kernel, so the two agree to accumulation order + the bf16 rounding of the output
        ys = L.conv2d_nhwc_host(x, k, b, relu=not transposed, transposed=transposed, precision="bf16-s3")
        assert np.abs(ys - ref).max() < 6e-2, np.abs(ys - ref).max()
        assert np.abs(ys - y1).max() < 2e-2, np.abs(ys - y1).max()
    # fused BatchNorm affine after the ReLU
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    y = L.conv2d_nhwc_host(x, k, b, scale=sc, shift=sh, relu=True, transposed=transposed)
    ref2 = np.maximum(torch_conv(x, k, b, False, transposed), 0) * sc + sh
    assert np.abs(y - ref2).max() < 3e-5


@pytest.mark.parametrize("cin,cout,hw,B", [
    (128, 128, (16, 96), 3),    # 2D tiles, exactly one tile row
    (96, 256, (8, 32), 3),      # 2D, H < 16: rows 8..15 of every tile are outside the image; 3 K chunks, 2 channel blocks
    (128, 128, (20, 64), 2),    # 2D, partial second tile row
    (32, 128, (48, 32), 5),     # 2D, one chunk (two k-halves), several tiles per image: the K stream crosses tile boundaries
    (64, 128, (48, 48), 3),     # flattened, several 512-position tiles
    (64, 256, (24, 24), 3),     # flattened, tiles straddle images
    (256, 128, (12, 12), 7),    # flattened 12 x 12 (the 384-pixel network's deepest level), 8 chunks
    (128, 384, (5, 6), 2),      # flattened, less than one tile, three channel blocks
    (32, 64, (32, 64), 3),      # 64-channel blocks (CT = 2; image tiles of 24 rows): one chunk, second tile row partial
    (128, 64, (16, 32), 4),     # 64-channel block, four chunks, less than one tile row
    (64, 64, (48, 32), 2),      # 64-channel block, two whole tile rows
    (64, 64, (50, 96), 2),      # three tile rows, the last with 2 of 24 rows inside the image
    (32, 64, (24, 32), 264),    # 64-channel blocks with a channel-split last round (two 32-channel sub-items per leftover item)
    (64, 192, (30, 46), 2),     # 64-channel blocks x 3, flattened
    # more work items than CUs, with a short last round: 264 items -> 33 per XCD on 32 blocks = one round + 1 leftover item per
    # XCD, which runs as four 32-channel sub-items (the channel-split last round); flattened: 285 tiles -> 36 per XCD, 4 leftover
    (32, 128, (16, 32), 264),
    (64, 128, (12, 12), 800),
    (256, 512, (12, 12), 3),    # four channel blocks = two groups of two (2.4 MB per pair): group by group, tile-major inside (down4.conv1)
    (96, 256, (16, 64), 70),    # one group of two on image tiles, more items than blocks: pairs side by side, parameters of both resident
    # W % 32 != 0 and wider than the flattened tiling: image tiles with a RAGGED last column tile (pixels at or beyond W read as zero
    # padding, are neither stored nor tracked) -- the 80 / 56 / 104 / 120-wide levels of 320 / 448 / 416 / 480-pixel slices
    (64, 128, (20, 80), 2),
    (32, 64, (26, 56), 3),
    (128, 256, (17, 104), 1),
    (96, 64, (24, 120), 2),
    (64, 128, (33, 51), 2),     # the narrowest such level: one whole and one 19-pixel column tile
])
def test_conv_c4_kernel_vs_torch(cin, cout, hw, B):
    """conv_c4_kernel (one wave per SIMD, 128 x 128 register tiles; unet_c4.hip) forced on, against the fp64 convolution
    and against conv_s3_kernel on the same layer (/root/reference/dosma/models/oaiunet2d.py:213-226)."""
    rng = np.random.default_rng(cin * 11 + cout + hw[0])
    H, W = hw
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    ref = np.maximum(torch_conv(x, k, b, False, False), 0) * sc + sh
    y4 = L.conv2d_nhwc_host(x, k, b, scale=sc, shift=sh, relu=True, precision="fp16x3-c4")
    assert y4.shape == ref.shape
    assert np.abs(y4 - ref).max() < 3e-5, np.abs(y4 - ref).max()
    y3 = L.conv2d_nhwc_host(x, k, b, scale=sc, shift=sh, relu=True, precision="fp16x3-s3")
    assert np.abs(y3 - ref).max() < 3e-5
    again = L.conv2d_nhwc_host(x, k, b, scale=sc, shift=sh, relu=True, precision="fp16x3-c4")
    assert np.array_equal(y4, again)  # no atomics, no timing dependence: same bits every run
    # the channel-split last round (leftover work items as four 32-channel sub-items) adds the same products in the same order
    whole = L.conv2d_nhwc_host(x, k, b, scale=sc, shift=sh, relu=True, precision="fp16x3-c4-nosplit")
    assert np.array_equal(y4, whole)


@pytest.mark.parametrize("cin,cout,hw,B", [
    (64, 32, (16, 32), 3),      # image tiles, exactly one tile row, two chunks (four k-steps), one channel block
    (128, 64, (20, 64), 2),     # image tiles, partial second tile row, two channel blocks
    (32, 32, (8, 32), 3),       # one chunk = two k-steps: the shortest item (the weight request pointer wraps in the prologue)
    (256, 128, (48, 96), 2),    # the 96 x 96 -> 192 x 192 shape family: three tile rows / columns, eight chunks, four channel blocks
    (64, 64, (48, 48), 3),      # flattened, several 512-position tiles
    (128, 96, (24, 24), 3),     # flattened, tiles straddle images, three channel blocks
    (512, 64, (12, 12), 7),     # flattened 12 x 12 (the deepest level's transposed convolution), 16 chunks
    (512, 128, (12, 12), 3),    # four channel blocks = one group of four (2.4 MB of weights stay in L2): tile-major
    (512, 256, (12, 12), 2),    # eight channel blocks = two groups of four: group by group, tile-major inside (the up3.deconv shape)
    (2048, 64, (5, 6), 2),      # 2.4 MB of weights PER channel block: no group fits -- channel-major item order
    (64, 32, (5, 6), 2),        # flattened, less than one tile
    (32, 32, (16, 32), 300),    # more work items than CUs (300 tiles): several items per block, the K stream crosses item boundaries
    (64, 64, (12, 12), 500),    # the same flattened: 56 tiles x 2 channel blocks per ... > 256 items
    # input grids with W % 32 != 0, wider than the flattened tiling: image tiles with a ragged last column tile
    (64, 32, (10, 60), 2),
    (128, 64, (18, 52), 3),
    (64, 96, (16, 104), 2),
])
def test_deconv_d4_kernel_vs_torch(cin, cout, hw, B):
    """deconv_d4_kernel (the transposed convolution on one wave per SIMD: 4 row-tiles x 4 phases x 32 channels; unet_d4.hip)
    forced on, against the fp64 transposed convolution and against conv_s3_kernel<32, *, DECONV> on the same layer
    (/root/reference/dosma/models/oaiunet2d.py:259-261: Conv2DTranspose(3x3, strides 2, SAME))."""
    rng = np.random.default_rng(cin * 13 + cout + hw[0])
    H, W = hw
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((3, 3, cout, cin)) / np.sqrt(2.25 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = torch_conv(x, k, b, False, True)
    y4 = L.conv2d_nhwc_host(x, k, b, relu=False, transposed=True, precision="fp16x3-c4")
    assert y4.shape == ref.shape == (B, 2 * H, 2 * W, cout)
    assert np.abs(y4 - ref).max() < 3e-5, np.abs(y4 - ref).max()
    y3 = L.conv2d_nhwc_host(x, k, b, relu=False, transposed=True, precision="fp16x3-s3")
    assert np.abs(y3 - ref).max() < 3e-5
    again = L.conv2d_nhwc_host(x, k, b, relu=False, transposed=True, precision="fp16x3-c4")
    assert np.array_equal(y4, again)  # no atomics, no timing dependence: same bits every run
    # the same bits whatever the batch a slice travels in (tiles of the flattened stack straddle images: slice 1 alone)
    if B <= 8:
        alone = L.conv2d_nhwc_host(x[1:2], k, b, relu=False, transposed=True, precision="fp16x3-c4")
        assert np.array_equal(alone[0], y4[1])
    # the epilogue's optional ReLU + affine (the network's transposed convolutions use neither; the operator entry offers both)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    ya = L.conv2d_nhwc_host(x, k, b, scale=sc, shift=sh, relu=True, transposed=True, precision="fp16x3-c4")
    assert np.abs(ya - (np.maximum(ref, 0) * sc + sh)).max() < 3e-5


def test_deconv_matches_the_scatter_definition():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 3, 4, 32)).astype(np.float32)
    k = rng.standard_normal((3, 3, 32, 32)).astype(np.float32) * 0.1
    b = rng.standard_normal(32).astype(np.float32)
    y = L.conv2d_nhwc_host(x, k, b, relu=False, transposed=True)
    assert np.abs(y - uo.deconv_naive(x, k, b)).max() < 2e-4


@pytest.fixture(scope="module")
def small_net():
    w = uo.make_weights(seed=3)
    tensors = weights_in_abi_order(w)
    return w, tensors


def weights_in_abi_order(w, depth=6):
    t = []
    for d in range(depth):
        t += [w[f"down{d}_conv1_kernel"], w[f"down{d}_conv1_bias"], w[f"down{d}_conv2_kernel"], w[f"down{d}_conv2_bias"],
              w[f"down{d}_bn_gamma"], w[f"down{d}_bn_beta"], w[f"down{d}_bn_mean"], w[f"down{d}_bn_var"]]
    for d in range(depth - 2, -1, -1):
        t += [w[f"up{d}_deconv_kernel"], w[f"up{d}_deconv_bias"], w[f"up{d}_conv1_kernel"], w[f"up{d}_conv1_bias"],
              w[f"up{d}_conv2_kernel"], w[f"up{d}_conv2_bias"],
              w[f"up{d}_bn_gamma"], w[f"up{d}_bn_beta"], w[f"up{d}_bn_mean"], w[f"up{d}_bn_var"]]
    t += [w["head_kernel"], w["head_bias"]]
    return t


def test_full_network_logits_vs_restatement(small_net, test_device):
    w, tensors = small_net
    rng = np.random.default_rng(0)
    S, H, W = 5, 64, 96
    vol = (rng.standard_normal((S, H, W)) * 120 + 300).astype(np.float32)
    # 5 slices -> batches 2,2,1; QMRI_TEST_DEVICE picks the HIP ordinal (a non-zero one on a multi-GPU box)
    eng = L.Unet2dEngine(tensors, H, W, max_batch=2, precision="fp16x3", device=test_device)
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
    ref = uo.forward(w, xw, dtype="float64")
    err = np.abs(logits - ref)
    assert err.max() < 1e-3, f"max logit error {err.max()}"
    assert np.array_equal(mask, (logits > 0).astype(np.uint8))
    agree = (mask == (ref > 0)).mean()
    assert agree > 0.9999
    # plain bf16 mode: same network, bf16-level logits, masks agree except next to the decision boundary
    eng.set_precision("bf16")
    logits16, mask16 = eng.forward_host(vol, whiten=True, eps=0.0)
    assert np.abs(logits16 - ref).max() < 0.25
    assert (mask16 == (ref > 0)).mean() > 0.995
    flips = mask16 != (ref > 0)
    assert np.abs(ref[flips]).max() < 0.25 if flips.any() else True
    eng.close()


def test_whitening_and_batching_are_consistent(small_net):
    w, tensors = small_net
    rng = np.random.default_rng(5)
    S, H, W = 4, 32, 32
    vol = rng.uniform(0, 1000, (S, H, W)).astype(np.float32)
    a = L.Unet2dEngine(tensors, H, W, max_batch=4)
    b = L.Unet2dEngine(tensors, H, W, max_batch=1)
    la, _ = a.forward_host(vol, whiten=True, eps=1e-8)
    lb, _ = b.forward_host(vol, whiten=True, eps=1e-8)
    assert np.array_equal(la, lb)  # slices are independent: batch size must not change a bit
    lc, _ = a.forward_host(uo.whiten_volume(vol, 1e-8).astype(np.float32), whiten=False)
    assert np.abs(la - lc).max() < 1e-3
    with pytest.raises(ValueError):
        a.forward_host(vol[:, :16, :])
    with pytest.raises(ValueError):
        L.Unet2dEngine(tensors, 50, 50)  # 50 -> 25 (odd) is pooled by 3, which does not divide it: the reference's graph
                                         # does not build either (Concatenate of 24 x 24 with 25 x 25, oaiunet2d.py:257-264)
    with pytest.raises(ValueError):
        L.Unet2dEngine(tensors, 64, 63)  # even height -> (2, 2) pooling on both axes, odd width


@pytest.mark.parametrize("precision,tol", [("fp16x3", 1e-3), ("bf16", 0.25)])
def test_odd_sizes_take_the_3x3_pooling_branch(small_net, precision, tol):
    """oaiunet2d.py:234-261: MaxPooling2D((3, 3)) / Conv2DTranspose(strides=(3, 3)) where the level's height is odd.
    72 x 144: levels 72, 36, 18, 9 (odd: pooled by 3), 3 (odd: by 3), 1."""
    w, tensors = small_net
    assert uo.level_factors(72, 144) == [2, 2, 2, 3, 3]
    rng = np.random.default_rng(72)
    vol = (rng.standard_normal((3, 72, 144)) * 90 + 200).astype(np.float32)
    eng = L.Unet2dEngine(tensors, 72, 144, max_batch=2, precision=precision)
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    ref = uo.forward(w, uo.whiten_volume(vol.astype(np.float64)).astype(np.float32), dtype="float64")
    assert logits.shape == ref.shape == (3, 72, 144, 4)
    assert np.abs(logits - ref).max() < tol, np.abs(logits - ref).max()
    if precision == "fp16x3":
        tr = eng.trace()
        assert "pool3:split" in tr and any(t.startswith("up3.deconv:igemm/stride3") for t in tr)
    eng.close()


# ----------------------------------------------------------------------------- drop-in model classes
def test_generate_mask_drop_in(small_net, tmp_path):
    """The reference's generate_mask contract (oaiunet2d.py:291-320): dict fc/tc/pc/men of uint8 {0,1}
    MedicalVolumes with the input's shape, orientation and affine -- for any input orientation."""
    from dosma_amd import MedicalVolume
    from dosma_amd.models import (IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, OAIUnet2D, get_model,
                                  model_from_config)
    from dosma_amd.models import weights as W

    w, _ = small_net
    rng = np.random.default_rng(9)
    H, Wd, S = 64, 32, 6
    sag = (rng.standard_normal((H, Wd, S)) * 80 + 200).astype(np.float32)  # (SI, AP, LR)
    aff = np.array([[0, 0, 1.5, -40.0], [0, -0.4, 0, 60.0], [-0.4, 0, 0, 70.0], [0, 0, 0, 1.0]])
    mv = MedicalVolume(sag, aff)
    assert mv.orientation == ("SI", "AP", "LR")
    model = IWOAIOAIUnet2DNormalized((H, Wd, 1), w, force_weights=True)
    out = model.generate_mask(mv)
    assert list(out) == ["fc", "tc", "pc", "men"]
    xw = uo.whiten_volume(sag.astype(np.float64)).astype(np.float32)
    ref = uo.forward(w, np.transpose(xw, (2, 0, 1)), dtype="float64") > 0  # (S, H, W, 4)
    for i, k in enumerate(out):
        m = out[k]
        assert isinstance(m, MedicalVolume) and m.dtype == np.uint8 and m.shape == mv.shape
        assert np.allclose(m.affine, mv.affine) and set(np.unique(m.volume)) <= {0, 1}
        assert (m.volume == np.transpose(ref[..., i], (1, 2, 0))).mean() > 0.9999
    # same data presented in another orientation -> same masks in that orientation
    mv_ax = mv.reformat(("AP", "LR", "SI"))
    out_ax = model(mv_ax)
    for k in out:
        assert out_ax[k].orientation == mv_ax.orientation
        assert np.array_equal(out_ax[k].reformat(mv.orientation).volume, out[k].volume)
    # un-normalised variant, registry, config wrapper, weights round trip through .npz
    path = tmp_path / "iwoai-2019-unet2d_fc-tc-pc-men_weights.npz"
    W.save_npz(path, w)
    m2 = get_model("iwoai-2019-t6", (H, Wd, 1), str(path))
    assert isinstance(m2, IWOAIOAIUnet2D)
    o2 = m2.generate_mask(mv)
    ref2 = uo.forward(w, np.transpose(sag, (2, 0, 1)), dtype="float64") > 0
    assert (o2["tc"].volume == np.transpose(ref2[..., 1], (1, 2, 0))).mean() > 0.9999
    with pytest.raises(ValueError):
        IWOAIOAIUnet2D((H, Wd, 1), str(tmp_path / "other.npz"))
    with pytest.raises(ValueError):
        IWOAIOAIUnet2D((H, Wd), w, force_weights=True)
    with pytest.raises(LookupError):
        get_model("nope", (H, Wd, 1), w)
    cfg = {"DOSMA_MODEL": "iwoai-2019-t6-normalized", "CATEGORIES": ["a", "b", "c", "d"], "WEIGHTS_FILE": w}
    m3 = model_from_config(cfg, input_shape=(H, Wd, 1))
    assert list(m3.generate_mask(mv)) == ["a", "b", "c", "d"]
    # single-class OAIUnet2D: one MedicalVolume, whiten(eps=1e-8)
    w1 = dict(w)
    w1["head_kernel"], w1["head_bias"] = w["head_kernel"][..., :1].copy(), w["head_bias"][:1].copy()
    m4 = OAIUnet2D((H, Wd, 1), w1)
    o4 = m4.generate_mask(mv)
    assert isinstance(o4, MedicalVolume) and o4.shape == mv.shape
    assert (o4.volume == out["fc"].volume).mean() > 0.999


@pytest.mark.gpu
def test_stanford_qdess_template(small_net):
    """stanford_qdess.py:158-201: 3D = RSS volume, 4D (..., 2) = the two echoes -> RSS first; whiten(eps=1e-8);
    dict pc / fc / tc / men."""
    from dosma_amd import MedicalVolume
    from dosma_amd.models import StanfordQDessUNet2D, get_model

    w, _ = small_net
    rng = np.random.default_rng(21)
    H, Wd, S = 64, 32, 5
    e = (rng.standard_normal((H, Wd, S, 2)) * 50 + 120).astype(np.float32)
    aff = np.array([[0, 0, 1.5, -40.0], [0, -0.4, 0, 60.0], [-0.4, 0, 0, 70.0], [0, 0, 0, 1.0]])
    model = StanfordQDessUNet2D((H, Wd, 1), w)
    rss = np.sqrt(np.sum(e.astype(np.float64) ** 2, axis=-1))
    out4 = model.generate_mask(MedicalVolume(e, aff))
    out3 = get_model("skm-tea-unet2d", (H, Wd, 1), w).generate_mask(MedicalVolume(rss, aff))
    assert list(out4) == ["pc", "fc", "tc", "men"] == list(out3)
    xw = uo.whiten_volume(rss, eps=1e-8).astype(np.float32)
    ref = uo.forward(w, np.transpose(xw, (2, 0, 1)), dtype="float64") > 0
    for i, k in enumerate(out4):
        assert out4[k].shape == (H, Wd, S) and out4[k].dtype == np.uint8
        assert np.array_equal(out4[k].volume, out3[k].volume)
        assert (out4[k].volume == np.transpose(ref[..., i], (1, 2, 0))).mean() > 0.9999
    with pytest.raises(ValueError):
        model.generate_mask(MedicalVolume(e[..., :1].repeat(3, -1), aff))


def _bf16_round(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw", [(32, 32, (16, 64)), (64, 32, (8, 32)), (32, 64, (24, 32)), (64, 64, (8, 96)),
                                         (32, 32, (40, 32))])
def test_register_weights_kernel_layers(cin, cout, hw):
    """Shapes that dispatch to conv_rw_kernel (unet_rw.hip: plain-bf16 mode, Cout 32 / 64, Cin 32 / 64, H % 8 == 0,
    W % 32 == 0), checked against the exact convolution of the bf16-rounded operands: what remains is fp32
    accumulation order and the bf16 rounding of the stored output (2^-9 relative)."""
    rng = np.random.default_rng(cin + 3 * cout + hw[0])
    B, (H, W) = 3, hw
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    xb, kb = _bf16_round(x), _bf16_round(k)
    for relu in (True, False):
        ref = torch_conv(xb, kb, b, relu=relu, transposed=False) * sc + sh
        y = L.conv2d_nhwc_host(x, k, b, scale=sc, shift=sh, relu=relu, precision="bf16")
        assert y.shape == ref.shape
        err = np.abs(y - ref)
        assert (err <= 2.0 ** -8 * np.abs(ref) + 2e-3).all(), err.max()
    # borders: every tile of the image touches the zero padding somewhere; an all-ones input makes a
    # misplaced halo pixel visible as a wrong count of taps
    ones = np.ones((1, H, W, cin), np.float32)
    k1 = np.full((3, 3, cin, cout), 2.0 ** -6, np.float32)
    y = L.conv2d_nhwc_host(ones, k1, np.zeros(cout, np.float32), relu=False, precision="bf16")
    cnt = np.full((H, W), 9.0)
    cnt[0, :] = cnt[-1, :] = 6.0
    cnt[:, 0] = cnt[:, -1] = 6.0
    cnt[0, 0] = cnt[0, -1] = cnt[-1, 0] = cnt[-1, -1] = 4.0
    assert np.array_equal(y[0, :, :, 0], cnt * cin * 2.0 ** -6)
    assert np.array_equal(y[0, :, :, cout - 1], cnt * cin * 2.0 ** -6)


def test_first_and_last_block_fallback_routes():
    """QMRI_ENC0=0 / QMRI_MID0=0 / QMRI_OUT0=0 (read once per process) send the first encoder block and the last convolution + classifier through
    the general kernels (first-layer kernel + conv_s3_kernel + pooling kernel; conv_s3_kernel with the fused classifier) -- the
    route sizes without 8 x 32 tiles take.  Same logits as the dedicated kernels, same distance from the restatement."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from dosma_amd import _lib as L
from oracle import unet_oracle as uo
from tests.test_unet_gpu import weights_in_abi_order
w = uo.make_weights(seed=11, bn="realistic")
rng = np.random.default_rng(1)
vol = (rng.standard_normal((3, 64, 96)) * 90 + 250).astype(np.float32)
eng = L.Unet2dEngine(weights_in_abi_order(w), 64, 96, max_batch=3, precision="fp16x3")
logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
tr = eng.trace()
ref = uo.forward(w, uo.whiten_volume(vol.astype(np.float64)).astype(np.float32), dtype="float64")
print("ERR", float(np.abs(logits - ref).max()), "TRACE", ",".join(tr))
np.save(sys.argv[1], logits)
''' % root
    import tempfile

    out = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, env in (("dedicated", {}), ("general", {"QMRI_ENC0": "0", "QMRI_OUT0": "0", "QMRI_MID0": "0"})):
            e = dict(os.environ, **env)
            path = os.path.join(d, tag + ".npy")
            txt = subprocess.check_output([sys.executable, "-c", code, path], env=e, cwd=root).decode()
            line = [ln for ln in txt.splitlines() if ln.startswith("ERR")][-1]
            out[tag] = (float(line.split()[1]), line.split("TRACE", 1)[1], np.load(path))
    assert "down0:enc0" in out["dedicated"][1] and "up0.conv2:out0+head" in out["dedicated"][1] and "up0.conv1:mid0" in out["dedicated"][1]
    assert "down0.conv1:c1/split" in out["general"][1] and "up0.conv2:s3/2d/bn32+head" in out["general"][1]
    assert "up0.conv1:s3/2d/bn32" in out["general"][1]
    assert out["dedicated"][0] < 1e-3 and out["general"][0] < 1e-3
    assert np.abs(out["dedicated"][2] - out["general"][2]).max() < 2e-4


@pytest.mark.parametrize("slices", [32, 33, 48])
def test_enc0_two_group_schedule_at_several_tiles_per_block(slices):
    """enc0_kernel's two groups of waves take a block's tiles alternately and hand over at shared barriers: 64 x 64 slices are
    16 tiles each, so 32 / 33 / 48 slices give every block 2 / 2 or 3 (the groups end in different half-periods) / 3 tiles on
    the 256 CUs of an MI355X.  Against the general route (QMRI_ENC0=0, read once per process: first-layer kernel +
    conv_s3_kernel + pooling kernel) on the same volume (/root/reference/dosma/models/oaiunet2d.py:213-243)."""
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from dosma_amd import _lib as L
from dosma_amd.models import weights as W
S = int(sys.argv[2])
eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), 64, 64, max_batch=S, precision="fp16x3")
vol = (np.random.default_rng(3).standard_normal((S, 64, 64)) * 2 + 0.5).astype(np.float32)
logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
print("TRACE", ",".join(eng.trace()))
np.save(sys.argv[1], logits)
''' % root
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, env in (("enc0", {}), ("general", {"QMRI_ENC0": "0"})):
            path = os.path.join(d, tag + ".npy")
            txt = subprocess.check_output([sys.executable, "-c", code, path, str(slices)], env=dict(os.environ, **env), cwd=root).decode()
            out[tag] = ([ln for ln in txt.splitlines() if ln.startswith("TRACE")][-1], np.load(path))
    assert "down0:enc0" in out["enc0"][0] and "down0.conv1:c1/split" in out["general"][0]
    assert np.abs(out["enc0"][1] - out["general"][1]).max() < 1e-4, np.abs(out["enc0"][1] - out["general"][1]).max()


@pytest.mark.parametrize("hw", [(32, 32), (160, 64), (224, 32), (96, 32)])
def test_dedicated_top_level_kernels_on_small_and_ragged_sizes(small_net, hw):
    """enc0 / mid0 / out0 (unet_enc0.hip) where their tiling is at its edges: one tile per slice (32 x 32), a single tile
    column (W = 32), heights that are not a multiple of out0's 12-row tiles (the last tile row is partly outside the
    slice: 160 = 13 x 12 + 4, 224 = 18 x 12 + 8)."""
    w, tensors = small_net
    H, W = hw
    rng = np.random.default_rng(H * 1000 + W)
    vol = (rng.standard_normal((3, H, W)) * 70 + 150).astype(np.float32)
    eng = L.Unet2dEngine(tensors, H, W, max_batch=3, precision="fp16x3")
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    ref = uo.forward(w, uo.whiten_volume(vol.astype(np.float64)).astype(np.float32), dtype="float64")
    assert np.abs(logits - ref).max() < 1e-3, np.abs(logits - ref).max()
    assert np.array_equal(mask, (logits > 0).astype(np.uint8))
    tr = eng.trace()
    assert "down0:enc0" in tr and "up0.conv1:mid0" in tr and "up0.conv2:out0+head" in tr, tr
    eng.close()


# ---- a14: the real generate_mask (fused GPU route) against the reference's OWN generate_mask (golden g9) ----
def passthrough_weights(a, b):
    """Keras-layout weights of the 6-level network that make it the per-pixel map logit_c = a_c * x + b_c -- the stand-in
    ``predict`` golden g9 was made with: the first block turns x into (relu(x), relu(-x)) on two channels, the skip
    connection carries them to the last block (every deeper layer is zero), the classifier recombines them.  BatchNorm
    with moving_variance = 1 - eps is the identity."""
    from dosma_amd.models import weights as W

    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    w = {}
    for name, shp in W.expected_shapes(W.NF, len(a)).items():
        if name.endswith("_gamma"):
            w[name] = np.ones(shp, np.float32)
        elif name.endswith("_var"):
            w[name] = np.full(shp, 1.0 - 1e-3, np.float32)
        else:
            w[name] = np.zeros(shp, np.float32)
    w["down0_conv1_kernel"][1, 1, 0, 0], w["down0_conv1_kernel"][1, 1, 0, 1] = 1.0, -1.0
    for name, off in (("down0_conv2_kernel", 0), ("up0_conv1_kernel", 32), ("up0_conv2_kernel", 0)):
        w[name][1, 1, off + 0, 0] = w[name][1, 1, off + 1, 1] = 1.0   # up0.conv1 reads [upsampled | skip]: skip = 32 ..
    w["head_kernel"][0, 0, 0, :], w["head_kernel"][0, 0, 1, :] = a, -a
    w["head_bias"][:] = b
    return w


def test_generate_mask_vs_reference_golden(golden):
    """The masks the product's ``generate_mask`` returns -- upload, reformat, whitening, network, threshold, class planes,
    reformat back, all through libqmri_hip.so -- equal the ones the REFERENCE's generate_mask returned for the same
    volumes (four orientations, anisotropic offset affine, every template, a non-default sigmoid threshold, the
    dual-echo input of the Stanford template), with the network replaced on both sides by the same per-pixel map
    (oaiunet2d.py:140-175, 291-320, 344-345; stanford_qdess.py:158-205).  Pixels whose logit is within 2e-3 of the
    cut are exempt (the fixture marks them): the two sides evaluate a_c * x + b_c in different arithmetic."""
    from dosma_amd import MedicalVolume
    from dosma_amd.models.oaiunet2d import IWOAIOAIUnet2D, IWOAIOAIUnet2DNormalized, OAIUnet2D
    from dosma_amd.models.stanford_qdess import StanfordQDessUNet2D

    z = golden("g9_generate_mask.npz")
    a4, b4 = z["a4"], z["b4"]
    templates = {
        "iwoai": (IWOAIOAIUnet2D, a4 / 300.0, b4 - 0.8, None),
        "iwoai_norm": (IWOAIOAIUnet2DNormalized, a4, b4, None),
        "oai": (OAIUnet2D, a4[:1], b4[:1], None),
        "stanford": (StanfordQDessUNet2D, a4, b4, None),
        "stanford_thr": (StanfordQDessUNet2D, a4, b4, 0.7),
    }
    H, W, _ = z["sag_vol"].shape
    checked = exempt = 0
    for tname, (cls, a, b, thr) in templates.items():
        model = cls((H, W, 1), passthrough_weights(a, b), force_weights=True)
        if thr is not None:
            model.sigmoid_threshold = thr
        cases = [(str(o), MedicalVolume(z[f"{o}_vol"], z[f"{o}_affine"]), f"{o}_{tname}") for o in z["orient_names"]]
        if tname == "stanford":
            cases.append(("dual", MedicalVolume(z["dual_vol"], z["sag_aff"]), "dual"))
        for oname, vol, tag in cases:
            out = model.generate_mask(vol)
            keys = [str(k) for k in z[f"{tag}_keys"]]
            items = list(out.items()) if keys else [("", out)]
            if keys:
                assert list(out.keys()) == keys, tag
            near = z[f"{tag}_near"]                                   # (S, H, W, C), sagittal frame
            for i, (k, m) in enumerate(items):
                assert m.volume.dtype == np.uint8 and m.volume.shape == vol.shape[:3], (tag, k)
                assert m.orientation == vol.orientation, (tag, k)
                if oname != "dual":
                    assert np.allclose(m.affine, z[f"{tag}_affine_{k}"], rtol=0, atol=1e-12), (tag, k)
                # compare in the sagittal frame, where the fixture's near-threshold map lives
                got = m.reformat(("SI", "AP", "LR")).volume
                ref = MedicalVolume(z[f"{tag}_mask_{k}"], m.affine).reformat(("SI", "AP", "LR")).volume
                ok = np.transpose(~near[..., i], (1, 2, 0))
                assert np.array_equal(got[ok], ref[ok]), (tag, k, int((got[ok] != ref[ok]).sum()))
                checked += int(ok.sum())
                exempt += int((~ok).sum())
    assert exempt < 5e-3 * checked, (exempt, checked)   # (the 2e-3 band around the cut holds ~0.1 % of the pixels)


# ---- the fp16 range of the split layout: raw intensities and large feature maps are rescaled, never clamped ----
def _shift_of(eng):
    tok = [t for t in eng.trace() if t.startswith("act_shift:")]
    return int(tok[-1].split(":")[1]) if tok else None


def test_raw_intensity_input_runs_in_range(small_net):
    """IWOAIOAIUnet2D feeds the network RAW intensities (oaiunet2d.py:322-323): 16-bit MRI values and the feature maps they
    produce are far outside the fp16 range of the hi + lo activation layout.  The engine runs the whole network scaled by a
    power of two (exact) instead of clamping: the logits agree with the fp64 restatement to fp32 rounding of their own
    magnitude, here and on the ragged-size route (general kernels)."""
    w, tensors = small_net
    rng = np.random.default_rng(21)
    for (S, H, W) in ((3, 64, 96), (2, 72, 144)):       # conv_s3 / dedicated kernels; the odd-size route (c1_split, general kernels, stride-3)
        vol = rng.uniform(1e3, 6.5e4, (S, H, W)).astype(np.float32)
        vol[:, : H // 4] = rng.integers(0, 65536, (S, H // 4, W)).astype(np.float32)   # the whole uint16 range, 65535 included
        eng = L.Unet2dEngine(tensors, H, W, max_batch=2, precision="fp16x3")
        logits, mask = eng.forward_host(vol, whiten=False)
        shift = _shift_of(eng)
        assert shift is not None and shift >= 9, eng.trace()       # max |x| = 65535 -> below 128
        ref = uo.forward(w, vol, dtype="float64")
        scale = np.abs(ref).max()
        assert scale > 1e3                                         # (feature maps well beyond 65504 without the rescaling)
        assert np.abs(logits - ref).max() < 1e-5 * scale + 1e-3, (np.abs(logits - ref).max(), scale)   # (1e-3 on unit-scale logits)
        assert (mask == (ref > 0)).mean() > 0.9999
        # the same engine on whitened data afterwards: back to exponent 0, the usual 1e-3
        lw, _ = eng.forward_host(vol, whiten=True)
        assert _shift_of(eng) == 0
        xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
        assert np.abs(lw - uo.forward(w, xw, dtype="float64")).max() < 1e-3
        eng.close()


def test_saturating_feature_maps_trigger_a_rescaled_repeat(small_net):
    """Unit-scale (whitened) input, but a first layer that amplifies by 2^18: the input gives no hint, the feature maps
    leave the fp16 range, the kernels raise the saturation flag and the forward is repeated at a higher exponent -- the
    result is the restatement's, and the model remembers the exponent for its next volume."""
    w, _ = small_net
    w = {k: np.array(v) for k, v in w.items()}
    w["down0_conv1_kernel"] = w["down0_conv1_kernel"] * np.float32(2.0 ** 18)
    w["down0_conv2_kernel"] = w["down0_conv2_kernel"] * np.float32(2.0 ** -18)   # the rest of the network sees usual magnitudes
    tensors = weights_in_abi_order(w)
    rng = np.random.default_rng(22)
    S, H, W = 2, 64, 64
    vol = (rng.standard_normal((S, H, W)) * 90 + 250).astype(np.float32)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
    ref = uo.forward(w, xw, dtype="float64")
    for kw in ({}, {"QMRI_ENC0": "0"}):      # the fused first block and the three-kernel route
        eng = L.Unet2dEngine(tensors, H, W, max_batch=2, precision="fp16x3")
        logits, _ = eng.forward_host(vol, whiten=True)
        first = _shift_of(eng)
        assert first is not None and first >= 6, eng.trace()
        assert np.abs(logits - ref).max() < 1e-3, np.abs(logits - ref).max()
        logits2, _ = eng.forward_host(vol, whiten=True)        # second volume: starts at the remembered exponent
        assert _shift_of(eng) == first and np.array_equal(logits, logits2)
        eng.close()
        break  # (QMRI_ENC0 is read once per process: the second route is covered by the ragged size above)
