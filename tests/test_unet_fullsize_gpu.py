"""Full-size parity of the 2D U-Net on the GPU: BASELINE.json configs[3] (384 x 384 slices) and the 512 x 512 slices of
configs[4], against the fp64 run of the restatement (oracle/unet_oracle.py; PARITY WITH KERAS UNPINNED, see its header).

What the reference pins at this size with data that is absent here: /root/reference/tests/models/test_oaiunet2d.py:19-41,
109-152 (exact masks of a 384 x 384 x 160 volume).  Here: seeded weights in Keras layouts -- with BatchNormalization
moving statistics spread like a trained network's (variance 5e-3 .. 1.5e2, means of either sign), not only He / identity
-- and the north_star tolerance on the logits (1e-3 abs) in the parity mode, for both activation-buffer sizes the bench
uses (max_batch 16 and 160), plus an assertion on WHICH kernels ran every layer (so that the 384-only dispatch -- 8 x 32
tiles at 384 / 192 / 96, the flattened tiling at 48 / 24 / 12, fused pool and head -- is what is tested)."""
import os

import numpy as np
import pytest

from dosma_amd import _lib as L
from oracle import unet_oracle as uo
from test_unet_gpu import weights_in_abi_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    w = uo.make_weights(seed=11, bn="realistic")
    return w, weights_in_abi_order(w)


def _volume(S, H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    blob = np.exp(-(((yy - H / 2) / (H / 4)) ** 2 + ((xx - W / 2) / (W / 3)) ** 2))  # structure, not only noise
    return (rng.standard_normal((S, H, W)) * 60 + 250 * blob[None] + 80).astype(np.float32)


@pytest.fixture(scope="module")
def ref384(net):
    w, _ = net
    vol = _volume(4, 384, 384, 384)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
    return vol, uo.forward(w, xw, dtype="float64")


def _expect_families(trace, H, W):
    """The first encoder block as one kernel (enc0) where 8 x 32 tiles cover the slice; conv_s3_kernel wherever it tiles the
    level (convolutions and transposed convolutions), the general kernel elsewhere."""
    by = dict(t.split(":", 1) for t in trace if ":" in t and not t.startswith(("pool", "head")))
    fused0 = H % 8 == 0 and W % 32 == 0
    for lvl in range(6):
        wl = W >> lvl
        fam = "s3/flat" if wl % 32 and wl + 2 <= 50 else "s3/2d"  # (W % 32 != 0 and wider: image tiles with a ragged last column)
        for name in (([f"down{lvl}.conv2"] if lvl or not fused0 else []) + ([f"down{lvl}.conv1"] if lvl else [])
                     + ([f"up{lvl}.conv1", f"up{lvl}.conv2"] if lvl < 5 and (lvl or not fused0) else [])):
            assert by[name].startswith(fam), (name, by[name], fam)
        if lvl < 5:  # the transposed convolution tiles the INPUT grid (level lvl + 1)
            win = W >> (lvl + 1)
            famd = "s3/flat" if win % 32 and win + 2 <= 50 else "s3/2d"
            assert by[f"up{lvl}.deconv"].startswith(famd), (lvl, by[f"up{lvl}.deconv"], famd)
    assert (by["down0"] == "enc0") if fused0 else (by["down0.conv1"] == "c1/split")
    assert by["up0.conv2"] == ("out0+head" if fused0 else by["up0.conv2"]) and by["up0.conv2"].endswith("+head")  # never goes to HBM
    assert by["up0.conv1"] == ("mid0" if fused0 else by["up0.conv1"])
    assert by["down1.conv2"].endswith("+pool") and by["down2.conv2"].endswith("+pool")


@pytest.mark.parametrize("max_batch", [16, 160])
def test_384_logits_parity_mode(net, ref384, max_batch):
    w, tensors = net
    vol, ref = ref384
    eng = L.Unet2dEngine(tensors, 384, 384, max_batch=max_batch, precision="fp16x3")
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    err = np.abs(logits - ref)
    assert err.max() < 1e-3, f"max |dlogit| {err.max():.3e} (logits span {np.abs(ref).max():.1f})"
    assert np.array_equal(mask, (logits > 0).astype(np.uint8))
    assert (mask == (ref > 0)).mean() > 0.99999
    tr = eng.trace()
    _expect_families(tr, 384, 384)
    fams = {t.split(":", 1)[1].split("+")[0] for t in tr if t.startswith(("down", "up")) and "conv" in t and "deconv" not in t}
    # the plain 3 x 3 convolutions run on conv_s3_kernel ("bn64" / "bn128") or conv_c4_kernel ("c4x64" / "c4x128")
    assert {"mid0", "out0"} <= fams and fams & {"s3/2d/bn64", "s3/2d/c4x64"} and fams & {"s3/2d/bn128", "s3/2d/c4x128"} \
        and fams & {"s3/flat/bn128", "s3/flat/c4x128"}, fams
    if os.environ.get("QMRI_C4", "1") == "1":
        # the product's dispatch (conv_s3_takes_c4, by layer shape only): every plain 3 x 3 convolution below the top level is
        # conv_c4_kernel's -- 64-channel blocks on 24-row image tiles, 128-channel blocks on image tiles and flattened levels --
        # and conv_s3_kernel was left with the transposed convolutions (round 4)
        assert {"s3/2d/c4x64", "s3/2d/c4x128", "s3/flat/c4x128"} <= fams, fams
        assert not fams & {"s3/2d/bn64", "s3/2d/bn128", "s3/flat/bn128"}, fams
        # ... and, since round 5, only where QMRI_D4 = 0 asks for it: the transposed convolutions are deconv_d4_kernel's
        want = "bn32" if os.environ.get("QMRI_D4", "1") == "0" else "d4x32"
        assert all(t.split(":", 1)[1].endswith(want) for t in tr if "deconv" in t), tr
    eng.close()


def test_384_logits_plain_bf16_mode(net, ref384):
    """The throughput mode at the benchmark size: bf16-level logits, masks equal except next to the decision boundary."""
    w, tensors = net
    vol, ref = ref384
    eng = L.Unet2dEngine(tensors, 384, 384, max_batch=16, precision="bf16")
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    span = np.abs(ref).max()
    err = np.abs(logits - ref)
    assert err.max() < 0.03 * span + 0.25, (err.max(), span)  # bf16 operands: ~2^-9 per product through 26 layers
    flips = mask != (ref > 0)
    assert flips.mean() < 5e-3
    assert np.abs(ref[flips]).max() < 0.03 * span + 0.25 if flips.any() else True
    eng.close()


def test_512_logits_parity_mode(net):
    """BASELINE configs[4] segments 512 x 512 slices: every level is a multiple of 32 wide down to 32, then 16 (flattened)."""
    w, tensors = net
    vol = _volume(2, 512, 512, 512)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
    ref = uo.forward(w, xw, dtype="float64")
    eng = L.Unet2dEngine(tensors, 512, 512, max_batch=32, precision="fp16x3")
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    assert np.abs(logits - ref).max() < 1e-3, np.abs(logits - ref).max()
    assert (mask == (ref > 0)).mean() > 0.99999
    _expect_families(eng.trace(), 512, 512)
    eng.close()


# slices compared at the bench's batch size: first / last, both sides of an XCD work-range boundary of the 8 x 32-tile levels
# (160 slices over 8 XCDs = 20 per XCD: 19 | 20, 79 | 80), and one whose flattened-level tiles straddle tile boundaries
# (a tile of the flattened levels is 256 / 512 flat positions: no slice starts on one except slice 0)
_BATCH_SAMPLE = [0, 19, 20, 79, 80, 121, 159]


def _bench_batch_parity(net, H, W, seed):
    """ONE pass of 160 slices through max_batch = 160 -- the shape bench.py times (`unet2d` at 384 x 384, `cfg5` at
    512 x 512; /root/reference/dosma/models/oaiunet2d.py:291-320 predicts the whole volume, reference test
    tests/models/test_oaiunet2d.py:19-41) -- compared on sampled slices with the restatement run on THOSE slices of the volume
    whitened with the whole volume's statistics (seg_model.py:114-127)."""
    w, tensors = net
    S = 160
    vol = _volume(S, H, W, seed)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)   # statistics of all 160 slices
    ref = uo.forward(w, xw[_BATCH_SAMPLE], dtype="float64")
    eng = L.Unet2dEngine(tensors, H, W, max_batch=S, precision="fp16x3")
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    assert logits.shape == (S, H, W, 4)
    err = np.abs(logits[_BATCH_SAMPLE] - ref)
    assert err.max() < 1e-3, f"max |dlogit| {err.max():.3e} per slice {err.reshape(len(_BATCH_SAMPLE), -1).max(1)}"
    band = np.abs(ref) < 1e-3                                            # the mask may differ only inside the tolerance band
    assert np.array_equal(mask[_BATCH_SAMPLE][~band], (ref > 0)[~band].astype(np.uint8))
    assert np.array_equal(mask, (logits > 0).astype(np.uint8))
    # every slice went through the network (none left at its buffer's previous content)
    assert np.isfinite(logits).all() and np.abs(logits).reshape(S, -1).max(1).min() > 0
    tr = eng.trace()
    _expect_families(tr, H, W)
    eng.close()
    return vol, logits, tr


def test_384_logits_at_bench_batch(net):
    """BASELINE configs[3] at the batch the bench times: 160 slices of 384 x 384 in one pass."""
    _bench_batch_parity(net, 384, 384, 3840)


def test_512_logits_at_cfg5_batch(net):
    """BASELINE configs[4]'s segmentation shape: 160 slices of 512 x 512 in one pass (bench.py `cfg5`)."""
    vol, logits, tr = _bench_batch_parity(net, 512, 512, 5120)
    by = dict(t.split(":", 1) for t in tr if ":" in t)
    if os.environ.get("QMRI_C4", "1") == "1":
        # which kernel a layer runs on is decided by its SHAPE: up1.conv1 (256 x 256, 128-channel concat input: 5.4 GB of
        # activations at 160 slices, beyond one 32-bit offset range) stays on conv_c4_kernel at this batch too -- its requests
        # address a work item's images through a descriptor base that moves with the item (round 4 fell back to
        # conv_s3_kernel<64> from 128 slices per pass up: ADVICE r4)
        assert by["up1.conv1"].startswith("s3/2d/c4x64") and by["up1.conv2"].startswith("s3/2d/c4x64"), by
    # ... and a slice's bits do not depend on the pass it travels in: the same volume in passes of 64 + 64 + 32 slices
    w, tensors = net
    eng = L.Unet2dEngine(tensors, 512, 512, max_batch=64, precision="fp16x3")
    split, _ = eng.forward_host(vol, whiten=True, eps=0.0)
    by64 = dict(t.split(":", 1) for t in eng.trace() if ":" in t)
    eng.close()
    assert np.array_equal(split, logits)
    assert by64["up1.conv1"] == by["up1.conv1"]


def test_512_forward_repeats_bit_for_bit(net):
    """Round 6's race: with conv_c4_kernel's barrier every THIRD k-step 2-12 % of the forwards of the 512 x 512 network came out with one
    wave's rows of one work item wrong (logits off by up to 0.5) -- step 0 of an item read its weights behind next_item's barrier, in the
    same barrier interval as the request of step 2 that rewrites their ring slot.  Found through 27 differing mask voxels in
    bench.py's cfg5 leg, not by this suite, whose repeat tests ran a handful of forwards.  Here: 80 device-resident forwards of a
    160-slice volume in passes of 32 (the configuration with the highest measured rate: >= 3.7 % per forward, i.e. >= 95 % to see the old
    cadence fail), every one equal to the first bit for bit.  (Every second step -- the default since -- measured 0 of 15 100.)"""
    import torch

    w, tensors = net
    dev = torch.device("cuda", 0)
    S, H = 160, 512
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn((S, H, H), device=dev, generator=gen) * 150 + 300
    logits = torch.empty((S, H, H, 4), device=dev)
    mask = torch.empty((S, H, H, 4), device=dev, dtype=torch.uint8)
    st = torch.cuda.current_stream(dev)
    eng = L.Unet2dEngine(tensors, H, H, max_batch=32, precision="fp16x3", device=0)
    first = None
    for rep in range(80):
        eng.forward_device(x.data_ptr(), S, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=st.cuda_stream)
        torch.cuda.synchronize()
        per_slice = logits.view(torch.int32).flatten(1).to(torch.int64).sum(1)  # (exact: integer sums of the logits' bit patterns)
        if first is None:
            first = per_slice.clone()
            assert torch.isfinite(logits).all()
        else:
            bad = (per_slice != first).nonzero().flatten().tolist()
            assert not bad, f"forward {rep}: slices {bad[:12]} differ from the first forward"
    eng.close()


def test_odd_level_widths_take_ragged_image_tiles(net):
    """224 x 224: 224 = 7 x 32 (whole image tiles), 112 and 56 are neither multiples of 32 nor <= 48 -- image tiles with a RAGGED
    last column tile on conv_c4_kernel / deconv_d4_kernel (round 5; the general kernel before that: 20-35 % of the forward on such
    sizes) --, 28 / 14 / 7 flattened: all three tilings in one network, same 1e-3 bar."""
    w, tensors = net
    vol = _volume(2, 224, 224, 224)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
    ref = uo.forward(w, xw, dtype="float64")
    eng = L.Unet2dEngine(tensors, 224, 224, max_batch=2, precision="fp16x3")
    logits, _ = eng.forward_host(vol, whiten=True, eps=0.0)
    assert np.abs(logits - ref).max() < 1e-3, np.abs(logits - ref).max()
    tr = eng.trace()
    _expect_families(tr, 224, 224)
    assert any(t.startswith("down1.conv2:s3/2d/c4x64+pool") for t in tr) and any(t.startswith("down2.conv1:s3/2d/c4x128") for t in tr)
    assert any(t.startswith("up1.deconv:s3/2d/d4x32") for t in tr) and any(t.startswith("down3.conv1:s3/flat") for t in tr)
    assert not any("igemm" in t for t in tr)
    eng.close()


@pytest.mark.parametrize("hw", [(160, 160), (320, 320), (448, 448), (96, 288), (64, 160), (352, 480), (32, 512), (512, 32), (384, 512)])
def test_parity_over_slice_shapes(net, hw):
    """Shapes between the pinned ones: every mix of tilings the dispatch can choose (image tiles where a level is a multiple of 32
    wide, the flattened tiling up to 48, the general kernel elsewhere; non-square slices; one-tile-wide levels).  Same bar: 1e-3
    abs on the logits, masks equal outside the tolerance band.  scripts/unet_size_sweep.py runs all 27 shapes
    (profiles/r05_unet_size_sweep.txt)."""
    w, tensors = net
    H, W = hw
    vol = _volume(1, H, W, H * 1000 + W)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
    ref = uo.forward(w, xw, dtype="float64")
    eng = L.Unet2dEngine(tensors, H, W, max_batch=3, precision="fp16x3")
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    eng.close()
    assert np.abs(logits - ref).max() < 1e-3, np.abs(logits - ref).max()
    assert not ((mask.astype(bool) != (ref > 0)) & (np.abs(ref) >= 1e-3)).any()


def test_forward_is_bitwise_repeatable(net):
    """No atomics and no data-dependent scheduling in the convolution kernels: the same volume through the same engine gives the
    same bits, whatever the timing of the LDS-DMA requests (a request landing after its counted wait would show up here first)."""
    w, tensors = net
    vol = _volume(24, 384, 384, 7)
    eng = L.Unet2dEngine(tensors, 384, 384, max_batch=24, precision="fp16x3")
    first, _ = eng.forward_host(vol, whiten=True, eps=0.0)
    for _ in range(4):
        again, _ = eng.forward_host(vol, whiten=True, eps=0.0)
        assert np.array_equal(first, again)
    eng.close()
    eng = L.Unet2dEngine(tensors, 384, 384, max_batch=5, precision="fp16x3")  # other batch boundaries, same per-slice bits
    split, _ = eng.forward_host(vol, whiten=True, eps=0.0)
    assert np.array_equal(first, split)
    eng.close()
