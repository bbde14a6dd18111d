"""CPU restatement of the reference's 2D U-Net forward pass -- TEST INFRASTRUCTURE ONLY.

**PARITY UNPINNED.**  The reference builds this network in Keras/TensorFlow
(``dosma/models/oaiunet2d.py:197-289``) and loads ``.h5`` weights; neither Keras/TF, the weights, nor
the ``unittest-data/`` masks its tests compare with (``tests/models/test_oaiunet2d.py:17-41, 109-152``)
exist in this image or under /root/reference, so this restatement cannot be checked against the
reference's own outputs here.  It follows the graph in the reference file line by line and the
documented TF/Keras layer semantics:

* ``Conv2D(C, (3,3), padding="same", activation="relu")`` -- NHWC cross-correlation, zero pad 1, kernel
  layout ``(kh, kw, Cin, Cout)`` (oaiunet2d.py:213-226, 266-279);
* ``BatchNormalization(axis=-1, momentum=0.95, epsilon=0.001)`` in inference mode, AFTER the second ReLU
  of each block: ``gamma * (x - mean) / sqrt(var + 1e-3) + beta`` (:228, :281); ``Dropout(0)`` = identity;
* ``MaxPooling2D((2,2))`` where the height is even, ``MaxPooling2D((3,3))`` where it is odd (:234-243; Keras' default
  stride = pool size, padding "valid"), and on the way up ``Conv2DTranspose(..., strides=(3,3))`` for those levels
  (:250-261): out[3o + k] = in[o] w[k];
* ``Conv2DTranspose(C, (3,3), padding="same", strides=(2,2))`` (:259-261): the gradient of a SAME
  stride-2 3x3 convolution, i.e. ``out[i] = sum_{o,k: 2o+k=i} in[o] w[k]`` for i in [0, 2H), kernel layout
  ``(kh, kw, Cout, Cin)``; equals ``torch.conv_transpose2d(stride=2, padding=0)`` cropped to ``[:2H, :2W]``;
* ``Concatenate(axis=3)([deconv, skip])`` (:257-264), skip = the block's post-BN output (:228-231);
* head ``Conv2D(4, (1,1), activation="sigmoid")`` (:285); mask = ``sigmoid > 0.5`` (:306) == ``logit > 0``;
* ``whiten_volume``: ``(x - mean(x)) / (std(x) + eps)`` over the whole volume (seg_model.py:114-127).

Used by tests/ as the checker for the HIP kernels (self-consistency: same seeded weights and inputs),
and by bench.py nowhere.
"""
import numpy as np

NF = (32, 64, 128, 256, 512, 1024)
BN_EPS = 1e-3
CLASSES = ("fc", "tc", "pc", "men")  # oaiunet2d.py:312


def layer_names(nf=NF):
    """Layer order = Keras graph-creation order (what ``load_weights`` of an .h5 iterates over)."""
    names = []
    for d in range(len(nf)):
        names += [f"down{d}_conv1", f"down{d}_conv2", f"down{d}_bn"]
    for d in range(len(nf) - 2, -1, -1):
        names += [f"up{d}_deconv", f"up{d}_conv1", f"up{d}_conv2", f"up{d}_bn"]
    names.append("head")
    return names


def make_weights(seed=0, nf=NF, n_classes=4, dtype=np.float32, bn="he"):
    """Seeded random weights in Keras layouts (He-normal kernels so activations stay O(1)).

    ``bn="he"``: BatchNormalization statistics near the identity (variance 0.5-1.5, mean 0.2-0.6).
    ``bn="realistic"``: moving statistics spread like a trained network's -- every block's second convolution
    gets a per-output-channel gain s in [0.1, 10] (kernel column and bias scaled; ReLU is positively homogeneous,
    so the pre-BN activation scales by s) and the BatchNormalization that follows carries the matching moving
    statistics: variance s^2 * U(0.5, 1.5) in [5e-3, 1.5e2], mean s * U(-0.6, 0.6) of either sign, gamma in
    [0.2, 2].  The block output stays O(1) (as after training) while the folded per-channel scale
    gamma / sqrt(var + eps) spans 0.02 ... 25 and the 1e-3 epsilon matters for the small-variance channels."""
    rng = np.random.default_rng(seed)
    w = {}
    bn_kind = bn
    if bn_kind not in ("he", "realistic"):
        raise ValueError(bn_kind)

    def conv(name, kh, cin, cout):
        w[f"{name}_kernel"] = (rng.standard_normal((kh, kh, cin, cout)) * np.sqrt(2.0 / (kh * kh * cin))).astype(dtype)
        w[f"{name}_bias"] = (0.05 * rng.standard_normal(cout)).astype(dtype)

    def bn_realistic(name, c):
        s = 10.0 ** rng.uniform(-1.0, 1.0, c)
        conv2 = name.replace("_bn", "_conv2")
        w[f"{conv2}_kernel"] = (w[f"{conv2}_kernel"] * s.astype(dtype)).astype(dtype)  # (kh, kw, Cin, Cout): per Cout
        w[f"{conv2}_bias"] = (w[f"{conv2}_bias"] * s.astype(dtype)).astype(dtype)
        w[f"{name}_var"] = (s * s * rng.uniform(0.5, 1.5, c)).astype(dtype)
        w[f"{name}_mean"] = (s * rng.uniform(-0.6, 0.6, c)).astype(dtype)
        w[f"{name}_gamma"] = rng.uniform(0.2, 2.0, c).astype(dtype)
        w[f"{name}_beta"] = (0.3 * rng.standard_normal(c)).astype(dtype)

    def bn(name, c):
        if bn_kind == "realistic":
            return bn_realistic(name, c)
        w[f"{name}_gamma"] = rng.uniform(0.7, 1.3, c).astype(dtype)
        w[f"{name}_beta"] = (0.1 * rng.standard_normal(c)).astype(dtype)
        w[f"{name}_mean"] = rng.uniform(0.2, 0.6, c).astype(dtype)
        w[f"{name}_var"] = rng.uniform(0.5, 1.5, c).astype(dtype)

    cin = 1
    for d, c in enumerate(nf):
        conv(f"down{d}_conv1", 3, cin, c)
        conv(f"down{d}_conv2", 3, c, c)
        bn(f"down{d}_bn", c)
        cin = c
    for d in range(len(nf) - 2, -1, -1):
        c = nf[d]
        # Conv2DTranspose kernel: (kh, kw, Cout, Cin)
        w[f"up{d}_deconv_kernel"] = (rng.standard_normal((3, 3, c, nf[d + 1])) * np.sqrt(1.0 / (2.25 * nf[d + 1]))).astype(dtype)
        w[f"up{d}_deconv_bias"] = (0.05 * rng.standard_normal(c)).astype(dtype)
        conv(f"up{d}_conv1", 3, 2 * c, c)
        conv(f"up{d}_conv2", 3, c, c)
        bn(f"up{d}_bn", c)
    w["head_kernel"] = (rng.standard_normal((1, 1, nf[0], n_classes)) * np.sqrt(1.0 / nf[0])).astype(dtype)
    w["head_bias"] = (0.05 * rng.standard_normal(n_classes)).astype(dtype)
    return w


def level_factors(H, W, depth=len(NF)):
    """Pooling factor between level d and d + 1 (oaiunet2d.py:234-243): 2 where the height is even, 3 where it is odd --
    applied to both axes; the graph only builds (Concatenate, :257-264) if the factor divides both."""
    out = []
    for d in range(depth - 1):
        f = 2 if H % 2 == 0 else 3
        if H % f or W % f:
            raise ValueError(f"the graph does not build for this size: level {d} is {H} x {W}, pooled by {f}")
        out.append(f)
        H, W = H // f, W // f
    return out


def whiten_volume(x, eps=0.0):
    """seg_model.py:114-127 (numpy semantics: float32 input -> float32 pairwise mean/std)."""
    x = np.asarray(x)
    if x.ndim != 3:
        raise ValueError(f"Input has {x.ndim} dimensions. Expected 3")
    return (x - np.mean(x)) / (np.std(x) + eps)


def _cpu_budget():
    """Cores this process may actually use: the affinity mask, cut by the cgroup's cpu.max quota where there is one."""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def forward(w, x, nf=NF, dtype="float32", return_features=False):
    """Logits (S, H, W, n_classes) of the network for slices ``x`` (S, H, W), torch CPU.

    ``dtype`` "float32" mirrors TF's fp32 inference; "float64" is the numerical ground truth used to
    judge what precision the GPU path needs.
    """
    import torch
    import torch.nn.functional as F

    # torch's CPU convolutions start one thread per VISIBLE core; under a cgroup CPU quota far below that (the GPU boxes: 16 of
    # several hundred) the spinning workers can take minutes for a forward that needs seconds -- cap them at the quota
    torch.set_num_threads(max(1, min(torch.get_num_threads(), _cpu_budget())))
    tdt = torch.float32 if dtype == "float32" else torch.float64
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(tdt)  # noqa: E731
    feats = {}

    def conv3(h, name, relu=True):
        k = t(w[f"{name}_kernel"]).permute(3, 2, 0, 1)  # (kh,kw,Cin,Cout) -> (Cout,Cin,kh,kw)
        h = F.conv2d(h, k, t(w[f"{name}_bias"]), padding=1)
        return F.relu(h) if relu else h

    def bn(h, name):
        scale = t(w[f"{name}_gamma"]) / torch.sqrt(t(w[f"{name}_var"]) + BN_EPS)
        shift = t(w[f"{name}_beta"]) - t(w[f"{name}_mean"]) * scale
        return h * scale[None, :, None, None] + shift[None, :, None, None]

    def deconv(h, name, stride):
        k = t(w[f"{name}_kernel"]).permute(3, 2, 0, 1)  # (kh,kw,Cout,Cin) -> (Cin,Cout,kh,kw)
        out = F.conv_transpose2d(h, k, t(w[f"{name}_bias"]), stride=stride, padding=0)
        return out[:, :, : stride * h.shape[2], : stride * h.shape[3]]

    h = t(x)[:, None, :, :]
    factors = level_factors(h.shape[2], h.shape[3], len(nf))
    skips = []
    with torch.no_grad():
        for d in range(len(nf)):
            h = conv3(h, f"down{d}_conv1")
            h = conv3(h, f"down{d}_conv2")
            h = bn(h, f"down{d}_bn")
            skips.append(h)
            feats[f"down{d}"] = h
            if d < len(nf) - 1:
                h = F.max_pool2d(h, factors[d])
        for d in range(len(nf) - 2, -1, -1):
            up = deconv(h, f"up{d}_deconv", factors[d])
            feats[f"up{d}_deconv"] = up
            h = torch.cat([up, skips[d]], dim=1)
            h = conv3(h, f"up{d}_conv1")
            h = conv3(h, f"up{d}_conv2")
            h = bn(h, f"up{d}_bn")
            feats[f"up{d}"] = h
        k = t(w["head_kernel"]).permute(3, 2, 0, 1)
        logits = F.conv2d(h, k, t(w["head_bias"]))
    out = logits.permute(0, 2, 3, 1).contiguous().numpy()
    if return_features:
        return out, {k_: v.permute(0, 2, 3, 1).contiguous().numpy() for k_, v in feats.items()}
    return out


def deconv_naive(x, kernel, bias):
    """The scatter definition of Conv2DTranspose(3x3, stride 2, SAME) for tiny arrays (pure numpy):
    out[b, 2o+kh, 2p+kw, co] += x[b, o, p, ci] * kernel[kh, kw, co, ci], cropped to (2H, 2W)."""
    B, H, W, Ci = x.shape
    Co = kernel.shape[2]
    out = np.zeros((B, 2 * H + 1, 2 * W + 1, Co), dtype=np.float64)
    for kh in range(3):
        for kw in range(3):
            contrib = np.einsum("bhwi,oi->bhwo", x.astype(np.float64), kernel[kh, kw].astype(np.float64))
            out[:, kh:kh + 2 * H:2, kw:kw + 2 * W:2, :] += contrib
    return out[:, : 2 * H, : 2 * W, :] + bias.astype(np.float64)
