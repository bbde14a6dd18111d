#!/usr/bin/env python3
"""CPU simulation: which operand precision do the UNet's MFMA convolutions need for |dlogit| < 1e-3?

Every convolution of the graph (oracle/unet_oracle.py) is evaluated in float64 on operands that were first
rounded the way an MFMA operand format rounds them; what is left is the operand-format error alone (fp32
accumulation adds ~1e-6).  Formats: one / two (hi + lo) fp16 or bf16 parts per operand; "x3" = hi*hi + hi*lo +
lo*hi (the lo*lo term is dropped), "a1w2" = single-part activations times two-part weights (2 MFMAs).

    python scripts/unet_precision_sim.py [H W S]
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from oracle import unet_oracle as uo  # noqa: E402


def rnd(t, fmt):
    if fmt == "f64":
        return t
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[fmt]
    return t.to(torch.float32).to(dt).to(torch.float64)


def split(t, fmt):
    hi = rnd(t, fmt)
    lo = rnd(t.to(torch.float32).to(torch.float64) - hi, fmt)
    return hi, lo


def split_scaled(t, fmt):
    """hi / lo parts of 2^k t with max |2^k t| in [2^13, 2^14) (the engine's `weight_shift`), and 2^-k."""
    m = float(t.abs().max())
    k = 14 - (int(np.floor(np.log2(m))) + 1) if m > 0 else 0
    hi, lo = split(t * 2.0 ** k, fmt)
    return hi, lo, 2.0 ** -k


f32 = lambda t: t.to(torch.float32).to(torch.float64)  # noqa: E731  (a value the kernel would hold in an fp32 register)

# Winograd F(2, 3) (Lavin & Gray 2016): 4 multiplications per 2 outputs of a 3-tap filter
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def wino_conv(x, k, fmt, two_d):
    """3x3 SAME convolution of x (B, C, H, W) with k (Co, C, 3, 3) through Winograd F(2x2, 3x3) (two_d) or F(2, 3) along
    x only: input / weight / output transforms in fp32, the TRANSFORMED operands split into fp16 hi + lo parts (weights
    pre-scaled by a power of two), product = hi*hi + hi*lo + lo*hi, accumulation exact (fp32 adds ~1e-6)."""
    B, C, H, W = x.shape
    Co = k.shape[0]
    xp = F.pad(x, (1, 1, 1, 1 if H % 2 == 0 else 2))  # (an odd height: one more zero row, cropped below)
    if W % 2:
        xp = F.pad(xp, (0, 1, 0, 0))
    Hp, Wp = xp.shape[2] - 2, xp.shape[3] - 2      # even
    if two_d:
        d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # (B, C, ty, tx, 4, 4)
        V = f32(torch.einsum("ai,bctuij,dj->bctuad", BT, d, BT))    # B^T d B  (additions only: exact in fp32 to 1 ulp)
        U = f32(torch.einsum("ai,ocij,dj->ocad", G, k, G))          # G g G^T
        Vh, Vl = split(V, fmt)
        Uh, Ul, inv = split_scaled(U, fmt)
        mm = lambda a, b: torch.einsum("bctuad,ocad->botuad", a, b)  # noqa: E731
        M = f32((mm(Vh, Uh) + mm(Vh, Ul) + mm(Vl, Uh)) * inv)
        Y = torch.einsum("pa,botuad,qd->botupq", AT, M, AT)         # (B, Co, ty, tx, 2, 2)
        y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, Co, Hp, Wp)
    else:
        d = xp.unfold(3, 4, 2)                                       # (B, C, Hp + 2, tx, 4)
        V = f32(torch.einsum("ai,bcyti->bcyta", BT, d))
        U = f32(torch.einsum("ai,ocji->ocja", G, k))                 # (Co, C, ky, 4)
        Vh, Vl = split(V, fmt)
        Uh, Ul, inv = split_scaled(U, fmt)
        Vw = lambda v: v.unfold(2, 3, 1)                             # noqa: E731  (B, C, Hp, tx, 4, ky)
        mm = lambda a, b: torch.einsum("bcytaj,ocja->boyta", Vw(a), b)  # noqa: E731
        M = f32((mm(Vh, Uh) + mm(Vh, Ul) + mm(Vl, Uh)) * inv)
        Y = torch.einsum("pa,boyta->boytp", AT, M)                   # (B, Co, Hp, tx, 2)
        y = Y.reshape(B, Co, Hp, Wp)
    return y[:, :, :H, :W]


def make_op(mode):
    """mode -> f(conv_fn, x, w): conv_fn(x, w) is the exact (float64) linear operator."""
    if mode == "exact":
        return lambda cf, x, w: cf(x, w)
    fmt, kind = mode.split(":")
    if kind == "x3s":  # the engine's parity mode: x3 with the weights pre-scaled by a power of two
        def f(cf, x, w):
            xh, xl = split(x, fmt)
            wh, wl, inv = split_scaled(w, fmt)
            return (cf(xh, wh) + cf(xh, wl) + cf(xl, wh)) * inv
        return f
    if kind in ("wino2d", "wino1d"):  # 3x3 convolutions through Winograd, transposed convolutions as in x3s
        direct = make_op(fmt + ":x3s")

        def f(cf, x, w):
            if getattr(cf, "is_conv3", False):
                return wino_conv(x, w, fmt, kind == "wino2d")
            return direct(cf, x, w)
        return f
    if kind == "x1":
        return lambda cf, x, w: cf(rnd(x, fmt), rnd(w, fmt))
    if kind == "a1w2":
        def f(cf, x, w):
            wh, wl = split(w, fmt)
            xa = rnd(x, fmt)
            return cf(xa, wh) + cf(xa, wl)
        return f
    if kind == "a2w1":
        def f(cf, x, w):
            xh, xl = split(x, fmt)
            wa = rnd(w, fmt)
            return cf(xh, wa) + cf(xl, wa)
        return f
    if kind == "x3":
        def f(cf, x, w):
            xh, xl = split(x, fmt)
            wh, wl = split(w, fmt)
            return cf(xh, wh) + cf(xh, wl) + cf(xl, wh)
        return f
    raise ValueError(mode)


def forward(w, x, op, nf=uo.NF):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.float64)  # noqa: E731

    def conv3(h, name, relu=True, first=False):
        k = t(w[f"{name}_kernel"]).permute(3, 2, 0, 1)
        def cf(a, b):
            return F.conv2d(a, b, None, padding=1)
        cf.is_conv3 = True
        y = (cf(h, k) if first else op(cf, h, k)) + t(w[f"{name}_bias"])[None, :, None, None]
        return F.relu(y) if relu else y

    def bn(h, name):
        scale = t(w[f"{name}_gamma"]) / torch.sqrt(t(w[f"{name}_var"]) + uo.BN_EPS)
        shift = t(w[f"{name}_beta"]) - t(w[f"{name}_mean"]) * scale
        return h * scale[None, :, None, None] + shift[None, :, None, None]

    def deconv(h, name):
        k = t(w[f"{name}_kernel"]).permute(3, 2, 0, 1)
        cf = lambda a, b: F.conv_transpose2d(a, b, None, stride=2)[:, :, : 2 * a.shape[2], : 2 * a.shape[3]]  # noqa: E731
        return op(cf, h, k) + t(w[f"{name}_bias"])[None, :, None, None]

    h = t(x)[:, None]
    skips = []
    with torch.no_grad():
        for d in range(len(nf)):
            h = conv3(h, f"down{d}_conv1", first=(d == 0))  # the first layer (Cin = 1) is exact fp32 in the engine
            h = conv3(h, f"down{d}_conv2")
            h = bn(h, f"down{d}_bn")
            skips.append(h)
            if d < len(nf) - 1:
                h = F.max_pool2d(h, 2)
        for d in range(len(nf) - 2, -1, -1):
            h = torch.cat([deconv(h, f"up{d}_deconv"), skips[d]], dim=1)
            h = conv3(h, f"up{d}_conv1")
            h = conv3(h, f"up{d}_conv2")
            h = bn(h, f"up{d}_bn")
        k = t(w["head_kernel"]).permute(3, 2, 0, 1)
        out = F.conv2d(h, k, t(w["head_bias"]))  # the head is exact fp32 in the engine
    return out.permute(0, 2, 3, 1).numpy()


def main():
    H, W, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 96, 2)
    torch.set_num_threads(8)
    for bn in ("he", "realistic"):
        w = uo.make_weights(seed=3, bn=bn)
        rng = np.random.default_rng(0)
        vol = (rng.standard_normal((S, H, W)) * 120 + 300).astype(np.float32)
        xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
        ref = forward(w, xw, make_op("exact"))
        print(f"BN stats: {bn}   logits: |max| {np.abs(ref).max():.2f}  std {ref.std():.2f}")
        modes = ("bf16:x1", "fp16:x1", "fp16:a1w2", "fp16:a2w1", "bf16:x3", "fp16:x3", "fp16:x3s", "fp16:wino1d", "fp16:wino2d")
        if len(sys.argv) >= 5:
            modes = tuple(sys.argv[4].split(","))
        for mode in modes:
            out = forward(w, xw, make_op(mode))
            err = np.abs(out - ref)
            print(f"  {mode:10s} max |dlogit| {err.max():.3e}   rms {np.sqrt((err ** 2).mean()):.3e}")


if __name__ == "__main__":
    main()
