"""Emulation of the W2XC_PRECISION_BF16 path for tests (TEST INFRASTRUCTURE): same dataflow as the HIP
engine -- fp32 first layer, activations rounded to bf16 (RNE) between layers, bf16 weights on the middle
layers and on a one-plane last layer that is computed inside the epilogue of the layer before it (fused:
3 or more layers, mid-layer plane counts), fp32 weights on any other last layer -- with float64 accumulation (the GPU accumulates in fp32 in MFMA order, so
individual bf16 roundings may differ by one ulp; tolerances in the tests account for that)."""
import numpy as np
import torch
import torch.nn.functional as F


def _bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def convert_bf16_emulated(layers, plane):
    n = len(layers)
    t = torch.from_numpy(np.ascontiguousarray(plane, dtype=np.float32)).to(torch.float64)[None, None]
    t = F.pad(t, (n, n, n, n), mode="replicate")
    for k, (nin, nout, w, b) in enumerate(layers):
        wt = torch.from_numpy(w).to(torch.float64)
        mid = lambda c: c in (32, 64, 128)
        fused_last = (n >= 3 and k == n - 1 and nout == 1 and mid(nin) and mid(layers[n - 2][0]))
        if 0 < k < n - 1 or fused_last:
            wt = _bf16(wt)
        t = F.conv2d(t, wt, torch.from_numpy(b.astype(np.float32)).to(torch.float64))
        t = torch.where(t > 0, t, np.float64(np.float32(0.1)) * t)
        if k < n - 1:
            t = _bf16(t)
    return t[0, 0].to(torch.float32).numpy()


# ---- W2XC_PRECISION_BF16X2 / BF16X3 (w2xc_split.hip) ----------------------------------------------------
PRODUCTS = {1: [(0, 0)], 2: [(1, 0), (0, 1), (0, 0)], 3: [(1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)]}   # (activation term, weight term)


def _split(t, terms, dtype=torch.bfloat16):
    """fp32 tensor -> list of `terms` 16-bit-valued float64 tensors: a0 = rnd(a), a1 = rnd(a - a0), ... (RNE)."""
    r = t.to(torch.float32)
    out = []
    for _ in range(terms):
        h = r.to(dtype).to(torch.float32)
        out.append(h.to(torch.float64))
        r = r - h          # exact in fp32
    return out


def convert_split_emulated(layers, plane, terms, n_in=1, fp16=False):
    """Dataflow of the split pipeline with float64 accumulation: fp32 first layer; every mid layer
    (cin, cout in {32,64,128}, not first) sums PRODUCTS[terms] of the 16-bit terms of its fp32 input and
    weights; a one-plane last layer behind a mid layer is fused into that layer's epilogue and uses the same split
    products, any other last layer is fp32.  fp16=True (W2XC_PRECISION_FP16X2): fp16 terms, activations clamped to +-65504,
    weights scaled by the power of two that puts max|w| into [2^14, 2^15) and the sum scaled back.
    `plane` is (h, w) or (n_in, h, w); returns all output planes."""
    n = len(layers)
    dt = torch.float16 if fp16 else torch.bfloat16
    x = np.ascontiguousarray(plane, dtype=np.float32)
    t = torch.from_numpy(x).reshape(1, n_in, x.shape[-2], x.shape[-1])
    t = F.pad(t.to(torch.float64), (n, n, n, n), mode="replicate").to(torch.float32)
    mid = lambda c: c in (32, 64, 128)
    for k, (nin, nout, w, b) in enumerate(layers):
        bias = torch.from_numpy(b.astype(np.float32)).to(torch.float64)
        wt = torch.from_numpy(w)
        fused_last = (n >= 3 and k == n - 1 and nout == 1 and mid(nin) and mid(layers[n - 2][0]))
        if (k > 0 and mid(nin) and mid(nout)) or fused_last:   # (two-term modes compute a 1-plane last layer the same way)
            scale = 1.0
            if fp16:
                mx = float(np.abs(w).max())
                scale = float(2.0 ** (15 - np.frexp(np.float32(mx))[1])) if mx > 0 else 1.0
                t = t.clamp(-65504.0, 65504.0)
            xs, ws = _split(t, terms, dt), _split(wt * np.float32(scale), terms, dt)
            acc = None
            for (ta, tb) in PRODUCTS[terms]:
                p = F.conv2d(xs[ta], ws[tb])
                acc = p if acc is None else acc + p
            y = acc / scale + bias.view(1, -1, 1, 1)
        else:
            y = F.conv2d(t.to(torch.float64), wt.to(torch.float64), bias)
        y = y.to(torch.float32)                       # the accumulator is fp32
        t = torch.where(y > 0, y, np.float32(0.1) * y)
    return t[0].numpy()
