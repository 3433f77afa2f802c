"""Emulation of the W2XC_PRECISION_BF16 path for tests (TEST INFRASTRUCTURE): same dataflow as the HIP
engine -- fp32 first layer, activations rounded to bf16 (RNE) between layers, bf16 weights on the middle
layers, fp32 last layer -- with float64 accumulation (the GPU accumulates in fp32 in MFMA order, so
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
        if 0 < k < n - 1:
            wt = _bf16(wt)
        t = F.conv2d(t, wt, torch.from_numpy(b.astype(np.float32)).to(torch.float64))
        t = torch.where(t > 0, t, np.float64(np.float32(0.1)) * t)
        if k < n - 1:
            t = _bf16(t)
    return t[0, 0].to(torch.float32).numpy()
