"""DESIGN.md 9.2, on the CPU: Winograd F(4x4,3x3) on the points 0, +-3/4, +-3/2 with every MFMA operand as an fp16 pair (hi, lo) and a product as
hi*hi + hi*lo + lo*hi accumulated in fp32 -- what the 1e-4 gate and the error against the fp64 truth become on the 7-layer scale2.0x topology (78x78 plane) and
on the single-layer standard-normal case of tests/test_gpu_parity.py, beside the fp32 kernel's own arithmetic.  numpy only:  python tools/fp16x3_emulation.py
Result (round 4), error against fp64 / range, gate use against the fp32 oracle (7 layers | one layer on standard-normal data):
  fp32 operands                         2.6e-6   0.15 | 0.36
  fp16 pairs as they are                1.7e-5   0.95 | 2.06   (FAILS: a third to a half of the lo halves are subnormal)
  fp16 pairs, operands scaled by 2^k    2.9e-6   0.16 | 0.50   (U and V each by the power of two that puts their largest magnitude at 2^13..2^14; unscaled behind the sums)
  hi * hi alone (plain fp16)            4.7e-3   252  | 848
largest |V| 50.5 unscaled (fp16 overflows at 65504): V's scale needs a bound of the layer's activations (e.g. a running maximum from the producer's epilogue)."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gen_model

def cook_toom(points, m=4, r=3):
    n = m + r - 1
    a = [float(p) for p in points]
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for k in range(n - 1):
        Nk = np.prod([a[k] - a[l] for l in range(n - 1) if l != k])
        for i in range(m): AT[i, k] = a[k] ** i
        for j in range(r): G[k, j] = a[k] ** j / Nk
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    M = np.zeros((m * r, n)); BT = np.zeros((n, n))
    for i in range(m):
        for j in range(r): M[i * r + j, :] = AT[i, :] * G[:, j]
    for p in range(n):
        rhs = np.array([1.0 if p == i + j else 0.0 for i in range(m) for j in range(r)])
        BT[:, p] = np.linalg.lstsq(M, rhs, rcond=None)[0]
    return AT, G, BT

def leaky(x): return np.where(x > 0, x, x * x.dtype.type(0.1))

def direct(x, w, b, dt):
    C, H, Wd = x.shape
    x = x.astype(dt); w = w.astype(dt)
    out = np.zeros((w.shape[0], H - 2, Wd - 2), dt)
    for r in range(3):
        for c in range(3): out += np.einsum('oc,chw->ohw', w[:, :, r, c], x[:, r:r + H - 2, c:c + Wd - 2]).astype(dt)
    return leaky(out + b.astype(dt)[:, None, None])

def split16(a):
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)

stats = {"vmax": 0.0, "lo_sub": 0, "lo_n": 0}
def wino(x, w, b, mats, mode, m=4):
    AT, G, BT = mats
    BT32 = BT.astype(np.float32); AT32 = AT.astype(np.float32)
    t = m + 2
    C, H, Wd = x.shape; O = w.shape[0]
    oh, ow = H - 2, Wd - 2
    nby, nbx = (oh + m - 1) // m, (ow + m - 1) // m
    xp = np.zeros((C, nby * m + 2, nbx * m + 2), np.float32); xp[:, :H, :Wd] = x
    iy = (np.arange(nby) * m)[:, None] + np.arange(t)[None, :]
    ix = (np.arange(nbx) * m)[:, None] + np.arange(t)[None, :]
    d = xp[:, iy][:, :, :, ix].transpose(0, 1, 3, 2, 4)
    V = np.einsum('ij,cyxjk->cyxik', BT32, d).astype(np.float32)
    V = np.einsum('cyxik,lk->cyxil', V, BT32).astype(np.float32)
    U = np.einsum('ij,ocjk,lk->ocil', G, w.astype(np.float64), G).astype(np.float32)
    if mode == "fp32":
        M = np.einsum('ocil,cyxil->oyxil', U, V).astype(np.float32)
    else:
        su = sv = np.float32(1.0)
        if mode.endswith("s"):   # operands scaled by powers of two into the upper fp16 range (the lo halves become normal numbers); unscaled behind the sums
            su = np.float32(2.0 ** np.floor(np.log2(16384.0 / np.abs(U).max()))); sv = np.float32(2.0 ** np.floor(np.log2(16384.0 / max(np.abs(V).max(), 1e-30))))
        Uh, Ul = split16(U * su); Vh, Vl = split16(V * sv)
        stats["vmax"] = max(stats["vmax"], float(np.abs(V).max()))
        nz = Vl != 0
        stats["lo_sub"] += int((np.abs(Vl[nz]) < 6.104e-5).sum()); stats["lo_n"] += int(nz.sum())
        M = np.einsum('ocil,cyxil->oyxil', Uh, Vh).astype(np.float32)     # (fp16 x fp16 products are exact in fp32; the sums are fp32 as in the MFMA)
        if mode.startswith("fp16x3"):
            M = (M + np.einsum('ocil,cyxil->oyxil', Uh, Vl).astype(np.float32) + np.einsum('ocil,cyxil->oyxil', Ul, Vh).astype(np.float32)).astype(np.float32)
        M = (M / (su * sv)).astype(np.float32)
    Y = np.einsum('ij,oyxjk->oyxik', AT32, M).astype(np.float32)
    Y = np.einsum('oyxik,lk->oyxil', Y, AT32).astype(np.float32)
    out = Y.transpose(0, 1, 3, 2, 4).reshape(O, nby * m, nbx * m)[:, :oh, :ow]
    return leaky(out + b.astype(np.float32)[:, None, None])

def gate(a, ref): return float((np.abs(a - ref) / (1e-4 * np.abs(ref) + 1e-5)).max())

mats = cook_toom([0, .75, -.75, 1.5, -1.5])
layers = gen_model.synth_layers(seed=102)
x0 = np.random.default_rng(7).random((1, 78, 78)).astype(np.float32)
sl = gen_model.synth_layers([32, 64], 200 + 32 * 7 + 64)
xs = np.random.default_rng(32 * 3 + 64 + 21).standard_normal((32, 23, 39)).astype(np.float32)
ref_s32 = direct(xs, sl[0][2], sl[0][3], np.float32)
x64 = x0.astype(np.float64); x32 = x0.copy()
for (ni, no, w, b) in layers:
    x64 = direct(x64, w, b, np.float64); x32 = direct(x32, w, b, np.float32)
rng = np.abs(x64).max()
for mode in ("fp32", "fp16x3", "fp16x3s", "fp16"):
    xw = x0.copy()
    for (ni, no, w, b) in layers: xw = wino(xw, w, b, mats, mode)
    ys = wino(xs, sl[0][2], sl[0][3], mats, mode)
    print("%-7s 7-layer: err / range vs fp64 %.2e, gate use vs the fp32 oracle %.3f | single layer, standard normal: gate use %.3f" %
          (mode, np.abs(xw - x64).max() / rng, gate(xw, x32), gate(ys, ref_s32)))
print("largest |V| %.3g (fp16 overflows at 65504); lo halves that are subnormal: %.1f %%" % (stats["vmax"], 100.0 * stats["lo_sub"] / max(stats["lo_n"], 1)))
