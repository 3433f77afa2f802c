#!/usr/bin/env python
"""Split-bf16 precisions vs the fp32 MFMA path on the GPU: error and per-layer time.
   python tools/split_check.py [--h 2160 --w 3840]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
from tools import gen_model
ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=2160); ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--model", default="scale2.0x")
ap.add_argument("--oracle", action="store_true", help="also compare with the CPU oracle (small planes)")
a = ap.parse_args()
w2xc = graft.load_package()
layers = gen_model.synth_layers(seed=gen_model.SEEDS[a.model])
ms = w2xc._ModelSet.from_layers(layers)
x = torch.rand(a.h, a.w, device="cuda")
st = torch.cuda.current_stream()
outs = {}
for name, prec in (("fp32", 0), ("bf16", 1), ("bf16x2", 2), ("fp16x2", 4), ("bf16x3", 3)):
    y = torch.empty_like(x)
    o = w2xc.make_opts(device=0, profile=1, precision=prec)
    for i in range(a.steps + 1):
        if i == 1: torch.cuda.synchronize(); ms.profile_reset(0)
        ms.convert_device(x.data_ptr(), a.w * 4, a.w, a.h, y.data_ptr(), a.w * 4, stream=st.cuda_stream, opts=o)
    torch.cuda.synchronize()
    t, n = ms.profile_read(0)
    per = [t[l] / max(n[l], 1) for l in range(len(layers))]
    outs[name] = y.double().cpu().numpy()
    ref = outs["fp32"]
    err = np.abs(outs[name] - ref)
    if a.oracle and a.h * a.w <= 400 * 400:
        from oracle import oracle as orc
        want = orc.Oracle(layers).convert(x.cpu().numpy(), njob=16).astype(np.float64)
        e2 = np.abs(outs[name] - want)
        print("        vs CPU oracle: max|err|/max %.3e  rms %.3e" % (e2.max() / np.abs(want).max(), np.sqrt((e2 ** 2).mean()) / np.abs(want).max()))
    print("%-7s total %7.3f ms  layers %s  max|err|/max|ref| %.3e  rms %.3e  nan %d" % (
        name, sum(per), " ".join("%.3f" % p for p in per), err.max() / np.abs(ref).max(), np.sqrt((err ** 2).mean()) / np.abs(ref).max(),
        int(np.isnan(outs[name]).sum())), flush=True)
