#!/usr/bin/env python
"""How much of a layer's time is the power limit?  The same 1080p -> 2160p frame (BASELINE configs[1]) with
  (a) seeded random weights and a random plane (what bench.py times),
  (b) the same weights and an all-zero plane with zero biases (activations are exactly 0 after layer 1: operand bits do not toggle),
  (c) all-zero weights, biases and plane,
per precision, alternating in ONE process.  The kernels execute the same instruction stream in the three cases (no value-dependent
branches), so a time difference is the shader clock the power management grants, not work.
   python tools/power_probe.py [--prec fp32,bf16] [--rounds 2] [--steps 5]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
from tools import gen_model
ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="fp32,bf16")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--h", type=int, default=2160); ap.add_argument("--w", type=int, default=3840)
a = ap.parse_args()
w2xc = graft.load_package()
P = {"fp32": w2xc.PRECISION_FP32, "bf16": w2xc.PRECISION_BF16}
base = gen_model.synth_layers(seed=102)
models = {
    "random": w2xc._ModelSet.from_layers(base),
    "zero-act": w2xc._ModelSet.from_layers([(ni, no, w, np.zeros_like(b)) for ni, no, w, b in base]),
    "zero-all": w2xc._ModelSet.from_layers([(ni, no, np.zeros_like(w), np.zeros_like(b)) for ni, no, w, b in base]),
}
planes = {"random": torch.rand(a.h, a.w, device="cuda"), "zero-act": torch.zeros(a.h, a.w, device="cuda"), "zero-all": torch.zeros(a.h, a.w, device="cuda")}
y = torch.empty(a.h, a.w, device="cuda")
st = torch.cuda.current_stream()
for rnd in range(a.rounds):
    for pname in a.prec.split(","):
        for case, ms in models.items():
            x = planes[case]
            o = w2xc.make_opts(device=0, profile=1, precision=P[pname])
            for i in range(a.steps + 1):
                if i == 1: torch.cuda.synchronize(); ms.profile_reset(0)
                ms.convert_device(x.data_ptr(), a.w * 4, a.w, a.h, y.data_ptr(), a.w * 4, stream=st.cuda_stream, opts=o)
            torch.cuda.synchronize()
            t, n = ms.profile_read(0)
            per = [t[i] / max(n[i], 1) for i in range(len(t))]
            print("round %d %-5s %-8s frame %.3f ms  layers: %s" % (rnd, pname, case, sum(per), " ".join("%.3f" % v for v in per)), flush=True)
