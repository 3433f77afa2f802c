#!/usr/bin/env python
"""Per-layer hipEvent times of the 1080p -> 2160p frame (BASELINE configs[1]) for each fp32 mid-layer kernel, alternating in ONE
process (boxes differ by 2-4 %): w2xc_opts.kernel = W2XC_KERNEL_MFMA / _WINOGRAD32 / _WINOGRAD.
   python tools/frame_ab.py [--kernels wino4,wino32,mfma] [--rounds 3] [--steps 5] [--topo 1,32,32,64,64,128,128,1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
from tools import gen_model
ap = argparse.ArgumentParser()
ap.add_argument("--kernels", default="wino4,wino32")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--h", type=int, default=2160); ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--topo", default="1,32,32,64,64,128,128,1")
ap.add_argument("--check", action="store_true", help="also print each kernel's max |difference| to the direct MFMA kernel over the output range")
a = ap.parse_args()
w2xc = graft.load_package()
K = {"mfma": w2xc.KERNEL_MFMA, "wino32": w2xc.KERNEL_WINOGRAD32, "wino4": w2xc.KERNEL_WINOGRAD4, "auto": w2xc.KERNEL_AUTO}
topo = [int(v) for v in a.topo.split(",")]
ms = w2xc._ModelSet.from_layers(gen_model.synth_layers(topo, 102))
x = torch.rand(a.h, a.w, device="cuda"); y = torch.empty_like(x)
st = torch.cuda.current_stream()
best = {}
ref = None
if a.check:
    ms.convert_device(x.data_ptr(), a.w * 4, a.w, a.h, y.data_ptr(), a.w * 4, stream=st.cuda_stream, opts=w2xc.make_opts(device=0, kernel=w2xc.KERNEL_MFMA))
    torch.cuda.synchronize()
    ref = y.clone()
for rnd in range(a.rounds):
    for name in a.kernels.split(","):
        o = w2xc.make_opts(device=0, profile=1, kernel=K[name])
        for i in range(a.steps + 1):
            if i == 1: torch.cuda.synchronize(); ms.profile_reset(0)
            ms.convert_device(x.data_ptr(), a.w * 4, a.w, a.h, y.data_ptr(), a.w * 4, stream=st.cuda_stream, opts=o)
        torch.cuda.synchronize()
        t, n = ms.profile_read(0)
        per = [t[i] / max(n[i], 1) for i in range(len(t))]
        extra = ""
        if ref is not None:
            extra = "  max|diff to mfma|/range %.3g" % ((y - ref).abs().max().item() / ref.abs().max().item())
        print("round %d %-7s frame %.3f ms  layers: %s%s" % (rnd, name, sum(per), " ".join("%.3f" % v for v in per), extra), flush=True)
        best[name] = min(best.get(name, 1e9), sum(per))
print("best: " + "  ".join("%s %.3f ms" % kv for kv in best.items()))
