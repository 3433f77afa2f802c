#!/usr/bin/env python
"""Time ONE mid layer (cin->cout) on the GPU: model 1->cin->cout->1, per-layer hipEvent times.
   W2XC_MFMA_VARIANT=<n> python tools/layer_bench.py --cin 128 --cout 128"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
from tools import gen_model
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=128); ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--h", type=int, default=2160); ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--zero", action="store_true", help="all-zero input plane (constant activations: data-dependent power check)")
ap.add_argument("--precision", type=int, default=0, help="0 fp32, 1 bf16, 2 bf16x2, 3 bf16x3, 4 fp16x2 (16-bit modes time a T->T mid layer: model 1->cin->cout->cout->1)")
a = ap.parse_args()
w2xc = graft.load_package()
topo = [1, a.cin, a.cout, 1] if a.precision < 1 else [1, a.cin, a.cout, a.cout, 1]
ms = w2xc._ModelSet.from_layers(gen_model.synth_layers(topo, 7))
x = torch.zeros(a.h, a.w, device="cuda") if a.zero else torch.rand(a.h, a.w, device="cuda"); y = torch.empty_like(x)
o = w2xc.make_opts(device=0, profile=1, precision=a.precision)
st = torch.cuda.current_stream()
for i in range(a.steps + 1):
    if i == 1: torch.cuda.synchronize(); ms.profile_reset(0)
    ms.convert_device(x.data_ptr(), a.w * 4, a.w, a.h, y.data_ptr(), a.w * 4, stream=st.cuda_stream, opts=o)
torch.cuda.synchronize()
t, n = ms.profile_read(0)
t2 = t[1] / n[1]
flops = 18.0 * a.cin * a.cout * (a.h + 2) * (a.w + 2)
print("precision=%d %d->%d %dx%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)  all layers: %s" % (a.precision, a.cin, a.cout, a.h, a.w, t2, flops / t2 / 1e9, flops / t2 / 1e9 / 1.573, " ".join("%.3f" % (t[i] / n[i]) for i in range(len(t)))))
