#!/usr/bin/env python
"""What row-sharding costs on the COMPUTE side, measured on one device: BASELINE configs[2]'s 16384 x 16384 plane (8192 x 8192 frame) cut into N = 1, 2, 4, 8
row shards exactly as `bench.py --gpus N` cuts it (rows [H r / N, H (r + 1) / N) plus the 28-row halo of the default kernels' banding-invariant geometry), every
shard timed ALONE with its view resident in HBM.  max over shards = the time an N-GPU node needs if nothing else is shared; N x max / T(1) - 1 = what the halo
recompute, the shorter launches and the per-launch ramps cost.  Not a scaling measurement (one device, no host contention, no PCIe): a bound on it."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from tools import gen_model
from bench import synth_luma, nn2x
w = g.load_package()
ms = w._ModelSet.from_layers(gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"]))
n = ms.n_layers
in_h = in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
H, W = 2 * in_h, 2 * in_w
y = synth_luma(2, in_h, in_w)
st = torch.cuda.current_stream()
o = w.make_opts(device=0)
res = {}
for N in (1, 2, 4, 8):
    ts = []
    for r in range(N):
        ra, rb = w.shard_rows(H, N, r)
        y0, y1 = w.shard_view(H, ra, rb, 4 * n)
        view = torch.from_numpy(np.ascontiguousarray(nn2x(y[y0 // 2:(y1 + 1) // 2])[y0 - 2 * (y0 // 2):][:y1 - y0])).cuda()
        out = torch.empty((rb - ra, W), dtype=torch.float32, device="cuda")
        run = lambda: ms.convert_rows_device(view.data_ptr(), W * 4, y1 - y0, y0, W, H, ra, rb, out.data_ptr(), W * 4, stream=st.cuda_stream, opts=o)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        ts.append(best)
        del view, out
    res[N] = ts
    print("N = %d: shard ms %s   max %.2f   N x max / T(1) = %.3f" % (N, " ".join("%.2f" % t for t in ts), max(ts), N * max(ts) / max(res[1])), flush=True)
print(json.dumps({"plane": [H, W], "shard_ms": res, "bound_on_strong_scaling_efficiency": {str(N): round(max(res[1]) / (N * max(res[N])), 4) for N in res}}))
