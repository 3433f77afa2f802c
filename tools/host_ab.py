#!/usr/bin/env python
"""Host plane in -> host plane out (the 1080p frame of BASELINE configs[1] through w2xc_convert_plane_rows, nearest-2x fused: what bench.py's host_to_host leg
times) for values of w2xc_opts.fusion, alternating in ONE process, beside the resident frame time:
   prog   = round 6: ONE launch of layer n - 1 that finishes the last layer itself, rows shipped by job flags (the default of the host entry points)
   gather = rounds 4 / 5: layer n - 1 + gather in three row chunks
   python tools/host_ab.py [--rounds 3] [--calls 15] [--pinned]"""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as graft
from tools import gen_model
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--calls", type=int, default=15)
ap.add_argument("--pinned", action="store_true")
ap.add_argument("--h", type=int, default=1080); ap.add_argument("--w", type=int, default=1920)
a = ap.parse_args()
w2xc = graft.load_package(); lib = w2xc.lib()
ms = w2xc._ModelSet.from_layers(gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"]))
h, w = a.h, a.w
y = np.random.default_rng(2).random((h, w), dtype=np.float32)
if a.pinned:
    src_t = torch.from_numpy(y).pin_memory(); dst_t = torch.empty((2 * h, 2 * w), dtype=torch.float32).pin_memory()
    src, dst = src_t.numpy(), dst_t.numpy()
else:
    src, dst = y, np.zeros((2 * h, 2 * w), np.float32)
d_in = torch.from_numpy(np.repeat(np.repeat(y, 2, 0), 2, 1)).cuda(); d_out = torch.empty_like(d_in); st = torch.cuda.current_stream()
def resident(n=10):
    o = w2xc.make_opts(device=0)
    ms.convert_device(d_in.data_ptr(), 2 * w * 4, 2 * w, 2 * h, d_out.data_ptr(), 2 * w * 4, stream=st.cuda_stream, opts=o); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): ms.convert_device(d_in.data_ptr(), 2 * w * 4, 2 * w, 2 * h, d_out.data_ptr(), 2 * w * 4, stream=st.cuda_stream, opts=o)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
F = {"prog": w2xc.FUSION_AUTO, "gather": w2xc.FUSION_GATHER_LAUNCH}
outs = {}
for rnd in range(a.rounds):
    res = resident()
    line = "round %d resident %.3f ms |" % (rnd, res)
    for name, f in F.items():
        o = w2xc.make_opts(device=0, device_mask=1, fusion=f)
        def call():
            rc = lib.w2xc_convert_plane_rows(ms.handle, src.ctypes.data, src.strides[0], 0, h, w, h, 1, 0, 2 * h, dst.ctypes.data, dst.strides[0], C.byref(o))
            assert rc == 0, w2xc.last_error()
        call(); call()
        ts = []
        for _ in range(a.calls):
            t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
        outs[name] = dst.copy()
        line += " %s median %.3f min %.3f (%.4f of resident) |" % (name, statistics.median(ts), min(ts), res / statistics.median(ts))
    print(line, flush=True)
print("max |prog - gather| =", float(np.abs(outs["prog"] - outs["gather"]).max()), " max |prog - resident| =", float(np.abs(outs["prog"] - d_out.cpu().numpy()).max()))
