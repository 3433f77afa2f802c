#!/usr/bin/env python
"""tools/stress_host_pipeline.py -- race screen for the host->host tile farm: many conversions of random sizes, band heights,
chunk sizes, staging-thread counts and precisions through w2xc_convert_plane / _nn2x, each compared bit for bit with the
device-pointer entry point (same kernels, no staging).  Any mismatch = a slot reused too early, an event waited on too late, a
chunk stitched to the wrong rows -- or, since round 6, a row shipped before the gather job that wrote it over PCIe had landed (the fp32 default of the
host entry points: layer n - 1 finishes the last layer itself and flags its rows; w2xc_opts.fusion = W2XC_FUSION_GATHER_LAUNCH = the chunked launches of
rounds 4 / 5, drawn half of the time).  Every 25th iteration is a 2160x3840 frame.   python tools/stress_host_pipeline.py [--iters 300]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
from tools import gen_model

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--max-h", type=int, default=700)
ap.add_argument("--max-w", type=int, default=900)
a = ap.parse_args()
w2xc = graft.load_package()
ms = w2xc._ModelSet.from_layers(gen_model.synth_layers(seed=102))
rng = np.random.default_rng(12345)
st = torch.cuda.current_stream()
bad = 0
for it in range(a.iters):
    h, w = int(rng.integers(1, a.max_h)), int(rng.integers(1, a.max_w))
    nn2x = bool(rng.integers(0, 2))
    if it % 25 == 24:
        h, w = (1080, 1920) if nn2x else (2160, 3840)
    fusion = [w2xc.FUSION_AUTO, w2xc.FUSION_GATHER_LAUNCH][int(rng.integers(0, 2))]
    prec = [w2xc.PRECISION_FP32, w2xc.PRECISION_FP32, w2xc.PRECISION_FP16X2, w2xc.PRECISION_BF16][int(rng.integers(0, 4))]
    band = [0, 0, 37, 128, 200][int(rng.integers(0, 5))]
    chunk = [16, 64, 512, 8192][int(rng.integers(0, 4))]
    w2xc.lib().w2xc_set_jobs(int(rng.integers(1, 9)))
    x = rng.random((h, w), dtype=np.float32)
    kw = dict(precision=prec, band_rows=band, fusion=fusion)
    got = ms.convert_nn2x(x, opts=w2xc.make_opts(host_chunk_kb=chunk, **kw)) if nn2x else ms.convert(x, opts=w2xc.make_opts(host_chunk_kb=chunk, **kw))
    up = 2 if nn2x else 1
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty((h * up, w * up), dtype=torch.float32, device="cuda")
    o = w2xc.make_opts(device=0, **kw)
    if nn2x:
        ms.convert_nn2x_device(d_in.data_ptr(), w * 4, w, h, d_out.data_ptr(), w * up * 4, stream=st.cuda_stream, opts=o)
    else:
        ms.convert_device(d_in.data_ptr(), w * 4, w, h, d_out.data_ptr(), w * 4, stream=st.cuda_stream, opts=o)
    st.synchronize()
    want = d_out.cpu().numpy()
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH it=%d %dx%d nn2x=%d prec=%d band=%d chunk=%s fusion=%d: max abs diff %g" % (it, h, w, nn2x, prec, band, chunk, fusion, np.abs(got - want).max()), flush=True)
print("stress_host_pipeline: %d iterations, %d mismatches" % (a.iters, bad))
sys.exit(1 if bad else 0)
