#!/usr/bin/env python
"""One 1080p -> 2160p host -> host call with w2xc_opts.verbose = 2: the phase timestamps of the unit (stderr), pageable and pinned planes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from tools import gen_model
w = g.load_package()
ms = w._ModelSet.from_layers(gen_model.synth_layers([1, 32, 32, 64, 64, 128, 128, 1], 102))
y = np.random.default_rng(1).random((1080, 1920), dtype=np.float32)
out = np.zeros((2160, 3840), np.float32)
lib = w.lib()
import ctypes as C
opts = w.make_opts(verbose=2)
def call(src, dst):
    rc = lib.w2xc_convert_plane_nn2x(ms.handle, src.ctypes.data, src.strides[0], 1920, 1080, dst.ctypes.data, dst.strides[0], C.byref(opts))
    assert rc == 0
for _ in range(3): call(y, out)
for _ in range(4):
    t0 = time.perf_counter(); call(y, out); print("pageable call %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
ps = torch.from_numpy(y).pin_memory(); pd = torch.empty((2160, 3840)).pin_memory()
for _ in range(2): call(ps.numpy(), pd.numpy())
for _ in range(3):
    t0 = time.perf_counter(); call(ps.numpy(), pd.numpy()); print("pinned call %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
