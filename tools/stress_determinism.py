#!/usr/bin/env python
"""Race screen for the counted-vmcnt / raw-barrier protocol of conv3x3_mfma2 and conv3x3_split (every precision,
the scale2.0x and the noise topologies, full-frame and odd sizes): the kernels are deterministic by
construction, so (1) repeated runs on the same input must be bit-identical, (2) under concurrent load on a
second stream too.  A DMA that lands late or a buffer overwritten early shows up as differing tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
from tools import gen_model
w2xc = graft.load_package()
ms = w2xc._ModelSet.from_layers(gen_model.synth_layers(seed=102))
bad = 0
PRECS = [("fp32", w2xc.PRECISION_FP32), ("fp16x2", w2xc.PRECISION_FP16X2), ("bf16x3", w2xc.PRECISION_BF16X3),
         ("bf16x2", w2xc.PRECISION_BF16X2), ("bf16", w2xc.PRECISION_BF16)]
for (pname, prec), (h, w) in [(p, s) for p in PRECS for s in [(2160, 3840), (1000, 1111), (257, 4097), (33, 70)]]:
    o = w2xc.make_opts(device=0, precision=prec)
    x = torch.rand(h, w, device="cuda")
    ref = torch.empty_like(x)
    st = torch.cuda.current_stream()
    ms.convert_device(x.data_ptr(), w * 4, w, h, ref.data_ptr(), w * 4, stream=st.cuda_stream, opts=o)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    noise = torch.rand(8192, 8192, device="cuda")
    for it in range(8):
        y = torch.empty_like(x)
        if it % 2:   # HBM-heavy traffic on another stream while the conv runs
            with torch.cuda.stream(side):
                for _ in range(20):
                    noise = noise * 1.0001 + 0.1
        ms.convert_device(x.data_ptr(), w * 4, w, h, y.data_ptr(), w * 4, stream=st.cuda_stream, opts=o)
        torch.cuda.synchronize()
        nd = int((y != ref).sum().item())
        if nd:
            bad += 1
            print("MISMATCH %s %dx%d run %d: %d pixels differ" % (pname, h, w, it, nd))
print("determinism stress:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
