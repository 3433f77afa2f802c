#!/usr/bin/env python
"""fp32 frame (1080p -> 2160p) with w2xc_opts.fusion = OFF / ON, alternating in one process: per-layer hipEvent times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
from tools import gen_model
w = g.load_package()
ms = w._ModelSet.from_layers(gen_model.synth_layers([1, 32, 32, 64, 64, 128, 128, 1], 102))
x = torch.rand(2160, 3840, device="cuda"); y = torch.empty_like(x); st = torch.cuda.current_stream()
best = {}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for name, f in (("off", w.FUSION_OFF), ("on", w.FUSION_ON)):
        o = w.make_opts(device=0, profile=1, fusion=f)
        for i in range(6):
            if i == 1: torch.cuda.synchronize(); ms.profile_reset(0)
            ms.convert_device(x.data_ptr(), 3840 * 4, 3840, 2160, y.data_ptr(), 3840 * 4, stream=st.cuda_stream, opts=o)
        torch.cuda.synchronize(); t, n = ms.profile_read(0)
        per = [t[i] / max(n[i], 1) for i in range(len(t))]
        print("round %d fusion %-3s frame %.3f ms  layers %s  (6+7: %.3f)" % (rnd, name, sum(per), " ".join("%.3f" % v for v in per), per[5] + per[6]), flush=True)
        best[name] = min(best.get(name, 1e9), sum(per))
print("best:", best)
