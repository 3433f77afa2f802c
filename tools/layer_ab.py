#!/usr/bin/env python
"""Per-layer hipEvent times of the resident 1080p -> 2160p frame for two option sets in ONE process (same box, alternating rounds):
   python tools/layer_ab.py            default (W2XC_FUSION_AUTO) against W2XC_FUSION_OFF"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from tools import gen_model
w = g.load_package()
ms = w._ModelSet.from_layers(gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"]))
x = torch.from_numpy(np.random.default_rng(1).random((2160, 3840), dtype=np.float32)).cuda()
y = torch.empty_like(x)
st = torch.cuda.current_stream()
sets = {"auto": w.make_opts(device=0, profile=1), "fusion_off": w.make_opts(device=0, profile=1, fusion=w.FUSION_OFF)}
for rnd in range(3):
    for name, o in sets.items():
        for i in range(7):
            if i == 2:
                torch.cuda.synchronize(); ms.profile_reset(0)
            ms.convert_device(x.data_ptr(), 3840 * 4, 3840, 2160, y.data_ptr(), 3840 * 4, stream=st.cuda_stream, opts=o)
        torch.cuda.synchronize()
        t, n = ms.profile_read(0)
        print("%-10s %s  sum %.3f   (%s)" % (name, " ".join("%.3f" % (t[i] / 5) for i in range(7)), sum(t) / 5, ",".join(ms.kernel_name(l, o) for l in range(7))))
