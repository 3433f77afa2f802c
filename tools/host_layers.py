import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from tools import gen_model
w = g.load_package()
ms = w._ModelSet.from_layers(gen_model.synth_layers([1, 32, 32, 64, 64, 128, 128, 1], 102))
y = np.random.default_rng(1).random((1080, 1920), dtype=np.float32)
out = np.zeros((2160, 3840), np.float32)
o = w.make_opts(profile=1)
for _ in range(3): ms.convert_nn2x(y, opts=o)
ms.profile_reset(0)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); ms.convert_nn2x(y, opts=o); ts.append((time.perf_counter() - t0) * 1e3)
t, n = ms.profile_read(0)
print("host mode: call median %.3f ms; per-layer GPU ms per call: %s (launches per call %s) sum %.3f" % (sorted(ts)[2], " ".join("%.3f" % (t[i] / 5) for i in range(7)), [n[i] // 5 for i in range(7)], sum(t) / 5))
x = torch.from_numpy(np.repeat(np.repeat(y, 2, 0), 2, 1)).cuda(); yy = torch.empty_like(x); st = torch.cuda.current_stream()
o2 = w.make_opts(device=0, profile=1)
for i in range(6):
    if i == 1: torch.cuda.synchronize(); ms.profile_reset(0)
    ms.convert_device(x.data_ptr(), 3840 * 4, 3840, 2160, yy.data_ptr(), 3840 * 4, stream=st.cuda_stream, opts=o2)
torch.cuda.synchronize(); t, n = ms.profile_read(0)
print("resident: per-layer %s sum %.3f" % (" ".join("%.3f" % (t[i] / 5) for i in range(7)), sum(t) / 5))
