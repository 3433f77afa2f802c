import sys; sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as g
from tools import gen_model
w = g.load_package()
ms = w._ModelSet.from_layers(gen_model.synth_layers([1, 32, 32, 64, 64, 128, 128, 1], 102))
x = torch.rand(2160, 3840, device="cuda"); yy = torch.empty_like(x); st = torch.cuda.current_stream()
for band in (0, 1080, 540, 0, 540):
    o2 = w.make_opts(device=0, profile=1, band_rows=band)
    for i in range(6):
        if i == 1: torch.cuda.synchronize(); ms.profile_reset(0)
        ms.convert_device(x.data_ptr(), 3840 * 4, 3840, 2160, yy.data_ptr(), 3840 * 4, stream=st.cuda_stream, opts=o2)
    torch.cuda.synchronize(); t, n = ms.profile_read(0)
    print("band_rows %4d: per-layer %s sum %.3f" % (band, " ".join("%.3f" % (t[i] / 5) for i in range(7)), sum(t) / 5))
