#!/usr/bin/env python
"""PCIe-inclusive timing of the HOST-pointer entry points (what the drop-in CLI path pays):
w2xc_convert_plane on the 2160x3840 CNN plane vs w2xc_convert_plane_nn2x on the 1080x1920 source."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
from tools import gen_model
import bench
w2xc = graft.load_package()
ms = w2xc._ModelSet.from_layers(gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"]))
up = bench.synth_frame_luma(2)
src = np.ascontiguousarray(up[::2, ::2])
for name, fn in (("convert_plane (2160x3840 host plane in/out)", lambda: ms.convert(up)),
                 ("convert_plane_nn2x (1080x1920 in, 2160x3840 out)", lambda: ms.convert_nn2x(src))):
    fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    print("%-52s %.1f ms  %.1f input-Mpix/s" % (name, t * 1e3, 1920 * 1080 / t / 1e6))
