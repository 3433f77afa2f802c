#!/usr/bin/env python
"""Frame and per-layer hipEvent times of the 1080p -> 2160p frame (BASELINE configs[1]) for values of w2xc_opts.fusion, alternating in ONE process
(boxes differ by 2-4 %): auto (round 6: layer n - 1 finishes the last layer itself, conv3x3_wino4 PROG) vs gather (its own conv3x3_last_gather launch).
   python tools/fusion_ab.py [--modes auto,gather] [--rounds 3] [--steps 8] [--h 2160 --w 3840]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
from tools import gen_model
ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="prog,gather")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--h", type=int, default=2160); ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--topo", default="1,32,32,64,64,128,128,1")
a = ap.parse_args()
w2xc = graft.load_package()
F = {"auto": w2xc.FUSION_AUTO, "gather": w2xc.FUSION_GATHER_LAUNCH, "prog": w2xc.FUSION_PROG, "off": w2xc.FUSION_OFF, "first": w2xc.FUSION_FIRST, "last": w2xc.FUSION_LAST}
ms = w2xc._ModelSet.from_layers(gen_model.synth_layers([int(v) for v in a.topo.split(",")], 102))
x = torch.rand(a.h, a.w, device="cuda"); y = torch.empty_like(x)
st = torch.cuda.current_stream()
outs, best = {}, {}
for rnd in range(a.rounds):
    for name in a.modes.split(","):
        o = w2xc.make_opts(device=0, profile=0, fusion=F[name])
        op = w2xc.make_opts(device=0, profile=1, fusion=F[name])
        run = lambda oo: ms.convert_device(x.data_ptr(), a.w * 4, a.w, a.h, y.data_ptr(), a.w * 4, stream=st.cuda_stream, opts=oo)
        run(o); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps): run(o)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.steps * 1e3
        ms.profile_reset(0)
        for _ in range(a.steps): run(op)
        torch.cuda.synchronize()
        t, n = ms.profile_read(0)
        per = [t[i] / max(n[i], 1) for i in range(len(t))]
        outs.setdefault(name, y.clone())
        print("round %d %-7s frame %.3f ms (wall)  layers: %s" % (rnd, name, wall, " ".join("%.3f" % v for v in per)), flush=True)
        best[name] = min(best.get(name, 1e9), wall)
print("best: " + "  ".join("%s %.3f ms" % kv for kv in best.items()))
names = list(outs)
for nm in names[1:]:
    print("max |%s - %s| = %g" % (names[0], nm, (outs[names[0]] - outs[nm]).abs().max().item()))
