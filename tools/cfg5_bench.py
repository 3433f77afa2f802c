#!/usr/bin/env python
"""configs[4] alone (3 -> 128 x 5 -> 3 on 2048 x 2048 through w2xc_convert_planes_device): per-launch hipEvent ms, the frame, and a patch against the
CPU oracle.  python tools/cfg5_bench.py [--rounds 3] [--fusion N]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as graft
from tools import gen_model
from oracle import oracle as orc
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--fusion", type=int, default=None)
ap.add_argument("--no-check", action="store_true")
a = ap.parse_args()
w2xc = graft.load_package()
wl = gen_model.synth_layers(gen_model.TOPOLOGY_WIDE, gen_model.SEEDS["wide"])
mw = w2xc._ModelSet.from_layers(wl)
h = w = 2048
xin = torch.rand((3, h, w), device="cuda"); xout = torch.empty((3, h, w), device="cuda")
st = torch.cuda.current_stream()
kw = {} if a.fusion is None else {"fusion": a.fusion}
po = w2xc.make_opts(device=0, profile=1, **kw)
fn = lambda: mw.convert_planes_device(3, xin.data_ptr(), h * w * 4, w * 4, w, h, xout.data_ptr(), h * w * 4, w * 4, stream=st.cuda_stream, opts=po)
fn(); fn(); torch.cuda.synchronize()
for r in range(a.rounds):
    mw.profile_reset(0)
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    lms, cnt = mw.profile_read(0)
    print("round %d frame %.3f ms (wall, with events)  launches: %s   (%s)" % (r, t * 1e3, " ".join("%.3f" % (lms[l] / max(cnt[l], 1)) for l in range(7)),
          ",".join(mw.kernel_name(l, po) for l in range(7))), flush=True)
if not a.no_check:
    xs = xin[:, 500:560, 700:760].cpu().numpy()
    tt = xs
    oo = orc.Oracle(wl)
    for l in range(7): tt = oo.filter(l, tt, njob=8)
    got = xout[:, 507:553, 707:753].cpu().numpy(); want = tt[:, 7:-7, 7:-7]
    print("patch max rel err vs oracle %.3g" % (np.abs(got - want).max() / np.abs(want).max()))
    # the plane's edges too (replicate padding inside layer 1): rows 0..45 x columns 0..45
    xs = np.pad(xin[:, 0:53, 0:53].cpu().numpy(), ((0, 0), (7, 0), (7, 0)), mode="edge"); tt = xs
    for l in range(7): tt = oo.filter(l, tt, njob=8)
    got = xout[:, 0:46, 0:46].cpu().numpy(); want = tt[:, 7:-7, 7:-7]
    print("corner max rel err vs oracle %.3g" % (np.abs(got - want).max() / np.abs(want).max()))
