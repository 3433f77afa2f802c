#!/usr/bin/env python
"""tools/run_configs.py -- run every configuration BASELINE.json lists, on one MI355X, and write a JSON report
(copied to profiles/r1_configs.json).  bench.py stays the headline (configs[1]); this is the evidence for the
other configs: timings from hipEvents / wall clock on resident planes, parity against the CPU oracle on
bounded samples (whole planes where the oracle finishes in seconds, patches otherwise).

  cfg1  noise1 topology, 256x256 luma plane: CPU oracle timed in full (nJob 4 and 32) + GPU parity/time
  cfg2  -> bench.py (scale2.0x on 1920x1080); only referenced here
  cfg3  scale2.0x on a synthetic 8192x8192 frame (CNN plane 16384^2), row-band entry point, 1 GPU here
  cfg4  noise2 + scale2.0x cascade on a 4096x4096 luma plane, bf16 path, tolerance vs the CPU fp32 cascade
  cfg5  widened 3-128-128-128-128-128-128-3 model on 3 planes of 2048x2048 (multi-plane wrapper)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as graft  # noqa: E402
from tools import gen_model
from oracle import oracle as orc  # noqa: E402

w2xc = graft.load_package()
out = {"device": torch.cuda.get_device_name(0), "host_cores": os.cpu_count()}
st = torch.cuda.current_stream()


def gpu_time(fn, steps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ---- cfg1 ------------------------------------------------------------------------------------------------------
layers = gen_model.synth_layers(seed=gen_model.SEEDS["noise1"])
x = np.random.default_rng(1).random((256, 256), dtype=np.float32)
o = orc.Oracle(layers)
cpu = {}
for nj in (4, 32):
    t0 = time.perf_counter()
    want = o.convert(x, njob=nj)
    cpu["njob%d_s" % nj] = round(time.perf_counter() - t0, 3)
ms = w2xc._ModelSet.from_layers(layers)
d_in = torch.from_numpy(x).cuda()
d_out = torch.empty_like(d_in)
opts = w2xc.make_opts(device=0)
t = gpu_time(lambda: ms.convert_device(d_in.data_ptr(), 1024, 256, 256, d_out.data_ptr(), 1024, stream=st.cuda_stream, opts=opts), 20)
out["cfg1_noise1_256x256"] = {"cpu_oracle": cpu, "cpu_Mpix_s_njob32": round(0.065536 / cpu["njob32_s"], 4),
                              "gpu_ms": round(t * 1e3, 3), "gpu_Mpix_s": round(0.065536 / t, 2),
                              "max_rel_err_vs_oracle": rel_err(d_out.cpu().numpy(), want)}
print("cfg1", out["cfg1_noise1_256x256"], flush=True)

# ---- cfg3 ------------------------------------------------------------------------------------------------------
slayers = gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"])
mscale = w2xc._ModelSet.from_layers(slayers)
H = W = 16384
rng = np.random.default_rng(3)
small = rng.integers(0, 256, size=(8192, 8192), dtype=np.uint8).astype(np.float32) / np.float32(255)
d_small = torch.from_numpy(small).cuda()
d_big = torch.empty((H, W), dtype=torch.float32, device="cuda")
t = gpu_time(lambda: mscale.convert_nn2x_device(d_small.data_ptr(), 8192 * 4, 8192, 8192, d_big.data_ptr(), W * 4,
                                                stream=st.cuda_stream, opts=opts), 2)
got = d_big[5000:5064, 9000:9064].cpu().numpy()
up = np.repeat(np.repeat(small[2490:2545, 4490:4545], 2, 0), 2, 1)   # plane rows 4980..5090, cols 8980..9090
sub = orc.Oracle(slayers).convert(up, block_splitting=False, njob=8)
out["cfg3_scale2x_8192x8192_1gpu"] = {"gpu_s": round(t, 4), "input_Mpix_s": round(8192 * 8192 / t / 1e6, 2),
                                      "entry": "w2xc_convert_plane_nn2x_device (nearest 2x fused into layer 1), workspace-banded",
                                      "patch_max_rel_err_vs_oracle": rel_err(got, sub[20:84, 20:84])}
# the same frame at the opt-in precisions (same entry point, same patch check)
for pname, prec in (("fp16x2", w2xc.PRECISION_FP16X2), ("bf16x3", w2xc.PRECISION_BF16X3)):
    po = w2xc.make_opts(device=0, precision=prec)
    t = gpu_time(lambda: mscale.convert_nn2x_device(d_small.data_ptr(), 8192 * 4, 8192, 8192, d_big.data_ptr(), W * 4,
                                                    stream=st.cuda_stream, opts=po), 2)
    got = d_big[5000:5064, 9000:9064].cpu().numpy()
    out["cfg3_scale2x_8192x8192_1gpu"][pname] = {"gpu_s": round(t, 4), "input_Mpix_s": round(8192 * 8192 / t / 1e6, 2),
                                                "patch_max_rel_err_vs_oracle": rel_err(got, sub[20:84, 20:84])}
print("cfg3", out["cfg3_scale2x_8192x8192_1gpu"], flush=True)
del d_big, d_small

# ---- cfg4 ------------------------------------------------------------------------------------------------------
nlayers = gen_model.synth_layers(seed=gen_model.SEEDS["noise2"])
mnoise = w2xc._ModelSet.from_layers(nlayers)
y = np.random.default_rng(4).random((4096, 4096), dtype=np.float32)
d_y = torch.from_numpy(y).cuda()
d_n = torch.empty_like(d_y)
d_s = torch.empty((8192, 8192), dtype=torch.float32, device="cuda")
res = {}
for prec, name in ((w2xc.PRECISION_BF16, "bf16"), (w2xc.PRECISION_FP16X2, "fp16x2"), (w2xc.PRECISION_FP32, "fp32")):
    po = w2xc.make_opts(device=0, precision=prec)

    def cascade():
        mnoise.convert_device(d_y.data_ptr(), 4096 * 4, 4096, 4096, d_n.data_ptr(), 4096 * 4, stream=st.cuda_stream, opts=po)
        mscale.convert_nn2x_device(d_n.data_ptr(), 4096 * 4, 4096, 4096, d_s.data_ptr(), 8192 * 4, stream=st.cuda_stream, opts=po)
    t = gpu_time(cascade, 3)
    # CPU fp32 cascade on a crop: noise on rows/cols 1000..1100 (+14 margin), nn2x, scale; compare the centre
    crop = np.ascontiguousarray(y[986:1114, 986:1114])
    n1 = orc.Oracle(nlayers).convert(crop, block_splitting=False, njob=8)[7:-7, 7:-7]        # valid: plane 993..1107
    s1 = orc.Oracle(slayers).convert(np.repeat(np.repeat(n1, 2, 0), 2, 1), block_splitting=False, njob=8)
    want = s1[14:-14, 14:-14]                                                                  # plane (2x) 2000..2200
    got = d_s[2000:2200, 2000:2200].cpu().numpy()
    mse = float(np.mean((got.astype(np.float64) - want) ** 2))
    res[name] = {"gpu_ms": round(t * 1e3, 2), "input_Mpix_s": round(4096 * 4096 / t / 1e6, 1),
                 "max_abs_err_vs_cpu_fp32": float(np.abs(got - want).max()), "max_abs_want": float(np.abs(want).max()),
                 "psnr_dB": round(10 * np.log10(1.0 / max(mse, 1e-30)), 1)}
out["cfg4_noise2_then_scale2x_4096x4096"] = res
print("cfg4", res, flush=True)
del d_s, d_n, d_y

# ---- cfg5 ------------------------------------------------------------------------------------------------------
wl = gen_model.synth_layers(gen_model.TOPOLOGY_WIDE, gen_model.SEEDS["wide"])
mw = w2xc._ModelSet.from_layers(wl)
h = w = 2048
xin = torch.rand((3, h, w), device="cuda")
xout = torch.empty((3, h, w), device="cuda")
po = w2xc.make_opts(device=0, profile=1)
fn = lambda: mw.convert_planes_device(3, xin.data_ptr(), h * w * 4, w * 4, w, h, xout.data_ptr(), h * w * 4, w * 4,
                                      stream=st.cuda_stream, opts=po)
fn()
torch.cuda.synchronize()
mw.profile_reset(0)
t = gpu_time(fn, 3)
lms, cnt = mw.profile_read(0)
per = []
for l in range(7):
    cin, cout = mw.planes(l)
    px = (h + 2 * (6 - l)) * (w + 2 * (6 - l))
    ms_l = lms[l] / max(cnt[l], 1)
    per.append({"layer": l + 1, "planes": "%d->%d" % (cin, cout), "kernel": mw.kernel_name(l), "ms": round(ms_l, 3),
                "tflops": round(18 * cin * cout * px / ms_l / 1e9, 1)})
xs = xin[:, 500:560, 700:760].cpu().numpy()
tt = np.pad(xs, ((0, 0), (0, 0), (0, 0)))
oo = orc.Oracle(wl)
for l in range(7):
    tt = oo.filter(l, tt, njob=8)
out["cfg5_wide_3_128x5_3_2048x2048"] = {"gpu_ms": round(t * 1e3, 2), "Mpix_s": round(h * w / t / 1e6, 1),
                                        "tflops_total": round(1488384 * h * w / t / 1e12, 1), "layers": per,
                                        "patch_max_rel_err_vs_oracle": rel_err(xout[:, 507:553, 707:753].cpu().numpy(), tt[:, 7:-7, 7:-7])}
print("cfg5", out["cfg5_wide_3_128x5_3_2048x2048"], flush=True)
dst = os.path.join(ROOT, "gpurun_out", "configs.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
