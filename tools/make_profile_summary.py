#!/usr/bin/env python
"""tools/make_profile_summary.py <gpurun_out/prof_TAG> <profiles/PREFIX> [fp32|bf16x2|bf16x3|fp16x2|bf16]
Turn the rocprofv3 outputs of tools/profile.sh into the committed summaries:
  PREFIX_kernel_stats.csv  (rocprofv3 --kernel-trace --stats)
  PREFIX_pmc_summary.txt   (per-kernel means of every PMC counter)
  PREFIX_roofline.json     (per-layer algorithmic FLOPs/bytes, HBM traffic, MFMA pipe utilisation)
FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE reports 1/2 of wide streaming reads
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024."""
import csv, json, os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash
base, prefix = sys.argv[1].rstrip("/") + "/", sys.argv[2]
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
T = {"fp32": 0, "bf16x2": 2, "bf16x3": 3, "fp16x2": 2, "bf16": 1}[prec]
PRODUCTS = {0: 1, 1: 1, 2: 3, 3: 6}[T]
H, W, NL = 2160, 3840, 7
PLANES = [(1, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 1)]
shutil.copy(base + "trace/trace_kernel_stats.csv", prefix + "_kernel_stats.csv")
here = os.path.dirname(os.path.abspath(__file__))
pmcs = [base + d + "/pmc_counter_collection.csv" for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds")]
extra = [base + d + "/pmc_counter_collection.csv" for d in ("pmc_issue", "pmc_fifo") if os.path.exists(base + d + "/pmc_counter_collection.csv")]
open(prefix + "_pmc_summary.txt", "w").write(subprocess.run([sys.executable, os.path.join(here, "pmc_summary.py")] + pmcs + extra, capture_output=True, text=True).stdout)

def mean_counter(path, sub, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v) if v else None

stats = {r["Name"]: r for r in csv.DictReader(open(base + "trace/trace_kernel_stats.csv"))}
out = {"command": "rocprofv3 --kernel-trace [--stats | --pmc ...] -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-host%s (tools/profile.sh)" % ("" if T == 0 else " --precision " + prec),
       "kernel_source_hash": kernel_source_hash(),   # bench.py only quotes `traffic` from this file while the sources still hash to this
       "note": __doc__.split("FETCH_SIZE", 1)[1].strip().replace("\n", " "), "kernels": {}}
for k, (cin, cout) in enumerate(PLANES, 1):
    if T == 0:
        sub = ("conv3x3_first<%d," % cin) if k == 1 else ("conv3x3_last<%d, %d" % (cin, cout)) if k == NL else ("conv3x3_mfma2<%d, %d," % (cin, cout))
    else:
        sub = ("conv3x3_first_split<%d," % cin) if k == 1 else ("conv3x3_last<%d, %d" % (cin, cout)) if k == NL else ("conv3x3_split<%d, %d," % (cin, cout))
        fused12 = any("conv3x3_first2_split" in n for n in stats)
        if fused12 and k == 1:
            continue                                   # computed inside layer 2's kernel
        if fused12 and k == 2:
            sub = "conv3x3_first2_split<%d," % cout   # layers 1 + 2 in one kernel
        if k == NL and any("conv3x3_last_gather" in n for n in stats):
            sub = "conv3x3_last_gather"   # two-term modes: the last layer is fused into layer NL-1's epilogue + this gather
    first2_fp32 = T == 0 and any("conv3x3_first2_wino4" in n for n in stats)
    if first2_fp32 and k == 1:
        continue                                       # computed inside layer 2's kernel (conv3x3_first2_wino4)
    if first2_fp32 and k == 2:
        sub = "conv3x3_first2_wino4"
    if T == 0 and k == NL and any("conv3x3_last_gather" in n for n in stats):
        sub = "conv3x3_last_gather"   # fp32: the last layer inside conv3x3_wino4's epilogue + this gather (w2xc_opts.fusion)
    names = [n for n in stats if sub in n and (n.startswith("void conv3x3") or n.startswith("conv3x3"))]
    if not names and T == 0 and 1 < k < NL:            # Winograd kernel for this shape (conv3x3_wino4<CIN, COUT, ..> / conv3x3_wino<CIN, COUT>)
        for sub in ("conv3x3_wino4<%d, %d," % (cin, cout), "conv3x3_wino<%d, %d" % (cin, cout)):
            names = [n for n in stats if sub in n]
            if names:
                break
    if not names:
        continue
    name = names[0]
    px = (H + 2 * (NL - k)) * (W + 2 * (NL - k))
    avg_ns = float(stats[name]["AverageNs"])
    fetch = mean_counter(pmcs[0], sub, "FETCH_SIZE"); write = mean_counter(pmcs[1], sub, "WRITE_SIZE")
    rd, wr = 2 * fetch * 1024, write * 1024
    # bytes per element: fp32 planes at the caller's boundary and into the last layer, T bf16 terms in between
    in_bpe = 4 if (T == 0 or k == 1 or k == NL) else 2 * T
    out_bpe = 4 if (T == 0 or k >= NL - 1) else 2 * T
    alg = (cin * in_bpe + cout * out_bpe) * px
    fused_fp32 = T == 0 and any("conv3x3_last_gather" in n for n in stats)
    w4 = any("conv3x3_wino4<%d, %d," % PLANES[NL - 2] in n for n in stats)   # conv3x3_wino4 sums a 64-plane block's partials on chip
    if sub == "conv3x3_last_gather":
        alg = ((PLANES[NL - 2][1] // (64 if w4 else 32) if T == 0 else 2) * 9 * 4 + 4) * px   # partial tap planes in (Cout / 64 or / 32 blocks, or two halves), one plane out
    if fused_fp32 and k == NL - 1:
        alg = (cin * 4 + (cout // (64 if w4 else 32)) * 9 * 4) * px     # fused: writes the partial tap planes instead of cout fp32 planes
    if first2_fp32 and k == 2:
        alg = (4 + cout * 4) * px                       # reads the one-plane input, layer 1's activations never reach HBM
    if T > 0 and k == 2 and any("conv3x3_first2_split" in n for n in stats):
        alg = (4 + cout * out_bpe) * px                 # reads the input plane, layer 1's activations never reach HBM
    if T > 0 and k == NL - 1 and any("conv3x3_last_gather" in n for n in stats):
        alg = (cin * in_bpe + 2 * 9 * 4) * px           # fused: writes the partial tap planes instead of cout fp32 planes
    wino = "conv3x3_wino" in name
    issued = 0.25 if ("conv3x3_wino4" in name or "first2_wino4" in name) else 16.0 / 36.0 if wino else 1.0   # F(4x4,3x3): 36 of 144 multiplies; F(2x2,3x3): 16 of 36
    e = {"layer": k, "avg_ns": avg_ns, "calls": int(stats[name]["Calls"]), "pixels": px, "executed_flops_over_algorithmic": issued,
         "algorithmic_flops": 18 * cin * cout * px, "algorithmic_tflops": 18 * cin * cout * px / avg_ns / 1e3,
         "tflops": 18 * cin * cout * px / avg_ns / 1e3 * issued,   # FLOPs the kernel issues / time
         "mfma_products_per_fma": PRODUCTS if 1 < k < NL else 1,
         "algorithmic_bytes": alg, "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_traffic_bytes": rd + wr,
         "traffic_over_algorithmic": (rd + wr) / alg, "achieved_GBps_algorithmic": alg / avg_ns}
    if sub == "conv3x3_first2_wino4":
        e["note"] = "layers 1 + 2 in one launch: layer 1 (18 FLOP per value, 576 per pixel) runs on the VALU inside this kernel and is not in the FLOP figures, which are layer 2's"
    if sub == "conv3x3_last_gather":   # the last layer's MFMA work runs inside layer NL-1's kernel; this kernel only adds 18 floats per pixel
        e["algorithmic_flops"] = 18 * px
        e["tflops"] = e["algorithmic_tflops"] = 18 * px / avg_ns / 1e3
        e["note"] = "last layer fused into the previous kernel's epilogue; this is the tap/half gather"
    grbm = mean_counter(pmcs[3], sub, "GRBM_GUI_ACTIVE"); busy = mean_counter(pmcs[2], sub, "SQ_VALU_MFMA_BUSY_CYCLES")
    if grbm and busy:
        e.update({"shader_clock_GHz": grbm / 8 / avg_ns, "mfma_pipe_utilisation": busy / (1024 * grbm / 8),
                  "avg_waves_per_simd": mean_counter(pmcs[2], sub, "SQ_WAVE_CYCLES") * 4 / (grbm / 8) / 1024,
                  "SQ_LDS_BANK_CONFLICT": mean_counter(pmcs[3], sub, "SQ_LDS_BANK_CONFLICT")})
    out["kernels"][name] = e
    print(k, name[:44], "%.2f ms %.1f TF issued  HBM %.2f GB (%.2fx alg)  mfma_util %.3f" % (avg_ns / 1e6, e["tflops"], (rd + wr) / 1e9, e["traffic_over_algorithmic"], e.get("mfma_pipe_utilisation", 0)))
json.dump(out, open(prefix + "_roofline.json", "w"), indent=1)
