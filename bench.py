#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric:
"Mpixels/sec end-to-end scale2.0x 7-layer conv, 1/2/4/8 GPU + host-CPU baseline").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one convertWithModels pass (7 layers: pad-7 by clamped loads, 7 kernel launches,
crop/stitch in the last kernel's store) over one frame per GPU:

  workload "scale2x_1080p" (BASELINE.json configs[1]): synthetic 1920x1080 RGB frame -> luma
  Y = 0.299R + 0.587G + 0.114B on /255 floats -> nearest-neighbour 2x (main.cpp:132-140) ->
  CNN plane 2160x3840 fp32, scale2.0x topology 1-32-32-64-64-128-128-1 with seeded synthetic
  weights (the shipped JSON models are stripped from the reference; see oracle/gen_model.py).

Input and output planes are resident in HBM when the timed region starts (device-pointer entry
point w2xc_convert_plane_device); `value` is whole-job input-image Mpix/s = N * 1920*1080 * K / t
(the CNN runs on 4x as many pixels).  Multi-GPU: one process per GPU, every rank converts its own
frame (the path shards into independent frames / row bands, no collective on the data path), so
per-GPU work is fixed: "scaling": "weak".  torch.distributed (RCCL) is used only for the barrier
and the MAX over ranks of the elapsed time.

Extra objects on the JSON line:
  roofline     dominant kernel = layer 6 (conv3x3_mfma, 128->128, 51% of the FLOPs): algorithmic
               FLOPs of one launch / its average duration from hipEvents recorded on the launch
               stream inside the timed region (w2xc_opts.profile), vs the 157.3 TFLOP/s fp32 MFMA peak.
  cpu_baseline the CPU oracle (reference-faithful restatement; OpenCV is unavailable so the real
               binary cannot be built) timed on this host's cores on a bounded sample of whole 512^2
               blocks of the same plane (the reference's block-split path costs the same per block).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PX = {  # 2*9*cin*cout per CNN pixel
    (1, 32): 576, (32, 32): 18432, (32, 64): 36864, (64, 64): 73728, (64, 128): 147456, (128, 128): 294912, (128, 1): 2304,
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0


def synth_frame_luma(seed, h=1080, w=1920):
    """seeded RGB uint8 frame -> Y plane in [0,1] (OpenCV RGB2YUV luma weights) -> nearest 2x"""
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8).astype(np.float32) / np.float32(255.0)
    y = (np.float32(0.299) * rgb[..., 0] + np.float32(0.587) * rgb[..., 1] + np.float32(0.114) * rgb[..., 2]).astype(np.float32)
    return np.repeat(np.repeat(y, 2, axis=0), 2, axis=1)


def cpu_baseline(layers, plane, budget_s=15.0):
    """Time the CPU oracle ("port": restatement of the reference algorithm, oracle/w2xc_oracle.c -- the
    real binary needs OpenCV, which is absent) on whole 512x512 blocks of the same plane.

    Threads: the reference partitions OUTPUT PLANES over nJob std::threads with floor(nOut/nJob) each
    and the remainder on the last thread (modelHandler.cpp:42-65).  With nJob > nOut every plane lands
    on the last thread, so on this topology (nOut = 32/32/64/64/128/128/1) nJob = 32 is the largest job
    count that still spreads every conv layer evenly; nJob = nproc on a 256-core host would serialise
    the 32- and 64-plane layers.  We therefore time nJob = min(32, cores) as the baseline (cores =
    threads actually busy) and, for context, the reference's default nJob = 4 (main.cpp:58-60)."""
    from oracle import oracle as orc
    ncpu = os.cpu_count() or 1
    njob = max(1, min(32, ncpu))
    o = orc.Oracle(layers)

    def one_block(idx, nj):
        y0 = (idx * 498) % max(plane.shape[0] - 498, 1)
        x0 = ((idx * 7) % 5) * 498 % max(plane.shape[1] - 498, 1)
        blk = np.ascontiguousarray(plane[y0:y0 + 498, x0:x0 + 498])   # pads to one full 512x512 block
        t0 = time.perf_counter()
        o.convert(blk, block_splitting=False, njob=nj)
        return time.perf_counter() - t0

    t1 = one_block(0, njob)
    n, total = 1, t1
    while total + t1 <= budget_s * 0.75 and n < 64:
        total += one_block(n, njob)
        n += 1
    t4 = one_block(n, 4) if total + 8 * t1 <= budget_s * 2 else None
    px_per_block = 498 * 498 / 4.0   # input-image pixels of one block (the CNN plane is the 2x image)
    res = {
        "value": round(n * px_per_block / total / 1e6, 6),
        "unit": "Mpix/s (input-image pixels)",
        "cores": njob,
        "host_cores": ncpu,
        "kind": "port",
        "sample": "%d whole 512x512 blocks (498x498 useful CNN pixels each) of the same plane in %.1f s, nJob=%d "
                  "(reference output-plane thread partition; the block-split path costs the same per block, so "
                  "this extrapolates to the plane)" % (n, total, njob),
        "label": "CPU restatement of reference algorithm (OpenCV unavailable)",
    }
    if t4 is not None:
        res["value_njob4_reference_default"] = round(px_per_block / t4 / 1e6, 6)
    return res


def pmc_traffic(kernel, cin, cout, H, W):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r1_roofline.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE).
    rocprofv3 cannot run inside this process, so this is the profile of the SAME command
    (tools/profile.sh); null when no matching profile is committed."""
    path = os.path.join(ROOT, "profiles", "r1_roofline.json")
    try:
        prof = json.load(open(path))
    except Exception:
        return None
    for name, k in prof.get("kernels", {}).items():
        if kernel in name and ("<%d, %d" % (cin, cout)) in name and k.get("pixels") == (H + 2) * (W + 2):
            return int(k["hbm_traffic_bytes"])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1080, help="input frame height (CNN plane is 2x)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true", help="skip the informational opt-in-precision runs after the timed region")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--band-rows", type=int, default=0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x2", "bf16x3", "fp16x2"],
                    help="bf16 = W2XC_PRECISION_BF16 (configs[3] arithmetic); bf16x2 / bf16x3 / fp16x2 = split products (fp32 values "
                         "as 2 / 3 bf16 terms or 2 fp16 terms on the 16-bit MFMAs).  NOT the headline number: the reference computes in fp32 "
                         "and `value` is only the BASELINE metric for the default")
    ap.add_argument("--workload", default="scale2x_1080p", choices=["scale2x_1080p", "plane", "image_u8"],
                    help="'plane': ONE --width x --height frame whose CNN plane is sharded into row bands over the "
                         "ranks (BASELINE.json configs[2] with --width 8192 --height 8192): strong scaling")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as graft
    from oracle import gen_model

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # one rank per GPU; (ranks share a GPU only in the gloo self-test)
    torch.cuda.set_device(dev_index)
    dist = None
    backend = os.environ.get("W2XC_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" for the 1-GPU self-test
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    w2xc = graft.load_package()
    layers = gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"])
    ms = w2xc._ModelSet.from_layers(layers)
    n_layers = ms.n_layers

    sharded = args.workload == "plane"
    plane = synth_frame_luma(seed=2 + (0 if sharded else rank), h=args.height, w=args.width)
    H, W = plane.shape
    stream = torch.cuda.current_stream()
    opts = w2xc.make_opts(device=dev_index, profile=1, band_rows=args.band_rows,
                          precision={"fp32": w2xc.PRECISION_FP32, "bf16": w2xc.PRECISION_BF16, "bf16x2": w2xc.PRECISION_BF16X2,
                                     "bf16x3": w2xc.PRECISION_BF16X3, "fp16x2": w2xc.PRECISION_FP16X2}[args.precision])
    if args.workload == "image_u8":
        # N2: uint8 RGB frame in HBM -> uint8 2x frame in HBM (colour conversion, bicubic U/V, CNN on Y, back to uint8)
        rgb = np.random.default_rng(2 + rank).integers(0, 256, size=(args.height, args.width, 3), dtype=np.uint8)
        d_in = torch.from_numpy(rgb).cuda()
        d_out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")

        def step():
            ms.scale2x_image_u8_device(d_in.data_ptr(), args.width * 3, args.width, args.height, d_out.data_ptr(), W * 3, 1,
                                       stream=stream.cuda_stream, opts=opts)
    elif sharded:
        # one plane, rank r owns output rows [ra, rb); its input view (rows + n_layers halo) is resident in HBM
        ra, rb = w2xc.shard_rows(H, world, rank)
        y0, y1 = w2xc.shard_view(H, ra, rb, n_layers)
        d_in = torch.from_numpy(np.ascontiguousarray(plane[y0:y1])).cuda()
        d_out = torch.empty((rb - ra, W), dtype=torch.float32, device="cuda")

        def step():
            ms.convert_rows_device(d_in.data_ptr(), W * 4, y1 - y0, y0, W, H, ra, rb, d_out.data_ptr(), W * 4,
                                   stream=stream.cuda_stream, opts=opts)
    else:
        d_in = torch.from_numpy(plane).cuda()
        d_out = torch.empty_like(d_in)

        def step():
            ms.convert_device(d_in.data_ptr(), W * 4, W, H, d_out.data_ptr(), W * 4, stream=stream.cuda_stream, opts=opts)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    ms.profile_reset(dev_index)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    layer_ms, launches = ms.profile_read(dev_index)
    ok = bool(torch.isfinite(d_out.float()).all().item())

    if rank == 0:
        in_px = args.height * args.width
        value = (1 if sharded else world) * in_px * args.steps / elapsed / 1e6
        # algorithmic FLOPs of one step per layer: every band launch of layer k computes
        # (band rows + 2(n-k)) x (W + 2(n-k)) pixels (valid conv on the haloed band)
        band_h = (rb - ra) if sharded else H
        flops_layer, nbands = [], []
        for l in range(n_layers):
            cin, cout = ms.planes(l)
            k = l + 1
            nb = max(launches[l] // max(args.steps, 1), 1)
            nbands.append(nb)
            px = nb * (band_h / nb + 2 * (n_layers - k)) * (W + 2 * (n_layers - k))
            flops_layer.append(2 * 9 * cin * cout * px)
        dom = int(np.argmax(flops_layer))   # dominant kernel: the layer with the most FLOPs
        dom_launches = max(launches[dom], 1)
        dom_ms = layer_ms[dom] / dom_launches
        bands = nbands[dom]
        dom_flops = flops_layer[dom] / bands
        # split modes: every algorithmic multiply-add is 3 (bf16x2, fp16x2) or 6 (bf16x3) 16-bit MFMA products
        products = {"fp32": 1, "bf16": 1, "bf16x2": 3, "bf16x3": 6, "fp16x2": 3}[args.precision]
        achieved = products * dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        per_layer = []
        for l in range(n_layers):
            ms_l = layer_ms[l] / max(launches[l], 1) * nbands[l]
            cin, cout = ms.planes(l)
            per_layer.append({"layer": l + 1, "kernel": ms.kernel_name(l, opts), "planes": "%d->%d" % (cin, cout),
                              "ms": round(ms_l, 4),
                              "tflops": round(flops_layer[l] / (ms_l * 1e-3) / 1e12, 2) if ms_l > 0 else None})
        out = {
            "metric": "Mpixels/sec end-to-end scale2.0x 7-layer conv",
            "value": round(value, 4),
            "unit": "Mpix/s (input-image pixels; CNN-plane pixels = 4x)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32",
                      "bf16": "bf16 activations/weights between layers, f32 accumulate (not the headline precision)",
                      "bf16x2": "f32 values as 2 bf16 terms (3 bf16 MFMA products per multiply-add, f32 accumulate) in layers 2..n-1; not the headline precision",
                      "fp16x2": "f32 values as 2 fp16 terms (3 fp16 MFMA products per multiply-add, f32 accumulate) in layers 2..n-1; not the headline precision",
                      "bf16x3": "f32 values as 3 bf16 terms (6 bf16 MFMA products per multiply-add, f32 accumulate) in layers 2..n-1; not the headline precision"}[args.precision],
            "data": "synthetic",
            "config": {"workload": ("plane (row-band sharded over ranks): " if sharded else "image_u8 (N2: u8 RGB in -> u8 2x RGB out, colour + bicubic U/V on the GPU): " if args.workload == "image_u8" else "scale2x_1080p: ") + "scale2.0x topology (1-32-32-64-64-128-128-1, synthetic seeded weights) on a "
                                   "%dx%d RGB frame -> Y plane nearest-2x -> %dx%d CNN plane, one frame per GPU per step, "
                                   "planes resident in HBM" % (args.width, args.height, W, H),
                       "cnn_plane": [H, W], "frames_per_step": 1 if sharded else world, "bands_per_frame": max(bands, 1),
                       "sharding": "independent frames per rank, no collective on the data path"},
            "roofline": {"bound": "mfma", "kernel": "%s (layer %d, %d->%d)" % ((ms.kernel_name(dom, opts), dom + 1) + ms.planes(dom)),
                         "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else 2500.0, "unit": "TFLOP/s",
                         "frac": round(achieved / (PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else 2500.0), 4),
                         "traffic": pmc_traffic(ms.kernel_name(dom), ms.planes(dom)[0], ms.planes(dom)[1], H, W)
                         if (dom == n_layers - 2 and bands == 1 and not sharded and args.precision == "fp32") else None,
                         "algorithmic_bytes": int((ms.planes(dom)[0] + ms.planes(dom)[1]) * 4 * dom_flops / (18 * ms.planes(dom)[0] * ms.planes(dom)[1])),
                         "avg_launch_ms": round(dom_ms, 4), "flops_per_launch": dom_flops, "mfma_products_per_fma": products},
            "layers": per_layer,
            "output_finite": ok,
        }
        if world == 1 and args.precision == "fp32" and args.workload == "scale2x_1080p" and not args.no_other_precisions:
            # the opt-in precisions on the same resident plane, right after the timed region (3 steps each): their
            # time and their distance from the fp32 result just measured.  Informational -- `value` above is fp32.
            try:
                other = {}
                ref = d_out.clone()
                rng = float(ref.abs().max().item())
                for name, prec in (("fp16x2", w2xc.PRECISION_FP16X2), ("bf16x3", w2xc.PRECISION_BF16X3),
                                   ("bf16x2", w2xc.PRECISION_BF16X2), ("bf16", w2xc.PRECISION_BF16)):
                    o2 = w2xc.make_opts(device=dev_index, band_rows=args.band_rows, precision=prec)
                    run = lambda: ms.convert_device(d_in.data_ptr(), W * 4, W, H, d_out.data_ptr(), W * 4, stream=stream.cuda_stream, opts=o2)
                    run()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        run()
                    torch.cuda.synchronize()
                    ms_p = (time.perf_counter() - t1) / 3 * 1e3
                    other[name] = {"ms_per_step": round(ms_p, 3), "Mpix_s": round(in_px / ms_p / 1e3, 2),
                                   "max_abs_diff_vs_fp32_path_over_range": float("%.3g" % ((d_out - ref).abs().max().item() / rng))}
                out["other_precisions"] = other
            except Exception as e:   # never let the side measurement break the headline line
                out["other_precisions"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(layers, plane, args.cpu_budget)
            out["speedup_vs_cpu"] = round(value / world / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
