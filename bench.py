#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric:
"Mpixels/sec end-to-end scale2.0x 7-layer conv, 1/2/4/8 GPU + host-CPU baseline").

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one convertWithModels pass (7 layers: pad-7 by clamped loads, 7 kernel launches, crop/stitch
in the last kernel's store) over one batch of synthetic input:

  N = 1   workload "scale2x_1080p" (BASELINE.json configs[1]): a synthetic 1920x1080 RGB frame -> luma
          Y = 0.299R + 0.587G + 0.114B on /255 floats -> nearest-neighbour 2x (main.cpp:132-140) -> CNN plane
          2160x3840 fp32, scale2.0x topology 1-32-32-64-64-128-128-1 with seeded synthetic weights (the shipped
          JSON models are stripped from the reference; tools/gen_model.py).
  N > 1   workload "plane" (BASELINE.json configs[2]): ONE synthetic 8192x8192 RGB frame whose 16384x16384 CNN
          plane is cut into N contiguous row ranges, one per rank (the reference's block walk,
          convertRoutine.cpp:84-169, made parallel; ranks never exchange data, the host is the gather):
          "scaling": "strong".  The weak-scaling figure (every rank its own 1080p frame) is reported beside it
          as `weak`.  The N = 1 line carries the one-GPU time of the same 8192x8192 plane (`plane_8192`) so the
          strong-scaling efficiency can be computed against the identical workload.

`value` (the bench contract: inputs resident in HBM when the timed region starts) times the device-pointer entry
point over K steps, bracketed by barrier + synchronize, MAX over ranks, no profiling events in the region.
`host_to_host` is SURVEY 8(d)'s definition of end-to-end -- the convertWithModels call with a HOST Y plane in and a
HOST plane out (pad, H2D, layers, D2H, stitch; pageable numpy planes, median of the per-call wall times), i.e. what
/root/reference/src/main.cpp:96,148 measures around its call -- and its ratio to `value`; PCIe-inclusive, so by the
contract it is reported beside `value`, never as `value`.

Extra objects on the JSON line:
  roofline     dominant kernel = layer 6 (128->128, 51% of the FLOPs): ALGORITHMIC FLOPs of one launch (18*Cin*Cout per
               pixel) / its average duration from hipEvents recorded on the launch stream (w2xc_opts.profile)
               in a second pass of the same K steps right after the timed region (so the events are not inside
               `value`'s region; `ms_per_step_profiled` shows they cost nothing), vs the 157.3 TFLOP/s fp32 MFMA peak.
               That layer runs a Winograd kernel (conv3x3_wino4, F(4x4,3x3): 36 instead of 144 multiplies per plane pair and
               4x4 block; conv3x3_wino, F(2x2,3x3): 16 instead of 36 per 2x2 block; all fp32).  `achieved` / `frac`
               are what the MFMA pipe really does: the FLOPs the kernel ISSUES (1/4 resp. 16/36 of the algorithmic ones) over time, against the peak -- always < 1, and
               reproducible from profiles/r6_kernel_stats.csv.  The algorithmic rate (SURVEY 8d's FLOPs over the same
               time) is carried beside it as `algorithmic_tflops` / `algorithmic_speedup_vs_direct_roofline` (> 1 means
               faster than ANY direct convolution could be on this MFMA).  w2xc_opts.kernel = W2XC_KERNEL_MFMA runs
               conv3x3_mfma2, where executed = algorithmic.  A fused last layer is counted at its useful FLOPs (2 x 9 x Cin per pixel).
               `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes of the same command
               (profiles/r6_roofline.json), only when that profile was taken from the kernel sources being run
               (hash check), else null.
  parity_patch_max_rel_err   one border and one interior 48x48 patch of the plane that was TIMED against the CPU oracle (the checker).
  cpu_baseline the CPU oracle (reference-faithful restatement; OpenCV is unavailable so the real binary cannot be
               built) timed on this host's cores on a bounded sample of whole 512^2 blocks of the same plane.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_16BIT_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
PRECISIONS = ["fp32", "bf16", "bf16x2", "bf16x3", "fp16x2"]


def synth_luma(seed, h, w):
    """seeded RGB uint8 frame -> Y plane in [0,1] (OpenCV RGB2YUV luma weights), generated in row strips"""
    rng = np.random.default_rng(seed)
    y = np.empty((h, w), np.float32)
    for r0 in range(0, h, 1024):
        r1 = min(h, r0 + 1024)
        rgb = rng.integers(0, 256, size=(r1 - r0, w, 3), dtype=np.uint8).astype(np.float32) / np.float32(255.0)
        y[r0:r1] = np.float32(0.299) * rgb[..., 0] + np.float32(0.587) * rgb[..., 1] + np.float32(0.114) * rgb[..., 2]
    return y


def nn2x(y):
    return np.repeat(np.repeat(y, 2, axis=0), 2, axis=1)


def kernel_source_hash():
    """sha256 over the kernel + engine sources: a committed PMC profile only describes the code it was taken from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "waifu2x-converter-cpp_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".hpp", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, cin, cout, H, W, precision="fp32"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, tools/make_profile_summary.py).  rocprofv3 cannot
    run inside this process, so this is the profile of the SAME command -- valid only for the sources it was taken
    from: returns (bytes, note)."""
    rel = "profiles/r6_roofline.json" if precision == "fp32" else "profiles/r6_%s_roofline.json" % precision
    path = os.path.join(ROOT, rel)
    try:
        prof = json.load(open(path))
    except Exception:
        return None, "no committed PMC profile (%s)" % rel
    if prof.get("kernel_source_hash") != kernel_source_hash():
        return None, "committed PMC profile is from other kernel sources (hash %s != %s): dropped" % (prof.get("kernel_source_hash"), kernel_source_hash())
    for name, k in prof.get("kernels", {}).items():
        if kernel in name and ("<%d, %d" % (cin, cout)) in name and k.get("pixels") == (H + 2) * (W + 2):
            return int(k["hbm_traffic_bytes"]), rel + " (rocprofv3 --pmc, same command, same sources)"
    return None, "no matching kernel in " + rel


def probe_matrix_clock(lib_path, device, ms=30):
    """What this box's matrix pipes run at: libw2xc_probe.so (csrc/w2xc_probe.hip, a measurement aid beside the drop-in library) streams
    independent v_mfma_f32_16x16x4_f32 on every SIMD for ~`ms` ms; MHz = MFMAs per SIMD per second x 32 cycles.  None when the
    probe library is not there (it is never needed by the product path)."""
    import ctypes
    path = os.path.join(os.path.dirname(lib_path), "libw2xc_probe.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.w2xc_probe_mfma_mhz.restype = ctypes.c_double
        lib.w2xc_probe_mfma_mhz.argtypes = [ctypes.c_int, ctypes.c_int]
        mhz = float(lib.w2xc_probe_mfma_mhz(int(device), int(ms)))
    except (OSError, AttributeError):
        return None
    return round(mhz, 1) if mhz > 0 else None


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


def self_launch(n, ndev):
    """`python bench.py --gpus N` WITHOUT a launcher: re-exec this command line under torch.distributed.run, one rank per GPU
    (--master-addr 127.0.0.1, a free port), so that a plain invocation cannot fail for a launcher reason.  RCCL cannot put two ranks
    on one device: on a box with fewer than N devices the ranks share them and the barrier / MAX run over gloo (the record says so in
    `backend` and `rank_devices`; the data path has no collective either way)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < n:
        env.setdefault("W2XC_BENCH_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def parity_patches(layers, plane_rows_of, d_out, ra, rb, H, W, n_layers):
    """The plane that was TIMED against the CPU oracle (the checker, never the thing measured): one 48x48 patch on the plane's border
    and one in the interior of this rank's rows, each from a crop with the network's halo (convertRoutine.cpp:84-169's own argument).
    Returns max |gpu - oracle| / max |oracle| over both patches and whether every element is inside rtol 1e-4 + atol 1e-5."""
    from oracle import oracle as orc
    o = orc.Oracle(layers)
    ph = min(48, rb - ra)
    pw = min(48, W)
    spots = [(ra, 0), (ra + max(0, (rb - ra) // 2 - ph // 2), max(0, W // 2 - pw // 2))]
    worst, ok = 0.0, True
    for (y, x) in spots:
        y0, y1, x0, x1 = max(0, y - n_layers), min(H, y + ph + n_layers), max(0, x - n_layers), min(W, x + pw + n_layers)
        sub = o.convert(np.ascontiguousarray(plane_rows_of(y0, y1)[:, x0:x1]), block_splitting=False, njob=min(8, os.cpu_count() or 1))
        want = sub[y - y0:y - y0 + ph, x - x0:x - x0 + pw]
        got = d_out[y - ra:y - ra + ph, x:x + pw].cpu().numpy()
        worst = max(worst, float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)))
        ok = ok and bool(np.allclose(got, want, rtol=1e-4, atol=1e-5))
    return worst, ok, spots


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=0, help="input frame height (CNN plane is 2x); default 1080 (plane workload: 8192)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-other-precisions", dest="no_extras", action="store_true",
                    help="skip the informational legs after the timed region (other precisions, the 8192x8192 plane at N=1, the weak figure at N>1)")
    ap.add_argument("--no-host", action="store_true", help="skip the host->host leg")
    ap.add_argument("--host-steps", type=int, default=0, help="host->host calls to time (default: max(5, steps))")
    ap.add_argument("--jobs", type=int, default=0, help="modelUtility nJob = host staging threads of the host->host path (default: the reference's 4; measured 4 ~ 8 > 16 > 32)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--band-rows", type=int, default=0)
    ap.add_argument("--precision", default="fp32", choices=PRECISIONS,
                    help="bf16 = W2XC_PRECISION_BF16 (configs[3] arithmetic); bf16x2 / bf16x3 / fp16x2 = split products (fp32 values "
                         "as 2 / 3 bf16 terms or 2 fp16 terms on the 16-bit MFMAs).  NOT the headline number: the reference computes in fp32 "
                         "and `value` is only the BASELINE metric for the default")
    ap.add_argument("--workload", default="auto", choices=["auto", "scale2x_1080p", "plane", "image_u8"],
                    help="auto = scale2x_1080p at N=1, plane (8192x8192, BASELINE.json configs[2]) at N>1.  'plane': ONE --width x --height "
                         "frame whose CNN plane is sharded into row ranges over the ranks: strong scaling")
    ap.add_argument("--dump-out", default="", help="(tests) directory: every rank saves its output rows of the last step as rank<r>.npz")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as graft
    from tools import gen_model

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            self_launch(args.gpus, torch.cuda.device_count() if torch.cuda.is_available() else 0)   # (does not return)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # one rank per GPU; (ranks share a GPU only in the gloo self-test)
    torch.cuda.set_device(dev_index)
    dist = None
    backend = os.environ.get("W2XC_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" for the 1-GPU self-test
    # W2XC_BENCH_FORCE_PG=1 (test aid): create the process group even for ONE rank, so that the RCCL initialisation, barrier,
    # all_reduce and all_gather of the N > 1 path run on a single-GPU box (two ranks cannot share a device under RCCL)
    if world > 1 or os.environ.get("W2XC_BENCH_FORCE_PG") == "1":
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    w2xc = graft.load_package()
    lib = w2xc.lib()
    layers = gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"])
    ms = w2xc._ModelSet.from_layers(layers)
    n_layers = ms.n_layers
    njobs = args.jobs if args.jobs > 0 else lib.w2xc_get_jobs()   # the reference's default: 4 (modelHandler.hpp:99, the CLI's -j)
    lib.w2xc_set_jobs(njobs)

    workload = args.workload
    if workload == "auto":
        workload = "scale2x_1080p" if world == 1 else "plane"
    sharded = workload == "plane"
    in_h = args.height or (8192 if sharded else 1080)
    in_w = args.width or (8192 if sharded else 1920)
    H, W = 2 * in_h, 2 * in_w
    prec_code = {"fp32": w2xc.PRECISION_FP32, "bf16": w2xc.PRECISION_BF16, "bf16x2": w2xc.PRECISION_BF16X2,
                 "bf16x3": w2xc.PRECISION_BF16X3, "fp16x2": w2xc.PRECISION_FP16X2}
    stream = torch.cuda.current_stream()

    def mk_opts(profile=0, precision=None, band_rows=None):
        return w2xc.make_opts(device=dev_index, device_mask=1 << dev_index, profile=profile,
                              band_rows=args.band_rows if band_rows is None else band_rows,
                              precision=prec_code[precision or args.precision])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank(x):
        """every rank's value of x, on rank 0's line (a SCALE record then checks itself: N entries, the MAX is the one used)"""
        if dist is None:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(v.item()) for v in out]

    def time_steps(step, steps, warmup):
        """the bench contract: W untimed steps, then exactly K steps between barrier + synchronize, MAX over ranks"""
        for _ in range(warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync_all()
        return max_over_ranks(time.perf_counter() - t0)

    # ---- the resident workloads: inputs and outputs live in HBM, device-pointer entry points ------------------------
    def resident_frame(y_src, opts_of):
        """one whole frame per rank (weak scaling): CNN plane = nearest-2x of y_src, resident"""
        d_in = torch.from_numpy(nn2x(y_src)).cuda()
        d_out = torch.empty_like(d_in)
        h2, w2 = d_in.shape

        def step(opts):
            ms.convert_device(d_in.data_ptr(), w2 * 4, w2, h2, d_out.data_ptr(), w2 * 4, stream=stream.cuda_stream, opts=opts)
        return (lambda: step(opts_of(0))), (lambda: step(opts_of(1))), d_in, d_out

    y_src = synth_luma(seed=2 + (0 if sharded else rank), h=in_h, w=in_w)
    ra, rb = (w2xc.shard_rows(H, world, rank) if sharded else (0, H))
    if workload == "image_u8":
        # N2: uint8 RGB frame in HBM -> uint8 2x frame in HBM (colour conversion, bicubic U/V, CNN on Y, back to uint8)
        rgb = np.random.default_rng(2 + rank).integers(0, 256, size=(in_h, in_w, 3), dtype=np.uint8)
        d_in = torch.from_numpy(rgb).cuda()
        d_out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")

        def mk_step(profile):
            o = mk_opts(profile)
            return lambda: ms.scale2x_image_u8_device(d_in.data_ptr(), in_w * 3, in_w, in_h, d_out.data_ptr(), W * 3, 1,
                                                      stream=stream.cuda_stream, opts=o)
        step, step_prof = mk_step(0), mk_step(1)
    elif sharded:
        # one plane, rank r owns output rows [ra, rb); its input view (rows + n_layers halo of the 2x plane) is resident in HBM
        y0, y1 = w2xc.shard_view(H, ra, rb, 4 * n_layers)   # the wide halo: the default F(4x4) kernel's banding-invariant geometry (shards stitch bit-identically)
        view = nn2x(y_src[y0 // 2:(y1 + 1) // 2])[y0 - 2 * (y0 // 2):][:y1 - y0]
        d_in = torch.from_numpy(np.ascontiguousarray(view)).cuda()
        del view
        d_out = torch.empty((rb - ra, W), dtype=torch.float32, device="cuda")

        def mk_step(profile):
            o = mk_opts(profile)
            return lambda: ms.convert_rows_device(d_in.data_ptr(), W * 4, y1 - y0, y0, W, H, ra, rb, d_out.data_ptr(), W * 4,
                                                  stream=stream.cuda_stream, opts=o)
        step, step_prof = mk_step(0), mk_step(1)
    else:
        step, step_prof, d_in, d_out = resident_frame(y_src, lambda p: mk_opts(p))

    elapsed = time_steps(step, args.steps, args.warmup)
    # (a second, un-reduced look at the same K steps per rank would perturb the timed region: the per-rank figure is this rank's own
    #  wall time of the profiled pass below)
    # second pass, same K steps, with the per-layer hipEvents on the launch stream: per-kernel durations for `roofline`
    ms.profile_reset(dev_index)
    sync_all()
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        step_prof()
    sync_all()
    elapsed_prof_own = time.perf_counter() - tp0
    elapsed_prof = max_over_ranks(elapsed_prof_own)
    rank_ms = per_rank(elapsed_prof_own / args.steps * 1e3)
    rank_dev = per_rank(float(torch.cuda.current_device()))
    layer_ms, launches = ms.profile_read(dev_index)
    ok = bool(torch.isfinite(d_out.float()).all().item())
    parity = None
    if rank == 0 and not args.no_cpu_baseline and workload != "image_u8":
        # the CPU oracle as the CHECKER of the plane just timed (never inside a timed region)
        rows_of = lambda a, b: nn2x(y_src[a // 2:(b + 1) // 2])[a - 2 * (a // 2):][:b - a]
        parity = parity_patches(layers, rows_of, d_out, ra, rb, H, W, n_layers)
    if args.dump_out:
        os.makedirs(args.dump_out, exist_ok=True)
        np.savez(os.path.join(args.dump_out, "rank%d.npz" % rank), ra=ra, rb=rb, rows=d_out.cpu().numpy())

    # ---- host -> host: the convertWithModels call on HOST planes, PCIe and stitch included ----------------
    host = None
    if not args.no_host and workload != "image_u8":
        hsteps = args.host_steps or max(5, min(args.steps, 20))

        def host_leg(precision, pinned, steps):
            """rows [ra, rb) of the 2x conversion of y_src: host source plane in (nearest-2x fused into layer 1, so the 4x
            larger plane never crosses PCIe, main.cpp:132-148), host output rows out.  Per-call wall times."""
            sy0, sy1 = max(0, ra - 4 * n_layers) // 2, (min(H, rb + 4 * n_layers) + 1) // 2   # the wide halo, as for the resident shard above
            if pinned:
                src = torch.from_numpy(y_src[sy0:sy1]).pin_memory()
                dst = torch.empty((rb - ra, W), dtype=torch.float32).pin_memory()
                src_np, dst_np = src.numpy(), dst.numpy()
            else:
                src_np = np.ascontiguousarray(y_src[sy0:sy1])
                dst_np = np.zeros((rb - ra, W), np.float32)   # touched once: no first-touch page faults inside the timing
            o = mk_opts(0, precision)

            def call():
                rc = lib.w2xc_convert_plane_rows(ms.handle, src_np.ctypes.data, src_np.strides[0], sy0, sy1 - sy0, in_w, in_h, 1, ra, rb,
                                                 dst_np.ctypes.data, dst_np.strides[0], C.byref(o))
                if rc != 0:
                    raise RuntimeError("w2xc_convert_plane_rows: " + w2xc.last_error())
            call()
            call()
            times = []
            for _ in range(steps):
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                call()
                times.append(max_over_ranks(time.perf_counter() - t0))
            return times, dst_np

        res_ms = elapsed / args.steps * 1e3
        px_job = in_h * in_w * (1 if sharded else world)
        host = {"definition": "SURVEY 8(d): wall time of the convertWithModels-equivalent call, HOST luma plane in -> HOST 2x plane out "
                              "(w2xc_convert_plane_rows, nearest-2x fused, pinned staging rings, H2D / layers / D2H+stitch overlapped), "
                              "median of %d calls, nJob=%d staging threads" % (hsteps, njobs)}
        for label, pinned in (("pageable", False), ("pinned", True)):
            times, dst_np = host_leg(args.precision, pinned, hsteps)
            med = statistics.median(times) * 1e3
            host[label] = {"ms_median": round(med, 4), "ms_min": round(min(times) * 1e3, 4), "Mpix_s": round(px_job / med / 1e3, 4),
                           "ratio_vs_resident": round(res_ms / med, 4)}
            if label == "pageable":
                dev_rows = d_out.cpu().numpy()
                host["max_abs_diff_vs_resident_output"] = float(np.abs(dst_np - dev_rows).max())
        if world == 1 and not args.no_extras and args.precision == "fp32":
            t_res = time_steps(lambda: ms.convert_device(d_in.data_ptr(), W * 4, W, H, d_out.data_ptr(), W * 4, stream=stream.cuda_stream,
                                                         opts=mk_opts(0, "fp16x2")), 5, 2) / 5 * 1e3
            times, _ = host_leg("fp16x2", False, hsteps)
            med = statistics.median(times) * 1e3
            host["fp16x2_pageable"] = {"ms_median": round(med, 4), "resident_ms": round(t_res, 4), "ratio_vs_resident": round(t_res / med, 4)}

    # ---- informational legs --------------------------------------------------------------------------------------------
    extras = {}
    if not args.no_extras and args.precision == "fp32" and workload != "image_u8":
        try:
            if world == 1 and workload == "scale2x_1080p":
                # the opt-in precisions on the same resident plane (3 steps each): their time and their distance from the
                # fp32 result just measured.  Informational -- `value` above is fp32.
                other = {}
                step()
                torch.cuda.synchronize()
                ref = d_out.clone()
                rng = float(ref.abs().max().item())
                for name in ("fp16x2", "bf16x3", "bf16x2", "bf16"):
                    o2 = mk_opts(0, name)
                    run = lambda: ms.convert_device(d_in.data_ptr(), W * 4, W, H, d_out.data_ptr(), W * 4, stream=stream.cuda_stream, opts=o2)
                    t = time_steps(run, 3, 1) / 3 * 1e3
                    other[name] = {"ms_per_step": round(t, 3), "Mpix_s": round(in_h * in_w / t / 1e3, 2),
                                   "max_abs_diff_vs_fp32_path_over_range": float("%.3g" % ((d_out - ref).abs().max().item() / rng))}
                extras["other_precisions"] = other
                # the F(2x2,3x3) kernel on the same plane (w2xc_opts.kernel = W2XC_KERNEL_WINOGRAD32: conv3x3_wino on every mid layer, last layer as its own launch)
                o4 = w2xc.make_opts(device=dev_index, device_mask=1 << dev_index, band_rows=args.band_rows, kernel=w2xc.KERNEL_WINOGRAD32)
                run4 = lambda: ms.convert_device(d_in.data_ptr(), W * 4, W, H, d_out.data_ptr(), W * 4, stream=stream.cuda_stream, opts=o4)
                t4 = time_steps(run4, 5, 2) / 5 * 1e3
                extras["other_kernels"] = {"winograd_f2x2 (conv3x3_wino, w2xc_opts.kernel = W2XC_KERNEL_WINOGRAD32)": {
                    "ms_per_step": round(t4, 3), "Mpix_s": round(in_h * in_w / t4 / 1e3, 2),
                    "max_abs_diff_vs_default_path_over_range": float("%.3g" % ((d_out - ref).abs().max().item() / rng)),
                    "note": "4 multiplies per output where the default F(4x4,3x3) kernel does 2.25"}}
                # BASELINE.json configs[2] on ONE GPU (what N > 1 shards): 8192x8192 frame, host -> host and resident
                del ref
                yb = synth_luma(seed=2, h=8192, w=8192)
                o3 = mk_opts(0)
                outb = np.zeros((16384, 16384), np.float32)

                def call_big():
                    rc = lib.w2xc_convert_plane_nn2x(ms.handle, yb.ctypes.data, yb.strides[0], 8192, 8192, outb.ctypes.data, outb.strides[0], C.byref(o3))
                    if rc != 0:
                        raise RuntimeError(w2xc.last_error())
                call_big()
                tb = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    call_big()
                    tb.append(time.perf_counter() - t0)
                d_yb = torch.from_numpy(yb).cuda()
                d_ob = torch.empty((16384, 16384), dtype=torch.float32, device="cuda")
                run = lambda: ms.convert_nn2x_device(d_yb.data_ptr(), 8192 * 4, 8192, 8192, d_ob.data_ptr(), 16384 * 4, stream=stream.cuda_stream, opts=o3)
                tr = time_steps(run, 2, 1) / 2
                extras["plane_8192"] = {"workload": "BASELINE.json configs[2] on one GPU: 8192x8192 frame -> 16384x16384 CNN plane, workspace-banded",
                                        "resident_ms": round(tr * 1e3, 2), "resident_Mpix_s": round(8192 * 8192 / tr / 1e6, 3),
                                        "host_to_host_ms_median": round(statistics.median(tb) * 1e3, 2),
                                        "host_to_host_Mpix_s": round(8192 * 8192 / statistics.median(tb) / 1e6, 3),
                                        "host_to_host_ratio_vs_resident": round(tr / statistics.median(tb), 4),
                                        "host_output_finite": bool(np.isfinite(outb[::97, ::89]).all())}
                # the like-for-like denominator of a 1 -> N scaling curve: `--gpus N` (N > 1) shards THIS plane, so its N-GPU `value` divides
                # by this figure, not by the 1080p `value` above (the N > 1 line re-measures it on rank 0 and carries the quotient itself)
                extras["scale_reference"] = {"workload": "plane_8192 (BASELINE.json configs[2], what --gpus N > 1 shards), ONE GPU, planes resident",
                                             "Mpix_s": extras["plane_8192"]["resident_Mpix_s"], "ms_per_step": extras["plane_8192"]["resident_ms"]}
                # what ONE host sustains when it feeds eight units (an 8-GPU node's host side, here on the device(s) present): the same plane
                # through w2xc_opts.host_units = 8 with a ONE-layer 1 -> 1 model -- pageable plane -> pinned ring -> H2D and D2H -> pinned ring ->
                # pageable plane with nJob staging threads, a fraction of a millisecond of kernel time per unit
                try:
                    tiny = w2xc._ModelSet.from_layers(gen_model.synth_layers([1, 1], 3))
                    o8 = w2xc.make_opts(device=dev_index, device_mask=1 << dev_index, host_units=8)

                    def call_farm():
                        rc = lib.w2xc_convert_plane_nn2x(tiny.handle, yb.ctypes.data, yb.strides[0], 8192, 8192, outb.ctypes.data, outb.strides[0], C.byref(o8))
                        if rc != 0:
                            raise RuntimeError(w2xc.last_error())
                    call_farm()
                    tf = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        call_farm()
                        tf.append(time.perf_counter() - t0)
                    moved = yb.nbytes + outb.nbytes
                    extras["host_farm_8units_GBps"] = round(moved / min(tf) / 1e9, 2)
                    extras["host_farm_8units"] = {"seconds": round(min(tf), 4), "bytes_in_plus_out": moved, "njob": njobs, "devices": 1,
                                                  "note": "eight units' scatter + gather through the staging rings of ONE device (units on a shared device "
                                                          "serialise): a lower bound on what an 8-GPU node's host side sustains; the 16384^2 plane's "
                                                          "compute share per GPU at N = 8 is ~57 ms"}
                    del tiny
                except Exception as e:
                    extras["host_farm_8units_error"] = repr(e)
                del d_yb, d_ob, outb, yb
            elif world > 1 and sharded:
                # weak scaling beside the strong-scaling headline: every rank converts its own 1080p frame
                yw = synth_luma(seed=2 + rank, h=1080, w=1920)
                stepw, _, dw_in, dw_out = resident_frame(yw, lambda p: mk_opts(p, band_rows=0))
                tw = time_steps(stepw, args.steps, args.warmup)
                extras["weak"] = {"workload": "scale2x_1080p: one 1920x1080 frame per rank per step (BASELINE.json configs[1] x N)",
                                  "value": round(world * 1080 * 1920 * args.steps / tw / 1e6, 4), "unit": "Mpix/s", "ms_per_step": round(tw / args.steps * 1e3, 4),
                                  "scaling": "weak"}
                del dw_in, dw_out
                # the like-for-like denominator of THIS line's `value`: the same plane, whole, on rank 0's GPU alone (the other ranks wait at the
                # next barrier; outside every timed region).  scaling_efficiency = value / (N x this) -- the N = 1 line's 1080p `value` is another workload
                if dist is not None:
                    dist.barrier()
                if rank == 0:
                    yb = y_src
                    d_yb = torch.from_numpy(np.ascontiguousarray(yb)).cuda()
                    d_ob = torch.empty((H, W), dtype=torch.float32, device="cuda")
                    o1 = mk_opts(0)
                    run1 = lambda: ms.convert_nn2x_device(d_yb.data_ptr(), in_w * 4, in_w, in_h, d_ob.data_ptr(), W * 4, stream=stream.cuda_stream, opts=o1)
                    run1()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        run1()
                    torch.cuda.synchronize()
                    t1 = (time.perf_counter() - t0) / 2
                    ref = in_h * in_w / t1 / 1e6
                    extras["scale_reference"] = {"workload": "the same %dx%d frame, whole, on ONE GPU (rank 0 alone), planes resident" % (in_w, in_h),
                                                 "Mpix_s": round(ref, 3), "ms_per_step": round(t1 * 1e3, 2)}
                    extras["scaling_efficiency"] = round((in_h * in_w * args.steps / elapsed / 1e6) / (world * ref), 4)
                    extras["scaling_efficiency_is"] = "value / (n_gpus x scale_reference.Mpix_s): same plane, same run, same box%s" % (
                        "" if len(set(int(v) for v in rank_dev)) == world else " -- RANKS SHARE A DEVICE here (self-test): not a scaling figure")
                    del d_yb, d_ob
        except Exception as e:   # never let a side measurement break the headline line
            extras["extras_error"] = repr(e)
    if not args.no_extras and world == 1 and workload == "scale2x_1080p":
        # how much of ms_per_step is the shader clock the power management grants: the SAME launches on all-zero weights, biases and
        # plane (the instruction stream has no value-dependent branch; operand bits do not toggle).  tools/power_probe.py is the long form.
        try:
            ms0 = w2xc._ModelSet.from_layers([(ni, no, np.zeros_like(wl), np.zeros_like(bl)) for ni, no, wl, bl in layers])
            z_in = torch.zeros_like(d_in)
            z_out = torch.empty_like(d_out)
            pw = {}
            for name in ([args.precision] if args.precision != "fp32" else ["fp32", "bf16"]):
                oz = mk_opts(0, name)
                runz = lambda: ms0.convert_device(z_in.data_ptr(), W * 4, W, H, z_out.data_ptr(), W * 4, stream=stream.cuda_stream, opts=oz)
                pw[name] = round(time_steps(runz, 5, 2) / 5 * 1e3, 4)
                if name == args.precision:   # per-layer too, for the dominant kernel's fraction at the unthrottled clock (`roofline`)
                    ozp = mk_opts(1, name)
                    ms0.profile_reset(dev_index)
                    for _ in range(5):
                        ms0.convert_device(z_in.data_ptr(), W * 4, W, H, z_out.data_ptr(), W * 4, stream=stream.cuda_stream, opts=ozp)
                    torch.cuda.synchronize()
                    zt, zn = ms0.profile_read(dev_index)
                    pw["layers_ms"] = [round(zt[i] / max(zn[i], 1), 4) for i in range(len(zt))]
            extras["zero_operand_ms_per_step"] = dict(pw, note="same launches, all-zero weights / biases / plane: the time at the clock an idle "
                                                              "datapath is granted; the gap to ms_per_step (other_precisions for bf16) is power management, not schedule")
            del ms0, z_in, z_out
        except Exception as e:
            extras["zero_operand_error"] = repr(e)

    if rank == 0:
        in_px = in_h * in_w
        value = (1 if sharded else world) * in_px * args.steps / elapsed / 1e6
        # algorithmic FLOPs of one step per layer: every band launch of layer k computes
        # (band rows + 2(n-k)) x (W + 2(n-k)) pixels (valid conv on the haloed band)
        band_h = rb - ra
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
        opts = mk_opts(0)
        # split modes: every algorithmic multiply-add is 3 (bf16x2, fp16x2) or 6 (bf16x3) 16-bit MFMA products
        products = {"fp32": 1, "bf16": 1, "bf16x2": 3, "bf16x3": 6, "fp16x2": 3}[args.precision]
        peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_16BIT_MFMA_TFLOPS
        def issued(l):   # fraction of a layer's algorithmic multiplies its kernel issues (Winograd F(2x2,3x3): 16 of 36; F(4x4,3x3): 36 of 144)
            name = ms.kernel_name(l, opts)
            return 0.25 if "wino4" in name else 16.0 / 36.0 if "wino" in name else 1.0   # (conv3x3_wino4, conv3x3_first2_wino4: F(4x4); conv3x3_wino: F(2x2))
        algorithmic = products * dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # a fused last layer's MFMAs are issued by the dominant kernel too (taps-as-rows on 16x16x4 tiles: 16 rows x cin per pixel, 9 of them useful)
        fused_last = dom + 1 == n_layers - 1 and ms.kernel_name(dom + 1, opts) == "conv3x3_last_gather" and args.precision == "fp32"
        # counted at its 9 USEFUL rows (2 x 9 x Cin per pixel = the last layer's algorithmic FLOPs), not the 16 the tile issues
        fused_flops = (2.0 * 9 * ms.planes(dom + 1)[0] * dom_flops / (18.0 * ms.planes(dom)[0] * ms.planes(dom)[1])) if fused_last else 0.0
        achieved = algorithmic * issued(dom) + (fused_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0)
        per_layer = []
        for l in range(n_layers):
            ms_l = layer_ms[l] / max(launches[l], 1) * nbands[l]
            cin, cout = ms.planes(l)
            alg_bytes = (cin + cout) * 4 * flops_layer[l] / (18 * cin * cout)
            alg_tf = products * flops_layer[l] / (ms_l * 1e-3) / 1e12 if ms_l > 0 else None
            # (the dominant layer's launch also carries the fused last layer's useful FLOPs: the same accounting as `roofline`)
            fused_tf = fused_flops * nbands[l] / (ms_l * 1e-3) / 1e12 if (l == dom and ms_l > 0) else 0.0
            per_layer.append({"layer": l + 1, "kernel": ms.kernel_name(l, opts), "planes": "%d->%d" % (cin, cout),
                              "ms": round(ms_l, 4),
                              # what the MFMA pipe does: FLOPs the kernel issues / time, and its share of the peak
                              "tflops": round(alg_tf * issued(l) + fused_tf, 2) if ms_l > 0 else None,
                              "frac_of_peak": round((alg_tf * issued(l) + fused_tf) / peak, 4) if ms_l > 0 else None,
                              # SURVEY 8(d)'s algorithmic FLOPs over the same time
                              "algorithmic_tflops": round(alg_tf, 2) if ms_l > 0 else None,
                              "algorithmic_GBs_fp32_nhwc": round(alg_bytes / (ms_l * 1e-3) / 1e9, 1) if ms_l > 0 else None})
            if per_layer[-1]["kernel"] == "(in_next_layer)":
                per_layer[-1]["note"] = "layer 1 has no launch of its own: it is computed inside the next layer's kernel (conv3x3_first2_wino4), whose `ms` includes it"
            if per_layer[-1]["kernel"] == "conv3x3_last_gather":
                for kf in ("tflops", "frac_of_peak", "algorithmic_tflops", "algorithmic_GBs_fp32_nhwc"):
                    per_layer[-1][kf] = None   # (this launch adds 9 x Cout/64 partial values per pixel: the layer's multiplies are in the previous launch)
                per_layer[-1]["note"] = ("last layer fused: its MFMA work runs in the previous layer's epilogue (that layer's `ms` includes it, its FLOP "
                                         "figures do not); this launch only sums the partial tap planes")
        traffic, traffic_note = (pmc_traffic(ms.kernel_name(dom, opts), ms.planes(dom)[0], ms.planes(dom)[1], H, W, args.precision)
                                 if (dom == n_layers - 2 and bands == 1 and not sharded and args.precision in ("fp32", "bf16")) else (None, "no PMC profile for this configuration"))
        wl_name = {"scale2x_1080p": "scale2x_1080p (BASELINE.json configs[1])", "plane": "plane, row-sharded over ranks (BASELINE.json configs[2] at 8192x8192)",
                   "image_u8": "image_u8 (N2: u8 RGB in -> u8 2x RGB out, colour + bicubic U/V on the GPU)"}[workload]
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
            "value_is": ("`value` = planes resident in HBM when the timed region starts, as the bench contract defines it (a PCIe-inclusive rate is never `value`). "
                         "BASELINE.json's metric says END-TO-END, which SURVEY 8(d) defines as host plane in -> host plane out (what main.cpp:148 brackets): "
                         "that is `value_end_to_end` (= `value_host_to_host`, pageable planes), measured in the same run; quote it first wherever 'end-to-end' is meant"),
            "value_resident": round(value, 4),
            "value_host_to_host": host["pageable"]["Mpix_s"] if host else None,
            "value_end_to_end": host["pageable"]["Mpix_s"] if host else None,
            "config": {"workload": wl_name + ": scale2.0x topology (1-32-32-64-64-128-128-1, synthetic seeded weights) on a "
                                   "%dx%d RGB frame -> Y plane nearest-2x -> %dx%d CNN plane, %s" %
                                   (in_w, in_h, W, H, ("ONE frame per step, rank r computes rows [H*r/N, H*(r+1)/N) (+7-row halo), no exchange" if sharded
                                                       else "one frame per GPU per step")),
                       "cnn_plane": [H, W], "frames_per_step": 1 if sharded else world, "bands_per_unit": max(bands, 1),
                       "sharding": ("contiguous row ranges of one plane per rank, no collective on the data path, host-side gather" if sharded
                                    else "independent frames per rank, no collective on the data path")},
            "ms_per_step_profiled": round(elapsed_prof / args.steps * 1e3, 4),
            # self-check of a multi-rank record: the world size the process group reports, the devices the ranks sit on, and each
            # rank's own ms per step (profiled pass, barrier to barrier): N entries, all close to `ms_per_step`, or the run is not N-wide
            "ranks_seen": (dist.get_world_size() if dist is not None else 1), "backend": backend if dist is not None else None,
            "visible_devices": torch.cuda.device_count(), "rank_devices": [int(v) for v in rank_dev],
            "rank_ms_per_step": [round(v, 4) for v in rank_ms],
            "roofline": {"bound": "mfma", "kernel": "%s (layer %d, %d->%d)" % ((ms.kernel_name(dom, opts), dom + 1) + ms.planes(dom)),
                         "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_note,
                         # fp32 NHWC: Cin*4 read + Cout*4 written per pixel; with the last layer fused the launch writes Cout/32 x 9 partial tap values instead
                         "algorithmic_bytes": int((ms.planes(dom)[0] * 4 + (ms.planes(dom)[1] // (64 if ms.kernel_name(dom, opts) == "conv3x3_wino4" else 32) * 9 * 4 if fused_last else ms.planes(dom)[1] * 4))
                                                  * dom_flops / (18 * ms.planes(dom)[0] * ms.planes(dom)[1])),
                         "avg_launch_ms": round(dom_ms, 4), "mfma_products_per_fma": products,
                         "flops_per_launch": dom_flops * products * issued(dom) + fused_flops,
                         "fused_last_layer_flops_per_launch": fused_flops,
                         "algorithmic_flops_per_launch": dom_flops,
                         "algorithmic_tflops": round(algorithmic, 3),
                         "algorithmic_speedup_vs_direct_roofline": round(algorithmic / peak, 4),
                         "note": (("Winograd F(4x4,3x3) issues 36/144" if issued(dom) == 0.25 else "Winograd F(2x2,3x3) issues 16/36") +
                                  " of the algorithmic multiplies: `achieved` / `frac` are the MFMA pipe's own rate (issued "
                                  "FLOPs / time / peak); `algorithmic_tflops` is SURVEY 8(d)'s FLOPs over the same time -- above the peak, i.e. faster than a "
                                  "direct convolution can run on this MFMA" if issued(dom) < 1 else "direct convolution: issued = algorithmic FLOPs") +
                                 ("; the launch also carries the fused last layer (`fused_last_layer_flops_per_launch` = its 2 x 9 x Cin useful FLOPs per pixel, "
                                  "+3 %%; the 16-row tile it issues is not counted), in `achieved` and in layers[%d] alike, not in the algorithmic figures of THIS layer" % dom
                                  if fused_last else ""),
                         "timing": "hipEvents on the launch stream around every launch, second pass of the same %d steps" % args.steps},
            "layers": per_layer,
            "output_finite": ok,
            # the timed plane against the CPU oracle (checker): one border and one interior 48x48 patch, rtol 1e-4 + atol 1e-5 (north_star)
            "parity_patch_max_rel_err": (float("%.4g" % parity[0]) if parity else None),
            "parity_patch_within_gate": (parity[1] if parity else None),
            "parity_patch_spots": ([list(p) for p in parity[2]] if parity else None),
        }
        if host:
            out["host_to_host"] = host
        out.update(extras)
        mhz = probe_matrix_clock(w2xc.LIB_PATH, dev_index)
        if mhz:
            # gpurun boxes differ by a few percent on the same launch: what THIS box's matrix pipes deliver under an all-SIMD MFMA stream, beside the
            # nominal clock `roofline.peak` is priced at (the fraction itself stays at the nominal peak)
            out["matrix_clock"] = {"mfma_equiv_mhz": mhz, "nominal_mhz": 2400.0, "ratio": round(mhz / 2400.0, 4),
                                   "how": "libw2xc_probe.so: 30 ms of independent v_mfma_f32_16x16x4_f32 on random operands on every SIMD, MFMAs/s/SIMD x 32 cycles, after the timed passes"}
        zl = extras.get("zero_operand_ms_per_step", {}).get("layers_ms")
        if zl and zl[dom] > 0 and bands == 1:
            # the dominant kernel at the clock an idle datapath gets: what the schedule alone achieves (tools/power_probe.py)
            out["roofline"]["zero_operand_launch_ms"] = zl[dom]
            out["roofline"]["frac_zero_operands"] = round(out["roofline"]["frac"] * dom_ms / zl[dom], 4)
        if not args.no_cpu_baseline and world == 1 and workload != "image_u8":   # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(layers, nn2x(y_src), args.cpu_budget)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            if host:
                out["speedup_vs_cpu_host_to_host"] = round(host["pageable"]["Mpix_s"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
