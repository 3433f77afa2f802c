"""Thread-safety of the C ABI (SURVEY 8b "engine internally thread-safe per Model"; the reference's own workers write disjoint planes,
/root/reference/src/modelHandler.cpp:42-69, and its singleton is unguarded, :163-168).  ctypes releases the GIL for the duration of a
call, so Python threads really are concurrent callers of libw2xc_hip.so.  Every result must be BIT-identical to the same call made
alone: a (model, device) context serialises its callers, different models and the Model::filter cache are independent.

tools/thread_stress.cpp is the same stress as a C++ program; `make -C waifu2x-converter-cpp_amd/csrc tsan` builds it and the engine's
host code with -fsanitize=thread (the log of that run is profiles/r6_tsan.log)."""
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import ROOT, rand_plane, small_layers
from tools import gen_model

pytestmark = pytest.mark.gpu

SIZES = [(96, 160), (301, 423), (64, 64), (257, 130), (40, 500)]


@pytest.fixture(scope="module")
def gpu(w2xc):
    assert w2xc.device_count() >= 1, "no HIP device visible: libw2xc_hip has no CPU fallback"
    return w2xc


def run_threads(n, fn):
    errs = []

    def wrap(t):
        try:
            fn(t)
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errs.append((t, repr(e)))
    th = [threading.Thread(target=wrap, args=(t,)) for t in range(n)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not any(x.is_alive() for x in th), "a caller thread did not come back (deadlock?)"
    assert not errs, errs


def test_four_threads_one_model_and_two_models_on_one_device(gpu, scale_layers, noise1_layers):
    """4 threads x 20 calls of w2xc_convert_plane, mixed plane sizes: threads 0..1 and 2..3 share a model each pair in the second half,
    all four share ONE model in the first half; every plane equals the single-threaded result bit for bit"""
    a = gpu._ModelSet.from_layers(scale_layers)
    b = gpu._ModelSet.from_layers(noise1_layers)
    planes = [rand_plane(h, w, 100 + i) for i, (h, w) in enumerate(SIZES)]
    want = {id(a): [a.convert(p) for p in planes], id(b): [b.convert(p) for p in planes]}
    bad = []

    def one_model(t):
        for it in range(20):
            i = (t * 3 + it) % len(planes)
            if not np.array_equal(a.convert(planes[i]), want[id(a)][i]):
                bad.append(("one", t, it))
    run_threads(4, one_model)

    def two_models(t):
        ms = a if t < 2 else b
        for it in range(20):
            i = (t + 2 * it) % len(planes)
            got = ms.convert(planes[i]) if it % 3 else ms.convert_nn2x(planes[i][:48, :64])
            ref = want[id(ms)][i] if it % 3 else None
            if ref is not None and not np.array_equal(got, ref):
                bad.append(("two", t, it))
    run_threads(4, two_models)
    assert not bad, bad
    # the row-unit fan-out inside one call next to other callers: 3 units per call, two callers
    o3 = gpu.make_opts(host_units=3)
    run_threads(2, lambda t: [bad.append(("units", t, k)) for k in range(6) if not np.array_equal(a.convert(planes[1], opts=o3), want[id(a)][1])])
    assert not bad, bad


def test_device_entry_points_from_threads_with_their_own_models(gpu, scale_layers):
    """asynchronous device-pointer calls: one model and one stream per thread (the header's rule: calls on the same (model, device) share a
    stream); results equal the host entry point's"""
    import torch
    x = rand_plane(200, 333, 9)
    ref_model = gpu._ModelSet.from_layers(scale_layers)
    want = ref_model.convert(x)
    bad = []

    def worker(t):
        torch.cuda.set_device(0)
        ms = gpu._ModelSet.from_layers(scale_layers)
        st = torch.cuda.Stream()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        for it in range(10):
            d_out.zero_()
            torch.cuda.synchronize()
            ms.convert_device(d_in.data_ptr(), x.shape[1] * 4, x.shape[1], x.shape[0], d_out.data_ptr(), x.shape[1] * 4,
                              stream=st.cuda_stream, opts=gpu.make_opts(device=0))
            st.synchronize()
            if not np.array_equal(d_out.cpu().numpy(), want):
                bad.append((t, it))
    run_threads(4, worker)
    assert not bad, bad


def test_layer_filter_resident_chain_from_threads(gpu):
    """Model::filter chained by hand (test.cpp:72-85) with filter_resident, one model per pair of threads: the cache of what the previous
    call left on the device belongs to the (model, device) context and is used under its lock -- a chain interleaved with another
    thread's calls on the SAME model must still equal the chain run alone (a foreign call in between only costs the re-upload)"""
    layers = small_layers([1, 32, 64, 1], 77)
    x = rand_plane(61, 83, 4)
    solo = gpu._ModelSet.from_layers(layers)
    want = [x]
    for l in range(3):
        want = list(solo.filter(l, want))
    want = want[0]
    shared = [gpu._ModelSet.from_layers(layers), gpu._ModelSet.from_layers(layers)]
    bad = []

    def chain(t):
        ms = shared[t // 2]
        o = gpu.make_opts(filter_resident=1)
        for it in range(8):
            planes = [x]
            for l in range(3):
                planes = list(ms.filter(l, planes, opts=o))
            if not np.array_equal(planes[0], want):
                bad.append((t, it))
    run_threads(4, chain)
    assert not bad, bad


def test_image_pipeline_models_in_opposite_roles_do_not_deadlock(gpu, scale_layers, noise1_layers):
    """w2xc_process_image_u8 locks the noise model's and the scale model's contexts: thread 0 passes (A, B), thread 1 (B, A) -- with the
    locks taken one after the other this is the classic two-mutex deadlock; std::lock takes them together"""
    a = gpu._ModelSet.from_layers(scale_layers)
    b = gpu._ModelSet.from_layers(noise1_layers)
    img = np.random.default_rng(3).integers(0, 256, (40, 56, 3), dtype=np.uint8)
    want = [gpu.process_image_u8(img, noise=a, scale=b, iterations=1), gpu.process_image_u8(img, noise=b, scale=a, iterations=1)]
    bad = []

    def worker(t):
        for it in range(25):
            got = gpu.process_image_u8(img, noise=a if t == 0 else b, scale=b if t == 0 else a, iterations=1)
            if not np.array_equal(got, want[t]):
                bad.append((t, it))
    run_threads(2, worker)
    assert not bad, bad


def test_singleton_knobs_and_default_opts_from_threads(gpu, noise1_layers):
    """modelUtility's knobs (the reference's singleton is unguarded, modelHandler.cpp:163-168) and the process defaults are set and read
    under mutexes while conversions run"""
    ms = gpu._ModelSet.from_layers(noise1_layers)
    x = rand_plane(70, 90, 2)
    want = ms.convert(x)
    lib = gpu.lib()
    bad = []

    def worker(t):
        for it in range(30):
            if t == 0:
                lib.w2xc_set_jobs(1 + it % 6)
                lib.w2xc_set_block_size(256 + it, 256)
                lib.w2xc_set_default_opts(None)
            elif not np.array_equal(ms.convert(x), want):
                bad.append((t, it))
    try:
        run_threads(3, worker)
    finally:
        lib.w2xc_set_jobs(4)
        lib.w2xc_set_block_size(512, 512)
    assert not bad, bad


def test_cpp_thread_stress_program(gpu, tmp_path):
    """tools/thread_stress.cpp (the program `make tsan` instruments): the plain build must pass"""
    exe = str(tmp_path / "thread_stress")
    lib_dir = os.path.join(ROOT, "waifu2x-converter-cpp_amd", "lib")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "thread_stress.cpp"), "-I", os.path.join(ROOT, "include"),
                        "-L", lib_dir, "-lw2xc_hip", "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "4", "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "thread_stress: ok" in r.stdout
