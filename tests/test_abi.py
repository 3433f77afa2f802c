"""CPU tests of the drop-in boundary: libw2xc_hip.so loads and exports every symbol that
include/w2xc_hip.h declares, the model container / JSON loader / modelUtility knobs behave like
the reference's (src/modelHandler.{hpp,cpp}), and -- without a GPU -- the compute calls FAIL
(no CPU fallback anywhere in the product path)."""
import ctypes as C
import os
import re

# fp32 kernel of the 64- / 128-output-plane layers under W2XC_KERNEL_AUTO (w2xc_opts.kernel = W2XC_KERNEL_MFMA / _WINOGRAD / _WINOGRAD32 / _WINOGRAD4 picks per call)
MID_128 = "conv3x3_wino4"   # W2XC_KERNEL_AUTO = Winograd F(4x4,3x3) where it applies; no environment switches

import numpy as np
import pytest

from conftest import ROOT, rand_plane, small_layers
from tools import gen_model
from oracle import oracle as orc


def header_functions():
    src = open(os.path.join(ROOT, "include", "w2xc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(w2xc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(w2xc):
    names = header_functions()
    assert len(names) >= 20
    lib = C.CDLL(w2xc.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libw2xc_hip.so does not export %s" % n
    assert set(names) == set(w2xc.ABI_SYMBOLS), "python binding and header disagree"
    lib.w2xc_version.restype = C.c_char_p
    assert b"gfx950" in lib.w2xc_version()


def test_probe_library_is_separate_from_the_drop_in(w2xc):
    """bench.py's matrix-clock probe (csrc/w2xc_probe.hip) is a library of its own: libw2xc_hip.so carries no probe symbol,
    libw2xc_probe.so exports exactly the one entry point bench.py binds"""
    probe = os.path.join(os.path.dirname(w2xc.LIB_PATH), "libw2xc_probe.so")
    assert os.path.exists(probe), "make -C waifu2x-converter-cpp_amd/csrc builds it beside libw2xc_hip.so"
    assert hasattr(C.CDLL(probe), "w2xc_probe_mfma_mhz")
    assert not hasattr(C.CDLL(w2xc.LIB_PATH), "w2xc_probe_mfma_mhz")


def test_no_oracle_in_product_path():
    """the product package must never import, link or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "waifu2x-converter-cpp_amd")
    banned = re.compile(r"(import\s+oracle|from\s+oracle|w2xc_oracle|libw2xc_ref|oracle/|oracle\.py|_ref/)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not banned.search(text), "%s reaches into the oracle" % os.path.join(dirpath, f)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "lib", "libw2xc_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "w2xc_ref" not in out


def test_load_json_matches_reference_loader(w2xc, models_dir):
    p = os.path.join(models_dir, "scale2.0x_model.json")
    ms = w2xc._ModelSet.from_json(p)
    want = orc.load_model_json(p)
    assert ms.n_layers == 7
    assert [ms.planes(l) for l in range(7)] == [(1, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 1)]
    for l, (nin, nout, w, b) in enumerate(want):
        gi, go, gw, gb = ms.layer_arrays(l)
        assert (gi, go) == (nin, nout)
        assert np.array_equal(gw, w), "double->float narrowing differs in layer %d" % l
        assert np.array_equal(gb, b)
    ref = orc.RefBuild(p)   # the reference's own loader agrees on the topology
    assert [ref.planes(l) for l in range(ref.nlayers)] == [ms.planes(l) for l in range(7)]


def test_model_vector_semantics(w2xc, models_dir):
    models = []
    assert w2xc.modelUtility.generateModelFromJSON(os.path.join(models_dir, "noise1_model.json"), models)
    assert len(models) == 7
    assert models[0].getNInputPlanes() == 1 and models[0].getNOutputPlanes() == 32
    assert models[6].getNInputPlanes() == 128 and models[6].getNOutputPlanes() == 1
    # appending a second file extends the same vector, like the reference's push_back (:189-194)
    assert w2xc.modelUtility.generateModelFromJSON(os.path.join(models_dir, "noise2_model.json"), models)
    assert len(models) == 14


def test_loader_errors(w2xc, tmp_path, capfd):
    models = []
    assert not w2xc.modelUtility.generateModelFromJSON(str(tmp_path / "missing.json"), models)   # :175-179
    assert "couldn't open" in capfd.readouterr().err
    bad = tmp_path / "bad.json"
    bad.write_text('[{"kW":3,"kH":3,')
    with pytest.raises(w2xc.W2xcError) as e:
        w2xc._ModelSet.from_json(str(bad))
    assert e.value.code == w2xc.ERR_JSON
    nonsq = tmp_path / "nonsq.json"
    nonsq.write_text('[{"kW":3,"kH":5,"nInputPlane":1,"nOutputPlane":1,"bias":[0],"weight":[[[[0,0,0],[0,0,0],[0,0,0]]]]}]')
    with pytest.raises(w2xc.W2xcError) as e:
        w2xc._ModelSet.from_json(str(nonsq))     # the reference exit(-1)s (hpp:52-58); the library reports
    assert e.value.code == w2xc.ERR_UNSUPPORTED
    short = tmp_path / "short.json"
    short.write_text('[{"kW":3,"kH":3,"nInputPlane":2,"nOutputPlane":1,"bias":[0],"weight":[[[[0,0,0],[0,0,0],[0,0,0]]]]}]')
    with pytest.raises(w2xc.W2xcError) as e:
        w2xc._ModelSet.from_json(str(short))
    assert e.value.code == w2xc.ERR_JSON
    assert not models


def test_json_number_forms(w2xc, tmp_path):
    """strtod semantics (picojson.h:725-793): exponents, negative zero, long mantissas"""
    p = tmp_path / "n.json"
    p.write_text('[{"kW":3,"kH":3,"nInputPlane":1,"nOutputPlane":1,"bias":[-1.5e-3],'
                 '"weight":[[[[1e-1, -0.0, 2.5E+1],[0.1000000000000000055511151231257827, 3, -7e-40],[1.17549435e-38, 16777217, 0.30000001192092896]]]]}]')
    ms = w2xc._ModelSet.from_json(str(p))
    _, _, w, b = ms.layer_arrays(0)
    want = np.array([1e-1, -0.0, 2.5e1, 0.1, 3, -7e-40, 1.17549435e-38, 16777217, 0.30000001192092896], np.float64).astype(np.float32)
    assert np.array_equal(w.ravel().view(np.uint32), want.view(np.uint32))
    assert b[0] == -1.5e-3


def test_model_utility_singleton(w2xc):
    u = w2xc.modelUtility.getInstance()
    assert u is w2xc.modelUtility.getInstance()
    assert u.getNumberOfJobs() == 4                 # hpp:99
    assert u.getBlockSize() == (512, 512)           # hpp:99
    assert not u.setNumberOfJobs(0) and u.getNumberOfJobs() == 4     # cpp:200
    assert u.setNumberOfJobs(8) and u.getNumberOfJobs() == 8
    assert not u.setBlockSize((-1, 4)) and u.getBlockSize() == (512, 512)   # cpp:210
    assert u.setBlockSizeExp2Square(8) and u.getBlockSize() == (256, 256)   # cpp:215-220
    assert not u.setBlockSizeExp2Square(-1)
    assert u.setBlockSize((512, 512)) and u.setNumberOfJobs(4)


def test_opts_defaults(w2xc):
    o = w2xc.make_opts()
    assert o.struct_size == C.sizeof(w2xc.Opts)
    assert (o.precision, o.kernel, o.device, o.device_mask, o.band_rows, o.profile) == (0, 0, -1, 0, 0, 0)


def test_opts_struct_size_versions_the_abi(w2xc, noise1_layers):
    """a caller compiled against an OLDER (shorter) w2xc_opts passes its own struct_size: only that prefix is read, the
    newer fields take their defaults; a LARGER struct_size (newer caller, older library) reads only what the library knows"""
    ms = w2xc._ModelSet.from_layers(noise1_layers)
    lib = w2xc.lib()

    class OldOpts(C.Structure):      # the first three fields only: struct_size, precision, kernel
        _fields_ = [("struct_size", C.c_int), ("precision", C.c_int), ("kernel", C.c_int)]
    name = lib.w2xc_layer_kernel_name
    old = OldOpts(C.sizeof(OldOpts), w2xc.PRECISION_FP32, w2xc.KERNEL_DIRECT)
    # place the short struct at the END of a buffer whose following bytes are garbage: they must not be read as fields
    buf = (C.c_char * 64)(*([b"\xff"] * 64))
    C.memmove(buf, C.byref(old), C.sizeof(old))
    assert name(ms.handle, 5, C.cast(buf, C.POINTER(w2xc.Opts))) == b"conv3x3_direct"
    old.kernel = w2xc.KERNEL_AUTO
    C.memmove(buf, C.byref(old), C.sizeof(old))
    assert name(ms.handle, 5, C.cast(buf, C.POINTER(w2xc.Opts))) == MID_128.encode()
    # precision travels in the prefix too: a 16-bit mode picks the split kernels
    old.precision = w2xc.PRECISION_FP16X2
    C.memmove(buf, C.byref(old), C.sizeof(old))
    assert name(ms.handle, 5, C.cast(buf, C.POINTER(w2xc.Opts))) == b"conv3x3_split"
    # a newer, larger struct: the library copies sizeof(its own w2xc_opts) and ignores the tail
    class NewOpts(C.Structure):
        _fields_ = w2xc.Opts._fields_ + [("future_a", C.c_int), ("future_b", C.c_double)]
    new = NewOpts()
    lib.w2xc_opts_init(C.cast(C.byref(new), C.POINTER(w2xc.Opts)))
    new.struct_size = C.sizeof(NewOpts)
    new.kernel = w2xc.KERNEL_DIRECT
    new.future_a = -1
    assert name(ms.handle, 0, C.cast(C.byref(new), C.POINTER(w2xc.Opts))) == b"conv3x3_direct"
    o = w2xc.make_opts()
    assert o.filter_resident == 0 and o.fusion == w2xc.FUSION_AUTO and o.struct_size == C.sizeof(w2xc.Opts) == 56 and o.host_units == 0 and o.host_chunk_kb == 0 and o.host_numa == 0


def test_hostile_model_files_do_not_cross_the_abi(w2xc, tmp_path):
    """huge / non-finite plane counts and truncated files come back as error codes, never as exceptions or UB"""
    cases = {
        "huge": '[{"nInputPlane":1e300,"nOutputPlane":1,"kW":3,"kH":3,"bias":[0],"weight":[[[[0,0,0],[0,0,0],[0,0,0]]]]}]',
        "big": '[{"nInputPlane":2000000000,"nOutputPlane":2000000000,"kW":3,"kH":3,"bias":[0],"weight":[[[[0,0,0],[0,0,0],[0,0,0]]]]}]',
        "neg": '[{"nInputPlane":-4,"nOutputPlane":1,"kW":3,"kH":3,"bias":[0],"weight":[]}]',
        "many": '[{"nInputPlane":5000,"nOutputPlane":5000,"kW":3,"kH":3,"bias":[0],"weight":[[[[0,0,0],[0,0,0],[0,0,0]]]]}]',
        "trunc": '[{"nInputPlane":1,"nOutputPlane":1,"kW":3,"kH":3,"bias":[0],"weight":[[[[0,0,0],[0,0',
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".json")
        p.write_text(text)
        with pytest.raises(w2xc.W2xcError) as e:
            w2xc._ModelSet.from_json(str(p))
        assert e.value.code == w2xc.ERR_JSON, name


def test_argument_validation(w2xc, noise1_layers):
    ms = w2xc._ModelSet.from_layers(noise1_layers)
    lib = w2xc.lib()
    buf = np.zeros((4, 4), np.float32)
    assert lib.w2xc_convert_plane(ms.handle, buf.ctypes.data, 16, 0, 4, buf.ctypes.data, 16, 1, None) == w2xc.ERR_ARG
    assert lib.w2xc_convert_plane(ms.handle, buf.ctypes.data, 8, 4, 4, buf.ctypes.data, 16, 1, None) == w2xc.ERR_ARG
    assert lib.w2xc_convert_plane(ms.handle, None, 16, 4, 4, buf.ctypes.data, 16, 1, None) == w2xc.ERR_ARG
    with pytest.raises(w2xc.W2xcError) as e:
        ms.filter(1, np.zeros((5, 4, 4), np.float32))     # 5 planes into a 32-plane layer (:29-35)
    assert e.value.code == w2xc.ERR_PLANES
    assert ms.kernel_name(5) == MID_128
    # layers 1 + 2 (1 -> 32 -> 32) in one launch under the default kernels (N3): layer 1 has no launch of its own unless fusion is off
    assert ms.kernel_name(0) == "(in_next_layer)" and ms.kernel_name(1) == "conv3x3_first2_wino4"
    assert ms.kernel_name(0, w2xc.make_opts(fusion=w2xc.FUSION_OFF)) == "conv3x3_first" and ms.kernel_name(1, w2xc.make_opts(fusion=w2xc.FUSION_OFF)) == "conv3x3_wino"
    assert ms.kernel_name(1, w2xc.make_opts(kernel=w2xc.KERNEL_MFMA)) == "conv3x3_mfma" and ms.kernel_name(1, w2xc.make_opts(kernel=w2xc.KERNEL_DIRECT)) == "conv3x3_direct"
    # the one-plane last layer: inside conv3x3_wino4's epilogue (+ the tap gather) unless fusion is off or another mid kernel runs layer 6
    # (round 6: the 128 -> 128 launch finishes it too -- conv3x3_wino4 PROG -- so the layer has no launch; W2XC_FUSION_GATHER_LAUNCH = the separate gather of rounds 4 / 5)
    assert ms.kernel_name(6) == "conv3x3_last_gather"      # (the device entry points; the host entry points let the 128 -> 128 launch finish the layer: W2XC_FUSION_PROG everywhere)
    assert ms.kernel_name(6, w2xc.make_opts(fusion=w2xc.FUSION_PROG)) == "(in_previous_layer)"
    assert ms.kernel_name(6, w2xc.make_opts(fusion=w2xc.FUSION_GATHER_LAUNCH)) == "conv3x3_last_gather"
    assert ms.kernel_name(6, w2xc.make_opts(fusion=w2xc.FUSION_OFF)) == "conv3x3_last"
    assert ms.kernel_name(6, w2xc.make_opts(fusion=w2xc.FUSION_ON, kernel=w2xc.KERNEL_WINOGRAD4)) == "conv3x3_last_gather"
    assert ms.kernel_name(6, w2xc.make_opts(fusion=w2xc.FUSION_ON, kernel=w2xc.KERNEL_WINOGRAD)) == "conv3x3_last"   # (the F(2x2) kernel has no fused epilogue)
    assert ms.kernel_name(6, w2xc.make_opts(fusion=w2xc.FUSION_ON, kernel=w2xc.KERNEL_MFMA)) == "conv3x3_last"   # (no fused epilogue in that kernel)
    assert ms.kernel_name(5, w2xc.make_opts(kernel=w2xc.KERNEL_WINOGRAD4)) == "conv3x3_wino4"
    assert ms.kernel_name(5, w2xc.make_opts(fusion=w2xc.FUSION_ON)) == MID_128   # (fusion does not change the mid-layer kernel)
    assert ms.kernel_name(5, w2xc.make_opts(kernel=w2xc.KERNEL_MFMA)) == "conv3x3_mfma"          # per-call choice of the mid-layer kernel
    assert ms.kernel_name(5, w2xc.make_opts(kernel=w2xc.KERNEL_WINOGRAD)) == "conv3x3_wino"      # (alias of _WINOGRAD32 since conv3x3_wino16 was retired)
    assert ms.kernel_name(5, w2xc.make_opts(kernel=w2xc.KERNEL_WINOGRAD32)) == "conv3x3_wino"
    assert ms.kernel_name(0, w2xc.make_opts(kernel=w2xc.KERNEL_WINOGRAD)) == "conv3x3_first"     # (first / last layers have one fast kernel)
    assert ms.kernel_name(5, w2xc.make_opts(kernel=w2xc.KERNEL_DIRECT)) == "conv3x3_direct"


def test_farm_unit_argument_contract(w2xc, noise1_layers):
    """w2xc_convert_plane_rows validates the unit BEFORE it looks for a device: the source view must cover exactly the rows the
    row range reads -- [row_begin - n, row_end + n) of the plane, halved (rounded outwards) when the nearest-2x is fused -- so the
    row arithmetic of the multi-GPU farm is checkable without a GPU (a valid unit then fails with ERR_HIP here, never on the CPU)."""
    ms = w2xc._ModelSet.from_layers(noise1_layers)
    lib = w2xc.lib()
    n, h, w = ms.n_layers, 100, 64
    src = np.zeros((h, w), np.float32)
    ok_code = w2xc.OK if w2xc.device_count() > 0 else w2xc.ERR_HIP

    # the MINIMUM view (n halo rows) is a contract of the explicitly chosen F(2x2) kernels; W2XC_KERNEL_AUTO (the F(4x4) default) wants 4 n (below)
    o32 = w2xc.make_opts(kernel=w2xc.KERNEL_WINOGRAD32)

    def call(view_y0, view_h, nn2x, rb, re, opts=o32):
        out = np.zeros((max(re - rb, 1), w << nn2x), np.float32)
        v = src[view_y0:view_y0 + view_h]
        return lib.w2xc_convert_plane_rows(ms.handle, v.ctypes.data, v.strides[0], view_y0, view_h, w, h, nn2x, rb, re,
                                           out.ctypes.data, out.strides[0], C.byref(opts) if opts is not None else None)
    for nn2x in (0, 1):
        H = h << nn2x
        for parts in (1, 2, 3, 7):
            for p in range(parts):
                rb, re = w2xc.shard_rows(H, parts, p)
                y0, y1 = w2xc.shard_view(H, rb, re, n)                      # plane rows the unit reads (output coordinates)
                sy0, sy1 = y0 >> nn2x, (y1 + nn2x) >> nn2x                   # ... as source rows
                assert call(sy0, sy1 - sy0, nn2x, rb, re) == ok_code, (nn2x, parts, p)
                if sy0 > 0:
                    assert call(sy0 + 1, sy1 - sy0 - 1, nn2x, rb, re) == w2xc.ERR_ARG      # first halo row missing
                if sy1 < h:
                    assert call(sy0, sy1 - sy0 - 1, nn2x, rb, re) == w2xc.ERR_ARG          # last halo row missing
                # W2XC_KERNEL_AUTO (opts == NULL): the default F(4x4) kernel needs the WIDE halo (4 rows per layer) for banding-invariant results;
                # on a narrower view it is refused -- never a silent change of kernel and rounding
                wy0, wy1 = w2xc.shard_view(H, rb, re, 4 * n)
                ws0, ws1 = wy0 >> nn2x, (wy1 + nn2x) >> nn2x
                assert call(ws0, ws1 - ws0, nn2x, rb, re, None) == ok_code, (nn2x, parts, p)
                if (ws0, ws1) != (sy0, sy1):
                    assert call(sy0, sy1 - sy0, nn2x, rb, re, None) == w2xc.ERR_ARG and "4 halo rows per layer" in w2xc.last_error()
        assert call(0, h, nn2x, 10, 10) == w2xc.ERR_ARG                       # empty range
        assert call(0, h, nn2x, -1, 5) == w2xc.ERR_ARG
        assert call(0, h, nn2x, 0, H + 1) == w2xc.ERR_ARG
        assert call(0, h + 1, nn2x, 0, H) == w2xc.ERR_ARG                     # view longer than the plane
    assert call(0, h, 2, 0, h) == w2xc.ERR_ARG                                # nn2x is 0 or 1


def test_no_cpu_fallback_without_gpu(w2xc, noise1_layers):
    """On a box without a HIP device the product path must fail, never compute on the CPU."""
    if w2xc.device_count() > 0:
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    ms = w2xc._ModelSet.from_layers(noise1_layers)
    with pytest.raises(w2xc.W2xcError) as e:
        ms.convert(rand_plane(8, 8, 0))
    assert e.value.code == w2xc.ERR_HIP
    out = w2xc.Mat()
    models = [w2xc.Model(ms, i) for i in range(7)]
    assert w2xc.convertWithModels(w2xc.Mat(rand_plane(8, 8, 0)), out, models) is False
    assert out.array is None
    outs = []
    assert models[0].filter([w2xc.Mat(rand_plane(8, 8, 0))], outs) is False
    with pytest.raises(w2xc.W2xcError) as e:
        ms.scale2x_image_u8(np.zeros((4, 4, 3), np.uint8))
    assert e.value.code == w2xc.ERR_HIP


def test_arbitrary_model_lists_get_their_own_container(w2xc):
    a = w2xc._ModelSet.from_layers(small_layers([1, 4, 4, 1], 3))
    models = [w2xc.Model(a, i) for i in range(3)]
    assert w2xc._set_of(models) is a
    sub = w2xc._set_of(models[:2])
    assert sub is not a and sub.n_layers == 2 and sub.planes(1) == (4, 4)
    assert np.array_equal(sub.layer_arrays(1)[2], a.layer_arrays(1)[2])


def test_cli_shell_logic():
    """N4 (tools/w2xc_cli.py): ratio -> (2x iterations, shrink) exactly like main.cpp:107-114, the automatic
    output name of main.cpp:173-189, and the reference's flag set (main.cpp:26-60)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("w2xc_cli", os.path.join(ROOT, "tools", "w2xc_cli.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    assert cli.plan_scale(2.0) == (1, 0.0)
    assert cli.plan_scale(4.0) == (2, 0.0)
    assert cli.plan_scale(1.0) == (0, 0.0)
    assert cli.plan_scale(1.5) == (1, 0.75)
    assert cli.plan_scale(3.0) == (2, 0.75)
    assert cli.plan_scale(2.5) == (2, 0.625)
    assert cli.plan_scale(0.5) == (0, 0.0)      # iter -1, shrink 0.5 / 2^-1 = 1.0: same-size linear resize = identity (main.cpp:107-114)
    assert cli.plan_scale(0.3) == (0, 0.6)      # iter -1: the reference shrinks by 0.6, not 0.3
    assert cli.auto_output_name("/x/pic.v1.jpg", "noise_scale", 2, 2.0) == "/x/pic.v1(noise_scale)(Level2)(x2.000000).png"
    assert cli.auto_output_name("a.png", "scale", 1, 1.5) == "a(scale)(x1.500000).png"
    assert cli.auto_output_name("a.png", "noise", 1, 2.0) == "a(noise)(Level1).png"
    a = cli.build_parser().parse_args(["-i", "in.png"])
    assert (a.output_file, a.mode, a.noise_level, a.scale_ratio, a.model_dir, a.jobs) == ("(auto)", "noise_scale", 1, 2.0, "models", 4)
