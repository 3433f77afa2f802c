"""The C++ drop-in boundary: tests/cpp/dropin_main.cpp is ONE caller written against the reference's
public API (modelUtility::generateModelFromJSON, setNumberOfJobs, convertWithModels, Model::filter);
tests/cpp/Makefile compiles it twice -- against the reference's own sources (dropin_ref) and against
include/w2xc/*.hpp + libw2xc_hip.so (dropin_hip).  Same program, same arguments, compared outputs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_close, rand_plane
from tools import gen_model
from oracle import oracle as orc

CPP = os.path.join(ROOT, "tests", "cpp")
REF_BIN = os.path.join(CPP, "_build", "dropin_ref")
HIP_BIN = os.path.join(CPP, "_build", "dropin_hip")


@pytest.fixture(scope="module")
def bins(w2xc):
    if not (os.path.exists(HIP_BIN) and os.path.exists(REF_BIN)):
        subprocess.run(["make", "-C", CPP], check=True)
    assert os.path.exists(HIP_BIN), "dropin_hip did not build"
    assert os.path.exists(REF_BIN), "dropin_ref is prebuilt where /root/reference exists and travels with the repo"
    return REF_BIN, HIP_BIN


def run_convert(binary, model, plane, tmp, tag, split=1, env=None):
    fin, fout = str(tmp / (tag + "_in.f32")), str(tmp / (tag + "_out.f32"))
    plane.astype(np.float32).tofile(fin)
    h, w = plane.shape
    r = subprocess.run([binary, "convert", model, fin, str(w), str(h), fout, str(split)], capture_output=True, text=True,
                       env=None if env is None else dict(os.environ, **env))
    out = np.fromfile(fout, np.float32).reshape(h, w) if r.returncode == 0 else None
    return r.returncode, out, r.stderr


def test_reference_build_of_the_caller_matches_oracle(bins, models_dir, tmp_path):
    model = os.path.join(models_dir, "noise1_model.json")
    x = rand_plane(30, 41, 3)
    rc, out, err = run_convert(bins[0], model, x, tmp_path, "ref")
    assert rc == 0, err
    assert np.array_equal(out, orc.Oracle.from_json(model).convert(x))


def test_adapter_reports_failure_without_gpu(bins, w2xc, models_dir, tmp_path):
    if w2xc.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu-marked test")
    rc, out, err = run_convert(bins[1], os.path.join(models_dir, "noise1_model.json"), rand_plane(8, 8, 1), tmp_path, "hip")
    assert rc == 1 and out is None              # convertWithModels returned false, nothing was computed on the CPU
    assert "no HIP device" in err
    r = subprocess.run([bins[1], "convert", str(tmp_path / "missing.json"), "x", "1", "1", "y"], capture_output=True, text=True)
    assert r.returncode == 3 and "couldn't open" in r.stderr          # generateModelFromJSON false (:175-179)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,split", [(48, 64, 1), (70, 45, 0), (600, 700, 1)])
def test_same_caller_reference_vs_hip(bins, models_dir, tmp_path, h, w, split):
    model = os.path.join(models_dir, "scale2.0x_model.json")
    x = rand_plane(h, w, h + w)
    rc_r, ref, err_r = run_convert(bins[0], model, x, tmp_path, "ref", split)
    rc_h, hip, err_h = run_convert(bins[1], model, x, tmp_path, "hip", split)
    assert rc_r == 0, err_r
    assert rc_h == 0, err_h
    assert_close(hip, ref, "dropin %dx%d" % (h, w))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp16x2", "bf16x3"])
def test_same_caller_with_precision_from_environment(bins, models_dir, tmp_path, precision):
    """the SAME binary (compiled against include/w2xc/*.hpp, no options anywhere in its source) switched to a
    16-bit split mode by W2XC_PRECISION: still inside the fp32 tolerance against the reference build, and not
    bit-identical to its own fp32 run (i.e. the switch took effect)."""
    model = os.path.join(models_dir, "scale2.0x_model.json")
    x = rand_plane(90, 120, 17)
    rc_r, ref, err_r = run_convert(bins[0], model, x, tmp_path, "ref")
    rc_f, f32, err_f = run_convert(bins[1], model, x, tmp_path, "hip32")
    rc_h, hip, err_h = run_convert(bins[1], model, x, tmp_path, "hip16", env={"W2XC_PRECISION": precision})
    assert rc_r == 0 and rc_f == 0 and rc_h == 0, (err_r, err_f, err_h)
    assert_close(hip, ref, "dropin " + precision)
    assert not np.array_equal(hip, f32)


@pytest.mark.gpu
def test_same_caller_filter(bins, models_dir, tmp_path):
    model = os.path.join(models_dir, "noise1_model.json")
    x = np.random.default_rng(4).standard_normal((32, 20, 33)).astype(np.float32)
    fin = str(tmp_path / "p.f32")
    x.tofile(fin)
    outs = []
    for b, tag in zip(bins, ("ref", "hip")):
        fout = str(tmp_path / (tag + ".f32"))
        r = subprocess.run([b, "filter", model, "1", fin, "32", "33", "20", fout], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(fout, np.float32).reshape(32, 20, 33))
    assert_close(outs[1], outs[0], "dropin filter")
    r = subprocess.run([bins[1], "filter", model, "1", fin, "5", "33", "20", str(tmp_path / "x.f32")], capture_output=True, text=True)
    assert r.returncode == 1 and "number of input planes mismatch" in r.stderr


@pytest.mark.gpu
def test_same_caller_filter_chain_resident_or_not(bins, models_dir, tmp_path):
    """test.cpp:72-85: all 7 layers through Model::filter by hand, the SAME binary with and without W2XC_FILTER_RESIDENT=1 (planes
    handed straight back are taken from the copy still on the GPU) -- bit-identical to each other, inside the tolerance vs the
    reference build of the same caller"""
    model = os.path.join(models_dir, "noise1_model.json")
    x = rand_plane(37, 52, 6)
    fin = str(tmp_path / "c.f32")
    x.tofile(fin)
    outs = {}
    for tag, b, env in (("ref", bins[0], {}), ("hip", bins[1], {}), ("hip_resident", bins[1], {"W2XC_FILTER_RESIDENT": "1"})):
        fout = str(tmp_path / (tag + ".f32"))
        r = subprocess.run([b, "chain", model, fin, "52", "37", fout], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs[tag] = np.fromfile(fout, np.float32).reshape(1, 37, 52)
    assert np.array_equal(outs["hip"], outs["hip_resident"])
    assert_close(outs["hip"], outs["ref"], "dropin filter chain")
