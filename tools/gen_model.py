"""Synthetic model generator in the reference's JSON schema (seeded weights for tests, bench.py and the tools; it is a
generator, not the checker, so it lives outside oracle/).

The three shipped models (models/noise1_model.json, noise2_model.json, scale2.0x_model.json) are
stripped from /root/reference (.MISSING_LARGE_BLOBS) and there is no network, so parity and
benchmarks run on seeded synthetic weights of the real topology
(appendix/waifu2x-nocuda/lib/srcnn.lua:13-26: 7 x conv3x3, planes 1-32-32-64-64-128-128-1).

Schema (appendix/waifu2x-nocuda/export_model_nocuda.lua:12-19): a JSON array of
{kW, kH, nInputPlane, nOutputPlane, bias[nOut], weight[nOut][nIn][kH][kW]}.

Weights are drawn as float32 and written with repr-exact decimal strings so that
JSON text -> strtod double -> (float) (modelHandler.cpp:95-97) round-trips bit-exactly.
W ~ N(0, 2 / ((1 + 0.1^2) * 9 * Cin))  (He init with the LeakyReLU(0.1) gain, keeps activations
O(1) through 7 layers), bias ~ U(-0.05, 0.05).
"""
import json
import os
import sys

import numpy as np

TOPOLOGY_WAIFU2X = [1, 32, 32, 64, 64, 128, 128, 1]          # srcnn.lua:13-26
TOPOLOGY_WIDE = [3, 128, 128, 128, 128, 128, 128, 3]         # BASELINE.json configs[4]

SEEDS = {"noise1": 101, "scale2.0x": 102, "noise2": 104, "wide": 105}


def synth_layers(planes=TOPOLOGY_WAIFU2X, seed=102, init="he_leaky"):
    """[(nin, nout, W[o,i,3,3] float32, bias[o] float64)] -- same tuple form as oracle.load_model_json.

    init = "he_leaky"   W ~ N(0, 2 / (1.01 * 9 * Cin)), bias ~ U(-0.05, 0.05): activations stay O(1) (default)
           "upstream"   the init the real models were TRAINED from, appendix/waifu2x-nocuda/lib/srcnn.lua:5-9:
                        W ~ N(0, sqrt(2 / (9 * nOutputPlane))), bias 0 -- fan-OUT scaling, so the activation
                        amplitude drifts from layer to layer instead of staying O(1)
           "wide_range" a trained-model-like weight distribution: per-(o, i) kernel magnitudes log-uniform over 10^3
                        (most kernels tiny, a few large), 30 % of the kernels exactly zero (OpenCV skips zero taps;
                        the result must not change), biases up to +-0.5"""
    rng = np.random.default_rng(seed)
    layers = []
    for nin, nout in zip(planes[:-1], planes[1:]):
        if init == "upstream":
            w = (rng.standard_normal((nout, nin, 3, 3)) * np.sqrt(2.0 / (9.0 * nout))).astype(np.float32)
            b = np.zeros(nout, np.float64)
        elif init == "wide_range":
            sigma = np.sqrt(2.0 / ((1.0 + 0.1 ** 2) * 9.0 * nin))
            mag = 10.0 ** rng.uniform(-3.0, 0.0, size=(nout, nin, 1, 1))
            mag *= (rng.random((nout, nin, 1, 1)) >= 0.3)
            mag /= max(np.sqrt((mag ** 2).mean()), 1e-30)            # keep the layer's overall gain at the He level
            w = (rng.standard_normal((nout, nin, 3, 3)) * sigma * mag).astype(np.float32)
            b = rng.uniform(-0.5, 0.5, size=nout).astype(np.float64)
        else:
            sigma = np.sqrt(2.0 / ((1.0 + 0.1 ** 2) * 9.0 * nin))
            w = (rng.standard_normal((nout, nin, 3, 3)) * sigma).astype(np.float32)
            # biases are stored as double by the reference; keep them float32-representable values
            # widened to double plus a non-representable tail so the (float) cast is exercised
            b = rng.uniform(-0.05, 0.05, size=nout).astype(np.float64)
        layers.append((nin, nout, np.ascontiguousarray(w), np.ascontiguousarray(b)))
    return layers


def _f32_repr(x):
    # shortest decimal string that round-trips the float32 value (and hence, via strtod->(float), too)
    return float(np.format_float_scientific(np.float32(x), unique=True))


def write_json(layers, path):
    objs = []
    for nin, nout, w, b in layers:
        objs.append({
            "kW": 3, "kH": 3, "nInputPlane": nin, "nOutputPlane": nout,
            "bias": [float(v) for v in b],
            "weight": [[[[_f32_repr(w[o, i, r, c]) for c in range(3)] for r in range(3)]
                        for i in range(nin)] for o in range(nout)],
        })
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(objs, f)
    return path


def ensure_model(name, directory, planes=None):
    """Write models/<name>_model.json (seed from SEEDS) if missing; return its path."""
    path = os.path.join(directory, name + "_model.json")
    if not os.path.exists(path):
        planes = planes or (TOPOLOGY_WIDE if name == "wide" else TOPOLOGY_WAIFU2X)
        write_json(synth_layers(planes, SEEDS.get(name, 1)), path)
    return path


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "models"
    for n in ("noise1", "noise2", "scale2.0x"):
        print(ensure_model(n, out))
