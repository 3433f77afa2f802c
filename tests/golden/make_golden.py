"""Generate tests/golden/*.npz from the REFERENCE'S OWN code (oracle/_ref = /root/reference's
modelHandler.cpp + convertRoutine.cpp over the OpenCV shim).  Run in the build container, where
/root/reference exists:  python tests/golden/make_golden.py

Each fixture: planes (topology), seed (tools/gen_model.synth_layers), block (singleton block size),
input plane, and the reference's convertWithModels output.  Small on purpose (<100 KB each)."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tools import gen_model
from oracle import oracle as orc  # noqa: E402

CASES = [
    # name, planes, seed, (h, w), block, weight init (tools/gen_model.synth_layers), input amplitude
    ("waifu2x_48x64", gen_model.TOPOLOGY_WAIFU2X, 101, (48, 64), 512, "he_leaky", 1.0),
    ("waifu2x_split_60x41_b40", gen_model.TOPOLOGY_WAIFU2X, 102, (60, 41), 40, "he_leaky", 1.0),
    ("tiny3_17x19", [1, 4, 6, 1], 9, (17, 19), 512, "he_leaky", 1.0),
    ("odd_1x1", gen_model.TOPOLOGY_WAIFU2X, 104, (1, 1), 512, "he_leaky", 1.0),
    # weight statistics the He-init models do not have: the init the shipped models were trained from (srcnn.lua:5-9) and a
    # trained-model-like 10^3 dynamic range with exactly-zero kernels; a full-range and a dark plane
    ("upstream_init_40x56", gen_model.TOPOLOGY_WAIFU2X, 33, (40, 56), 512, "upstream", 1.0),
    ("wide_range_40x56", gen_model.TOPOLOGY_WAIFU2X, 33, (40, 56), 512, "wide_range", 1.0),
    ("upstream_init_dark_33x47", gen_model.TOPOLOGY_WAIFU2X, 34, (33, 47), 512, "upstream", 1.0 / 255.0),
]


def main():
    orc.build()
    for name, planes, seed, (h, w), block, init, amp in CASES:
        layers = gen_model.synth_layers(planes, seed, init=init)
        with tempfile.TemporaryDirectory() as d:
            p = gen_model.write_json(layers, os.path.join(d, "m.json"))
            ref = orc.RefBuild(p)
            x = np.random.default_rng(seed + 1000).random((h, w), dtype=np.float32) * np.float32(amp)
            y = ref.convert(x, block=(block, block))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), planes=np.array(planes), seed=seed, block=block,
                            init=init, input=x, output=y)
        print(name, y.shape, float(np.abs(y).max()))


if __name__ == "__main__":
    main()
