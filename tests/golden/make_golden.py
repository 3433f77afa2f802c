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
    # name, planes, seed, (h, w), block
    ("waifu2x_48x64", gen_model.TOPOLOGY_WAIFU2X, 101, (48, 64), 512),
    ("waifu2x_split_60x41_b40", gen_model.TOPOLOGY_WAIFU2X, 102, (60, 41), 40),
    ("tiny3_17x19", [1, 4, 6, 1], 9, (17, 19), 512),
    ("odd_1x1", gen_model.TOPOLOGY_WAIFU2X, 104, (1, 1), 512),
]


def main():
    orc.build()
    for name, planes, seed, (h, w), block in CASES:
        layers = gen_model.synth_layers(planes, seed)
        with tempfile.TemporaryDirectory() as d:
            p = gen_model.write_json(layers, os.path.join(d, "m.json"))
            ref = orc.RefBuild(p)
            x = np.random.default_rng(seed + 1000).random((h, w), dtype=np.float32)
            y = ref.convert(x, block=(block, block))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), planes=np.array(planes), seed=seed, block=block,
                            input=x, output=y)
        print(name, y.shape, float(np.abs(y).max()))


if __name__ == "__main__":
    main()
