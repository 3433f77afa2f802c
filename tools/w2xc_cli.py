#!/usr/bin/env python
"""tools/w2xc_cli.py -- N4: a parity shell for the reference CLI (/root/reference/src/main.cpp) on top of
libw2xc_hip, for machines without OpenCV.  Same flags (main.cpp:26-60), same mode / ratio logic
(:83-121, iter = ceil(log2 ratio), shrink iff int(ratio) != 2^iter, :107-114,158-167), same automatic output
name (:173-189).  Image I/O is PIL instead of cv::imread/imwrite; everything between -- convertTo,
RGB2YUV on BGR data (Q3), noise pass, nearest/bicubic 2x + CNN, linear shrink, YUV2RGB, saturate to uint8 --
runs on the GPU in one call (w2xc_process_image_u8_ex)."""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def plan_scale(ratio):
    """(iterations, shrink_ratio) exactly as main.cpp:107-114 computes them (shrink 0.0 = none)."""
    if not ratio > 0:
        raise ValueError("scale_ratio must be positive")
    it = int(math.ceil(math.log2(ratio)))        # may be NEGATIVE for ratio < 1: the reference does not clamp it (:107-108)
    shrink = 0.0
    if int(ratio) != 2.0 ** it:                   # static_cast<int>(ratio) != std::pow(2, iter)
        shrink = ratio / 2.0 ** it                # e.g. ratio 0.3 -> iter -1 -> shrink 0.6 (NOT 0.3), ratio 0.5 -> shrink 1.0
    if shrink == 1.0:
        shrink = 0.0                              # cv::resize to the same size with INTER_LINEAR is the identity
    return max(it, 0), shrink                     # the 2x loop runs max(iter, 0) times (:126)


def auto_output_name(input_file, mode, noise_level, scale_ratio):
    """main.cpp:173-189 (std::to_string(double) prints six decimals)."""
    dot = input_file.rfind(".")
    name = input_file[:dot] if dot >= 0 else input_file
    name += "(" + mode + ")"
    if "noise" in mode:
        name += "(Level" + str(noise_level) + ")"
    if "scale" in mode:
        name += "(x" + "%.6f" % scale_ratio + ")"
    return name + ".png"


def build_parser():
    ap = argparse.ArgumentParser(description="waifu2x reimplementation using libw2xc_hip (MI355X)")
    ap.add_argument("-i", "--input_file", required=True, help="path to input image file (you should input full path)")
    ap.add_argument("-o", "--output_file", default="(auto)", help="path to output image file (you should input full path)")
    ap.add_argument("-m", "--mode", default="noise_scale", choices=["noise", "scale", "noise_scale"], help="image processing mode")
    ap.add_argument("--noise_level", type=int, default=1, choices=[1, 2], help="noise reduction level")
    ap.add_argument("--scale_ratio", type=float, default=2.0, help="custom scale ratio")
    ap.add_argument("--model_dir", default="models", help="path to custom model directory (don't append last / )")
    ap.add_argument("-j", "--jobs", type=int, default=4, help="number of threads launching at the same time")
    # not a reference flag: w2xc_opts.precision of the engine (fp32 = the reference's arithmetic on the fp32 MFMA)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "fp16x2", "bf16x2", "bf16"],
                    help="engine arithmetic for the CNN layers (extension; default fp32)")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    from PIL import Image
    import __graft_entry__ as graft
    w2xc = graft.load_package()

    img = np.asarray(Image.open(args.input_file).convert("RGB"))
    bgr = np.ascontiguousarray(img[:, :, ::-1])                       # cv::imread(IMREAD_COLOR) order (Q3)
    w2xc.modelUtility.getInstance().setNumberOfJobs(args.jobs)        # :79

    noise = scale = None
    if args.mode in ("noise", "noise_scale"):                         # :83-89
        noise = w2xc._ModelSet.from_json(os.path.join(args.model_dir, "noise%d_model.json" % args.noise_level))
    iterations, shrink = 0, 0.0
    if args.mode in ("scale", "noise_scale"):                         # :103-121
        iterations, shrink = plan_scale(args.scale_ratio)
        scale = w2xc._ModelSet.from_json(os.path.join(args.model_dir, "scale2.0x_model.json"))
        print("start scaling")
    if noise is None and iterations == 0 and not shrink:
        out = bgr                                                      # ratio 1.0 in scale mode: nothing to do
    elif iterations == 0 and noise is None:
        raise SystemExit("scale_ratio %g needs no 2x step; the reference would only shrink, which is not supported without a model pass" % args.scale_ratio)
    else:
        prec = {"fp32": w2xc.PRECISION_FP32, "bf16": w2xc.PRECISION_BF16, "bf16x2": w2xc.PRECISION_BF16X2, "bf16x3": w2xc.PRECISION_BF16X3, "fp16x2": w2xc.PRECISION_FP16X2}[args.precision]
        opts = w2xc.make_opts(precision=prec)      # always explicit: an explicit --precision beats the W2XC_PRECISION env default
        out = w2xc.process_image_u8(bgr, noise, scale if iterations else None, iterations, opts, shrink)
    name = args.output_file
    if name == "(auto)":
        name = auto_output_name(args.input_file, args.mode, args.noise_level, args.scale_ratio)
    Image.fromarray(np.ascontiguousarray(out[:, :, ::-1])).save(name)
    print("process successfully done!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
