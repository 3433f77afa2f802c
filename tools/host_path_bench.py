#!/usr/bin/env python
"""tools/host_path_bench.py -- the host -> host leg of bench.py alone, for several nJob (staging thread) counts.
   python tools/host_path_bench.py [--precision fp32] [--jobs 4,8,16,32]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp32")
ap.add_argument("--jobs", default="4,8,16,32")
a = ap.parse_args()
for j in a.jobs.split(","):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--precision", a.precision,
                          "--jobs", j, "--steps", "10", "--host-steps", "15"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    r = json.loads(out)
    h = r["host_to_host"]
    print("nJob %2s: resident %.3f ms | pageable median %.3f min %.3f (%.4f) | pinned median %.3f (%.4f)" % (
        j, r["ms_per_step"], h["pageable"]["ms_median"], h["pageable"]["ms_min"], h["pageable"]["ratio_vs_resident"],
        h["pinned"]["ms_median"], h["pinned"]["ratio_vs_resident"]), flush=True)
