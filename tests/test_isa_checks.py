"""Static checks on the gfx950 code objects the build produced (CPU-only: llvm-objdump / llvm-readelf on waifu2x-converter-cpp_amd/lib/*.o).

* the hazard "VALU writes an SGPR -> a VMEM instruction reads it: 5 wait states" for VMEM instructions inside asm statements, which the compiler's hazard
  recogniser does not look into (tools/check_sgpr_vmem_hazard.py; round 5's objects had 72 such places);
* no VGPR spill and no scratch in the kernels of the default fp32 frame -- a scratch reload inside a stage waits with vmcnt(0) for every transfer in flight.
"""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "waifu2x-converter-cpp_amd", "lib")
KERNEL_OBJS = ["w2xc_kernels.o", "w2xc_split_t1.o", "w2xc_split_t2.o", "w2xc_split_t3.o", "w2xc_split_t4.o", "w2xc_split_t5.o", "w2xc_wino.o",
               "w2xc_wino4_p.o", "w2xc_wino4_n.o", "w2xc_wino4_f.o", "w2xc_first2_wino4.o", "w2xc_color.o"]   # (csrc/Makefile: the objects built from .hip sources)
OBJS = [os.path.join(LIB, o) for o in KERNEL_OBJS if os.path.exists(os.path.join(LIB, o))]


def _built():
    if len(OBJS) < len(KERNEL_OBJS):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as graft
        graft.build()
    return [os.path.join(LIB, o) for o in KERNEL_OBJS]


@pytest.mark.parametrize("obj", KERNEL_OBJS)
def test_no_sgpr_vmem_hazard_inside_asm_statements(obj):
    _built()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_sgpr_vmem_hazard.py"), os.path.join(LIB, obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "hazards found: 0" in r.stdout


def _resources(obj):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), os.path.join(LIB, obj)], capture_output=True, text=True, check=True).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(\S+)\s+vgpr\s+(\d+) sgpr\s+(\d+) vspill\s+(\d+) sspill\s+(\d+) scratch\s+(\d+)", line)
        if m:
            rows[m.group(1)] = dict(vgpr=int(m.group(2)), vspill=int(m.group(4)), scratch=int(m.group(6)))
    return rows


@pytest.mark.parametrize("obj,pattern,allow_scratch", [
    ("w2xc_wino4_p.o", r"conv3x3_wino4", False),
    ("w2xc_wino4_n.o", r"conv3x3_wino4", False),
    ("w2xc_first2_wino4.o", r"conv3x3_first2_wino4", False),
    # (the PROG instantiations -- ...ELb1ELb1EE -- call w4_prog_job, a real function: its frame is the only scratch in that object)
    ("w2xc_wino4_f.o", r"conv3x3_wino4I.*ELb1ELb0EEv", False),
    ("w2xc_kernels.o", r"conv3x3_(first|last|mfma2)", False),
])
def test_hot_kernels_do_not_spill_vector_registers(obj, pattern, allow_scratch):
    if not os.path.exists(os.path.join(LIB, obj)):
        _built()
    rows = {k: v for k, v in _resources(obj).items() if re.search(pattern, k)}
    assert rows, "no kernel matched %s in %s" % (pattern, obj)
    bad = {k: v for k, v in rows.items() if v["vspill"] != 0 or (v["scratch"] != 0 and not allow_scratch)}
    assert not bad, bad
