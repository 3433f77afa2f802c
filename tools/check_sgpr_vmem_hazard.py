#!/usr/bin/env python
"""tools/check_sgpr_vmem_hazard.py <object.o> -- gfx9 hazard "VALU writes an SGPR -> a VMEM instruction reads it (address / descriptor / offset): 5 wait
states" for VMEM instructions that sit inside inline asm, which the compiler's hazard recogniser does not look into: every global_* / buffer_* with scalar
operands is checked against the instructions in front of it (v_readlane / v_readfirstlane / VALU compares writing SGPRs; s_nop N counts N + 1 states)."""
import re, subprocess, sys, tempfile, os
obj = sys.argv[1]
T = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % T, obj, "%s/copy.o" % T])   # (explicit output: the input stays untouched)
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=%s/fat.bin" % T,
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=%s/k.co" % T])
txt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "%s/k.co" % T]).decode()
def sregs(tok):
    m = re.match(r's\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r's(\d+)$', tok)
    return {int(m.group(1))} if m else set()
bad = 0
for part in re.split(r'\n(?=[0-9a-f]{16} <)', txt):
    m = re.match(r'[0-9a-f]{16} <([^>]+)>', part)
    if not m: continue
    ins = [re.sub(r'\s*//.*', '', l).strip() for l in part.split('\n')[1:]]
    ins = [l for l in ins if l]
    for i, l in enumerate(ins):
        if not re.match(r'(global_|buffer_|scratch_)', l): continue
        ops = [o.strip() for o in l.split(None, 1)[1].split(',')] if ' ' in l else []
        used = set()
        for o in ops:
            for tok in o.split():
                used |= sregs(tok)
        if not used: continue
        states = 0
        for j in range(i - 1, max(i - 8, -1), -1):
            p = ins[j]
            if states >= 5: break
            mm = re.match(r'(v_readlane_b32|v_readfirstlane_b32)\s+(s\d+)', p)
            wr = sregs(mm.group(2)) if mm else set()
            mm2 = re.match(r'v_cmp\S*\s+(s\[\d+:\d+\])', p)
            if mm2: wr |= sregs(mm2.group(1))
            if wr & used:
                bad += 1
                print("%s: %s  <- %d wait states after: %s" % (m.group(1)[:50], l, states, p))
                break
            mn = re.match(r's_nop (\d+)', p)
            states += (int(mn.group(1)) + 1) if mn else 1
print("hazards found: %d" % bad)
sys.exit(1 if bad else 0)
