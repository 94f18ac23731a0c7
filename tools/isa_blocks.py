#!/usr/bin/env python3
"""Splits a kernel's gfx950 assembly (hipcc -S) into basic blocks and counts instruction classes per block:
   python3 tools/isa_blocks.py file.s [kernel-substring]
Prints label, line, #VALU (and how many are half-rate / transcendental by the calibration table of DESIGN 5b), #SALU,
#VMEM, #LDS, #branches and the branch targets - enough to attribute a kernel's instruction budget to its source blocks."""
import re, sys
HALF = re.compile(r"^v_(cmp|cmpx|cndmask|min|max|med3|cvt|bfe|perm|and_or|mad_u32|mul_lo|mul_hi|lshl_add|lshl_or|add_lshl|div_|pk_|bfi|alignbit|sad|mad_i32|mad_u64|add3|xad|or3|readlane|readfirstlane|writelane|ldexp|frexp|fract|trunc|floor|rndne|ceil)")
TRANS = re.compile(r"^v_(rcp|sqrt|rsq|exp|log|sin|cos)")
def main():
    path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else None
    lines = open(path).read().split("\n")
    blocks = []; cur = None; active = want is None
    for i, l in enumerate(lines):
        s = l.strip()
        if re.match(r"^[A-Za-z_.$][\w.$]*:", s) and not s.startswith(".L") and want:
            active = want in s
        if not active: continue
        m = re.match(r"^(\.LBB[\w]+|[A-Za-z_][\w.$]*):", s)
        if m:
            cur = dict(label=m.group(1), line=i + 1, valu=0, half=0, trans=0, salu=0, vmem=0, lds=0, smem=0, br=[], wait=0); blocks.append(cur); continue
        if cur is None or not s or s.startswith(";") or s.startswith("."): continue
        op = s.split()[0]
        if op.startswith("v_"):
            cur["valu"] += 1
            if TRANS.match(op): cur["trans"] += 1
            elif HALF.match(op): cur["half"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur["br"].append(op.replace("s_cbranch_", "") + ">" + s.split()[-1]); cur["salu"] += 1
        elif op.startswith("s_waitcnt"): cur["wait"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"): cur["smem"] += 1
        elif op.startswith("s_"): cur["salu"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): cur["vmem"] += 1
        elif op.startswith("ds_"): cur["lds"] += 1
    tot = dict(valu=0, salu=0, vmem=0, lds=0)
    for b in blocks:
        for k in tot: tot[k] += b[k]
        print(f"{b['label']:14s} L{b['line']:6d} valu {b['valu']:4d} (half {b['half']:3d} trans {b['trans']:2d}) salu {b['salu']:3d} vmem {b['vmem']:2d} lds {b['lds']:2d} smem {b['smem']:2d} wait {b['wait']:2d}  {' '.join(b['br'])}")
    print("total", tot)
main()
