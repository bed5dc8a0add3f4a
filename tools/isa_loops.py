#!/usr/bin/env python3
"""Per kernel in a hipcc -S listing: every loop (label ... backward branch to it) with its MFMA / scratch / waitcnt(0) counts.
Usage: isa_loops.py file.s [kernel-name-substring]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
func, labels, lines = None, {}, []
def report():
    if func is None or want not in func:
        return
    print("==", func)
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\S+)", l) or re.match(r"\s+s_branch\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i]
            mf = sum("v_mfma" in b for b in body)
            if mf == 0:
                continue
            sc = sum(b.strip().startswith("scratch_") for b in body)
            vm0 = sum(bool(re.search(r"vmcnt\(0\)", b)) for b in body)
            gl = sum("global_load" in b for b in body)
            gs = sum("global_store" in b for b in body)
            dr = sum("ds_read" in b for b in body)
            dw = sum("ds_write" in b for b in body)
            bar = sum("s_barrier" in b for b in body)
            nop = sum(b.strip().startswith("s_nop") for b in body)
            print(f"  loop {m.group(1)}: {len(body)} lines, mfma {mf}, scratch {sc}, vmcnt(0) {vm0}, gload {gl}, gstore {gs}, ds_read {dr}, ds_write {dw}, barrier {bar}, s_nop {nop}")
for l in src:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        report()
        func, labels, lines = m.group(1), {}, []
        continue
    if func is None:
        continue
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        labels[m.group(1)] = len(lines)
    lines.append(l)
report()
