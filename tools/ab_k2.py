#!/usr/bin/env python3
"""A/B timing of K2 builds on ONE box: alternates tools/k2_mode_run.py between shared libraries (NLML_HPE_LIB), so that the
chip-to-chip and warm-up differences of a power-limited kernel cancel.
usage: python tools/ab_k2.py <rounds> <mode> <fused|features> <iters> libA.so libB.so [...]   ("-" = the in-tree library)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, mode, path, iters = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
libs = sys.argv[5:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "-":
            env["NLML_HPE_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k2_mode_run.py"), mode, path, "65536", iters],
                             env=env, capture_output=True, text=True, timeout=300)
        m = re.search(r"([\d.]+) ms", out.stdout)
        if not m:
            print(l, "FAILED", out.stdout[-300:], out.stderr[-300:], flush=True)
            continue
        res[l].append(float(m.group(1)))
        print(f"round {r} {l}: {m.group(1)} ms", flush=True)
for l in libs:
    v = sorted(res[l])
    if v:
        print(f"{l}: median {v[len(v)//2]:.4f} ms  min {v[0]:.4f}  ({65536/v[len(v)//2]/1e3:.2f} Mfaces/s)  n={len(v)}")
