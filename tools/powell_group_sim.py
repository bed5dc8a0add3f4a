#!/usr/bin/env python3
"""Would CU GROUPS with the objective's weights resident in LDS shorten the Powell launch in the reference order?
Model with MEASURED inputs, no GPU needed (build only if it says <= 0.25 s):

  * evaluation counts of BASELINE config 3 in the reference order: profiles/r04_powell_nfev_reference.npy (4,096 faces, 6.674 M evaluations);
  * today's round cost by live machines (tools/powell_round.py reference, round 4's final build, microseconds): 14.9 / 20.1 / 27.1 /
    32.9 / 40.3 with 1..5, 60.3 with 8, 118.2 with 16 -- i.e. ~7.4 us per evaluation at the issue rate (0.79 of the f64 vector peak);
  * proposal: G CUs per group, each holding the Wm columns of its share of numpy's 16 pairwise leaves in LDS (a leaf = 80 / 88 / 92
    columns; the split must follow leaf boundaries or np.sum's order -- and the bits -- change), all faces of the group evaluated by all
    G CUs, two hand-offs per round through L2 (coefficients out, partial sums back).

A round of the group with L live machines then costs  base + handoffs + L * eval_us * (largest column share / 1404).
    python tools/powell_group_sim.py"""
import os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nf = np.load(os.path.join(ROOT, "profiles", "r04_powell_nfev_reference.npy")).astype(int)
LEAVES = [80] + [88] * 14 + [92]               # the 16 leaves of np.sum over 1,404 elements (tucker_ref.h tr_leaf_start / tr_leaf_len)
assert sum(LEAVES) == 1404
EVAL_US = 7.4                                   # one evaluation on one CU at the measured issue rate (118.2 us / 16)
BASE_US = 4.0                                   # f-vectors, state-machine call, factor table: what a round costs beside the evaluations


def share(G):
    """largest column count of a CU when 16 leaves are dealt to G CUs in contiguous runs"""
    per = [len(a) for a in np.array_split(np.arange(16), G)]
    i, worst = 0, 0
    for n in per:
        worst = max(worst, sum(LEAVES[i:i + n])); i += n
    return worst


def launch_ms(G, handoff_us):
    groups = 256 // G
    per = int(np.ceil(len(nf) / groups))
    frac = share(G) / 1404.0
    worst = 0.0
    for b in range(groups):
        v = np.sort(nf[b * per:(b + 1) * per])
        t, prev = 0.0, 0
        for i, r in enumerate(v):
            live = len(v) - i
            t += (r - prev) * (BASE_US + 2 * handoff_us + live * EVAL_US * frac)
            prev = r
        worst = max(worst, t)
    return worst / 1e3, share(G) * 135 * 4 / 1024.0


print(f"faces {len(nf)}, evaluations {nf.sum()}; today (one CU per 16 faces, Wm streamed from L2): 283 ms measured")
print(f"issue-rate floor (every evaluation at {EVAL_US} us / CU, perfect balance, no round overhead): {nf.sum() * EVAL_US / 256 / 1e3:.1f} ms")
for G in (2, 4, 6, 8, 16):
    row = []
    for h in (0.0, 1.5, 3.0):
        ms, kb = launch_ms(G, h)
        row.append(f"{ms:6.1f}")
    print(f"G = {G:2d} CUs per group ({256 // G:3d} groups x {int(np.ceil(len(nf) / (256 // G))):3d} faces, largest LDS slab {kb:5.1f} KB): launch "
          f"{row[0]} ms with free hand-offs, {row[1]} ms at 1.5 us per hand-off, {row[2]} ms at 3 us")
