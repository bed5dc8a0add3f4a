#!/usr/bin/env python3
"""Would a device-side FACE QUEUE shorten the Powell launch (VERDICT r3 item 5 ii)?  Simulation with MEASURED inputs, no GPU needed:
  * the per-face evaluation counts of BASELINE config 3 in the reference order (profiles/r04_powell_nfev_reference.npy, written by
    tools/powell_nfev_dump.py on the GPU: 4,096 faces, mean 1,629, max 5,966, 6.674 M in all);
  * the cost of one round by number of live machines (tools/powell_round.py reference, microseconds): 17.3 / 21.5 / 29.1 / 35.1 /
    43.2 with 1..5, 63.9 with 8, 125.3 with 16 (linear in between).
Static dealing (today): workgroup b owns faces 16 b .. 16 b + 15 for the whole launch; the launch ends with the slowest workgroup.
Queue: every workgroup has S machine slots; a slot whose machine finished takes the next face from a global counter.
    python tools/powell_queue_sim.py"""
import heapq, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nf = np.load(os.path.join(ROOT, "profiles", "r04_powell_nfev_reference.npy")).astype(int)
pts = {0: 0.0, 1: 17.3, 2: 21.5, 3: 29.1, 4: 35.1, 5: 43.2, 8: 63.9, 16: 125.3}
xs = sorted(pts)
cost = lambda n: float(np.interp(n, xs, [pts[k] for k in xs]))


def static(S=16):
    tmax = tsum = 0.0
    for b in range(0, len(nf), S):
        v, t, prev = np.sort(nf[b:b + S]), 0.0, 0
        for i, r in enumerate(v):
            t += (r - prev) * cost(len(v) - i)
            prev = r
        tmax, tsum = max(tmax, t), tsum + t
    return tmax / 1e3, tsum / 1e3 / (len(nf) / S)


def queue(S, nwg=256):
    nxt, rem = 0, [[] for _ in range(nwg)]
    heap = [(0.0, b) for b in range(nwg)]
    tend = 0.0
    while heap:
        t, b = heapq.heappop(heap)
        r = [v for v in rem[b] if v > 0]
        while len(r) < S and nxt < len(nf):
            r.append(nf[nxt]); nxt += 1
        if not r:
            tend = max(tend, t); continue
        m = min(r)
        rem[b] = [v - m for v in r]
        heapq.heappush(heap, (t + m * cost(len(r)), b))
    return tend / 1e3


print(f"faces {len(nf)}, evaluations {nf.sum()} (mean {nf.mean():.0f}, max {nf.max()})")
print("static dealing, 16 faces per workgroup: launch %.1f ms (mean workgroup %.1f ms)" % static(16))
for S in (4, 6, 8, 10, 12, 16):
    print(f"face queue, {S:2d} slots per workgroup: launch {queue(S):.1f} ms")
print(f"floors: every evaluation at the 16-live rate {nf.sum() * 125.3 / 16 / 256 / 1e3:.1f} ms; the slowest face alone {nf.max() * 17.3 / 1e3:.1f} ms")
