"""CPU study: what pose error would a split-precision matrix-core formulation of K2 have?

x = x_hi + x_lo, w = w_hi + w_lo (pieces in f16 or bf16), product = w_hi*x_hi + w_hi*x_lo + w_lo*x_hi
(+ optional more terms), accumulated in f32.  Weights of a layer may be pre-scaled by a power of two.
Imports the oracle, hence it lives under tests/; a development aid, never part of the product path.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import encoder_heads as eh
from nlml_hpe_amd import synth, weights


def to_bf16(a):
    u = np.asarray(a, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split(a, kind, pieces):
    out, rem = [], np.asarray(a, np.float32).copy()
    for _ in range(pieces):
        p = rem.astype(np.float16).astype(np.float32) if kind == "f16" else to_bf16(rem)
        out.append(p)
        rem = rem - p
    return out


def lin(h, w, b, kind, pieces, terms):
    # power-of-two weight scale so the largest |w| lands in [1,2)
    s = 2.0 ** -np.floor(np.log2(np.abs(w).max()))
    wp = split(w * np.float32(s), kind, pieces)
    hp = split(h, kind, pieces)
    acc = np.zeros((h.shape[0], w.shape[0]), np.float32)
    for (i, j) in terms:
        if i < pieces and j < pieces:
            acc += (hp[i].astype(np.float64) @ wp[j].T.astype(np.float64)).astype(np.float32)   # f32 accumulate (approx)
    return acc * np.float32(1.0 / s) + b


def forward(x, p, kind, pieces, terms):
    h = np.asarray(x, np.float32)
    n = len(p.enc)
    for li, (w, b) in enumerate(p.enc):
        h = lin(h, w, b, kind, pieces, terms)
        if li < n - 2: h = np.maximum(h, 0)
        elif li == n - 2: h = np.tanh(h)
    outs = []
    for g, name in enumerate(eh.HEAD_NAMES):
        z = h[:, 3 * g:3 * g + 3]
        for li, (w, b) in enumerate(p.heads[name]):
            z = lin(z, w, b, kind, pieces, terms)
            if li < 4: z = np.maximum(z, 0)
        outs.append(z)
    return np.concatenate(outs, 1)


B = int(os.environ.get("B", "4096"))
heads = weights.load_head_state_dicts("models")
sd = synth.encoder_state_dict(1404, 0)
p = eh.Params(sd, heads)
x = synth.features(B, 1404, 1)
truth = np.degrees(eh.forward_numpy(x, p, np.float64))
f32 = np.degrees(eh.forward_numpy(x, p, np.float32))
print(f"f32 numpy vs f64: max {np.abs(f32 - truth).max():.3e} deg")
T3 = [(0, 0), (0, 1), (1, 0)]
T4 = T3 + [(1, 1)]
T6 = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
for kind, pieces, terms, label in (("f16", 2, T3, "f16 x2 pieces, 3 products"), ("f16", 2, T4, "f16 x2 pieces, 4 products"),
                                   ("bf16", 2, T3, "bf16 x2 pieces, 3 products"), ("bf16", 3, T6, "bf16 x3 pieces, 6 products"),
                                   ("f16", 1, [(0, 0)], "plain f16"), ("bf16", 1, [(0, 0)], "plain bf16")):
    out = np.degrees(forward(x, p, kind, pieces, terms).astype(np.float64))
    e = np.abs(out - truth)
    print(f"{label:32s}: max {e.max():.3e} deg  mean {e.mean():.3e}")
