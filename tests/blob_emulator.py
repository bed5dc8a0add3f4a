"""Numpy model of how encoder_heads.hip walks the packed blob (layout.h) -- test helper.

It decodes the blob with the SAME index arithmetic the kernel uses (job offsets from the header,
fragment order [step][block][lane][4], bias in accumulator-register order, LDS column offsets of
every stage) and evaluates the network in float64.  If this model reproduces the oracle, the host
packer and the kernel's addressing scheme agree; the GPU tests then only have to show that the
kernel implements this walk.
"""
import numpy as np

# (nb, jobs, k8) per stage; E0's k8 comes from the header
STAGES = [(4, 8, None), (4, 4, 128), (2, 4, 64), (1, 4, 32), (1, 2, 16), (1, 1, 8),
          (1, 12, 1), (2, 12, 16), (1, 12, 32), (1, 6, 16), (1, 3, 8)]


def _header(blob):
    u = blob[:256].view(np.uint32)
    assert u[0] == 0x4E4C4D4C and u[1] == 2
    return {"F": int(u[2]), "mode": int(u[3]), "k8_e0": int(u[4]), "total16": int(u[5]),
            "w_off": u[6:17].astype(np.int64), "b_off": u[17:28].astype(np.int64), "job_w16": u[28:39].astype(np.int64)}


def _job(blob_f, hdr, stage, job, x_in, k8, nb):
    """x_in f64[32 faces, >= 8*k8] -> acc f64[nb*32 rows, 32 faces] following the MFMA walk."""
    woff = (hdr["w_off"][stage] + job * hdr["job_w16"][stage]) * 4
    w = blob_f[woff: woff + k8 * nb * 64 * 4].reshape(k8, nb, 64, 4).astype(np.float64)
    boff = (hdr["b_off"][stage] + job * nb * 8) * 4
    b = blob_f[boff: boff + nb * 32].reshape(nb, 2, 16).astype(np.float64)
    acc = np.zeros((nb, 32, 32))
    for n in range(nb):
        for h in range(2):
            for q in range(16):
                acc[n, (q & 3) + 8 * (q >> 2) + 4 * h, :] = b[n, h, q]
    for s in range(k8):
        for j in range(4):
            for h in range(2):
                k = 8 * s + 4 * h + j
                # lanes (r, h): A[r][k] = w[s, n, 32h + r, j]; B[k][face] = x_in[face, k]
                acc += w[s, :, 32 * h:32 * h + 32, j][:, :, None] * x_in[None, None, :, k]
    return acc.reshape(nb * 32, 32)


def forward(blob: np.ndarray, x: np.ndarray) -> tuple:
    """x f32[32,F] (one tile) -> (out f64[32,3], latent f64[32,9])."""
    hdr = _header(blob)
    bf = blob.view(np.float32)
    F = hdr["F"]
    k8 = hdr["k8_e0"]
    xin = np.zeros((32, 8 * k8))
    xin[:, :F] = x
    relu = lambda v: np.maximum(v, 0)
    h1 = np.zeros((32, 1024))
    for job in range(8):                    # jobs 0-3: pass A (neurons 0..511), 4-7: pass B; job = 4*pass + wave
        h1[:, 128 * job:128 * job + 128] = relu(_job(bf, hdr, 0, job, xin, k8, 4)).T
    h2 = np.zeros((32, 512))
    for wv in range(4):
        h2[:, 128 * wv:128 * wv + 128] = relu(_job(bf, hdr, 1, wv, h1, 128, 4)).T
    h3 = np.zeros((32, 256))
    for wv in range(4):
        h3[:, 64 * wv:64 * wv + 64] = relu(_job(bf, hdr, 2, wv, h2, 64, 2)).T
    h4 = np.zeros((32, 128))
    for wv in range(4):
        h4[:, 32 * wv:32 * wv + 32] = relu(_job(bf, hdr, 3, wv, h3, 32, 1)).T
    h5 = np.zeros((32, 64))
    for wv in range(2):
        h5[:, 32 * wv:32 * wv + 32] = np.tanh(_job(bf, hdr, 4, wv, h4, 16, 1)).T
    lat = _job(bf, hdr, 5, 0, h5, 8, 1).T                      # [32 faces, 32 cols]: head g at cols 8g..8g+2
    latent = np.stack([lat[:, 8 * (n // 3) + n % 3] for n in range(9)], axis=1)
    ha = np.zeros((32, 384)); hb = np.zeros((32, 768)); hc = np.zeros((32, 384)); hd = np.zeros((32, 192))
    for job in range(12):
        g, nb = job >> 2, job & 3
        ha[:, 128 * g + 32 * nb: 128 * g + 32 * nb + 32] = relu(_job(bf, hdr, 6, job, lat[:, 8 * g:8 * g + 8], 1, 1)).T
    for job in range(12):
        g, p = job >> 2, job & 3
        hb[:, 256 * g + 64 * p: 256 * g + 64 * p + 64] = relu(_job(bf, hdr, 7, job, ha[:, 128 * g:128 * g + 128], 16, 2)).T
    for job in range(12):
        g, nb = job >> 2, job & 3
        hc[:, 128 * g + 32 * nb: 128 * g + 32 * nb + 32] = relu(_job(bf, hdr, 8, job, hb[:, 256 * g:256 * g + 256], 32, 1)).T
    for job in range(6):
        g, nb = job >> 1, job & 1
        hd[:, 64 * g + 32 * nb: 64 * g + 32 * nb + 32] = relu(_job(bf, hdr, 9, job, hc[:, 128 * g:128 * g + 128], 16, 1)).T
    out = np.zeros((32, 3))
    for g in range(3):
        out[:, g] = _job(bf, hdr, 10, g, hd[:, 64 * g:64 * g + 64], 8, 1)[0, :]
    return out, latent
