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
    assert u[0] == 0x4E4C4D4C and u[1] == 3
    return {"F": int(u[2]), "mode": int(u[3]), "k8_e0": int(u[4]), "total16": int(u[5]),
            "w_off": u[6:17].astype(np.int64), "b_off": u[17:28].astype(np.int64), "job_w16": u[28:39].astype(np.int64),
            "inv_scale": blob[:256].view(np.float32)[39:50].astype(np.float64)}


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


# ---- split-f16 mode (NLML_MODE_F16X2, layout.h namespace hx) -------------------------------------------------
# (nb, jobs, k16) per stage; E0's k16 comes from the header
STAGES_HX = [(4, 8, None), (4, 4, 64), (2, 4, 32), (1, 4, 16), (1, 2, 8), (2, 1, 4),
             (1, 12, 1), (2, 12, 8), (1, 12, 16), (1, 6, 8), (1, 3, 4)]


def _split(v):
    """f32 values -> (hi, lo) as the kernel forms them: hi = f16(v), lo = f16(v - hi); returned as f64."""
    v = np.asarray(v, np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def _job_hx(blob, hdr, stage, job, x_in, k16, nb):
    """x_in f32[32 faces, >= 16*k16] -> f64[nb*32 rows, 32 faces] = inv_scale * (bias' + sum of the three split products)."""
    woff = int(hdr["w_off"][stage] + job * hdr["job_w16"][stage]) * 16
    w = blob[woff: woff + k16 * nb * 2 * 64 * 16].view(np.float16).reshape(k16, nb, 2, 64, 8).astype(np.float64)
    boff = int(hdr["b_off"][stage] + job * nb * 8) * 16
    b = blob[boff: boff + nb * 32 * 4].view(np.float32).reshape(nb, 2, 16).astype(np.float64)
    acc = np.zeros((nb, 32, 32))
    for n in range(nb):
        for h in range(2):
            for q in range(16):
                acc[n, (q & 3) + 8 * (q >> 2) + 4 * h, :] = b[n, h, q]
    xh, xl = _split(x_in)
    for s in range(k16):
        for h in range(2):
            for j in range(8):
                k = 16 * s + 8 * h + j
                whi = w[s, :, 0, 32 * h:32 * h + 32, j][:, :, None]
                wlo = w[s, :, 1, 32 * h:32 * h + 32, j][:, :, None]
                acc += wlo * xh[None, None, :, k] + whi * xl[None, None, :, k] + whi * xh[None, None, :, k]
    return acc.reshape(nb * 32, 32) * hdr["inv_scale"][stage]


def forward_f16x2(blob: np.ndarray, x: np.ndarray) -> tuple:
    """Split-f16 blob walk: x f32[32,F] (one face block) -> (out f64[32,3], latent f64[32,9]).

    Activations pass between layers as f32 (the kernel rounds accumulator*inv_scale to f32 before splitting)."""
    hdr = _header(blob)
    assert hdr["mode"] in (2, 3)      # NLML_MODE_F16X2 / NLML_MODE_F16X2S: one image
    F, k16 = hdr["F"], hdr["k8_e0"]
    f32 = lambda v: np.asarray(v, np.float32)
    xin = np.zeros((32, 16 * k16), np.float32)
    xin[:, :F] = x
    relu = lambda v: np.maximum(v, 0)
    h1 = np.zeros((32, 1024), np.float32)
    for job in range(8):
        h1[:, 128 * job:128 * job + 128] = f32(relu(_job_hx(blob, hdr, 0, job, xin, k16, 4)).T)
    h2 = np.zeros((32, 512), np.float32)
    for wv in range(4):
        h2[:, 128 * wv:128 * wv + 128] = f32(relu(_job_hx(blob, hdr, 1, wv, h1, 64, 4)).T)
    h3 = np.zeros((32, 256), np.float32)
    for wv in range(4):
        h3[:, 64 * wv:64 * wv + 64] = f32(relu(_job_hx(blob, hdr, 2, wv, h2, 32, 2)).T)
    h4 = np.zeros((32, 128), np.float32)
    for wv in range(4):
        h4[:, 32 * wv:32 * wv + 32] = f32(relu(_job_hx(blob, hdr, 3, wv, h3, 16, 1)).T)
    h5 = np.zeros((32, 64), np.float32)
    for wv in range(2):
        h5[:, 32 * wv:32 * wv + 32] = f32(np.tanh(_job_hx(blob, hdr, 4, wv, h4, 8, 1)).T)
    lat = f32(_job_hx(blob, hdr, 5, 0, h5, 4, 2).T)             # [32 faces, 64 cols]: head g at cols 16g..16g+2
    latent = np.stack([lat[:, 16 * (n // 3) + n % 3] for n in range(9)], axis=1).astype(np.float64)
    ha = np.zeros((32, 384), np.float32); hb = np.zeros((32, 768), np.float32)
    hc = np.zeros((32, 384), np.float32); hd = np.zeros((32, 192), np.float32)
    for job in range(12):
        g, nb = job >> 2, job & 3
        ha[:, 128 * g + 32 * nb: 128 * g + 32 * nb + 32] = f32(relu(_job_hx(blob, hdr, 6, job, lat[:, 16 * g:16 * g + 16], 1, 1)).T)
    for job in range(12):
        g, p = job >> 2, job & 3
        hb[:, 256 * g + 64 * p: 256 * g + 64 * p + 64] = f32(relu(_job_hx(blob, hdr, 7, job, ha[:, 128 * g:128 * g + 128], 8, 2)).T)
    for job in range(12):
        g, nb = job >> 2, job & 3
        hc[:, 128 * g + 32 * nb: 128 * g + 32 * nb + 32] = f32(relu(_job_hx(blob, hdr, 8, job, hb[:, 256 * g:256 * g + 256], 16, 1)).T)
    for job in range(6):
        g, nb = job >> 1, job & 1
        hd[:, 64 * g + 32 * nb: 64 * g + 32 * nb + 32] = f32(relu(_job_hx(blob, hdr, 9, job, hc[:, 128 * g:128 * g + 128], 8, 1)).T)
    out = np.zeros((32, 3))
    for g in range(3):
        out[:, g] = _job_hx(blob, hdr, 10, g, hd[:, 64 * g:64 * g + 64], 4, 1)[0, :]
    return out, latent
