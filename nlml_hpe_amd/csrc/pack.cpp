// pack.cpp -- host-side packer: reference state-dict weights -> MFMA fragment-order blob.
// Layout contract: layout.h.  Weight sources follow NLML_HPE_Model_Builder.py:33-53 (encoder)
// and :76-92 (heads); [out,in] row-major as torch.nn.Linear stores them.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "layout.h"

namespace nlml {

namespace {

struct JobSrc {
  const float* W;   // [N][K] row-major
  const float* b;   // [N]
  int K;            // true K
  int N;            // true N
  int row0;         // first neuron of the job (for plain jobs)
  int special;      // 0: rows row0 + i;  1: E5 latent placement;  2: H4 single row
  int lat_step;     // E5: columns reserved per head in the latent image (8 in f32 mode, 16 in bf16 mode)
};

// neuron index feeding accumulator row i (0 .. 32*nb-1) of the job, or -1 for a zero row
inline int row_of(const JobSrc& j, int i) {
  switch (j.special) {
    case 1: {  // E5: latent n = 3g + c lands on row step*g + c -> head g reads columns step*g .. +step-1
      int g = i / j.lat_step, c = i % j.lat_step;
      return (g < 3 && c < 3) ? 3 * g + c : -1;
    }
    case 2:  // H4: the single output neuron on row 0
      return i == 0 ? 0 : -1;
    default: {
      int n = j.row0 + i;
      return n < j.N ? n : -1;
    }
  }
}

}  // namespace

static bool split_f16(int mode) { return mode == NLML_MODE_F16X2 || mode == NLML_MODE_F16X2S; }

// Layer 0's K is padded (zero weights) to whole PAIRS of x slabs (2 x XS_COLS columns), so the kernel's
// slab loop has no partial-slab case and its two staging register sets alternate statically.
static int e0_k8(int F) { return (F + 2 * XS_COLS - 1) / (2 * XS_COLS) * (2 * XS_STEPS); }
// bf16 mode: K steps of 16, x slabs of 64 columns, slab pairs => K padded to 128 columns
static int e0_k16(int F) { return (F + 2 * bf::XS_COLS - 1) / (2 * bf::XS_COLS) * (2 * bf::XS_STEPS); }

// split-f16 mode: K steps of 16, x slabs of 32 columns, slab pairs => K padded to 64 columns
static int e0_k16x2(int F) { return (F + 2 * hx::XS_COLS - 1) / (2 * hx::XS_COLS) * (2 * hx::XS_STEPS); }

static const StageDesc& stage_of(int mode, int s) {
  return mode == NLML_MODE_BF16 ? bf::kStages[s] : (split_f16(mode) ? hx::kStages[s] : kStages[s]);
}
static int e0_steps(int mode, int F) {
  return mode == NLML_MODE_BF16 ? e0_k16(F) : (split_f16(mode) ? e0_k16x2(F) : e0_k8(F));
}
static int pieces_of(int mode) { return split_f16(mode) ? hx::PIECES : 1; }


// round-to-nearest-even f32 -> f16 bits and back (host side of the split; subnormals kept, overflow -> inf)
static uint16_t to_f16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);           // NaN
  if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // >= 65520 rounds to inf
  if (u < 0x33000001u) return (uint16_t)sign;                       // <= 2^-25 rounds to zero
  int e = (int)(u >> 23) - 127;
  uint32_t m = (u & 0x7fffffu) | 0x800000u;                         // 24-bit significand
  int shift = e < -14 ? 13 + (-14 - e) : 13;                        // bits dropped (subnormal: more)
  uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  uint32_t out = e < -14 ? q : (((uint32_t)(e + 15) << 10) + (q - 0x400u));   // carry ripples into the exponent
  return (uint16_t)(sign | out);
}
static float from_f16(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float v;
  if (e == 0) v = (float)m * 5.9604644775390625e-08f;               // 2^-24
  else if (e == 31) { uint32_t u = 0x7f800000u | (m << 13); std::memcpy(&v, &u, 4); }
  else { uint32_t u = ((e + 112u) << 23) | (m << 13); std::memcpy(&v, &u, 4); }
  uint32_t u;
  std::memcpy(&u, &v, 4);
  u |= sign;
  std::memcpy(&v, &u, 4);
  return v;
}

// round-to-nearest-even f32 -> bf16 (NaN stays NaN)
static uint16_t to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

size_t blob_bytes_for(int F, int mode) {
  size_t units = sizeof(Header) / 16;
  for (int s = 0; s < NUM_STAGES; ++s) {
    const StageDesc& d = stage_of(mode, s);
    int k8 = (s == ST_E0) ? e0_steps(mode, F) : d.k8;
    units += (size_t)d.jobs * k8 * d.nb * 64 * pieces_of(mode);      // weights: 16 bytes per lane (and piece)
    units += (size_t)d.jobs * d.nb * 2 * 4;        // bias: 2 halves x 16 floats
  }
  units += 4096;  // 64 KiB tail pad: the K loops prefetch up to 7 steps (<= 8 KiB) past a job's end
  // NLML_MODE_F16X2S: the NLML_MODE_F16X2 image, 256 bytes, then a complete NLML_MODE_F32 image -- what the re-evaluation launch of
  // the strict-fast mode reads (faces whose activations leave f16's range, when a tile has many: encoder_heads.hip).  The mode of a
  // blob is still read off its size.
  if (mode == NLML_MODE_F16X2S) units += 16 + blob_bytes_for(F, NLML_MODE_F32) / 16;
  return units * 16;
}

size_t strict_f32_image_offset(int F) { return blob_bytes_for(F, NLML_MODE_F16X2) + 256; }

int pack_blob(int F, int mode, const float* const enc_w[6], const float* const enc_b[6],
              const float* const head_w[3][5], const float* const head_b[3][5],
              void* blob, size_t blob_bytes) {
  if (F <= 0 || !blob) return fail(NLML_E_BADARG, "pack: bad F or null blob");
  const bool bf16 = mode == NLML_MODE_BF16, f16x2 = split_f16(mode);
  const int kw = (bf16 || f16x2) ? 16 : 8;        // K step width
  const int pieces = pieces_of(mode);
  const int per_half = kw / 2;         // k values per lane half
  const size_t need = blob_bytes_for(F, mode);
  if (blob_bytes < need) return fail(NLML_E_BADARG, "pack: blob buffer too small");
  for (int i = 0; i < 6; ++i)
    if (!enc_w[i] || !enc_b[i]) return fail(NLML_E_BADARG, "pack: null encoder tensor");
  for (int g = 0; g < 3; ++g)
    for (int i = 0; i < 5; ++i)
      if (!head_w[g][i] || !head_b[g][i]) return fail(NLML_E_BADARG, "pack: null head tensor");

  std::memset(blob, 0, blob_bytes);
  Header* hdr = reinterpret_cast<Header*>(blob);
  float* base = reinterpret_cast<float*>(blob);
  hdr->magic = BLOB_MAGIC;
  hdr->version = BLOB_VERSION;
  hdr->F = (uint32_t)F;
  hdr->mode = (uint32_t)mode;
  hdr->k8_e0 = (uint32_t)e0_steps(mode, F);

  const int encN[6] = {1024, 512, 256, 128, 64, 9};
  const int encK[6] = {F, 1024, 512, 256, 128, 64};
  const int headN[5] = {128, 256, 128, 64, 1};
  const int headK[5] = {3, 128, 256, 128, 64};

  size_t cur = sizeof(Header) / 16;  // 16-byte units
  for (int s = 0; s < NUM_STAGES; ++s) {
    const StageDesc& d = stage_of(mode, s);
    const int k8 = (s == ST_E0) ? e0_steps(mode, F) : d.k8;
    const size_t job_w16 = (size_t)k8 * d.nb * 64 * pieces;
    hdr->w_off[s] = (uint32_t)cur;
    // split-f16 mode: one power-of-two scale per stage puts the largest |w| in [128, 256), so the lo pieces of all
    // but the tiniest weights are normal f16 numbers; bias is scaled alike, the kernel multiplies by inv_scale.
    float scale = 1.0f;
    if (f16x2) {
      float mx = 0.0f;
      const int nmat = s <= ST_E5 ? 1 : 3;
      for (int g = 0; g < nmat; ++g) {
        const float* Wsrc = s <= ST_E5 ? enc_w[s] : head_w[g][s - ST_H0];
        const size_t cnt = s <= ST_E5 ? (size_t)encN[s] * encK[s] : (size_t)headN[s - ST_H0] * headK[s - ST_H0];
        for (size_t i = 0; i < cnt; ++i) {
          const float a = std::fabs(Wsrc[i]);
          if (a > mx && a <= 3.0e38f) mx = a;
        }
      }
      if (mx > 0.0f) {
        int ex = 0;
        std::frexp(mx, &ex);                    // mx = m * 2^ex, m in [0.5, 1)
        int e = 8 - ex;                         // mx * 2^e in [128, 256)
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        scale = std::ldexp(1.0f, e);
      }
    }
    hdr->inv_scale[s] = 1.0f / scale;
    hdr->job_w16[s] = (uint32_t)job_w16;
    const size_t b_off = cur + (size_t)d.jobs * job_w16;
    hdr->b_off[s] = (uint32_t)b_off;

    for (int j = 0; j < d.jobs; ++j) {
      JobSrc src{};
      src.lat_step = (bf16 || f16x2) ? 16 : 8;
      if (s <= ST_E5) {
        src.W = enc_w[s]; src.b = enc_b[s]; src.K = encK[s]; src.N = encN[s];
        src.row0 = j * d.nb * 32;
        src.special = (s == ST_E5) ? 1 : 0;
      } else {
        const int li = s - ST_H0;
        const int per_head = d.jobs / 3;
        const int g = j / per_head, sub = j % per_head;
        src.W = head_w[g][li]; src.b = head_b[g][li]; src.K = headK[li]; src.N = headN[li];
        src.row0 = sub * d.nb * 32;
        src.special = (s == ST_H4) ? 2 : 0;
      }
      float* w = base + (cur + (size_t)j * job_w16) * 4;
      for (int st = 0; st < k8; ++st)
        for (int nb = 0; nb < d.nb; ++nb)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = row_of(src, nb * 32 + (lane & 31));
            float* dst = w + ((((size_t)st * d.nb + nb) * pieces) * 64 + lane) * 4;      // 16 bytes per lane
            uint16_t* dst16 = reinterpret_cast<uint16_t*>(dst);
            uint16_t* dst16_lo = dst16 + 64 * 8;                                        // piece 1 follows piece 0
            for (int e = 0; e < per_half; ++e) {
              const int k = kw * st + per_half * (lane >> 5) + e;
              const float v = (n >= 0 && k < src.K) ? src.W[(size_t)n * src.K + k] : 0.0f;
              if (f16x2) {
                const float vs = v * scale;                       // exact: power of two
                const uint16_t hi = to_f16(vs);
                dst16[e] = hi;
                dst16_lo[e] = to_f16(vs - from_f16(hi));          // the difference is exact in f32
              } else if (bf16) {
                dst16[e] = to_bf16(v);
              } else {
                dst[e] = v;
              }
            }
          }
      float* b = base + (b_off + (size_t)j * d.nb * 8) * 4;
      for (int nb = 0; nb < d.nb; ++nb)
        for (int h = 0; h < 2; ++h)
          for (int q = 0; q < 16; ++q) {
            const int n = row_of(src, nb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h);
            b[(nb * 2 + h) * 16 + q] = n >= 0 ? src.b[n] * scale : 0.0f;
          }
    }
    cur = b_off + (size_t)d.jobs * d.nb * 8;
  }
  hdr->total16 = (uint32_t)(need / 16);
  if (mode == NLML_MODE_F16X2S) {   // the f32 image behind the split-f16 one
    const size_t off = strict_f32_image_offset(F);
    if (int rc = pack_blob(F, NLML_MODE_F32, enc_w, enc_b, head_w, head_b, reinterpret_cast<char*>(blob) + off, blob_bytes - off)) return rc;
  }
  if (f16x2) {   // the split-f16 kernels compute these offsets from k16_e0 instead of reading them (encoder_heads_f16x2_dev.h)
    for (int s = 0; s < NUM_STAGES; ++s) {
      const uint32_t k = hdr->k8_e0;
      const uint32_t jw = (s == ST_E0) ? (uint32_t)hx::kStages[s].nb * 64 * hx::PIECES * k
                                       : (uint32_t)hx::kStages[s].k8 * hx::kStages[s].nb * 64 * hx::PIECES;
      uint32_t w = sizeof(Header) / 16;
      for (int t = 0; t < s; ++t) {
        const uint32_t jt = (t == ST_E0) ? (uint32_t)hx::kStages[t].nb * 64 * hx::PIECES * k
                                         : (uint32_t)hx::kStages[t].k8 * hx::kStages[t].nb * 64 * hx::PIECES;
        w += hx::kStages[t].jobs * (jt + hx::kStages[t].nb * 8);
      }
      if (hdr->w_off[s] != w || hdr->job_w16[s] != jw || hdr->b_off[s] != w + hx::kStages[s].jobs * jw)
        return fail(NLML_E_BADARG, "pack: blob layout differs from the split-f16 kernels' offset formula");
    }
  }
  return 0;
}

}  // namespace nlml
