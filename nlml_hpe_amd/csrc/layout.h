// layout.h -- shared between the host packer (pack.cpp) and the device kernels.
//
// Packed weight blob ("fragment order") for the fused encoder+heads kernel.
//
// The network (NLML_HPE_Model_Builder.py:33-53,76-92) is cut into 11 STAGES; each stage is a
// list of JOBS; one job = NB blocks of 32 output neurons that share one input slice, computed
// by ONE wave for the faces of the tile with v_mfma_f32_32x32x2_f32:
//
//     D[n][face] += A[n][k] * B[k][face],   A = weights (rows), B = activations (columns)
//
// A wave walks K in steps of 8.  In step s lane l (r = l&31, h = l>>5) needs, for each of the
// job's neuron blocks nb, the four weights W[32*nb + r][8s + 4h + j], j = 0..3: MFMA j of the
// step contracts k = 8s+j (lanes 0-31) and k = 8s+4+j (lanes 32-63).  The blob stores exactly
// those 16 bytes per lane, lane-contiguous, so every weight load is one fully coalesced
// 1-KiB global_load_dwordx4 per (step, block):
//
//     wfrag[job][s][nb][lane] : float4
//
// and the bias in the accumulator's register order (C/D map of the 32x32 MFMA: register q of
// lane half h is row (q&3) + 8*(q>>2) + 4*h), so the accumulators are INITIALISED with it:
//
//     bias[job][nb][h][q] : float, q < 16
//
// A tile is 64 faces = two MFMA column blocks ("face blocks") that share every weight fragment
// (each 1-KiB weight load feeds 8 MFMAs), which halves the weight stream per FLOP against a
// 32-face tile -- the stream, not the matrix pipe, was the limit at 32 (profiles/).
//
// Blob = header (256 B) | stage 0 weights | stage 0 bias | stage 1 ... | tail pad.
// Header words (uint32): see Header below.  All offsets are in units of 16 bytes from blob start.
#pragma once
#include <stdint.h>

namespace nlml {

constexpr uint32_t BLOB_MAGIC = 0x4E4C4D4Cu;  // "NLML"
constexpr uint32_t BLOB_VERSION = 3;

constexpr int TILE_FACES = 64;   // faces per workgroup tile = 2 MFMA column blocks of 32
constexpr int NUM_WAVES = 4;
constexpr int NUM_STAGES = 11;   // E0..E5, H0..H4

// Stage ids
enum { ST_E0 = 0, ST_E1, ST_E2, ST_E3, ST_E4, ST_E5, ST_H0, ST_H1, ST_H2, ST_H3, ST_H4 };

struct StageDesc {
  int nb;        // neuron blocks per job (accumulators held at once)
  int jobs;      // jobs in the stage
  int K;         // true contraction length (E0: F, filled at run time)
  int k8;        // K steps of 8 (ceil(K/8))
};

// Static part of the stage table (E0's K/k8 depend on F).
//                                   nb jobs   K   k8
constexpr StageDesc kStages[NUM_STAGES] = {
    /*E0  F   ->1024 relu*/ {4, 8, 0, 0},   // jobs 0-3: neurons 0..511 (pass A), 4-7: 512..1023 (pass B)
    /*E1 1024-> 512 relu*/ {4, 4, 1024, 128},
    /*E2  512-> 256 relu*/ {2, 4, 512, 64},
    /*E3  256-> 128 relu*/ {1, 4, 256, 32},
    /*E4  128->  64 tanh*/ {1, 2, 128, 16},
    /*E5   64->   9 none*/ {1, 1, 64, 8},
    /*H0 3x(3->128) relu*/ {1, 12, 3, 1},
    /*H1 3x(128->256)   */ {2, 12, 128, 16},
    /*H2 3x(256->128)   */ {1, 12, 256, 32},
    /*H3 3x(128->64)    */ {1, 6, 128, 16},
    /*H4 3x(64->1)  none*/ {1, 3, 64, 8},
};

struct Header {
  uint32_t magic;
  uint32_t version;
  uint32_t F;          // encoder input width
  uint32_t mode;       // NLML_MODE_*
  uint32_t k8_e0;      // K steps of layer 0: ceil(F/64)*8 (K zero-padded to whole pairs of 32-column x slabs)
  uint32_t total16;    // blob size in 16-byte units (incl. tail pad)
  uint32_t w_off[NUM_STAGES];  // weights of stage, 16-byte units
  uint32_t b_off[NUM_STAGES];  // bias of stage, 16-byte units
  uint32_t job_w16[NUM_STAGES];  // 16-byte units per job (weights)
  float inv_scale[NUM_STAGES];   // NLML_MODE_F16X2: 2^-e of the stage's power-of-two weight scale (1.0 in the other modes)
  uint32_t reserved[64 - 6 - 4 * NUM_STAGES];
};
static_assert(sizeof(Header) == 256, "header is 256 bytes");

// LDS activation images: [faces][stride] f32, stride = 4*odd => conflict-free ds_read_b128
// (16-lane groups) and ds_write_b128 (8-lane groups); see DESIGN.md "LDS images".
// The encoder trunk (E0..E3) runs on all 64 faces; E4, E5 and the heads run per 32-face block.
constexpr int S_H1H = 516;  // one HALF of E0's output: 512 neurons (pass A or pass B)
constexpr int S_H2 = 516;   // E1 out 512
constexpr int S_H3 = 260;   // E2 out 256
constexpr int S_H4 = 132;   // E3 out 128
constexpr int S_H5 = 68;    // E4 out 64 (tanh)
constexpr int S_LAT = 36;   // E5 out: latent, head g at columns 8g..8g+2, rest exact zeros
constexpr int S_HA = 388;   // H0 out 3 x 128   (32 faces)
constexpr int S_HB = 772;   // H1 out 3 x 256   (32 faces)
constexpr int S_HC = 388;   // H2 out 3 x 128   (32 faces)
constexpr int S_HD = 196;   // H3 out 3 x 64    (32 faces)
constexpr int S_XS = 36;    // E0 input slab: 32 columns of x (+4 pad)
constexpr int XS_COLS = 32;
constexpr int XS_STEPS = XS_COLS / 8;

// LDS offsets (floats).  Lifetimes are sequential; see DESIGN.md for the overlap argument.
constexpr int O_H1H = 0;                          // 64*516 = 33024; pass A, then pass B in place
constexpr int O_XS = 64 * S_H1H;                  // 3 slabs x 64*36 = 6912  -> ends 39936 (dead before LAT is written)
constexpr int O_H2 = 0;                           // written after E1's last K loop (barrier)
constexpr int O_H3 = 0;                           // written after E2's K loop (barrier): 64*260 = 16640
constexpr int O_H4 = O_H3 + 64 * S_H3;            // 16640 .. 25088
constexpr int O_H5 = O_H4 + 64 * S_H4;            // 25088 .. 29440
constexpr int O_LAT = 38656;                      // 64*36 = 2304 -> ends 40960; survives both head passes
constexpr int O_HA = 0;                           // 32*388 = 12416
constexpr int O_HB = O_HA + 32 * S_HA;            // 12416 .. 37120
constexpr int O_HC = 0;
constexpr int O_HD = O_HB;
constexpr int LDS_FLOATS = 40960;                 // 160 KiB
static_assert(O_XS + 3 * 64 * S_XS <= LDS_FLOATS, "LDS map");
static_assert(O_H5 + 64 * S_H5 <= O_LAT, "LDS map");
static_assert(O_HB + 32 * S_HB <= O_LAT, "LDS map");
static_assert(O_LAT + 64 * S_LAT <= LDS_FLOATS, "LDS map");

// ------------------------------------------------------------------------------------------
// bf16 throughput mode (NLML_MODE_BF16): same stages, jobs and bias format; weights are bf16 and a K step
// is 16 (one v_mfma_f32_32x32x16_bf16 per neuron block and face block): lane l (r = l&31, h = l>>5) holds
// the 8 weights W[32*nb + r][16s + 8h + j], j = 0..7 = 16 bytes, so offsets and job sizes keep the same
// 16-byte units with k16 steps in place of k8 steps.  Activations live in LDS as bf16 [face][k] rows with
// stride k+8 elements (16-byte chunks per row odd => conflict-free ds_read_b128).  Layer 0's output for 64
// faces is 128 KB in bf16, so layer 0 runs in ONE pass (8 neuron blocks per wave) and x is read once.
namespace bf {

constexpr StageDesc kStages[NUM_STAGES] = {   // k8 field = K steps of 16 here
    /*E0  F   ->1024 relu*/ {8, 4, 0, 0},
    /*E1 1024-> 512 relu*/ {4, 4, 1024, 64},
    /*E2  512-> 256 relu*/ {2, 4, 512, 32},
    /*E3  256-> 128 relu*/ {1, 4, 256, 16},
    /*E4  128->  64 tanh*/ {1, 2, 128, 8},
    /*E5   64->   9 none*/ {2, 1, 64, 4},     // latent n = 3g+c on row 16g+c: 48 rows = 2 blocks
    /*H0 3x(3->128) relu*/ {1, 12, 3, 1},
    /*H1 3x(128->256)   */ {2, 12, 128, 8},
    /*H2 3x(256->128)   */ {1, 12, 256, 16},
    /*H3 3x(128->64)    */ {1, 6, 128, 8},
    /*H4 3x(64->1)  none*/ {1, 3, 64, 4},
};
constexpr int XS_COLS = 64;             // x slab: 64 columns = 4 K steps of 16
constexpr int XS_STEPS = XS_COLS / 16;
// strides in bf16 elements
constexpr int S_H1 = 1032, S_H2 = 520, S_H3 = 264, S_H4 = 136, S_H5 = 72, S_LAT = 72;   // LAT: 2 blocks = 64 columns
constexpr int S_HA = 392, S_HB = 776, S_HC = 392, S_HD = 200, S_XS = 72;
// LDS offsets in BYTES
constexpr int O_H1 = 0;                                   // 64*1032*2 = 132096
constexpr int O_XS = 64 * S_H1 * 2;                       // 3 slabs x 9216 -> ends 159744
constexpr int O_H2 = 0;                                   // after barrier: 66560
constexpr int O_H3 = 64 * S_H2 * 2;                       // 66560 .. 100352
constexpr int O_H4 = O_H3 + 64 * S_H3 * 2;                // .. 117760
constexpr int O_H5 = O_H4 + 64 * S_H4 * 2;                // .. 126976
constexpr int O_LAT = 163840 - 64 * S_LAT * 2;            // 154624 .. 163840 (x slabs are dead by then)
constexpr int O_HA = 0;                                   // 32*392*2 = 25088
constexpr int O_HB = O_HA + 32 * S_HA * 2;                // .. 74752
constexpr int O_HC = 0;
constexpr int O_HD = O_HB;
static_assert(O_XS + 3 * 64 * S_XS * 2 <= 163840, "LDS map");
static_assert(O_H5 + 64 * S_H5 * 2 <= O_LAT, "LDS map");
static_assert(O_HB + 32 * S_HB * 2 <= O_LAT, "LDS map");

}  // namespace bf

// ------------------------------------------------------------------------------------------
// Split-f16 parity mode (NLML_MODE_F16X2): every f32 operand is carried as TWO f16 pieces, v = hi + lo with
// hi = f16(v), lo = f16(v - hi) (22 significand bits), and a product w*x is evaluated on the f16 matrix cores as
// w_hi*x_hi + w_hi*x_lo + w_lo*x_hi (three v_mfma_f32_32x32x16_f16, f32 accumulate; the dropped w_lo*x_lo term is
// 2^-22 relative).  The weights of a stage are pre-scaled by a power of two (exact) so their lo pieces stay in
// f16's normal range; the accumulators are scaled back (exactly) before the activation.  Bytes per weight and per
// activation are the same 4 as in the f32 mode, so the jobs, the two-pass layer 0 and the LDS budget follow the
// f32 kernel, while a K step is 16 as in the bf16 mode.
//   weights   wfrag[job][k16 step][nb][piece][lane] : 8 x f16 = 16 bytes per lane, piece 0 = hi, 1 = lo
//   LDS image [piece][face][k] f16, row stride K+8 elements (odd number of 16-byte chunks)
namespace hx {

constexpr StageDesc kStages[NUM_STAGES] = {   // k8 field = K steps of 16
    /*E0  F   ->1024 relu*/ {4, 8, 0, 0},     // jobs 0-3: neurons 0..511 (pass A), 4-7: 512..1023 (pass B)
    /*E1 1024-> 512 relu*/ {4, 4, 1024, 64},
    /*E2  512-> 256 relu*/ {2, 4, 512, 32},
    /*E3  256-> 128 relu*/ {1, 4, 256, 16},
    /*E4  128->  64 tanh*/ {1, 2, 128, 8},
    /*E5   64->   9 none*/ {2, 1, 64, 4},     // latent n = 3g+c on row 16g+c
    /*H0 3x(3->128) relu*/ {1, 12, 3, 1},
    /*H1 3x(128->256)   */ {2, 12, 128, 8},
    /*H2 3x(256->128)   */ {1, 12, 256, 16},
    /*H3 3x(128->64)    */ {1, 6, 128, 8},
    /*H4 3x(64->1)  none*/ {1, 3, 64, 4},
};
constexpr int PIECES = 2;
constexpr int XS_COLS = 32;             // x slab: 32 columns = 2 K steps of 16
constexpr int XS_STEPS = XS_COLS / 16;
// row strides in f16 elements
constexpr int S_H1H = 520, S_H2 = 520, S_H3 = 264, S_H4 = 136, S_H5 = 72, S_LAT = 56;   // LAT: columns 16g+c, g<3
constexpr int S_HA = 392, S_HB = 776, S_HC = 392, S_HD = 200, S_XS = 40;
// plane sizes (bytes) and LDS offsets (bytes); the lo plane of an image follows its hi plane
constexpr int P_H1H = 64 * S_H1H * 2, P_H2 = 64 * S_H2 * 2, P_H3 = 64 * S_H3 * 2, P_H4 = 64 * S_H4 * 2;
constexpr int P_H5 = 64 * S_H5 * 2, P_LAT = 64 * S_LAT * 2, P_XS = 64 * S_XS * 2;
constexpr int P_HA = 32 * S_HA * 2, P_HB = 32 * S_HB * 2, P_HC = 32 * S_HC * 2, P_HD = 32 * S_HD * 2;
constexpr int LDS_BYTES = 163840;
constexpr int O_H1H = 0;                                  // 2 x 66560 = 133120; pass A, then pass B in place
constexpr int O_XS = O_H1H + 2 * P_H1H;                   // 3 slabs x 2 planes x 5120 = 30720 -> ends 163840
constexpr int O_H2 = 0;                                   // written after layer 1's last K loop (barrier)
constexpr int O_H3 = 0;                                   // written after layer 2's K loop (barrier): 67584
constexpr int O_H4 = O_H3 + 2 * P_H3;                     // .. 102400
constexpr int O_H5 = O_H4 + 2 * P_H4;                     // .. 120832
constexpr int O_LAT = LDS_BYTES - 2 * P_LAT;              // 149504 .. 163840; survives both head passes
constexpr int O_HA = 0;                                   // 2 x 25088 = 50176
constexpr int O_HB = O_HA + 2 * P_HA;                     // .. 149504
constexpr int O_HC = 0;
constexpr int O_HD = O_HB;
static_assert(O_XS + 3 * 2 * P_XS <= LDS_BYTES, "LDS map");
static_assert(O_H5 + 2 * P_H5 <= O_LAT, "LDS map");
static_assert(O_HB + 2 * P_HB <= O_LAT, "LDS map");
static_assert(O_HD + 2 * P_HD <= O_LAT, "LDS map");

}  // namespace hx

}  // namespace nlml
