/*
 * nlml_hpe.h -- C ABI of libnlml_hpe_hip.so: the MI355X (gfx950) implementation of the
 * NLML_HPE batched-inference hot path.
 *
 * The reference (MahdiGhafoorian/NLML_HPE) is pure Python on stock ATen / numpy and has no
 * FFI of its own; each entry point below names the reference call it replaces (file:line
 * under /root/reference) so a maintainer can bind it from the same place (ctypes stubs in
 * INTEGRATION.md).  Conventions for every function:
 *
 *   - plain pointers and sizes only; all data pointers are DEVICE pointers unless the
 *     parameter name starts with "h_" (host);
 *   - the caller owns every buffer; nothing is allocated or freed inside a launch function;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); launches are
 *     asynchronous on it and may be captured into a hipGraph;
 *   - return value 0 on success, otherwise a negative NLML_E_* code or a positive
 *     hipError_t; nlml_last_error() returns a thread-local message for the last failure;
 *   - thread-safe for distinct streams; no global mutable state.
 */
#ifndef NLML_HPE_H
#define NLML_HPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLML_ABI_VERSION 2   /* 2: the TD entry points without _ex default to NLML_TD_ORDER_REFERENCE */

#define NLML_E_BADARG   (-1)  /* null pointer, negative size, misaligned buffer           */
#define NLML_E_BADBLOB  (-2)  /* packed weight blob has wrong magic / version / F         */
#define NLML_E_SHAPE    (-3)  /* architecture other than the reference's (see pack)      */

/* Fixed architecture constants of the reference model. */
#define NLML_NUM_LANDMARKS 468   /* helpers/FeatureExtractor.py:106                      */
#define NLML_F_REFERENCE   1404  /* configs/config_EncoderTrainer.yaml:19                */
#define NLML_LATENT        9     /* 3 x (1,3) matrices, NLML_HPE_Model_Builder.py:187-195 */
#define NLML_TUCKER_Q      135   /* 5*3*3*3 rows of W, TD_main.py:232-238                */

int         nlml_abi_version(void);
const char* nlml_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  IPD landmark normalisation.
 * Replaces Read_Landmarks_and_Normalizing_using_IPD (helpers/FeatureExtractor.py:30-66) plus
 * the f32 cast of its callers (:101,:142,:187), batched over faces.
 *   raw      f32[B,468,3]  FaceMesh coordinates (x,y,z interleaved)
 *   normalize != 0: out = f32( (f64(v) - f64(raw[1][c])) / ipd ),  ipd = ||raw[33]-raw[263]||
 *                   in f64, replaced by 1e-6 when exactly 0 (:47-48); == 0: out = raw.
 *   out      f32[B,1404]
 *   valid    u8[B] or NULL: 0 where the OUTPUT row is all zero -- the reference's "no face"
 *            sentinel (FeatureExtractor.py:105-106; callers skip: NLML_HPE_Test.py:257-260).
 * Bit-exact with the reference (f64 subtract and divide, one rounding to f32).
 */
int nlml_normalize_ipd(const float* raw, int64_t B, int normalize,
                       float* out, uint8_t* valid, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  Encoder + three heads, fused forward.
 * Replaces CombinedAnglePredictionModel.forward (NLML_HPE_Model_Builder.py:115-126), i.e.
 * LandmarkEncoder.forward (:55-68) and 3 x AnglePredictionNetwork.forward (:104-105), called
 * as model(x) at NLML_HPE_Test.py:272 / generatePose_on_video.py:209.
 *
 * Weights are packed ONCE on the host into the kernel's MFMA fragment order:
 *   enc_w[i]/enc_b[i], i<6 : encoder.{0,2,4,6,8,10}.{weight,bias}, weight [out,in] row-major
 *                            with shapes (1024,F) (512,1024) (256,512) (128,256) (64,128) (9,64)
 *   head_w[g][i]/head_b[g][i], g<3 (yaw,pitch,roll), i<5 : model.{0,2,4,6,8}.{weight,bias},
 *                            shapes (128,3) (256,128) (128,256) (64,128) (1,64)
 * (the state-dict layout of models/Encoder.pth and models/{yaw,pitch,roll}_network.pth).
 * mode: NLML_MODE_F32 = f32 storage + f32 MFMA, layers 0 to 3 summed in blocks of 128 k (the strict parity mode: at the
 *       reference's operating range -- poses to +-60 deg, FX3c, 16,384 faces -- its distance from the exact result is
 *       1.25e-5 / 4.0e-5 / 8.7e-5 deg in p50 / p99 / max, no worse than the reference's own 1.7e-5 / 5.5e-5 / 9.9e-5);
 *       NLML_MODE_BF16 = bf16 weights and activations + bf16 MFMA, f32 accumulate (throughput mode:
 *       ~5x the faces/s; its error is ~0.1 deg max / 0.02 deg mean and is never claimed as parity);
 *       NLML_MODE_F16X2 = split-f16 mode, OPT-IN, 1.10x THE REFERENCE'S ERROR: every f32 weight and activation is carried as two f16
 *       pieces (hi + lo, 22 significand bits) and a product runs as three f16 MFMAs with f32 accumulation.
 *       ~1e-5 deg from the reference on small poses; at the reference's operating range (FX3c) 1.86e-5 / 5.9e-5 / 1.22e-4 deg
 *       from the exact result in p50 / p99 / max = 1.10 / 1.08 / 1.24x the reference's own distance, 0.024 % of the faces beyond 1e-4
 *       deg; ~3x NLML_MODE_F32's faces/s (the default of this build's host layer until round 3; NLML_MODE_F16X2S since).  No input-range limit: a
 *       face whose activations leave f16's range (|v| >= 65520 -- e.g. the reference's ipd == 0 -> 1e-6 branch,
 *       FeatureExtractor.py:47-48) is re-evaluated inside the same launch in f32 on the vector ALUs from the same
 *       blob (csrc/encoder_heads_f16x2_rescue.h); NaN/Inf inputs give a non-finite pose, as in the reference.
 *       COST OF THAT SLOW PATH: rescued faces go four at a time, each group streaming the whole 9.6 MB blob through its CU
 *       (~0.15 ms).  A handful of such faces per launch is free; a batch in which EVERY face overflows (un-normalised
 *       pixel-scale landmarks against trained-scale weights) runs ~40x slower: 6.1 ms instead of 0.16 ms for 4,096 faces
 *       (0.67 M faces/s; tests/test_gpu_parity.py::test_split_f16_every_face_of_a_tile_overflows).  Feed such data to
 *       NLML_MODE_F32, which has no range limit and no cliff.  Rescued faces use the blob's weights as hi + lo, i.e. 22
 *       significand bits, not the original f32 weights: ~2x torch-f32's distance from the exact result on such inputs.
 *       NLML_MODE_F16X2S = the same operands and the same three MFMAs per product, but the two SMALL products of a K step
 *       (w_lo*x_hi, w_hi*x_lo) accumulate in registers of their own in layers 0 to 2 and join the big sum once per K
 *       block: the matrix instruction truncates its products to the running sum's exponent, which is what costs
 *       NLML_MODE_F16X2 its distance (profiles/r03_mfma_f16_numerics_probe.txt; that mode has room for the second accumulator set only from
 *       layer 1's second K half on).  THE DEFAULT of the host layer.  Parity at the operating range (FX3c, 16,384 faces): 1.50e-5 /
 *       4.65e-5 / 9.2e-5 deg from the exact result = 0.88 / 0.85 / 0.93x the PINNED reference's own distance (fixture generated in the
 *       build container) but 1.14 / 1.12 / 1.10x torch-f32's on the GPU box's host (1,048,576 faces, profiles/r04_parity_soak_1M.json:
 *       the reference's distance depends on its host's GEMM blocking); 0.03-0.07 % of the faces differ from the reference's batched
 *       OUTPUT by more than 1e-4 deg (NLML_MODE_F32: 0.002-0.012 %, what the reference shows against itself): the looser of the two
 *       parity-class modes -- NLML_MODE_F32 is the one to use where 1e-4 deg must hold face by face.  ~0.9x NLML_MODE_F16X2's faces/s (an eight-wave kernel,
 *       csrc/encoder_heads_f16x2_w8.hip; layer 0 runs in two passes over x to make room for the second accumulator set).
 *       Packed image: NLML_MODE_F16X2's, 256 bytes, then a complete NLML_MODE_F32 image (the size still names the mode).
 *       Range behaviour: a tile with faces beyond f16's range is re-evaluated whole on the f32 matrix cores from the f32 image by a
 *       second launch that the forward entry points enqueue behind the kernel (csrc/encoder_heads.hip, re-evaluation mode; it ends at
 *       once for every other tile; only the out-of-range faces are written): the strict parity kernel's bits on those faces, and a
 *       batch in which EVERY face overflows runs 3x slower, not 40x.
 * The forward entry points recognise the mode of a blob by its size.
 */
#define NLML_MODE_F32   0
#define NLML_MODE_BF16  1
#define NLML_MODE_F16X2 2
#define NLML_MODE_F16X2S 3

size_t nlml_encoder_heads_packed_bytes(int F, int mode);
int    nlml_encoder_heads_pack(int F, int mode,
                               const float* const h_enc_w[6], const float* const h_enc_b[6],
                               const float* const h_head_w[3][5], const float* const h_head_b[3][5],
                               void* h_blob, size_t blob_bytes);

/*   x       f32[B,F], row stride ldx floats (ldx >= F)
 *   blob    device copy of the packed blob (16-byte aligned)
 *   out     f32[B,3]  (yaw, pitch, roll) in RADIANS -- the three [B,1] outputs of the
 *           reference side by side (rad->deg is left to the caller, Model_Builder.py:125)
 *   latent  f32[B,9] or NULL: the encoder output before the split (:58)
 *   valid   u8[B] or NULL: 0 where the input row is all zero -- the "no face" sentinel the
 *           reference's callers test per face (NLML_HPE_Test.py:257-260); the pose is still
 *           computed for such rows, as the reference's forward would.
 */
int nlml_encoder_heads_fwd(const float* x, int64_t ldx, int64_t B, int F,
                           const void* blob, size_t blob_bytes,
                           float* out, float* latent, uint8_t* valid, void* stream);

/* Diagnostic build of nlml_encoder_heads_fwd (x 16-byte aligned, F % 4 == 0).  It also writes
 *   pre_tanh f32[B,64] (or NULL): the Linear(128,64) outputs BEFORE the Tanh (Model_Builder.py:49-50);
 *            everything up to there is pure f32 fma, so it is compared bit for bit against the C
 *            oracle's fmaf chain (tests/test_gpu_parity.py);
 *   stamps   u64[ceil(B/64),4,16] (or NULL): per-wave s_memtime at the stage boundaries, for the
 *            cycle-share breakdown in profiles/ (never a run-time figure). */
int nlml_encoder_heads_fwd_debug(const float* x, int64_t ldx, int64_t B, int F,
                                 const void* blob, size_t blob_bytes,
                                 float* out, float* latent, float* pre_tanh,
                                 unsigned long long* stamps, void* stream);

/* Fused K1+K2: raw landmarks in, pose out; the normalised features never touch HBM.
 *   raw f32[B,468,3]; valid u8[B] or NULL as in nlml_normalize_ipd. F must be 1404. */
int nlml_landmarks_to_pose(const float* raw, int64_t B, int normalize,
                           const void* blob, size_t blob_bytes,
                           float* out, float* latent, uint8_t* valid, void* stream);

/* The same forward for SMALL batches in NLML_MODE_F16X2 / NLML_MODE_F16X2S: the three big layers as one launch each over (neuron blocks x
 * 64-face tiles) plus one launch for the tail, instead of one CU per tile, so 64 or 2,000 faces use the whole chip (a
 * 64-face video tick: 0.06 ms instead of 0.17 ms); the results are bit-identical to nlml_encoder_heads_fwd / nlml_landmarks_to_pose with the same blob.
 * Activations pass between the launches through `workspace` (device memory, 16-byte aligned, at least
 * nlml_encoder_heads_small_workspace_bytes(B, F) bytes, contents irrelevant before and after); stream order is the only
 * synchronisation, so the sequence can be captured into a hipGraph.  Above ~4,500 faces the fused entry points are faster.
 */
size_t nlml_encoder_heads_small_workspace_bytes(int64_t B, int F);
int nlml_encoder_heads_fwd_small(const float* x, int64_t ldx, int64_t B, int F,
                                 const void* blob, size_t blob_bytes, float* out, float* latent, uint8_t* valid,
                                 void* workspace, size_t ws_bytes, void* stream);
int nlml_landmarks_to_pose_small(const float* raw, int64_t B, int normalize,
                                 const void* blob, size_t blob_bytes, float* out, float* latent, uint8_t* valid,
                                 void* workspace, size_t ws_bytes, void* stream);

/* The same forward in NLML_MODE_F16X2S as TRUNK LAUNCH + STREAMED TAIL LAUNCH (+ the f32 re-evaluation launch): layers 0-2 by the
 * eight-wave kernel, which ends with layer 2's output in `workspace` as MFMA operand fragments (1 KB per face), then layers E3..E5 and
 * the three heads (NLML_HPE_Model_Builder.py:45-53,76-92) by a kernel in which a wave keeps a 32-face block's activations in registers
 * through the whole tail while the tail's 1.06 MB of weights pass through LDS once per 256 faces (csrc/encoder_heads_f16x2_tailws.hip)
 * -- in the fused kernel they stream once per 64 faces through thirteen barrier-separated stages on half the CU's waves.
 * MEASURED 1.4-2.2 % faster than the fused kernel at 65,536 faces (0.801 against 0.815 ms, same box; the trunk launch alone costs 0.90 of the
 * fused kernel = its share of the L2 -> CU bytes; below ~60,000 faces the path LOSES: -5 % at 32,768, -16 % at 16,384): inside
 * the box-to-box spread, so nothing picks it by default
 * (NLML_K2_STREAMED_MIN=<faces> makes the _ws entry points route batches from that size on through it).
 * Bit-identical to nlml_encoder_heads_fwd / nlml_landmarks_to_pose with the same blob.  Input layout: F % 4 == 0, rows 16-byte aligned
 * (ldx % 4 == 0) -- else NLML_E_BADARG.  `workspace`: nlml_encoder_heads_workspace_bytes(B, F) bytes, 16-byte aligned.
 */
int nlml_encoder_heads_fwd_streamed(const float* x, int64_t ldx, int64_t B, int F,
                                    const void* blob, size_t blob_bytes, float* out, float* latent, uint8_t* valid,
                                    void* workspace, size_t ws_bytes, void* stream);
int nlml_landmarks_to_pose_streamed(const float* raw, int64_t B, int normalize,
                                    const void* blob, size_t blob_bytes, float* out, float* latent, uint8_t* valid,
                                    void* workspace, size_t ws_bytes, void* stream);

/* THE FORWARD WITH A WORKSPACE: picks the fastest of the paths above for the batch size and the blob's mode (split-f16 modes: the
 * layer-per-launch path up to 4,096 faces; the fused kernel otherwise and for the other modes, which ignore the workspace; the
 * trunk + streamed-tail path only when the environment asks for it, NLML_K2_STREAMED_MIN=<faces>).  Same bits whichever path runs.  For hosts that
 * want one call for every batch size; the packaged host layer makes the same choice in Python (nlml_hpe_amd/model.py, `small_batch_max`)
 * and calls the plain / _small forms, which is also what bench.py times.
 * `workspace`: at least nlml_encoder_heads_workspace_bytes(B, F) bytes (>= the _small and _streamed paths' needs), 16-byte aligned.
 */
size_t nlml_encoder_heads_workspace_bytes(int64_t B, int F);
int nlml_encoder_heads_fwd_ws(const float* x, int64_t ldx, int64_t B, int F,
                              const void* blob, size_t blob_bytes, float* out, float* latent, uint8_t* valid,
                              void* workspace, size_t ws_bytes, void* stream);
int nlml_landmarks_to_pose_ws(const float* raw, int64_t B, int normalize,
                              const void* blob, size_t blob_bytes, float* out, float* latent, uint8_t* valid,
                              void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3  Tucker objective, batched over face-evaluations.
 * Replaces objective() (TD_Tester.py:31-58): f = a*cos(b*w+c)+d (:25-28) in f64 rounded to
 * f32 (:37,40,43); x_hat = einsum('ijklm,i,j,k,l->m', W, u, f_y, f_p, f_r) in f64 (:46);
 * err = 0.5*sum((x-x_hat)^2) (:49).
 *   Wm         f32[135,1404]  = W.reshape(-1,1404) of outputs/features/Trained_data.npz
 *   x          f32[N,1404]    feature rows, row stride ldx
 *   x_index    i32[N] or NULL: evaluation n uses row x_index[n] of x (several evaluations of
 *              one face share its row); NULL = row n
 *   params     f64[N,8]       (w_y, w_p, w_r, u_id[5]) per evaluation (TD_Tester.py:32-33)
 *   cos_params f64[3,3,4]     optimized_{yaw,pitch,roll}[0:3,:] rows (a,b,c,d) (TD_Inference.py:56)
 *   err        f64[N]
 *   x_hat      f64[N,1404] or NULL
 *
 * OPERATION ORDER.  The reference's objective is a fixed sequence of separately rounded f64 operations, and the Powell
 * minimisation on top of it is sensitive to its last bits (the minimum is flat), so the order is part of the contract:
 *   NLML_TD_ORDER_REFERENCE  (the default of the entry points without _ex: the PARITY mode)
 *       np.einsum('ijklm,i,j,k,l->m')'s own loop -- for (i,j,k,l) in nesting order and every m,
 *       x_hat[m] = ((((W*u_i)*f_yj)*f_pk)*f_rl) + x_hat[m], each operation rounded on its own -- and numpy's pairwise np.sum
 *       (TD_Tester.py:46,49): err and x_hat are BIT-IDENTICAL to the reference's (FX4).  Vector ALUs, 5 operations per (q, m).
 *       Alignment: Wm and x 4-byte aligned, any ldx >= 1404.
 *   NLML_TD_ORDER_FAST       (opt-in)
 *       x_hat = c^T Wm with c = ((u*f_y)*f_p)*f_r as a GEMM on the f64 matrix cores, one fma chain per output; agrees with the
 *       reference's objective to <= 1e-12 relative (measured ~2e-16), ~5x the evaluations/s.
 *       Alignment: Wm and x 16-byte aligned and ldx % 4 == 0 (16-byte vector loads), else NLML_E_BADARG.
 */
#define NLML_TD_ORDER_FAST      0
#define NLML_TD_ORDER_REFERENCE 1
int nlml_tucker_objective(const float* Wm, const float* x, int64_t ldx, const int32_t* x_index,
                          const double* params, const double* cos_params, int64_t N,
                          double* err, double* x_hat, void* stream);          /* == _ex(..., NLML_TD_ORDER_REFERENCE, stream) */
int nlml_tucker_objective_ex(const float* Wm, const float* x, int64_t ldx, const int32_t* x_index,
                             const double* params, const double* cos_params, int64_t N,
                             double* err, double* x_hat, int order, void* stream);

/* ------------------------------------------------------------------------------------------
 * TD end-to-end: batched, lock-step Powell minimisation of the K3 objective, entirely on device.
 * Replaces Test() (TD_Tester.py:162-199): scipy.optimize.minimize(objective, zeros(8),
 * method='Powell') with scipy's defaults (xtol = ftol = 1e-4, maxiter = maxfev = 8000), one
 * independent minimisation per face; the reference spends 2-4 s per face in it.
 *   Wm, cos_params  as for nlml_tucker_objective
 *   x        f32[N,1404] one feature row per face, row stride ldx
 *   x0       f64[N,8] starting points or NULL for zeros (TD_Tester.py:166)
 *   result   f64[N,8]  minimiser (w_y, w_p, w_r in RADIANS, u_id[5]); degrees are the caller's
 *            np.degrees(result.x)[:3] (:196-199)
 *   fval f64[N], nfev i32[N], nit i32[N], status i32[N] (1 converged, 2 maxfev, 3 maxiter, 4 nan):
 *            scipy's res.fun / res.nfev / res.nit; each may be NULL.
 * The control flow is scipy 1.15.3's (restated in nlml_hpe_amd/csrc/powell.h, checked against scipy step for step on the CPU).
 *   NLML_TD_ORDER_REFERENCE (default, parity mode): every machine receives the reference's objective values bit for bit, so it
 *       walks scipy's own trajectory: same evaluation counts, same final angles (FX5: identical bits; 1e-4 deg is the bar).
 *       (Up to the cos() of the f-vectors: the device's f64 cos and the host libm's may differ in the last place, which can flip the
 *       f32 rounding of an f-vector entry about once per 1e8 values -- tests/test_gpu_parity.py sweeps 1e6 angles.)
 *   NLML_TD_ORDER_FAST (opt-in, ~3x the faces/s): the same algorithm on the matrix-core objective.  The minimum is flat and
 *       Powell's termination is rounding-sensitive, so the END POINT moves under ANY re-ordering of the objective's sums (scipy
 *       itself: tests/test_powell_sm.py): 6e-3 deg from scipy's on clean grid faces (FX5); on BASELINE config 3's 4,096 noisy
 *       grid faces median 1.8e-3 deg (per face, largest of the three angles), 10 % of the faces > 0.02 deg, 0.3 % > 1 deg, max 8.7 deg.  Not a parity mode.
 */
int nlml_tucker_powell(const float* Wm, const float* x, int64_t ldx, const double* cos_params, int64_t N,
                       const double* x0, double* result, double* fval, int32_t* nfev, int32_t* nit,
                       int32_t* status, void* stream);                         /* == _ex(..., NLML_TD_ORDER_REFERENCE, stream) */
int nlml_tucker_powell_ex(const float* Wm, const float* x, int64_t ldx, const double* cos_params, int64_t N,
                          const double* x0, double* result, double* fval, int32_t* nfev, int32_t* nit,
                          int32_t* status, int order, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4  Video post-processing for S concurrent streams, one frame tick per call.
 * Replaces, per stream (generatePose_on_video.py): round(np.degrees(.), 2) (:211), the exponential
 * smoothing s = alpha*new + (1-alpha)*s seeded by the first prediction (:215-224, alpha 0.4 at :179),
 * and visualize_axes_on_face (:73-124): face centre from landmarks 1/33/263 x frame size, the
 * 100-px jump gate, and the three axis end points (size 80).
 *   pose_rad  f32[S,3]      this tick's model output (radians)
 *   raw       f32[S,468,3]  this tick's FaceMesh landmarks (only 1, 33, 263 are read)
 *   valid     u8[S] or NULL 0 = no face in this stream's frame: state and outputs untouched (:193-196); a stream whose
 *                           pose_rad holds NaN/Inf is skipped the same way (the reference would raise at :121)
 *   state     f64[S,6]      persistent: smoothed yaw/pitch/roll, previous centre x/y, prediction count;
 *                           zero-initialise before the first tick
 *   smoothed  f64[S,3] degrees; centre f64[S,2] pixels; endpoints f64[S,3,2] pixels (x,y of the red,
 *             green, blue axis tips; the reference draws int() of them)
 */
int nlml_video_post(const float* pose_rad, const float* raw, const uint8_t* valid, int64_t S,
                    double frame_w, double frame_h, double alpha, double max_jump, double size,
                    double* state, double* smoothed, double* centre, double* endpoints, void* stream);
/* The same with   updated  u8[S] or NULL: 1 where this tick was applied to the stream, 0 where it was skipped (no face, or a
 * non-finite pose) and state / smoothed / centre / endpoints still hold the previous tick's values -- what a caller that saves the
 * outputs must record as "no new result" (the reference drops such a frame, generatePose_on_video.py:193-196). */
int nlml_video_post_ex(const float* pose_rad, const float* raw, const uint8_t* valid, int64_t S,
                       double frame_w, double frame_h, double alpha, double max_jump, double size,
                       double* state, double* smoothed, double* centre, double* endpoints, uint8_t* updated, void* stream);

/* Artefact producers that are pure tensor algebra.
 *
 * nlml_cosine_table replaces the two nested loops over cosine() that build the heads' training inputs
 * (NLML_HPE_MLPHeadsTrainer.py:71-73,179-205): out[i][j] = a_j*cos(b_j*w_i + c_j) + d_j in f64.
 *   angles_rad f32[n]   np.radians(np.arange(min, max, interval).astype(np.float32)) (:179-181)
 *   cos_params f64[R,4] rows (a,b,c,d) = optimized_{yaw,pitch,roll} of outputs/features/Trained_data.npz
 *   out        f64[n,R]
 *
 * nlml_mode5_product replaces W = tl.tensordot(core, transpose(feature_matrix), axes=(4,0)) (TD_main.py:232-238):
 * W[q][m] = sum_r core[q][r] * U_feat[m][r], an r-ascending f32 fma chain per output.
 *   core   f32[Q,R5]  the Tucker core with its four leading modes flattened (Q = 135 for the shipped model)
 *   U_feat f32[M,R5]  mode-5 factor matrix (M = 1404 features)
 *   W      f32[Q,M]
 */
int nlml_cosine_table(const float* angles_rad, int64_t n, const double* cos_params, int R, double* out, void* stream);
int nlml_mode5_product(const float* core, const float* U_feat, int Q, int R5, int M, float* W, void* stream);

/* Host-side stepping of the same Powell state machine (no GPU involved): the caller evaluates
 * the objective.  Used to check the restated control flow against scipy on the CPU.
 *   h_state: caller-allocated buffer of nlml_powell_state_bytes() bytes.
 *   nlml_powell_step returns 1 and fills h_xeval[8] when it needs f(h_xeval) (pass it as `fin` to
 *   the next call; `fin` of the first call is ignored), 0 when finished. */
size_t nlml_powell_state_bytes(void);
int    nlml_powell_init(void* h_state, const double* h_x0, double xtol, double ftol);
int    nlml_powell_step(void* h_state, double fin, double* h_xeval);
int    nlml_powell_result(const void* h_state, double* h_x, double* h_fval, int* h_nfev, int* h_nit, int* h_status);

#ifdef __cplusplus
}
#endif
#endif /* NLML_HPE_H */
