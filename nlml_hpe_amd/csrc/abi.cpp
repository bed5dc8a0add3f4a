// abi.cpp -- the extern "C" surface declared in include/nlml_hpe.h: argument checks, then the
// launchers in the .hip files.  No allocation, no synchronisation, no global mutable state.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "layout.h"
#include "powell.h"

namespace nlml {
static thread_local char g_err[256] = "";
int fail(int code, const char* msg) {
  std::snprintf(g_err, sizeof g_err, "%s", msg ? msg : "unknown error");
  return code ? code : NLML_E_BADARG;
}
}  // namespace nlml

using namespace nlml;

static bool known_mode(int mode) { return mode == NLML_MODE_F32 || mode == NLML_MODE_BF16 || mode == NLML_MODE_F16X2 || mode == NLML_MODE_F16X2S; }
static bool split_f16(int mode) { return mode == NLML_MODE_F16X2 || mode == NLML_MODE_F16X2S; }

extern "C" {

int nlml_abi_version(void) { return NLML_ABI_VERSION; }
const char* nlml_last_error(void) { return g_err; }

int nlml_normalize_ipd(const float* raw, int64_t B, int normalize, float* out, uint8_t* valid, void* stream) {
  if (B < 0 || (B > 0 && (!raw || !out))) return fail(NLML_E_BADARG, "normalize_ipd: null buffer or negative B");
  if ((reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(out)) & 15)
    return fail(NLML_E_BADARG, "normalize_ipd: raw/out must be 16-byte aligned");
  return launch_normalize_ipd(raw, B, normalize, out, valid, stream);
}

size_t nlml_encoder_heads_packed_bytes(int F, int mode) {
  if (F <= 0 || !known_mode(mode)) return 0;
  return blob_bytes_for(F, mode);
}

int nlml_encoder_heads_pack(int F, int mode, const float* const h_enc_w[6], const float* const h_enc_b[6],
                            const float* const h_head_w[3][5], const float* const h_head_b[3][5], void* h_blob,
                            size_t blob_bytes) {
  if (!known_mode(mode)) return fail(NLML_E_BADARG, "pack: unsupported mode");
  if (!h_enc_w || !h_enc_b || !h_head_w || !h_head_b) return fail(NLML_E_BADARG, "pack: null table");
  return pack_blob(F, mode, h_enc_w, h_enc_b, h_head_w, h_head_b, h_blob, blob_bytes);
}

// The blob lives in device memory; its mode is recognised by its size (the sizes of the modes never coincide:
// the latent and head-input stages are padded differently in each).
static int check_blob_args(int64_t B, int F, const void* blob, size_t blob_bytes, const float* out, int* mode) {
  if (B < 0 || F <= 0) return fail(NLML_E_BADARG, "encoder_heads: negative B or bad F");
  if (B > 0 && (!blob || !out)) return fail(NLML_E_BADARG, "encoder_heads: null buffer");
  if (reinterpret_cast<uintptr_t>(blob) & 15) return fail(NLML_E_BADARG, "encoder_heads: blob must be 16-byte aligned");
  if (blob_bytes == blob_bytes_for(F, NLML_MODE_F32)) *mode = NLML_MODE_F32;
  else if (blob_bytes == blob_bytes_for(F, NLML_MODE_BF16)) *mode = NLML_MODE_BF16;
  else if (blob_bytes == blob_bytes_for(F, NLML_MODE_F16X2)) *mode = NLML_MODE_F16X2;
  else if (blob_bytes == blob_bytes_for(F, NLML_MODE_F16X2S)) *mode = NLML_MODE_F16X2S;
  else return fail(NLML_E_BADBLOB, "encoder_heads: blob size does not match F in any mode");
  return 0;
}

int nlml_encoder_heads_fwd(const float* x, int64_t ldx, int64_t B, int F, const void* blob, size_t blob_bytes,
                           float* out, float* latent, uint8_t* valid, void* stream) {
  int mode = 0;
  if (int rc = check_blob_args(B, F, blob, blob_bytes, out, &mode)) return rc;
  if (B > 0 && (!x || ldx < F)) return fail(NLML_E_BADARG, "encoder_heads: null x or ldx < F");
  if (mode == NLML_MODE_BF16) return launch_encoder_heads_bf16(x, ldx, nullptr, 0, B, F, blob, out, latent, valid, stream);
  if (split_f16(mode))
    return launch_encoder_heads_f16x2(x, ldx, nullptr, 0, B, F, blob, out, latent, valid, mode == NLML_MODE_F16X2S, stream);
  return launch_encoder_heads_f32(x, ldx, nullptr, 0, B, F, blob, out, latent, valid, nullptr, nullptr, stream);
}

int nlml_encoder_heads_fwd_debug(const float* x, int64_t ldx, int64_t B, int F, const void* blob, size_t blob_bytes,
                                 float* out, float* latent, float* pre_tanh, unsigned long long* stamps, void* stream) {
  int mode = 0;
  if (int rc = check_blob_args(B, F, blob, blob_bytes, out, &mode)) return rc;
  if (mode != NLML_MODE_F32) return fail(NLML_E_BADARG, "debug build: f32 blob only");
  if (B > 0 && (!x || ldx < F)) return fail(NLML_E_BADARG, "encoder_heads: null x or ldx < F");
  return launch_encoder_heads_f32(x, ldx, nullptr, 0, B, F, blob, out, latent, nullptr, pre_tanh, stamps, stream);
}

int nlml_landmarks_to_pose(const float* raw, int64_t B, int normalize, const void* blob, size_t blob_bytes,
                           float* out, float* latent, uint8_t* valid, void* stream) {
  int mode = 0;
  if (int rc = check_blob_args(B, NLML_F_REFERENCE, blob, blob_bytes, out, &mode)) return rc;
  if (B > 0 && !raw) return fail(NLML_E_BADARG, "landmarks_to_pose: null raw");
  if (mode == NLML_MODE_BF16)
    return launch_encoder_heads_bf16(nullptr, 0, raw, normalize, B, NLML_F_REFERENCE, blob, out, latent, valid, stream);
  if (split_f16(mode))
    return launch_encoder_heads_f16x2(nullptr, 0, raw, normalize, B, NLML_F_REFERENCE, blob, out, latent, valid,
                                      mode == NLML_MODE_F16X2S, stream);
  return launch_encoder_heads_f32(nullptr, 0, raw, normalize, B, NLML_F_REFERENCE, blob, out, latent, valid, nullptr, nullptr, stream);
}

// ---- split-f16 mode, one launch per layer (small batches) -------------------------------------------------------
size_t nlml_encoder_heads_small_workspace_bytes(int64_t B, int F) { return small_workspace_bytes(B, F); }

static int check_small(int64_t B, int F, const void* blob, size_t blob_bytes, const float* out, int* split) {
  int mode = 0;
  if (int rc = check_blob_args(B, F, blob, blob_bytes, out, &mode)) return rc;
  if (!split_f16(mode)) return fail(NLML_E_BADARG, "small-batch path: NLML_MODE_F16X2 / NLML_MODE_F16X2S blob only");
  *split = mode == NLML_MODE_F16X2S;
  return 0;
}

int nlml_encoder_heads_fwd_small(const float* x, int64_t ldx, int64_t B, int F, const void* blob, size_t blob_bytes,
                                 float* out, float* latent, uint8_t* valid, void* workspace, size_t ws_bytes, void* stream) {
  int split = 0;
  if (int rc = check_small(B, F, blob, blob_bytes, out, &split)) return rc;
  if (B > 0 && (!x || ldx < F)) return fail(NLML_E_BADARG, "encoder_heads: null x or ldx < F");
  return launch_encoder_heads_f16x2_small(x, ldx, nullptr, 0, B, F, blob, out, latent, valid, workspace, ws_bytes, split, stream);
}

int nlml_landmarks_to_pose_small(const float* raw, int64_t B, int normalize, const void* blob, size_t blob_bytes,
                                 float* out, float* latent, uint8_t* valid, void* workspace, size_t ws_bytes, void* stream) {
  int split = 0;
  if (int rc = check_small(B, NLML_F_REFERENCE, blob, blob_bytes, out, &split)) return rc;
  if (B > 0 && !raw) return fail(NLML_E_BADARG, "landmarks_to_pose: null raw");
  return launch_encoder_heads_f16x2_small(nullptr, 0, raw, normalize, B, NLML_F_REFERENCE, blob, out, latent, valid, workspace,
                                          ws_bytes, split, stream);
}

// ---- strict-fast mode, LARGE batches: trunk launch + streamed tail launch (encoder_heads_f16x2_w8.hip TRUNK, encoder_heads_f16x2_tailws.hip) ----
static int check_streamed(int64_t B, int F, const void* blob, size_t blob_bytes, const float* out) {
  int mode = 0;
  if (int rc = check_blob_args(B, F, blob, blob_bytes, out, &mode)) return rc;
  if (mode != NLML_MODE_F16X2S) return fail(NLML_E_BADARG, "streamed-tail path: NLML_MODE_F16X2S blob only");
  return 0;
}
int nlml_encoder_heads_fwd_streamed(const float* x, int64_t ldx, int64_t B, int F, const void* blob, size_t blob_bytes,
                                    float* out, float* latent, uint8_t* valid, void* workspace, size_t ws_bytes, void* stream) {
  if (int rc = check_streamed(B, F, blob, blob_bytes, out)) return rc;
  if (B > 0 && (!x || ldx < F)) return fail(NLML_E_BADARG, "encoder_heads: null x or ldx < F");
  return launch_encoder_heads_f16x2_tailws(x, ldx, nullptr, 0, B, F, blob, out, latent, valid, workspace, ws_bytes, stream);
}

int nlml_landmarks_to_pose_streamed(const float* raw, int64_t B, int normalize, const void* blob, size_t blob_bytes,
                                    float* out, float* latent, uint8_t* valid, void* workspace, size_t ws_bytes, void* stream) {
  if (int rc = check_streamed(B, NLML_F_REFERENCE, blob, blob_bytes, out)) return rc;
  if (B > 0 && !raw) return fail(NLML_E_BADARG, "landmarks_to_pose: null raw");
  return launch_encoder_heads_f16x2_tailws(nullptr, 0, raw, normalize, B, NLML_F_REFERENCE, blob, out, latent, valid, workspace,
                                           ws_bytes, stream);
}

// ---- the forward with a caller-provided workspace: the fastest path for the batch size and the blob's mode -------------------------
// (measured crossovers, tools/k2_crossover.py and bench.py extra.k2_batch_sweep)
static const int64_t kSmallMax = 4096;     // split-f16 modes: up to here the layer-per-launch path over 64-face tiles
// The trunk + streamed-tail path (encoder_heads_f16x2_tailws.hip) is bit-identical to the fused kernel and measured 1.4-2.2 % faster at 65,536
// faces (0.801 against 0.815 ms, same box, alternating) -- inside the box-to-box spread, for a 64 MB workspace and a
// second big launch -- so the dispatcher does not pick it by itself.  NLML_K2_STREAMED_MIN=<faces> routes batches from that size
// on through it (the explicit _streamed entry points always do).
static int64_t streamed_min() {
  static const int64_t v = [] { const char* e = getenv("NLML_K2_STREAMED_MIN"); return e && e[0] ? (int64_t)atoll(e) : (int64_t)-1; }();
  return v;
}

size_t nlml_encoder_heads_workspace_bytes(int64_t B, int F) {
  const size_t a = small_workspace_bytes(B, F), b = tailws_workspace_bytes(B, F);
  return a > b ? a : b;
}

static int fwd_ws(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F, const void* blob, size_t blob_bytes,
                  float* out, float* latent, uint8_t* valid, void* workspace, size_t ws_bytes, void* stream) {
  int mode = 0;
  if (int rc = check_blob_args(B, F, blob, blob_bytes, out, &mode)) return rc;
  if (B > 0 && !raw && (!x || ldx < F)) return fail(NLML_E_BADARG, "encoder_heads: null x or ldx < F");
  if (mode == NLML_MODE_BF16) return launch_encoder_heads_bf16(x, ldx, raw, normalize, B, F, blob, out, latent, valid, stream);
  if (!split_f16(mode)) return launch_encoder_heads_f32(x, ldx, raw, normalize, B, F, blob, out, latent, valid, nullptr, nullptr, stream);
  const int split = mode == NLML_MODE_F16X2S;
  if (B <= kSmallMax)
    return launch_encoder_heads_f16x2_small(x, ldx, raw, normalize, B, F, blob, out, latent, valid, workspace, ws_bytes, split, stream);
  if (split && streamed_min() >= 0 && B >= streamed_min() && tailws_supported(raw ? raw : x, raw ? NLML_F_REFERENCE : ldx, F))
    return launch_encoder_heads_f16x2_tailws(x, ldx, raw, normalize, B, F, blob, out, latent, valid, workspace, ws_bytes, stream);
  return launch_encoder_heads_f16x2(x, ldx, raw, normalize, B, F, blob, out, latent, valid, split, stream);
}

int nlml_encoder_heads_fwd_ws(const float* x, int64_t ldx, int64_t B, int F, const void* blob, size_t blob_bytes,
                              float* out, float* latent, uint8_t* valid, void* workspace, size_t ws_bytes, void* stream) {
  return fwd_ws(x, ldx, nullptr, 0, B, F, blob, blob_bytes, out, latent, valid, workspace, ws_bytes, stream);
}

int nlml_landmarks_to_pose_ws(const float* raw, int64_t B, int normalize, const void* blob, size_t blob_bytes,
                              float* out, float* latent, uint8_t* valid, void* workspace, size_t ws_bytes, void* stream) {
  if (B > 0 && !raw) return fail(NLML_E_BADARG, "landmarks_to_pose: null raw");
  return fwd_ws(nullptr, 0, raw, normalize, B, NLML_F_REFERENCE, blob, blob_bytes, out, latent, valid, workspace, ws_bytes, stream);
}

// The matrix-core order reads Wm and the x rows with 16-byte vector loads (tucker_common.h load11 / tucker_few)
static int check_td_fast_alignment(const float* Wm, const float* x, int64_t ldx, const char* who) {
  if ((reinterpret_cast<uintptr_t>(Wm) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || (ldx & 3)) {
    static thread_local char msg[160];
    snprintf(msg, sizeof msg, "%s: NLML_TD_ORDER_FAST needs Wm and x 16-byte aligned and ldx %% 4 == 0", who);
    return fail(NLML_E_BADARG, msg);
  }
  return 0;
}

int nlml_tucker_objective_ex(const float* Wm, const float* x, int64_t ldx, const int32_t* x_index,
                             const double* params, const double* cos_params, int64_t N, double* err, double* x_hat,
                             int order, void* stream) {
  if (order != NLML_TD_ORDER_FAST && order != NLML_TD_ORDER_REFERENCE) return fail(NLML_E_BADARG, "tucker_objective: unknown order");
  if (N < 0) return fail(NLML_E_BADARG, "tucker_objective: negative N");
  if (N > 0 && (!Wm || !x || !params || !cos_params || !err)) return fail(NLML_E_BADARG, "tucker_objective: null buffer");
  if (N > 0 && ldx < NLML_F_REFERENCE) return fail(NLML_E_BADARG, "tucker_objective: ldx < 1404");
  if (N > 0 && order == NLML_TD_ORDER_FAST)
    if (int rc = check_td_fast_alignment(Wm, x, ldx, "tucker_objective")) return rc;
  return launch_tucker_objective(Wm, x, ldx, x_index, params, cos_params, N, err, x_hat, order, stream);
}

int nlml_tucker_objective(const float* Wm, const float* x, int64_t ldx, const int32_t* x_index,
                          const double* params, const double* cos_params, int64_t N, double* err, double* x_hat,
                          void* stream) {
  return nlml_tucker_objective_ex(Wm, x, ldx, x_index, params, cos_params, N, err, x_hat, NLML_TD_ORDER_REFERENCE, stream);
}

int nlml_video_post_ex(const float* pose_rad, const float* raw, const uint8_t* valid, int64_t S, double frame_w,
                       double frame_h, double alpha, double max_jump, double size, double* state, double* smoothed,
                       double* centre, double* endpoints, uint8_t* updated, void* stream) {
  if (S < 0) return fail(NLML_E_BADARG, "video_post: negative S");
  if (S > 0 && (!pose_rad || !raw || !state || !smoothed || !centre || !endpoints))
    return fail(NLML_E_BADARG, "video_post: null buffer");
  return launch_video_post(pose_rad, raw, valid, S, frame_w, frame_h, alpha, max_jump, size, state, smoothed, centre,
                           endpoints, updated, stream);
}

int nlml_video_post(const float* pose_rad, const float* raw, const uint8_t* valid, int64_t S, double frame_w,
                    double frame_h, double alpha, double max_jump, double size, double* state, double* smoothed,
                    double* centre, double* endpoints, void* stream) {
  return nlml_video_post_ex(pose_rad, raw, valid, S, frame_w, frame_h, alpha, max_jump, size, state, smoothed, centre, endpoints,
                            nullptr, stream);
}

int nlml_cosine_table(const float* angles_rad, int64_t n, const double* cos_params, int R, double* out, void* stream) {
  if (n < 0 || R < 0) return fail(NLML_E_BADARG, "cosine_table: negative size");
  if (n > 0 && R > 0 && (!angles_rad || !cos_params || !out)) return fail(NLML_E_BADARG, "cosine_table: null buffer");
  return launch_cosine_table(angles_rad, n, cos_params, R, out, stream);
}

int nlml_mode5_product(const float* core, const float* U_feat, int Q, int R5, int M, float* W, void* stream) {
  if (Q < 0 || R5 < 0 || M < 0) return fail(NLML_E_BADARG, "mode5_product: negative size");
  if (Q > 0 && M > 0 && (!W || (R5 > 0 && (!core || !U_feat)))) return fail(NLML_E_BADARG, "mode5_product: null buffer");
  return launch_mode5_product(core, U_feat, Q, R5, M, W, stream);
}

// ---- host-side stepping of the Powell state machine (powell.h) --------------------------------
size_t nlml_powell_state_bytes(void) { return sizeof(PowellState); }

int nlml_powell_init(void* h_state, const double* h_x0, double xtol, double ftol) {
  if (!h_state || !h_x0) return fail(NLML_E_BADARG, "powell_init: null pointer");
  powell_init(*static_cast<PowellState*>(h_state), h_x0, xtol, ftol);
  return 0;
}

int nlml_powell_step(void* h_state, double fin, double* h_xeval) {
  if (!h_state || !h_xeval) return fail(NLML_E_BADARG, "powell_step: null pointer");
  PowellState& s = *static_cast<PowellState*>(h_state);
  const bool need = powell_step(s, fin);
  if (need) std::memcpy(h_xeval, s.xeval, sizeof s.xeval);
  return need ? 1 : 0;
}

int nlml_powell_result(const void* h_state, double* h_x, double* h_fval, int* h_nfev, int* h_nit, int* h_status) {
  if (!h_state || !h_x) return fail(NLML_E_BADARG, "powell_result: null pointer");
  const PowellState& s = *static_cast<const PowellState*>(h_state);
  std::memcpy(h_x, s.x, sizeof s.x);
  if (h_fval) *h_fval = s.fval;
  if (h_nfev) *h_nfev = s.nfev;
  if (h_nit) *h_nit = s.iter;
  if (h_status) *h_status = s.status;
  return 0;
}

int nlml_tucker_powell_ex(const float* Wm, const float* x, int64_t ldx, const double* cos_params, int64_t N,
                          const double* x0, double* result, double* fval, int32_t* nfev, int32_t* nit,
                          int32_t* status, int order, void* stream) {
  if (order != NLML_TD_ORDER_FAST && order != NLML_TD_ORDER_REFERENCE) return fail(NLML_E_BADARG, "tucker_powell: unknown order");
  if (N < 0) return fail(NLML_E_BADARG, "tucker_powell: negative N");
  if (N > 0 && (!Wm || !x || !cos_params || !result)) return fail(NLML_E_BADARG, "tucker_powell: null buffer");
  if (N > 0 && ldx < NLML_F_REFERENCE) return fail(NLML_E_BADARG, "tucker_powell: ldx < 1404");
  if (N > 0 && order == NLML_TD_ORDER_FAST)
    if (int rc = check_td_fast_alignment(Wm, x, ldx, "tucker_powell")) return rc;
  return launch_tucker_powell(Wm, x, ldx, cos_params, N, x0, result, fval, nfev, nit, status, order, stream);
}

int nlml_tucker_powell(const float* Wm, const float* x, int64_t ldx, const double* cos_params, int64_t N,
                       const double* x0, double* result, double* fval, int32_t* nfev, int32_t* nit,
                       int32_t* status, void* stream) {
  return nlml_tucker_powell_ex(Wm, x, ldx, cos_params, N, x0, result, fval, nfev, nit, status, NLML_TD_ORDER_REFERENCE, stream);
}

}  // extern "C"
