// abi_internal.h -- error plumbing and cross-file declarations behind include/nlml_hpe.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace nlml {

// Records msg in the thread-local error slot and returns code (never 0).
int fail(int code, const char* msg);

// pack.cpp
size_t blob_bytes_for(int F, int mode);
int pack_blob(int F, int mode, const float* const enc_w[6], const float* const enc_b[6],
              const float* const head_w[3][5], const float* const head_b[3][5],
              void* blob, size_t blob_bytes);

// encoder_heads.hip
int launch_encoder_heads_f32(const float* x, int64_t ldx, const float* raw, int normalize,
                             int64_t B, int F, const void* blob, float* out, float* latent,
                             uint8_t* valid, float* pre_tanh, unsigned long long* stamps, void* stream,
                             int reeval_over = -1);   // >= 0: re-evaluation launch behind a split-f16 one (encoder_heads.hip)
// the f32 image inside an NLML_MODE_F16X2S blob (pack.cpp): byte offset from the blob's start
size_t strict_f32_image_offset(int F);
// faces of a tile the strict-fast kernels re-evaluate themselves on the vector ALUs (as NLML_MODE_F16X2 does); a tile with more goes
// to the f32 re-evaluation launch.  0: every face beyond f16's range is re-evaluated on the f32 matrix cores -- one rule, one
// accuracy class (the strict parity kernel's bits), and no slow-path code inside the strict kernels.
constexpr int STRICT_INKERNEL_RESCUE_MAX = 0;
// encoder_heads_bf16_w8.hip (throughput mode, eight waves per workgroup)
int launch_encoder_heads_bf16(const float* x, int64_t ldx, const float* raw, int normalize,
                              int64_t B, int F, const void* blob, float* out, float* latent,
                              uint8_t* valid, void* stream);
// encoder_heads_f16x2.hip (split-f16 parity mode)
int launch_encoder_heads_f16x2(const float* x, int64_t ldx, const float* raw, int normalize,
                               int64_t B, int F, const void* blob, float* out, float* latent,
                               uint8_t* valid, int split, void* stream);   // split: NLML_MODE_F16X2S
// encoder_heads_f16x2_w8.hip (NLML_MODE_F16X2S, eight waves per workgroup: the strict-fast mode's fused kernel)
int launch_encoder_heads_f16x2_w8(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                                  const void* blob, float* out, float* latent, uint8_t* valid, void* stream);
// the same forward as TWO launches (+ the re-evaluation launch): the trunk (layers 0-2, the eight-wave kernel ending with layer 2's output
// in `workspace`) and the streamed tail (encoder_heads_f16x2_tailws.hip: layers E3.. and the heads with the weights through LDS once per
// 256 faces); bit-identical to the fused kernel
bool tailws_supported(const float* x, int64_t ldx, int F);
size_t tailws_workspace_bytes(int64_t B, int F);
int launch_encoder_heads_f16x2_tailws(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                                      const void* blob, float* out, float* latent, uint8_t* valid, void* workspace,
                                      size_t ws_bytes, void* stream);
// encoder_heads_f16x2_tailws.hip: h3 = layer 2's output as operand fragments, nfb 32-face blocks of it -> poses (and the latent)
int launch_tail_ws(const void* blob, const void* h3, int64_t nfb, int64_t B, float* out, float* latent, void* stream);
// encoder_heads_f16x2_small.hip (split-f16 mode, big layers as separate launches + one tail launch, for small batches)
size_t small_workspace_bytes(int64_t B, int F);
int launch_encoder_heads_f16x2_small(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                                     const void* blob, float* out, float* latent, uint8_t* valid, void* workspace,
                                     size_t ws_bytes, int split, void* stream);
// artefacts.hip
int launch_cosine_table(const float* angles, int64_t n, const double* cos_params, int R, double* out, void* stream);
int launch_mode5_product(const float* core, const float* U, int Q, int R5, int M, float* W, void* stream);
// normalize_ipd.hip
int launch_normalize_ipd(const float* raw, int64_t B, int normalize, float* out, uint8_t* valid,
                         void* stream);
// tucker_objective.hip
int launch_tucker_objective(const float* Wm, const float* x, int64_t ldx, const int32_t* x_index,
                            const double* params, const double* cos_params, int64_t N,
                            double* err, double* x_hat, int order, void* stream);

// tucker_powell.hip
int launch_tucker_powell(const float* Wm, const float* x, int64_t ldx, const double* cos_params, int64_t N,
                         const double* x0, double* result, double* fval, int32_t* nfev, int32_t* nit,
                         int32_t* status, int order, void* stream);

// video_post.hip
int launch_video_post(const float* pose_rad, const float* raw, const uint8_t* valid, int64_t S, double frame_w,
                      double frame_h, double alpha, double max_jump, double size, double* state, double* smoothed,
                      double* centre, double* endpoints, uint8_t* updated, void* stream);

}  // namespace nlml
