// torch_ops.cpp -- torch.ops.nlml_hpe.* registered from COMPILED code (SURVEY.md 7 "Design stance", 8b "Underlying op").
//
// A TORCH_LIBRARY shim over the C ABI of include/nlml_hpe.h, nothing else: it checks shapes and dtypes (a wrong shape must never
// reach a hand-written kernel), allocates the outputs with torch's allocator, takes torch's CURRENT HIP stream of the operand's
// device and calls the same nlml_* entry point the ctypes binding calls -- so the results are the same bits, and the host side of a
// call is one dispatcher hop instead of a Python wrapper (measured at a 64-face tick, tools/host_hop.py: 15.4 us per call through the
// round-3 Python-registered op, 11.1 us through the Python wrapper, 4.0 us for the bare C call).  Kernels are registered for the GPU
// backend only (ROCm tensors carry torch's CUDA dispatch key): a CPU tensor has no kernel to land on and the dispatcher raises --
// there is no CPU fallback.  Meta kernels give the output shapes for tracing.
//
// Built by csrc/Makefile into nlml_hpe_amd/libnlml_torch_ops.so (links libnlml_hpe_hip.so next to it) and loaded by ops.py with
// torch.ops.load_library; a missing library raises at import.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <string>
#include <tuple>

#include "../../include/nlml_hpe.h"

namespace {

constexpr int64_t F_REF = NLML_F_REFERENCE;

void need(const at::Tensor& t, const char* name, at::ScalarType dt) {
  TORCH_CHECK(t.is_cuda(), name, ": expected a GPU tensor (there is no CPU fallback), got ", t.device());
  TORCH_CHECK(t.scalar_type() == dt, name, ": expected ", dt, ", got ", t.scalar_type());
}

// a buffer whose data_ptr() / numel() are handed to the C ABI as (pointer, size): it must be one dense run of bytes
void need_dense(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor (its data_ptr() and numel() go to the kernel as pointer and size)");
}

void same_device(const at::Tensor& a, const at::Tensor& b, const char* nb) {
  TORCH_CHECK(a.device() == b.device(), nb, " is on ", b.device(), " but the first operand is on ", a.device(),
              ": all operands must share one GPU");
}

void check(int rc, const char* what) {
  if (rc != 0) {
    const char* msg = nlml_last_error();
    TORCH_CHECK(false, what, " failed (code ", rc, "): ", msg ? msg : "");
  }
}

// the launch goes to the operand's device, on torch's current stream of THAT device
struct OnDevice {
  c10::hip::HIPGuardMasqueradingAsCUDA guard;
  void* stream;
  explicit OnDevice(const at::Tensor& t)
      : guard(t.device()), stream(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream()) {}
};

int td_order(const std::string& order) {
  if (order == "reference") return NLML_TD_ORDER_REFERENCE;
  if (order == "fast") return NLML_TD_ORDER_FAST;
  TORCH_CHECK(false, "unknown TD order '", order, "'; expected 'fast' or 'reference'");
}

at::Tensor normalize_ipd(const at::Tensor& raw_, bool normalize) {
  need(raw_, "raw", at::kFloat);
  TORCH_CHECK(raw_.dim() == 3 && raw_.size(1) == 468 && raw_.size(2) == 3, "raw: expected [B,468,3], got ", raw_.sizes());
  const at::Tensor raw = raw_.contiguous();
  const int64_t B = raw.size(0);
  at::Tensor out = at::empty({B, F_REF}, raw.options());
  OnDevice dev(raw);
  check(nlml_normalize_ipd(raw.data_ptr<float>(), B, normalize ? 1 : 0, out.data_ptr<float>(), nullptr, dev.stream), "nlml_normalize_ipd");
  return out;
}

at::Tensor encoder_heads_fwd(const at::Tensor& x_, const at::Tensor& blob, int64_t F) {
  need(x_, "x", at::kFloat);
  need(blob, "packed_w", at::kByte);
  need_dense(blob, "packed_w");
  same_device(x_, blob, "packed_w");
  TORCH_CHECK(x_.dim() == 2 && x_.size(1) == F, "x: expected [B,", F, "], got ", x_.sizes());
  const at::Tensor x = x_.stride(1) == 1 ? x_ : x_.contiguous();
  const int64_t B = x.size(0), ldx = B > 1 ? x.stride(0) : F;
  at::Tensor out = at::empty({B, 3}, x.options());
  OnDevice dev(x);
  check(nlml_encoder_heads_fwd(x.data_ptr<float>(), ldx, B, (int)F, blob.data_ptr(), (size_t)blob.numel(), out.data_ptr<float>(),
                               nullptr, nullptr, dev.stream), "nlml_encoder_heads_fwd");
  return out;
}

at::Tensor landmarks_to_pose(const at::Tensor& raw_, const at::Tensor& blob, bool normalize) {
  need(raw_, "raw", at::kFloat);
  need(blob, "packed_w", at::kByte);
  need_dense(blob, "packed_w");
  same_device(raw_, blob, "packed_w");
  TORCH_CHECK(raw_.dim() == 3 && raw_.size(1) == 468 && raw_.size(2) == 3, "raw: expected [B,468,3], got ", raw_.sizes());
  const at::Tensor raw = raw_.contiguous();
  const int64_t B = raw.size(0);
  at::Tensor out = at::empty({B, 3}, raw.options());
  OnDevice dev(raw);
  check(nlml_landmarks_to_pose(raw.data_ptr<float>(), B, normalize ? 1 : 0, blob.data_ptr(), (size_t)blob.numel(),
                               out.data_ptr<float>(), nullptr, nullptr, dev.stream), "nlml_landmarks_to_pose");
  return out;
}

at::Tensor encoder_heads_fwd_small(const at::Tensor& x_, const at::Tensor& blob, int64_t F, const at::Tensor& ws) {
  need(x_, "x", at::kFloat);
  need(blob, "packed_w", at::kByte);
  need_dense(blob, "packed_w");
  need(ws, "workspace", at::kByte);
  need_dense(ws, "workspace");
  same_device(x_, blob, "packed_w");
  same_device(x_, ws, "workspace");
  TORCH_CHECK(x_.dim() == 2 && x_.size(1) == F, "x: expected [B,", F, "], got ", x_.sizes());
  const at::Tensor x = x_.stride(1) == 1 ? x_ : x_.contiguous();
  const int64_t B = x.size(0), ldx = B > 1 ? x.stride(0) : F;
  at::Tensor out = at::empty({B, 3}, x.options());
  OnDevice dev(x);
  check(nlml_encoder_heads_fwd_small(x.data_ptr<float>(), ldx, B, (int)F, blob.data_ptr(), (size_t)blob.numel(), out.data_ptr<float>(),
                                     nullptr, nullptr, ws.data_ptr(), (size_t)ws.numel(), dev.stream), "nlml_encoder_heads_fwd_small");
  return out;
}

at::Tensor landmarks_to_pose_small(const at::Tensor& raw_, const at::Tensor& blob, bool normalize, const at::Tensor& ws) {
  need(raw_, "raw", at::kFloat);
  need(blob, "packed_w", at::kByte);
  need_dense(blob, "packed_w");
  need(ws, "workspace", at::kByte);
  need_dense(ws, "workspace");
  same_device(raw_, blob, "packed_w");
  same_device(raw_, ws, "workspace");
  TORCH_CHECK(raw_.dim() == 3 && raw_.size(1) == 468 && raw_.size(2) == 3, "raw: expected [B,468,3], got ", raw_.sizes());
  const at::Tensor raw = raw_.contiguous();
  const int64_t B = raw.size(0);
  at::Tensor out = at::empty({B, 3}, raw.options());
  OnDevice dev(raw);
  check(nlml_landmarks_to_pose_small(raw.data_ptr<float>(), B, normalize ? 1 : 0, blob.data_ptr(), (size_t)blob.numel(),
                                     out.data_ptr<float>(), nullptr, nullptr, ws.data_ptr(), (size_t)ws.numel(), dev.stream),
        "nlml_landmarks_to_pose_small");
  return out;
}

// the video tick's forward: pose AND the "a face was found" mask (row not all-zero) in one call; `ws` empty = the fused launch,
// else the layer-per-launch path through that workspace
std::tuple<at::Tensor, at::Tensor> landmarks_to_pose_valid(const at::Tensor& raw_, const at::Tensor& blob, bool normalize,
                                                           const std::optional<at::Tensor>& ws) {
  need(raw_, "raw", at::kFloat);
  need(blob, "packed_w", at::kByte);
  need_dense(blob, "packed_w");
  same_device(raw_, blob, "packed_w");
  TORCH_CHECK(raw_.dim() == 3 && raw_.size(1) == 468 && raw_.size(2) == 3, "raw: expected [B,468,3], got ", raw_.sizes());
  const at::Tensor raw = raw_.contiguous();
  const int64_t B = raw.size(0);
  at::Tensor out = at::empty({B, 3}, raw.options()), valid = at::empty({B}, raw.options().dtype(at::kByte));
  OnDevice dev(raw);
  if (ws.has_value()) {
    need(*ws, "workspace", at::kByte);
    need_dense(*ws, "workspace");
    same_device(raw_, *ws, "workspace");
    check(nlml_landmarks_to_pose_small(raw.data_ptr<float>(), B, normalize ? 1 : 0, blob.data_ptr(), (size_t)blob.numel(),
                                       out.data_ptr<float>(), nullptr, valid.data_ptr<uint8_t>(), ws->data_ptr(), (size_t)ws->numel(),
                                       dev.stream), "nlml_landmarks_to_pose_small");
  } else {
    check(nlml_landmarks_to_pose(raw.data_ptr<float>(), B, normalize ? 1 : 0, blob.data_ptr(), (size_t)blob.numel(),
                                 out.data_ptr<float>(), nullptr, valid.data_ptr<uint8_t>(), dev.stream), "nlml_landmarks_to_pose");
  }
  return {out, valid};
}

void check_td(const at::Tensor& Wm, const at::Tensor& x, const at::Tensor& cosp) {
  need(Wm, "Wm", at::kFloat);
  need(x, "x", at::kFloat);
  need(cosp, "cos_params", at::kDouble);
  same_device(x, Wm, "Wm");
  same_device(x, cosp, "cos_params");
  TORCH_CHECK(Wm.dim() == 2 && Wm.size(0) == 135 && Wm.size(1) == F_REF, "Wm: expected [135,1404], got ", Wm.sizes());
  TORCH_CHECK(x.dim() == 2 && x.size(1) == F_REF, "x: expected [N,1404], got ", x.sizes());
  TORCH_CHECK(cosp.dim() == 3 && cosp.size(0) == 3 && cosp.size(1) == 3 && cosp.size(2) == 4, "cos_params: expected [3,3,4], got ", cosp.sizes());
}

at::Tensor tucker_objective(const at::Tensor& Wm_, const at::Tensor& x_, const at::Tensor& params_, const at::Tensor& cosp_,
                            std::string order) {
  check_td(Wm_, x_, cosp_);
  need(params_, "params", at::kDouble);
  same_device(x_, params_, "params");
  TORCH_CHECK(params_.dim() == 2 && params_.size(1) == 8, "params: expected [N,8], got ", params_.sizes());
  TORCH_CHECK(x_.size(0) == params_.size(0), "x has ", x_.size(0), " rows but params has ", params_.size(0));
  const at::Tensor Wm = Wm_.contiguous(), x = x_.contiguous(), params = params_.contiguous(), cosp = cosp_.contiguous();
  const int64_t N = params.size(0);
  at::Tensor err = at::empty({N}, params.options());
  OnDevice dev(x);
  check(nlml_tucker_objective_ex(Wm.data_ptr<float>(), x.data_ptr<float>(), F_REF, nullptr, params.data_ptr<double>(),
                                 cosp.data_ptr<double>(), N, err.data_ptr<double>(), nullptr, td_order(order), dev.stream),
        "nlml_tucker_objective_ex");
  return err;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> tucker_powell(const at::Tensor& Wm_, const at::Tensor& x_,
                                                                                      const at::Tensor& cosp_, std::string order) {
  check_td(Wm_, x_, cosp_);
  const at::Tensor Wm = Wm_.contiguous(), x = x_.contiguous(), cosp = cosp_.contiguous();
  const int64_t N = x.size(0);
  const auto f64 = x.options().dtype(at::kDouble), i32 = x.options().dtype(at::kInt);
  at::Tensor res = at::empty({N, 8}, f64), fun = at::empty({N}, f64), nfev = at::empty({N}, i32), nit = at::empty({N}, i32),
             status = at::empty({N}, i32);
  OnDevice dev(x);
  check(nlml_tucker_powell_ex(Wm.data_ptr<float>(), x.data_ptr<float>(), F_REF, cosp.data_ptr<double>(), N, nullptr,
                              res.data_ptr<double>(), fun.data_ptr<double>(), nfev.data_ptr<int32_t>(), nit.data_ptr<int32_t>(),
                              status.data_ptr<int32_t>(), td_order(order), dev.stream), "nlml_tucker_powell_ex");
  return {res, fun, nfev, nit, status};
}

// one video tick, everything in place: `state` f64[S,6] and the caller's output buffers smoothed f64[S,3] (degrees), centre f64[S,2],
// endpoints f64[S,3,2], updated u8[S] -- a stream that is skipped (no face / non-finite pose) keeps its previous outputs, as the
// C entry point specifies
void video_post(const at::Tensor& pose_, const at::Tensor& raw_, const std::optional<at::Tensor>& valid_, double frame_w, double frame_h,
                double alpha, double max_jump, double size, at::Tensor state, at::Tensor sm, at::Tensor centre, at::Tensor ep,
                at::Tensor upd) {
  need(pose_, "pose_rad", at::kFloat);
  need(raw_, "raw", at::kFloat);
  need(state, "state", at::kDouble);
  need(sm, "smoothed", at::kDouble);
  need(centre, "centre", at::kDouble);
  need(ep, "endpoints", at::kDouble);
  need(upd, "updated", at::kByte);
  const at::Tensor* const others[] = {&raw_, &state, &sm, &centre, &ep, &upd};
  for (const at::Tensor* t : others) same_device(pose_, *t, "video_post operand");
  const int64_t S = pose_.size(0);
  TORCH_CHECK(pose_.dim() == 2 && pose_.size(1) == 3, "pose_rad: expected [S,3], got ", pose_.sizes());
  TORCH_CHECK(raw_.dim() == 3 && raw_.size(0) == S && raw_.size(1) == 468 && raw_.size(2) == 3, "raw: expected [S,468,3], got ", raw_.sizes());
  TORCH_CHECK(state.is_contiguous() && state.numel() == S * 6 && sm.is_contiguous() && sm.numel() == S * 3 && centre.is_contiguous() &&
                  centre.numel() == S * 2 && ep.is_contiguous() && ep.numel() == S * 6 && upd.is_contiguous() && upd.numel() == S,
              "video_post: state [S,6], smoothed [S,3], centre [S,2], endpoints [S,3,2], updated [S], all contiguous");
  const at::Tensor pose = pose_.contiguous(), raw = raw_.contiguous();
  at::Tensor valid;
  if (valid_.has_value()) {
    need(*valid_, "valid", at::kByte);
    same_device(pose_, *valid_, "valid");
    TORCH_CHECK(valid_->dim() == 1 && valid_->size(0) == S, "valid: expected [S]");
    valid = valid_->contiguous();
  }
  OnDevice dev(pose);
  check(nlml_video_post_ex(pose.data_ptr<float>(), raw.data_ptr<float>(), valid.defined() ? valid.data_ptr<uint8_t>() : nullptr, S,
                           frame_w, frame_h, alpha, max_jump, size, state.data_ptr<double>(), sm.data_ptr<double>(),
                           centre.data_ptr<double>(), ep.data_ptr<double>(), upd.data_ptr<uint8_t>(), dev.stream), "nlml_video_post_ex");
}

at::Tensor cosine_table(const at::Tensor& angles_, const at::Tensor& cosp_) {
  need(angles_, "angles_rad", at::kFloat);
  need(cosp_, "cos_params", at::kDouble);
  same_device(angles_, cosp_, "cos_params");
  TORCH_CHECK(angles_.dim() == 1 && cosp_.dim() == 2 && cosp_.size(1) == 4, "expected angles [n] and cos_params [R,4]");
  const at::Tensor angles = angles_.contiguous(), cosp = cosp_.contiguous();
  at::Tensor out = at::empty({angles.size(0), cosp.size(0)}, cosp.options());
  OnDevice dev(angles);
  check(nlml_cosine_table(angles.data_ptr<float>(), angles.size(0), cosp.data_ptr<double>(), (int)cosp.size(0), out.data_ptr<double>(),
                          dev.stream), "nlml_cosine_table");
  return out;
}

// ---- shapes only (Meta backend: tracing / fake tensors) -------------------------------------------------------------
at::Tensor normalize_ipd_meta(const at::Tensor& raw, bool) { return at::empty({raw.size(0), F_REF}, raw.options()); }
at::Tensor pose_meta3(const at::Tensor& x, const at::Tensor&, int64_t) { return at::empty({x.size(0), 3}, x.options()); }
at::Tensor pose_metab(const at::Tensor& x, const at::Tensor&, bool) { return at::empty({x.size(0), 3}, x.options()); }
at::Tensor pose_meta3w(const at::Tensor& x, const at::Tensor&, int64_t, const at::Tensor&) { return at::empty({x.size(0), 3}, x.options()); }
at::Tensor pose_metabw(const at::Tensor& x, const at::Tensor&, bool, const at::Tensor&) { return at::empty({x.size(0), 3}, x.options()); }
at::Tensor tucker_objective_meta(const at::Tensor&, const at::Tensor&, const at::Tensor& params, const at::Tensor&, std::string) {
  return at::empty({params.size(0)}, params.options());
}
std::tuple<at::Tensor, at::Tensor> pose_valid_meta(const at::Tensor& raw, const at::Tensor&, bool, const std::optional<at::Tensor>&) {
  return {at::empty({raw.size(0), 3}, raw.options()), at::empty({raw.size(0)}, raw.options().dtype(at::kByte))};
}
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> tucker_powell_meta(const at::Tensor&, const at::Tensor& x, const at::Tensor&,
                                                                                           std::string) {
  const int64_t N = x.size(0);
  const auto f64 = x.options().dtype(at::kDouble), i32 = x.options().dtype(at::kInt);
  return {at::empty({N, 8}, f64), at::empty({N}, f64), at::empty({N}, i32), at::empty({N}, i32), at::empty({N}, i32)};
}
void video_post_meta(const at::Tensor&, const at::Tensor&, const std::optional<at::Tensor>&, double, double, double, double, double, at::Tensor,
                     at::Tensor, at::Tensor, at::Tensor, at::Tensor) {}   // everything in place: nothing to shape
at::Tensor cosine_table_meta(const at::Tensor& angles, const at::Tensor& cosp) { return at::empty({angles.size(0), cosp.size(0)}, cosp.options()); }

}  // namespace

TORCH_LIBRARY(nlml_hpe, m) {
  m.def("normalize_ipd(Tensor raw, bool normalize) -> Tensor");
  m.def("encoder_heads_fwd(Tensor x, Tensor packed_w, int F) -> Tensor");
  m.def("landmarks_to_pose(Tensor raw, Tensor packed_w, bool normalize) -> Tensor");
  m.def("encoder_heads_fwd_small(Tensor x, Tensor packed_w, int F, Tensor workspace) -> Tensor");
  m.def("landmarks_to_pose_small(Tensor raw, Tensor packed_w, bool normalize, Tensor workspace) -> Tensor");
  m.def("landmarks_to_pose_valid(Tensor raw, Tensor packed_w, bool normalize, Tensor? workspace) -> (Tensor, Tensor)");
  m.def("tucker_objective(Tensor Wm, Tensor x, Tensor params, Tensor cos_params, str order=\"reference\") -> Tensor");
  m.def("tucker_powell(Tensor Wm, Tensor x, Tensor cos_params, str order=\"reference\") -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("video_post(Tensor pose_rad, Tensor raw, Tensor? valid, float frame_w, float frame_h, float alpha, float max_jump, float size, "
        "Tensor(a!) state, Tensor(b!) smoothed, Tensor(c!) centre, Tensor(d!) endpoints, Tensor(e!) updated) -> ()");
  m.def("cosine_table(Tensor angles_rad, Tensor cos_params) -> Tensor");
}

TORCH_LIBRARY_IMPL(nlml_hpe, CUDA, m) {   // ROCm tensors dispatch on torch's CUDA key
  m.impl("normalize_ipd", &normalize_ipd);
  m.impl("encoder_heads_fwd", &encoder_heads_fwd);
  m.impl("landmarks_to_pose", &landmarks_to_pose);
  m.impl("encoder_heads_fwd_small", &encoder_heads_fwd_small);
  m.impl("landmarks_to_pose_small", &landmarks_to_pose_small);
  m.impl("landmarks_to_pose_valid", &landmarks_to_pose_valid);
  m.impl("tucker_objective", &tucker_objective);
  m.impl("tucker_powell", &tucker_powell);
  m.impl("video_post", &video_post);
  m.impl("cosine_table", &cosine_table);
}

TORCH_LIBRARY_IMPL(nlml_hpe, Meta, m) {
  m.impl("normalize_ipd", &normalize_ipd_meta);
  m.impl("encoder_heads_fwd", &pose_meta3);
  m.impl("landmarks_to_pose", &pose_metab);
  m.impl("encoder_heads_fwd_small", &pose_meta3w);
  m.impl("landmarks_to_pose_small", &pose_metabw);
  m.impl("tucker_objective", &tucker_objective_meta);
  m.impl("landmarks_to_pose_valid", &pose_valid_meta);
  m.impl("tucker_powell", &tucker_powell_meta);
  m.impl("video_post", &video_post_meta);
  m.impl("cosine_table", &cosine_table_meta);
}
