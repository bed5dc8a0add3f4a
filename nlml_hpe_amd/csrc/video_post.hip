// video_post.hip -- K4: per-frame post-processing of the video entry point for S concurrent streams.
//
// Replaces, per stream and frame (generatePose_on_video.py):
//   :211      yaw,pitch,roll = round(np.degrees(pred.item()), 2)
//   :215-224  exponential smoothing  s = 0.4*new + 0.6*s, the first prediction seeds s
//   :73-124   visualize_axes_on_face: centre = mean of landmarks 1/33/263 scaled by the frame size,
//             jump gate (keep the previous centre if it moved > 100 px), three axis end points
//             (size 80) from yaw (negated) / pitch / roll
// with the state (smoothed angles, previous centre, prediction count) kept on the device between
// frames, so a tick of S streams is one launch and no per-face D2H sync.  A stream whose frame has
// no face (valid == 0) is skipped exactly like the reference's `continue` (:193-196): its state
// does not change and its outputs are left untouched; so is a stream whose pose is not finite.  `updated` (optional) tells the
// caller which streams' ticks were applied, so a skipped stream is never reported with stale outputs as if they were new.
//
// One thread per stream, all f64 (the reference computes in Python floats).  round(x, 2) is
// rint(x*100)/100 (half-to-even), which equals Python's correctly-rounded round() except when
// x*100 lies within an ulp of a .5 tie.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"

namespace nlml {

__global__ void video_post_kernel(const float* __restrict__ pose_rad, const float* __restrict__ raw,
                                  const uint8_t* __restrict__ valid, int64_t S, double frame_w, double frame_h,
                                  double alpha, double max_jump, double size, double* __restrict__ state,
                                  double* __restrict__ smoothed, double* __restrict__ centre,
                                  double* __restrict__ endpoints, uint8_t* __restrict__ updated) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  if (updated) updated[s] = 0;              // set to 1 below only if this stream's tick is applied
  if (valid && !valid[s]) return;
  // A non-finite pose (NaN/Inf landmarks in) is treated like "no face": the stream's state is left alone.  The reference
  // has no such frame to copy -- its int(x1) at generatePose_on_video.py:121 raises on NaN and ends the loop -- and letting
  // it through would poison the stream's EMA for good (s = 0.4*NaN + 0.6*s).
  if (!(isfinite(pose_rad[s * 3 + 0]) && isfinite(pose_rad[s * 3 + 1]) && isfinite(pose_rad[s * 3 + 2]))) return;
  double* st = state + s * 6;               // [sm_yaw, sm_pitch, sm_roll, prev_tdx, prev_tdy, count]
  const double kDeg = 57.29577951308232;    // 180/pi, np.degrees
  double ang[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double deg = (double)pose_rad[s * 3 + k] * kDeg;
    const double nw = rint(deg * 100.0) / 100.0;                                  // round(..., 2), :211
    ang[k] = (st[5] < 1.0) ? nw : alpha * nw + (1.0 - alpha) * st[k];             // :215-222
    st[k] = ang[k];
    smoothed[s * 3 + k] = ang[k];
  }
  const float* lm = raw + s * NLML_F_REFERENCE;
  const double nx = lm[3], ny = lm[4], lx = lm[99], ly = lm[100], rx = lm[789], ry = lm[790];   // landmarks 1, 33, 263
  const double new_tdx = (nx + lx + rx) * frame_w / 3.0;                          // :91-92
  const double new_tdy = (ny + ly + ry) * frame_h / 3.0;
  double tdx = new_tdx, tdy = new_tdy;
  if (st[5] >= 1.0) {
    const double dx = new_tdx - st[3], dy = new_tdy - st[4];
    if (sqrt(dx * dx + dy * dy) > max_jump) { tdx = st[3]; tdy = st[4]; }         // :99-107
  }
  st[3] = tdx; st[4] = tdy; st[5] += 1.0;
  if (updated) updated[s] = 1;
  centre[s * 2 + 0] = tdx; centre[s * 2 + 1] = tdy;
  const double kRad = 3.141592653589793 / 180.0;
  const double pitch = ang[1] * kRad, yaw = -(ang[0] * kRad), roll = ang[2] * kRad;   // :74-76
  double* ep = endpoints + s * 6;
  ep[0] = size * (cos(yaw) * cos(roll)) + tdx;                                        // :110-119
  ep[1] = size * (cos(pitch) * sin(roll) + cos(roll) * sin(pitch) * sin(yaw)) + tdy;
  ep[2] = size * (-cos(yaw) * sin(roll)) + tdx;
  ep[3] = size * (cos(pitch) * cos(roll) - sin(pitch) * sin(yaw) * sin(roll)) + tdy;
  ep[4] = size * (sin(yaw)) + tdx;
  ep[5] = size * (-cos(yaw) * sin(pitch)) + tdy;
}

int launch_video_post(const float* pose_rad, const float* raw, const uint8_t* valid, int64_t S, double frame_w,
                      double frame_h, double alpha, double max_jump, double size, double* state, double* smoothed,
                      double* centre, double* endpoints, uint8_t* updated, void* stream) {
  if (S == 0) return 0;
  const dim3 grid((unsigned)((S + 63) / 64)), block(64);
  hipLaunchKernelGGL(video_post_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), pose_rad, raw, valid, S,
                     frame_w, frame_h, alpha, max_jump, size, state, smoothed, centre, endpoints, updated);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
