// powell.h -- Powell's direction-set minimiser with Brent line searches as a RESUMABLE state machine.
//
// The reference's slow path, TD_Tester.Test (TD_Tester.py:162-199), calls
//     scipy.optimize.minimize(objective, zeros(8), method='Powell')          (:191-194)
// with scipy's defaults (xtol = ftol = 1e-4, maxiter = maxfev = 8000, unbounded).  scipy's driver
// is sequential Python that calls the objective ~1.4k-3.3k times per face.  To run thousands of
// faces in lock step on the GPU the algorithm is restated here so that every objective call becomes
// a SUSPEND point: `powell_step` returns the next trial point, the caller evaluates the objective
// (for a whole batch of faces at once, tucker_powell.hip) and resumes the machine with the value.
//
// Algorithm restated (published algorithm of scipy 1.15.3, scipy/optimize/_optimize.py; the module
// is a dependency of the reference, not part of its tree): _minimize_powell (direction loop,
// termination test, extrapolated point and direction replacement), _linesearch_powell (unbounded
// branch), bracket (golden-ratio expansion with parabolic steps, grow limit 110), Brent.optimize
// (parabolic interpolation / golden section, tol = 100*xtol, maxiter 500) and the
// "recover from bracket error" rule (best of the three bracket points).  All arithmetic is f64
// with the same operation order and no fused multiply-add, so that, GIVEN THE SAME OBJECTIVE
// VALUES, the machine visits bit-identical trial points (checked against scipy on the CPU,
// tests/test_powell_sm.py).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NLML_HD __host__ __device__ __forceinline__
#else
#define NLML_HD inline
#endif

namespace nlml {

constexpr int PW_N = 8;  // (w_y, w_p, w_r, u_id[5]) -- TD_Tester.py:166

struct PowellState {
  // --- outer Powell ---
  int pc;          // resume label of the outer machine
  int ls_pc;       // resume label of the line-search machine
  int i, iter, bigind, nfev, maxfun, maxiter, status, any;
  double xtol, ftol;
  double x[PW_N], x1[PW_N], direc[PW_N][PW_N];
  double fval, fx, fx2, delta, t, temp;
  // --- line search (bracket + Brent) along xi from p ---
  double p[PW_N], xi[PW_N], d1[PW_N];
  double alpha_min, fret;
  double xa, xb, xc, fa, fb, fc, w, fw, wlim, tmp1, tmp2, val, denom;
  int biter, bracket_ok, brk;
  double bx, bw, bv, bfx, bfw, bfv, a, b, deltax, rat, u, fu, tol1, tol2, xmid, pp, dx_temp;
  int it;
  // --- the point whose objective value the caller must supply next ---
  double xeval[PW_N];
};

enum { PW_RUNNING = 0, PW_CONVERGED = 1, PW_MAXFEV = 2, PW_MAXITER = 3, PW_NAN = 4 };

#if defined(__clang__)
#define NLML_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define NLML_FP_STRICT
#endif

NLML_HD void powell_init(PowellState& s, const double* x0, double xtol = 1e-4, double ftol = 1e-4) {
  s.pc = 0; s.ls_pc = 0; s.iter = 0; s.nfev = 0; s.status = PW_RUNNING;
  s.maxfun = PW_N * 1000; s.maxiter = PW_N * 1000;
  s.xtol = xtol; s.ftol = ftol;
  for (int k = 0; k < PW_N; ++k) {
    s.x[k] = x0[k];
    for (int j = 0; j < PW_N; ++j) s.direc[k][j] = (k == j) ? 1.0 : 0.0;
  }
}

// Line search from s.p along s.xi with Brent tolerance tol.  Returns true when it needs the
// objective at s.xeval (resume with its value in fin); false when finished, with s.fret,
// s.alpha_min set.  Mirrors scipy's bracket() + Brent.optimize() + bracket-error recovery.
NLML_HD bool linesearch_step(PowellState& s, double fin, double tol) {
  NLML_FP_STRICT
  const double gold = 1.618034, verysmall = 1e-21, grow_limit = 110.0, cg = 0.3819660, mintol = 1.0e-11;
#define LS_EVAL(LABEL, ALPHA)                                                        \
  do {                                                                               \
    for (int k_ = 0; k_ < PW_N; ++k_) s.xeval[k_] = s.p[k_] + (ALPHA) * s.xi[k_];    \
    s.ls_pc = LABEL;                                                                 \
    return true;                                                                     \
    case LABEL:;                                                                     \
  } while (0)

  switch (s.ls_pc) {
    case 0:
      // ---------------- bracket(func, xa=0, xb=1) ----------------
      s.xa = 0.0; s.xb = 1.0;
      LS_EVAL(1, s.xa); s.fa = fin;
      LS_EVAL(2, s.xb); s.fb = fin;
      if (s.fa < s.fb) {  // switch so fa > fb
        double tx = s.xa; s.xa = s.xb; s.xb = tx;
        double tf = s.fa; s.fa = s.fb; s.fb = tf;
      }
      s.xc = s.xb + gold * (s.xb - s.xa);
      LS_EVAL(3, s.xc); s.fc = fin;
      s.biter = 0;
      while (s.fc < s.fb) {
        s.tmp1 = (s.xb - s.xa) * (s.fb - s.fc);
        s.tmp2 = (s.xb - s.xc) * (s.fb - s.fa);
        s.val = s.tmp2 - s.tmp1;
        s.denom = (fabs(s.val) < verysmall) ? 2.0 * verysmall : 2.0 * s.val;
        s.w = s.xb - ((s.xb - s.xc) * s.tmp2 - (s.xb - s.xa) * s.tmp1) / s.denom;
        s.wlim = s.xb + grow_limit * (s.xc - s.xb);
        if (s.biter > 1000) break;  // scipy raises here; unreachable in practice
        s.biter += 1;
        s.brk = 0;
        if ((s.w - s.xc) * (s.xb - s.w) > 0.0) {
          LS_EVAL(4, s.w); s.fw = fin;
          if (s.fw < s.fc) {
            s.xa = s.xb; s.xb = s.w; s.fa = s.fb; s.fb = s.fw;
            s.brk = 1;
          } else if (s.fw > s.fb) {
            s.xc = s.w; s.fc = s.fw;
            s.brk = 1;
          } else {
            s.w = s.xc + gold * (s.xc - s.xb);
            LS_EVAL(5, s.w); s.fw = fin;
          }
        } else if ((s.w - s.wlim) * (s.wlim - s.xc) >= 0.0) {
          s.w = s.wlim;
          LS_EVAL(6, s.w); s.fw = fin;
        } else if ((s.w - s.wlim) * (s.xc - s.w) > 0.0) {
          LS_EVAL(7, s.w); s.fw = fin;
          if (s.fw < s.fc) {
            s.xb = s.xc; s.xc = s.w;
            s.w = s.xc + gold * (s.xc - s.xb);
            s.fb = s.fc; s.fc = s.fw;
            LS_EVAL(8, s.w); s.fw = fin;
          }
        } else {
          s.w = s.xc + gold * (s.xc - s.xb);
          LS_EVAL(9, s.w); s.fw = fin;
        }
        if (s.brk) break;
        s.xa = s.xb; s.xb = s.xc; s.xc = s.w;
        s.fa = s.fb; s.fb = s.fc; s.fc = s.fw;
      }
      {
        const bool cond1 = (s.fb < s.fc && s.fb <= s.fa) || (s.fb < s.fa && s.fb <= s.fc);
        const bool cond2 = (s.xa < s.xb && s.xb < s.xc) || (s.xc < s.xb && s.xb < s.xa);
        const bool cond3 = isfinite(s.xa) && isfinite(s.xb) && isfinite(s.xc);
        s.bracket_ok = (cond1 && cond2 && cond3) ? 1 : 0;
      }
      if (!s.bracket_ok) {  // _recover_from_bracket_error: best of the three points
        if (isnan(s.xa) || isnan(s.xb) || isnan(s.xc) || isnan(s.fa) || isnan(s.fb) || isnan(s.fc)) {
          s.alpha_min = NAN; s.fret = NAN;
        } else {
          s.alpha_min = s.xa; s.fret = s.fa;                       // argmin keeps the first minimum
          if (s.fb < s.fret) { s.alpha_min = s.xb; s.fret = s.fb; }
          if (s.fc < s.fret) { s.alpha_min = s.xc; s.fret = s.fc; }
        }
        s.ls_pc = 0;
        return false;
      }
      // ---------------- Brent.optimize() ----------------
      s.bx = s.bw = s.bv = s.xb;
      s.bfw = s.bfv = s.bfx = s.fb;
      if (s.xa < s.xc) { s.a = s.xa; s.b = s.xc; } else { s.a = s.xc; s.b = s.xa; }
      s.deltax = 0.0;
      s.rat = 0.0;
      s.it = 0;
      while (s.it < 500) {
        s.tol1 = tol * fabs(s.bx) + mintol;
        s.tol2 = 2.0 * s.tol1;
        s.xmid = 0.5 * (s.a + s.b);
        if (fabs(s.bx - s.xmid) < (s.tol2 - 0.5 * (s.b - s.a))) break;   // converged
        if (fabs(s.deltax) <= s.tol1) {
          s.deltax = (s.bx >= s.xmid) ? s.a - s.bx : s.b - s.bx;          // golden section step
          s.rat = cg * s.deltax;
        } else {                                                          // parabolic step
          s.tmp1 = (s.bx - s.bw) * (s.bfx - s.bfv);
          s.tmp2 = (s.bx - s.bv) * (s.bfx - s.bfw);
          s.pp = (s.bx - s.bv) * s.tmp2 - (s.bx - s.bw) * s.tmp1;
          s.tmp2 = 2.0 * (s.tmp2 - s.tmp1);
          if (s.tmp2 > 0.0) s.pp = -s.pp;
          s.tmp2 = fabs(s.tmp2);
          s.dx_temp = s.deltax;
          s.deltax = s.rat;
          if ((s.pp > s.tmp2 * (s.a - s.bx)) && (s.pp < s.tmp2 * (s.b - s.bx)) &&
              (fabs(s.pp) < fabs(0.5 * s.tmp2 * s.dx_temp))) {
            s.rat = s.pp * 1.0 / s.tmp2;
            s.u = s.bx + s.rat;
            if ((s.u - s.a) < s.tol2 || (s.b - s.u) < s.tol2) s.rat = (s.xmid - s.bx >= 0) ? s.tol1 : -s.tol1;
          } else {
            s.deltax = (s.bx >= s.xmid) ? s.a - s.bx : s.b - s.bx;
            s.rat = cg * s.deltax;
          }
        }
        if (fabs(s.rat) < s.tol1) s.u = (s.rat >= 0) ? s.bx + s.tol1 : s.bx - s.tol1;   // move by at least tol1
        else s.u = s.bx + s.rat;
        LS_EVAL(10, s.u); s.fu = fin;
        if (s.fu > s.bfx) {
          if (s.u < s.bx) s.a = s.u; else s.b = s.u;
          if ((s.fu <= s.bfw) || (s.bw == s.bx)) {
            s.bv = s.bw; s.bw = s.u; s.bfv = s.bfw; s.bfw = s.fu;
          } else if ((s.fu <= s.bfv) || (s.bv == s.bx) || (s.bv == s.bw)) {
            s.bv = s.u; s.bfv = s.fu;
          }
        } else {
          if (s.u >= s.bx) s.a = s.bx; else s.b = s.bx;
          s.bv = s.bw; s.bw = s.bx; s.bx = s.u;
          s.bfv = s.bfw; s.bfw = s.bfx; s.bfx = s.fu;
        }
        s.it += 1;
      }
      s.alpha_min = s.bx;
      s.fret = s.bfx;
      s.ls_pc = 0;
      return false;
  }
#undef LS_EVAL
  return false;
}

// The dominant call, as straight-line code: a line search suspended inside Brent's loop (label 10) is resumed with f(u), finishes
// that iteration and suspends at the next trial point.  The generic machine does the same through ~60 dependent accesses to
// the state (on the device: LDS round trips of ONE lane, ~6 k cycles a call); here everything the iteration needs is read up
// front and written back once.  Same operations in the same order as linesearch_step's loop body.  Returns true when the
// iteration suspended again at label 10 (state committed); false when it would leave the loop (converged, iteration limit) --
// then NOTHING has been written and the caller runs the generic machine on the untouched state.
NLML_HD bool brent_resume_fast(PowellState& s, double fin, double tol) {
  NLML_FP_STRICT
  const double cg = 0.3819660, mintol = 1.0e-11;
  double bx = s.bx, bw = s.bw, bv = s.bv, bfx = s.bfx, bfw = s.bfw, bfv = s.bfv, a = s.a, b = s.b;
  double deltax = s.deltax, rat = s.rat, u = s.u;
  double pk[PW_N], xk[PW_N];
  for (int k = 0; k < PW_N; ++k) { pk[k] = s.p[k]; xk[k] = s.xi[k]; }
  const int it = s.it + 1;
  const double fu = fin;
  if (fu > bfx) {
    if (u < bx) a = u; else b = u;
    if ((fu <= bfw) || (bw == bx)) {
      bv = bw; bw = u; bfv = bfw; bfw = fu;
    } else if ((fu <= bfv) || (bv == bx) || (bv == bw)) {
      bv = u; bfv = fu;
    }
  } else {
    if (u >= bx) a = bx; else b = bx;
    bv = bw; bw = bx; bx = u;
    bfv = bfw; bfw = bfx; bfx = fu;
  }
  if (!(it < 500)) return false;
  const double tol1 = tol * fabs(bx) + mintol;
  const double tol2 = 2.0 * tol1;
  const double xmid = 0.5 * (a + b);
  if (fabs(bx - xmid) < (tol2 - 0.5 * (b - a))) return false;   // converged: the generic machine finishes the search
  if (fabs(deltax) <= tol1) {
    deltax = (bx >= xmid) ? a - bx : b - bx;
    rat = cg * deltax;
  } else {
    double tmp1 = (bx - bw) * (bfx - bfv);
    double tmp2 = (bx - bv) * (bfx - bfw);
    double pp = (bx - bv) * tmp2 - (bx - bw) * tmp1;
    tmp2 = 2.0 * (tmp2 - tmp1);
    if (tmp2 > 0.0) pp = -pp;
    tmp2 = fabs(tmp2);
    const double dx_temp = deltax;
    deltax = rat;
    if ((pp > tmp2 * (a - bx)) && (pp < tmp2 * (b - bx)) && (fabs(pp) < fabs(0.5 * tmp2 * dx_temp))) {
      rat = pp * 1.0 / tmp2;
      u = bx + rat;
      if ((u - a) < tol2 || (b - u) < tol2) rat = (xmid - bx >= 0) ? tol1 : -tol1;
    } else {
      deltax = (bx >= xmid) ? a - bx : b - bx;
      rat = cg * deltax;
    }
  }
  if (fabs(rat) < tol1) u = (rat >= 0) ? bx + tol1 : bx - tol1;
  else u = bx + rat;
  s.bx = bx; s.bw = bw; s.bv = bv; s.bfx = bfx; s.bfw = bfw; s.bfv = bfv; s.a = a; s.b = b;
  s.deltax = deltax; s.rat = rat; s.u = u; s.fu = fu; s.it = it;
  for (int k = 0; k < PW_N; ++k) s.xeval[k] = pk[k] + u * xk[k];
  return true;
}

// Advance the minimiser.  First call: fin is ignored.  Returns true when the caller must evaluate
// the objective at s.xeval and call again with the value; false when finished (s.x, s.fval,
// s.nfev, s.iter, s.status hold the result: scipy's res.x, res.fun, res.nfev, res.nit).
NLML_HD bool powell_step(PowellState& s, double fin) {
  NLML_FP_STRICT
  // scipy's function wrapper raises _MaxFuncCallError BEFORE evaluation number maxfun+1; the
  // driver loop catches it and stops with the current x.
#define PW_EVAL_AT_XEVAL(LABEL)                          \
  do {                                                   \
    if (s.nfev >= s.maxfun) { s.status = PW_MAXFEV; s.pc = -1; return false; } \
    s.nfev += 1;                                         \
    s.pc = LABEL;                                        \
    return true;                                         \
    case LABEL:;                                         \
  } while (0)
  // run the inner machine until it finishes; every suspension of it suspends us at the same label
#define PW_LINESEARCH(LABEL)                                                  \
  do {                                                                        \
    s.ls_pc = 0;                                                              \
    fin = 0.0;                                                                \
    case LABEL:                                                               \
    if (linesearch_step(s, fin, s.xtol * 100)) {                              \
      if (s.nfev >= s.maxfun) { s.status = PW_MAXFEV; s.pc = -1; return false; } \
      s.nfev += 1;                                                            \
      s.pc = LABEL;                                                           \
      return true;                                                            \
    }                                                                         \
  } while (0)

#ifndef NLML_POWELL_NO_FAST_PATH
  if (s.ls_pc == 10 && (s.pc == 2 || s.pc == 4) && brent_resume_fast(s, fin, s.xtol * 100)) {
    if (s.nfev >= s.maxfun) { s.status = PW_MAXFEV; s.pc = -1; return false; }   // as PW_LINESEARCH does after a suspension
    s.nfev += 1;
    return true;
  }
#endif
  switch (s.pc) {
    case 0:
      for (int k = 0; k < PW_N; ++k) s.xeval[k] = s.x[k];
      PW_EVAL_AT_XEVAL(1);
      s.fval = fin;
      for (int k = 0; k < PW_N; ++k) s.x1[k] = s.x[k];
      s.iter = 0;
      while (true) {
        s.fx = s.fval;
        s.bigind = 0;
        s.delta = 0.0;
        for (s.i = 0; s.i < PW_N; ++s.i) {
          s.fx2 = s.fval;
          s.any = 0;
          for (int k = 0; k < PW_N; ++k) { s.xi[k] = s.direc[s.i][k]; s.p[k] = s.x[k]; s.any |= (s.xi[k] != 0.0); }
          if (s.any) {  // a zero direction is skipped (_linesearch_powell: "if not np.any(xi)")
            PW_LINESEARCH(2);
            for (int k = 0; k < PW_N; ++k) { s.d1[k] = s.alpha_min * s.xi[k]; s.x[k] = s.p[k] + s.d1[k]; }
            s.fval = s.fret;
          }
          if ((s.fx2 - s.fval) > s.delta) { s.delta = s.fx2 - s.fval; s.bigind = s.i; }
        }
        s.iter += 1;
        if (2.0 * (s.fx - s.fval) <= s.ftol * (fabs(s.fx) + fabs(s.fval)) + 1e-20) { s.status = PW_CONVERGED; break; }
        if (s.nfev >= s.maxfun) { s.status = PW_MAXFEV; break; }
        if (s.iter >= s.maxiter) { s.status = PW_MAXITER; break; }
        if (isnan(s.fx) && isnan(s.fval)) { s.status = PW_NAN; break; }
        // extrapolated point
        for (int k = 0; k < PW_N; ++k) { s.d1[k] = s.x[k] - s.x1[k]; s.x1[k] = s.x[k]; s.xeval[k] = s.x[k] + s.d1[k]; }
        PW_EVAL_AT_XEVAL(3);
        s.fx2 = fin;
        if (s.fx > s.fx2) {
          s.t = 2.0 * (s.fx + s.fx2 - 2.0 * s.fval);
          s.temp = (s.fx - s.fval - s.delta);
          s.t *= s.temp * s.temp;
          s.temp = s.fx - s.fx2;
          s.t -= s.delta * s.temp * s.temp;
          if (s.t < 0.0) {
            s.any = 0;
            for (int k = 0; k < PW_N; ++k) { s.xi[k] = s.d1[k]; s.p[k] = s.x[k]; s.any |= (s.xi[k] != 0.0); }
            if (s.any) {
              PW_LINESEARCH(4);
              s.any = 0;
              for (int k = 0; k < PW_N; ++k) {
                s.d1[k] = s.alpha_min * s.xi[k];
                s.x[k] = s.p[k] + s.d1[k];
                s.any |= (s.d1[k] != 0.0);
              }
              s.fval = s.fret;
              if (s.any) {
                for (int k = 0; k < PW_N; ++k) { s.direc[s.bigind][k] = s.direc[PW_N - 1][k]; s.direc[PW_N - 1][k] = s.d1[k]; }
              }
            }
          }
        }
      }
      if (s.status == PW_CONVERGED) {
        s.any = isnan(s.fval) ? 1 : 0;
        for (int k = 0; k < PW_N; ++k) s.any |= isnan(s.x[k]) ? 1 : 0;
        if (s.any) s.status = PW_NAN;
      }
      s.pc = -1;
      return false;
    default:
      return false;
  }
#undef PW_EVAL_AT_XEVAL
#undef PW_LINESEARCH
}

}  // namespace nlml
