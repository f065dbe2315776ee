// exact_math.h — the two shortened-but-exact instruction sequences of the cell arithmetic: 1.0f / x and sqrt((double)x).
// Part of the device code of liblbm_d2q9.so (included by common.h) AND included by scripts/experiments/recip_exhaustive.hip
// and sqrt_exhaustive.hip, which enumerate every float through THESE functions on the GPU (tests/test_gpu_parity.py runs
// both): an edit here is checked against the IEEE division / the correctly rounded root on all inputs, not a private copy.
// gfx950 only; compile with -ffp-contract=off like the rest of the library (the fma calls below are explicit).
#pragma once

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// sqrt((double)x) for a float x (d2q9-bgk.c:667 promotes u_sq to double), correctly rounded.  hipcc's sqrt(double) is
// v_rsq_f64, one joint refinement of g ~ sqrt(x) and h ~ 1/(2 sqrt(x)), two Newton corrections of g and a rescaling of
// arguments below 2^-767: 11 + 5 instructions.  A converted float is never that small, and for a 24-bit significand the
// refinement is not needed: the raw estimates followed by two corrections already give the correctly rounded result for
// EVERY non-negative float — 2 139 095 041 values enumerated on gfx950, scripts/experiments/sqrt_exhaustive.hip (one
// correction after the refinement would do as well; one correction alone is wrong for half of them).  7 instructions.
__device__ __forceinline__ double sqrt_of_float(float xf)
{
  const double x = static_cast<double>(xf);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  const double h = y * 0.5;
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return __builtin_amdgcn_class(x, 0x260) ? x : g;     // +-0 and +inf map to themselves
}

// 1.0f / x, correctly rounded (:561).  hipcc's division is twelve instructions (two div_scale, rcp, six fma, div_fmas,
// div_fixup).  v_rcp_f32 and ONE Newton step give the same bits whenever the magnitude of their result is at least
// 2^-126 (i.e. the result is a normal number; a NaN fails the comparison): checked against the division on all 2^32 bit
// patterns on gfx950 (scripts/experiments/recip_exhaustive.hip) — so the test is one compare on the result, and a wave
// that holds anything else (x zero, denormal, above 2^126, infinite or NaN) takes the division.
#ifndef LBM_RECIP_DIVISION
#define LBM_RECIP_DIVISION 0          // 1: timing / cross-check builds that always divide
#endif
__device__ __forceinline__ float recip_newton(float x)
{
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float recip_exact(float x)
{
  if (LBM_RECIP_DIVISION) return 1.0f / x;
  const float q = recip_newton(x);
  if (__builtin_amdgcn_ballot_w64(__builtin_fabsf(q) >= 0x1p-126f) == __builtin_amdgcn_read_exec()) return q;
  return 1.0f / x;
}
__device__ __forceinline__ f2 recip_exact(f2 x)
{
  f2 r;
  if (LBM_RECIP_DIVISION) { r.x = 1.0f / x.x; r.y = 1.0f / x.y; return r; }
  r.x = recip_newton(x.x); r.y = recip_newton(x.y);
  const unsigned long long ok = __builtin_amdgcn_ballot_w64(__builtin_fabsf(r.x) >= 0x1p-126f) & __builtin_amdgcn_ballot_w64(__builtin_fabsf(r.y) >= 0x1p-126f);
  if (ok == __builtin_amdgcn_read_exec()) return r;
  r.x = 1.0f / x.x; r.y = 1.0f / x.y;
  return r;
}

}  // namespace
