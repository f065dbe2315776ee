// common.h — constants, argument block of the one-step kernels, loads/stores, reductions, the cell arithmetic (relax_cell), source-row selection
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "exact_math.h"      // f2, recip_exact, sqrt_of_float: the shortened exact sequences, enumerated by scripts/experiments/*_exhaustive.hip

namespace {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kCellsPerLane = 4;     // one 16-byte access per population per lane
constexpr int kHaloGuard = 4;        // floats of guard on each side of a halo-buffer row

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access

struct StepArgs {
  const float* src;            // source grid, plane 0 row 0
  float* dst;                  // destination grid
  const uint32_t* mask;        // obstacle bitfield, bit c of the partition-local cell index
  int mask_words;              // words allocated for it
  size_t ps;                   // plane stride in floats
  int nx, nyl;                 // row length, rows owned by this partition
  int quad_begin, quad_end;    // 4-cell groups [begin,end) of the partition handled by this launch
  int quad_begin2, quad_end2;  // optional second range (boundary launch: last row), empty if begin2>=end2
  int iters;                   // 1024-cell chunks per block
  // sources outside the partition (row-partitioned runs); nullptr = periodic wrap inside the plane
  const float* south_halo;     // populations 2,5,6 of the row below row 0   [3][nxp], data at +kHaloGuard
  const float* north_halo;     // populations 4,7,8 of the row above row nyl-1
  float* send_south;           // row 0's populations 4,7,8 for the southern neighbour (next step)
  float* send_north;           // row nyl-1's populations 2,5,6 for the northern neighbour
  int nxp;                     // halo-buffer row pitch = nx + 2*kHaloGuard
  int release_sends;           // the send buffers are a peer's memory: end the block with a system-scope release
  float omega;
  float accel_w1, accel_w2;    // d2q9-bgk.c:445-446
  int accel_row;               // local row that is global row ny-2, or -1: epilogue accelerate for the NEXT step
  double* partials_out;        // this launch's per-block sums
  const double* prev_partials; // previous step's per-block sums, folded by block 0 of this launch
  int n_prev;
  double* sums;                // per-step totals of this run
  int* counter;                // index of the next entry of sums
};

__device__ __forceinline__ f4 load4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 load4u(const float* p) { return *reinterpret_cast<const f4u*>(p); }

template <bool NT>
__device__ __forceinline__ void store4(float* p, f4 v)
{
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = v;
}

// Sum over the 64 lanes of a FULL wave, every lane gets the total.  Data-parallel-primitive moves (v_mov_b32 dpp: a
// register-to-register lane permutation at VALU latency) instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32
// — an LDS-crossbar round trip of ~100 cycles per level: the classic reduction quad_perm, quad_perm, row_half_mirror,
// row_mirror (now every lane of a 16-lane row holds the row's sum), row_bcast15 into rows 1 and 3, row_bcast31 into
// rows 2 and 3 (now row 3 holds the total), read lane 63.  Fixed order: deterministic.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_moved(double v)
{
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = static_cast<int>(b & 0xffffffffull), hi = static_cast<int>(b >> 32);
  // old = 0: lanes of rows outside ROW_MASK (and lanes whose source lane does not exist) receive +0.0
  const unsigned int mlo = static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false));
  const unsigned int mhi = static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(mhi) << 32) | mlo);
}

__device__ __forceinline__ double wave_sum(double v)
{
  v += dpp_moved<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v += dpp_moved<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v += dpp_moved<0x141, 0xF>(v);    // row_half_mirror
  v += dpp_moved<0x140, 0xF>(v);    // row_mirror
  v += dpp_moved<0x142, 0xA>(v);    // row_bcast15 -> rows 1, 3
  v += dpp_moved<0x143, 0xC>(v);    // row_bcast31 -> rows 2, 3
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(b & 0xffffffffull), 63));
  const unsigned int hi = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(b >> 32), 63));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}

// ---- several per-lane doubles at once (the per-step sums of a K-step kernel) ------------------------------------------------
// N wave_sum()s cost N x 6 dependent butterfly stages.  Halving butterflies do the same work in log2(N) + the rest: gfx950's
// v_permlane32_swap (lanes l <-> l ^ 32) and v_permlane16_swap (l <-> l ^ 16) hand the upper half of one value to the
// lanes that keep the other, so each stage halves the number of values a lane carries; a third halving (N = 8) goes over
// row_ror:8 (l <-> l ^ 8).  What is left is ONE value per lane, summed over the lanes of its class by DPP stages whose
// every lane has a source (no `old` operand to initialise).  ~25 (N = 4) / ~40 (N = 8) instructions instead of ~45 per value.
// All 64 lanes of the wave must be active.  Result: every lane holds the wave total of a[wave_sum_slot<N>(lane)].
template <int N> __device__ __forceinline__ int wave_sum_slot(int lane)
{
  if constexpr (N == 4) return lane >> 4;                                            // row r: a[r]
  else return ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + 4 * (lane >> 5);
}

template <int CTRL>
__device__ __forceinline__ double dpp_full(double v)                                   // CTRL reads an existing lane for every lane
{
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = static_cast<unsigned int>(__builtin_amdgcn_mov_dpp(static_cast<int>(b & 0xffffffffull), CTRL, 0xF, 0xF, false));
  const unsigned int hi = static_cast<unsigned int>(__builtin_amdgcn_mov_dpp(static_cast<int>(b >> 32), CTRL, 0xF, 0xF, false));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}

// lanes < 32: a(l) + a(l + 32); lanes >= 32: b(l - 32) + b(l)
__device__ __forceinline__ double halve_swap32(double a, double b)
{
  const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
  const auto lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned int>(ab), static_cast<unsigned int>(bb), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned int>(ab >> 32), static_cast<unsigned int>(bb >> 32), false, false);
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[0]) << 32) | lo[0]) +
         __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[1]) << 32) | lo[1]);
}

// even rows (16 lanes each): a(l) + a(l + 16); odd rows: b(l - 16) + b(l)
__device__ __forceinline__ double halve_swap16(double a, double b)
{
  const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
  const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned int>(ab), static_cast<unsigned int>(bb), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned int>(ab >> 32), static_cast<unsigned int>(bb >> 32), false, false);
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[0]) << 32) | lo[0]) +
         __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[1]) << 32) | lo[1]);
}

__device__ __forceinline__ double wave_sum_vec4(double a0, double a1, double a2, double a3)
{
  double v = halve_swap16(halve_swap32(a0, a2), halve_swap32(a1, a3));               // row r: a[r] over lanes l, l^16, l^32, l^48
  v += dpp_full<0xB1>(v);           // quad_perm [1,0,3,2]
  v += dpp_full<0x4E>(v);           // quad_perm [2,3,0,1]
  v += dpp_full<0x141>(v);          // row_half_mirror
  v += dpp_full<0x140>(v);          // row_mirror
  return v;
}

__device__ __forceinline__ double wave_sum_vec8(const double (&a)[8])
{
  const double c0 = halve_swap16(halve_swap32(a[0], a[4]), halve_swap32(a[2], a[6]));   // row r: a[2 (r & 1) + 4 (r >> 1)]
  const double c1 = halve_swap16(halve_swap32(a[1], a[5]), halve_swap32(a[3], a[7]));   //        a[1 + ...]
  const bool upper = (threadIdx.x & 8) != 0;                                             // lanes 8-15 of a row keep c1
  const double keep = upper ? c1 : c0, send = upper ? c0 : c1;
  double v = keep + dpp_full<0x128>(send);                                               // row_ror:8 = lane l ^ 8
  v += dpp_full<0xB1>(v);
  v += dpp_full<0x4E>(v);
  v += dpp_full<0x141>(v);          // row_half_mirror stays inside the eight lanes
  return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release + acquire fence on ALL memory, i.e. an
// s_waitcnt vmcnt(0) first: placed after a kernel's last global stores (the reductions at the end of every step kernel)
// it keeps the whole block — its LDS, its wave slots — waiting a store round trip (1-2 k cycles) for nothing, and placed
// between the sub-steps of lbm_multi_kernel it would wait for the NEXT tile's prefetched loads.  Here the fences name the
// LDS address space: s_waitcnt lgkmcnt(0), s_barrier, and the compiler keeps LDS accesses on their side of it.
__device__ __forceinline__ void lds_barrier()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- bounded waits on flag words of a rank's exported window (peer-to-peer loop, kernels/p2p.h; the fold block of lbm_multi_kernel) ----
// Polls are RELAXED system-scope loads of the uncached window (an acquire load per poll would invalidate this XCD's
// caches on every iteration, under the feet of the launch running beside it); ONE acquire fence follows the last
// of them.  What the flags guard is read by a later kernel of the stream in any case.
// acquire = false: nothing the flags guard is READ afterwards (the "ready" words, which only hold back this rank's stores).
__device__ __forceinline__ void p2p_wait_flags(const unsigned long long* flags, const unsigned long long* parity_words, int nflags, unsigned long long epoch,
                                               unsigned long long parity, long long timeout_ticks, int* err, bool acquire = true)
{
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
  const long long t0 = wall_clock64();
  for (int f = 0; f < nflags; ++f) {
    while (__hip_atomic_load(flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      if (wall_clock64() - t0 > timeout_ticks) {
        __hip_atomic_store(err, 1 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      __builtin_amdgcn_s_sleep(2);                             // ~128 cycles between polls: the flag is a remote write away
    }
    if (parity_words && __hip_atomic_load(parity_words + 2 * f + (epoch & 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != parity) {
      __hip_atomic_store(err, 100 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
  if (acquire) __atomic_thread_fence(__ATOMIC_ACQUIRE);        // system scope
}

// Deterministic block sum (fixed tree): every thread gets the total.
__device__ __forceinline__ double block_sum(double v, double* lds /* kBlock/64 doubles */)
{
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  lds_barrier();
  if ((threadIdx.x & 63) == 0) lds[wave] = v;
  lds_barrier();
  double t = lds[0];
#pragma unroll
  for (int w = 1; w < kBlock / 64; ++w) t += lds[w];
  return t;
}

// Block 0 of every step launch does no lattice work: it folds the PREVIOUS
// step's per-block sums into sums[counter++] (d2q9-bgk.c:367) while the other blocks stream, so the
// fold's latency (a dependent load + two barriers) is off the critical path of the tiny grids.
__device__ __forceinline__ void fold_previous(const StepArgs& a, double* red);

// One cell (T = float) or an x-pair of cells (T = f2, packed v_pk_*_f32): moments, equilibrium and
// relaxation (d2q9-bgk.c:546-666).  Every value is produced by the reference's own sequence of
// roundings; two things differ in form only:
//   - opposite directions share work.  u[3] = -u[1], u[4] = -u[2], u[7] = -u[5], u[6] = -u[8] exactly
//     (:596-603; a sum and its mirror round alike), so "u*ic_sq" of one is the negation of the other's and
//     "u*u*ic_sq", "- u_sq", "* 0.5f*densinv*ic_sq" are equal: computed once for the pair of
//     directions, with "rho - a" standing for "rho + (-a)" (18 of 113 operations less);
//   - the independent chains are written interleaved.  gfx950 needs a wait state between a packed
//     instruction and a consumer of its result, which hipcc fills with s_nop (4 cycles, as much as
//     the instruction itself) when the next instruction in program order is that consumer.
// msq = m^2 (un-normalised momentum squared), rinv = 1/rho.
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ float splat_as(float, float v) { return v; }
__device__ __forceinline__ f2 splat_as(f2, float v) { return f2{v, v}; }
template <typename T>
__device__ __forceinline__ void relax_core(const T (&t)[9], float omega, T (&o)[9], T& msq_out, T& rinv_out)
{
  const T csq_inv = splat_as(T{}, 3.0f);                                                     // :497
  const T w0 = splat_as(T{}, 4.0f / 9.0f), w1 = splat_as(T{}, 1.0f / 9.0f), w2 = splat_as(T{}, 1.0f / 36.0f);   // :499-501
  const T om = splat_as(T{}, omega), half = splat_as(T{}, 0.5f);
  T rho = t[0] + t[1];                                           // :546-554, :570-574, :576-580 side by side
  T mx = t[1] + t[5];
  T my = t[2] + t[5];
  rho += t[2]; mx += t[8]; my += t[6];
  rho += t[3]; mx -= t[3]; my -= t[4];
  rho += t[4]; mx -= t[6]; my -= t[7];
  rho += t[5]; mx -= t[7]; my -= t[8];
  rho += t[6];
  const T mxx = mx * mx, myy = my * my;
  const T e5 = mx + my, e8 = mx - my;                            // :600,603 (u[7] = -e5, u[6] = -e8)
  rho += t[7];
  const T msq = mxx + myy;                                       // :589
  T a[4] = {mx * csq_inv, my * csq_inv, e5 * csq_inv, e8 * csq_inv};        // :610-617 for k = 1, 2, 5, 8
  rho += t[8];
  const T rinv = recip_exact(rho);                               // :561
  T d[4] = {a[0] * mx, a[1] * my, a[2] * e5, a[3] * e8};         // :624-631
  const T h = half * rinv * csq_inv;                             // "0.5f*densinv*ic_sq" of :638-646
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = d[i] - msq;
  const T hm = h * msq;
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = h * d[i];
  T s[9];                                                        // :638-646
  s[0] = rho - hm;
  s[1] = rho + a[0]; s[3] = rho - a[0];
  s[2] = rho + a[1]; s[4] = rho - a[1];
  s[5] = rho + a[2]; s[7] = rho - a[2];
  s[8] = rho + a[3]; s[6] = rho - a[3];
  s[1] += d[0]; s[3] += d[0]; s[2] += d[1]; s[4] += d[1];
  s[5] += d[2]; s[7] += d[2]; s[8] += d[3]; s[6] += d[3];
  s[0] = w0 * s[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) s[k] = ((k < 5) ? w1 : w2) * s[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = s[k] - t[k];                // :658-666
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = om * s[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = t[k] + s[k];
  msq_out = msq;
  rinv_out = rinv;
}

// One cell; returns sqrt(m^2)/rho in double (:667).
__device__ __forceinline__ double relax_cell(const float (&t)[9], float omega, float (&o)[9])
{
  float msq, rinv;
  relax_core<float>(t, omega, o, msq, rinv);
  return sqrt_of_float(msq) * static_cast<double>(rinv);   // :667
}

// Row bases of the three source rows of destination row y, per population (d2q9-bgk.c:511-512,
// 526-538): here -> 0,1,3 (+k*ps); south row -> 2,5,6; north row -> 4,7,8.  Rows outside the
// partition come from the halo messages (row-partitioned runs) or wrap periodically.
struct RowPtrs {
  const float *here, *s2, *s5, *s6, *n4, *n7, *n8;
};

__device__ __forceinline__ RowPtrs source_rows(const StepArgs& a, int y)
{
  RowPtrs r;
  const size_t ps = a.ps;
  const int nx = a.nx;
  r.here = a.src + static_cast<size_t>(y) * nx;
  if (y > 0 || a.south_halo == nullptr) {
    const int ys = (y > 0) ? y - 1 : a.nyl - 1;                        // periodic wrap (:245-247 with one rank)
    const float* q = a.src + static_cast<size_t>(ys) * nx;
    r.s2 = q + 2 * ps; r.s5 = q + 5 * ps; r.s6 = q + 6 * ps;
  } else {
    const float* q = a.south_halo + kHaloGuard;
    r.s2 = q; r.s5 = q + a.nxp; r.s6 = q + 2 * a.nxp;
  }
  if (y < a.nyl - 1 || a.north_halo == nullptr) {
    const int yn = (y < a.nyl - 1) ? y + 1 : 0;
    const float* q = a.src + static_cast<size_t>(yn) * nx;
    r.n4 = q + 4 * ps; r.n7 = q + 7 * ps; r.n8 = q + 8 * ps;
  } else {
    const float* q = a.north_halo + kHaloGuard;
    r.n4 = q; r.n7 = q + a.nxp; r.n8 = q + 2 * a.nxp;
  }
  return r;
}



__device__ __forceinline__ void bounce_or_relax(const float (&t)[9], const float (&o)[9], bool blocked, float (&out)[9])
{
  out[0] = blocked ? t[0] : o[0];                                       // bounce-back d2q9-bgk.c:687-695
  out[1] = blocked ? t[3] : o[1];
  out[2] = blocked ? t[4] : o[2];
  out[3] = blocked ? t[1] : o[3];
  out[4] = blocked ? t[2] : o[4];
  out[5] = blocked ? t[7] : o[5];
  out[6] = blocked ? t[8] : o[6];
  out[7] = blocked ? t[5] : o[7];
  out[8] = blocked ? t[6] : o[8];
}

__device__ __forceinline__ void accelerate_cell(float (&out)[9], float w1, float w2)   // d2q9-bgk.c:457-469
{
  if (out[3] - w1 > 0.0f && out[6] - w2 > 0.0f && out[7] - w2 > 0.0f) {
    out[1] += w1; out[5] += w2; out[8] += w2;
    out[3] -= w1; out[6] -= w2; out[7] -= w2;
  }
}

// i / d for 0 <= i < 1024, 4 <= d <= 44: one full-rate 24-bit multiply instead of an integer division
__device__ __forceinline__ int small_div(int i, int d)
{
  return static_cast<int>(__umul24(static_cast<unsigned>(i), static_cast<unsigned>((65536 + d - 1) / d)) >> 16);
}

// base + 32-bit byte offset: the form hipcc turns into "global_load/store v_off, s[base]" (needs < 4 GiB per grid)
template <typename V, typename B>
__device__ __forceinline__ auto& at_byte(B* base, uint32_t byte_off)
{
  using P = std::conditional_t<std::is_const_v<B>, const V, V>;
  using C = std::conditional_t<std::is_const_v<B>, const char, char>;
  return *reinterpret_cast<P*>(reinterpret_cast<C*>(base) + byte_off);
}

// Two x-adjacent cells: relaxation / bounce-back select, next step's accelerate_flow, sum|u| terms.
// p[k] = streamed-in population k of the pair; mbits = their two obstacle bits.
// Returns the pair's sum|u| contribution; bit j of `skip` drops cell j from it (obstacles, cells of the
// ghost ring: they do not count, and the double-precision sqrt is a tenth of the cell's instructions).
// tile_accel is block-uniform: false for the tiles whose frame does not meet row ny-2, which then skip
// the accelerate_flow code instead of predicating it away in every pair.
// TERMS: how the sum|u| term of a cell, sqrt((double)msq) * (double)rinv (:667), is formed.  Populations never depend on it.
//   kTermsDouble (LBM_FLAG_EXACT_AVVELS; the tile and sweep kernels' default): in double precision, every term correctly
//     rounded before the product: v_rsq_f64 + 9 double-precision instructions per CELL.
//   kTermsCompensated (lbm_multi_kernel's default): without double-precision arithmetic.  The root as an unevaluated float
//     sum s + c (s = msq * rsq(msq); c = the Newton correction of s from the residual msq - s*s, which a fused multiply-add
//     delivers exactly), the product with rinv as p + lo (p = s * rinv, its rounding error by another fused multiply-add,
//     plus c * rinv): relative error ~2^-44 per cell where the double form has 2^-53 — av_vels, a float, comes out bit for
//     bit the same in every test here — in 7 packed float instructions and two v_rsq_f32 per PAIR.  Only p is widened per
//     pair; the lo parts are summed per lane in float (a lane adds at most a few per launch) and widened once.  Why it
//     matters: the launch runs AT the socket power limit (scripts/power_trace.py: 1380 of 1400 W, shader clock 2.2 of
//     2.4 GHz); double-precision instructions it does not execute come back as clock.  A cell at rest (msq = 0) gives 0:
//     the rsq argument is held at the smallest normal number.  msq = inf (a diverged run) gives NaN where the reference
//     gives inf.
//   kTermsFloat (LBM_FLAG_FAST_AVVELS): v_sqrt_f32 and one multiply; av_vels then agrees to ~1e-7 (an ulp of its float).
constexpr int kTermsDouble = 0, kTermsFloat = 1, kTermsCompensated = 2;

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }     // v_pk_fma_f32

template <int TERMS>
__device__ __forceinline__ double finish_pair_lo(const f2 (&t)[9], uint32_t mbits, float omega, bool tile_accel, bool accel, float w1, float w2,
                                                 uint32_t skip, f2 (&out)[9], float& lo_sum)
{
  f2 o[9], msq, rinv;
  relax_core<f2>(t, omega, o, msq, rinv);                               // :546-666 on both cells at once (v_pk_*_f32)
  // bounce-back select (d2q9-bgk.c:687-695) and the next step's accelerate_flow (:457-469), per cell
  static constexpr int opp[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
#pragma unroll
  for (int k = 0; k < 9; ++k) out[k] = o[k];
  if (__builtin_amdgcn_ballot_w64(mbits != 0u) != 0ull) {          // wave-uniform: most waves hold no obstacle
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool blocked = (mbits >> j) & 1u;
#pragma unroll
      for (int k = 0; k < 9; ++k) out[k][j] = blocked ? t[opp[k]][j] : o[k][j];
    }
  }
  if (tile_accel) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float r[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) r[k] = out[k][j];
      if (accel && !((mbits >> j) & 1u)) accelerate_cell(r, w1, w2);
#pragma unroll
      for (int k = 0; k < 9; ++k) out[k][j] = r[k];
    }
  }
  double term = 0.0;
  if (skip != 3u) {
    if constexpr (TERMS == kTermsFloat) {
      const f2 root = f2{__builtin_amdgcn_sqrtf(msq.x), __builtin_amdgcn_sqrtf(msq.y)};
      const f2 tf = root * rinv;
      term = static_cast<double>(((skip & 1u) ? 0.0f : tf.x) + ((skip & 2u) ? 0.0f : tf.y));
    } else if constexpr (TERMS == kTermsCompensated) {
      const f2 y = f2{__builtin_amdgcn_rsqf(__builtin_fmaxf(msq.x, 0x1p-126f)), __builtin_amdgcn_rsqf(__builtin_fmaxf(msq.y, 0x1p-126f))};
      const f2 s = msq * y;                                  // ~sqrt(msq)
      const f2 r = fma2(-s, s, msq);                         // msq - s*s
      const f2 c = (r * y) * 0.5f;                           // sqrt(msq) ~ s + c
      const f2 p = s * rinv;
      const f2 e = fma2(s, rinv, -p);                        // s * rinv = p + e exactly
      const f2 lo = fma2(c, rinv, e);
      term = static_cast<double>((skip & 1u) ? 0.0f : p.x) + static_cast<double>((skip & 2u) ? 0.0f : p.y);
      lo_sum += ((skip & 1u) ? 0.0f : lo.x) + ((skip & 2u) ? 0.0f : lo.y);
    } else {
      const double t0 = sqrt_of_float(msq.x) * static_cast<double>(rinv.x);   // :667
      const double t1 = sqrt_of_float(msq.y) * static_cast<double>(rinv.y);
      term = ((skip & 1u) ? 0.0 : t0) + ((skip & 2u) ? 0.0 : t1);
    }
  }
  return term;
}

// The same with the pair's sum as one double (the tile and sweep kernels: TERMS = false / true = double / float).
template <int TERMS = kTermsDouble>
__device__ __forceinline__ double finish_pair(const f2 (&t)[9], uint32_t mbits, float omega, bool tile_accel, bool accel, float w1, float w2,
                                              uint32_t skip, f2 (&out)[9])
{
  float lo = 0.0f;
  const double hi = finish_pair_lo<TERMS>(t, mbits, omega, tile_accel, accel, w1, w2, skip, out, lo);
  return TERMS == kTermsCompensated ? hi + static_cast<double>(lo) : hi;
}


}  // namespace
