// lbm_kernels.hip — hand-written gfx950 (MI355X / CDNA4) kernels + device half of the C ABI
// (include/lbm_d2q9.h) for the D2Q9-BGK timestep path of ag14774/MPILattice-Boltzmann.
//
// What one launch of lbm_step_* does (reference lines, relative to the reference tree):
//   pull-stream            d2q9-bgk.c:526-538      9 populations from the 3x3 neighbourhood
//   moments + equilibrium  d2q9-bgk.c:546-646
//   BGK relaxation         d2q9-bgk.c:658-666      fluid cells
//   bounce-back            d2q9-bgk.c:687-695      obstacle cells
//   sum |u|                d2q9-bgk.c:667,684      -> per-block partial, double
//   accelerate_flow        d2q9-bgk.c:442-478      fused as an EPILOGUE on global row ny-2: the row is
//                                                  written already accelerated for the next step
//   av_vels[tt-1]          d2q9-bgk.c:367          block 0 folds the previous launch's partials
//
// Kernels in this file (all produce the same bits; lbm_run / the partitioned loop pick by grid size):
//   lbm_step_kernel / _narrow / _lds   one step per launch (4 cells, 1 cell per lane; LDS-staged variant)
//   lbm_multi_kernel<K>                K steps per pass over HBM, 64x16 tiles, intermediate states in LDS
//   lbm_tile_kernel<T,H>               up to H steps per launch for the launch-latency-bound small grids
//
// Layout in HBM: struct-of-arrays, 9 planes of ny_local*nx floats (plane stride padded, see
// plane_stride_floats()), two grids (source / destination, swapped per step like :376-378), the
// obstacle map as a bitfield (1 bit per cell).  Every population value is consumed by exactly one
// cell per step, so the kernel is a pure stream: 9 x 16-byte loads + 9 x 16-byte stores per lane
// (4 cells per lane), no MFMA, bounded by HBM bandwidth.
//
// Arithmetic is written in the reference's operation order and this file is compiled with
// -ffp-contract=off, so the post-step populations are BIT-IDENTICAL to the reference's
// (gcc -std=c99 never fuses either); 1.0f/x and sqrt are correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).  Only the summation order of sum|u| differs
// (tree in double instead of a serial float accumulator).
//
// gfx950 only: 64-wide wavefronts are assumed throughout (wave reductions step through 32..1).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lbm_d2q9.h"
#include "lbm_internal.h"

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

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Deterministic block sum (fixed tree): every thread gets the total.
__device__ __forceinline__ double block_sum(double v, double* lds /* kBlock/64 doubles */)
{
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[wave] = v;
  __syncthreads();
  double t = lds[0];
#pragma unroll
  for (int w = 1; w < kBlock / 64; ++w) t += lds[w];
  return t;
}

// Block 0 of every step launch does no lattice work: it folds the PREVIOUS
// step's per-block sums into sums[counter++] (d2q9-bgk.c:367) while the other blocks stream, so the
// fold's latency (a dependent load + two barriers) is off the critical path of the tiny grids.
__device__ __forceinline__ void fold_previous(const StepArgs& a, double* red);

// One cell: moments, equilibrium, relaxation in the reference's operation order (d2q9-bgk.c:546-666).
// t[] = streamed-in populations, o[] = relaxed populations; returns sqrt(m^2)/rho in double (:667).
__device__ __forceinline__ void relax_cell_core(const float (&t)[9], float omega, float (&o)[9], float& msq_out, float& rinv_out);

__device__ __forceinline__ double relax_cell(const float (&t)[9], float omega, float (&o)[9])
{
  float msq, rinv;
  relax_cell_core(t, omega, o, msq, rinv);
  return sqrt(static_cast<double>(msq)) * static_cast<double>(rinv);   // :667
}

// The same without the sum|u| term: msq = m^2 (un-normalised momentum squared), rinv = 1/rho.
__device__ __forceinline__ void relax_cell_core(const float (&t)[9], float omega, float (&o)[9], float& msq_out, float& rinv_out)
{
  const float csq_inv = 3.0f;                                   // :497
  const float w0 = 4.0f / 9.0f, w1 = 1.0f / 9.0f, w2 = 1.0f / 36.0f;   // :499-501
  float rho = t[0];                                             // :546-554
  rho += t[1]; rho += t[2]; rho += t[3]; rho += t[4];
  rho += t[5]; rho += t[6]; rho += t[7]; rho += t[8];
  const float rinv = 1.0f / rho;                                // :561
  float mx = t[1] + t[5];                                       // :570-574
  mx += t[8]; mx -= t[3]; mx -= t[6]; mx -= t[7];
  float my = t[2] + t[5];                                       // :576-580
  my += t[6]; my -= t[4]; my -= t[7]; my -= t[8];
  const float msq = mx * mx + my * my;                          // :589
  float e[9];
  e[1] = mx;       e[2] = my;        e[3] = -mx;       e[4] = -my;        // :596-599
  e[5] = mx + my;  e[6] = -mx + my;  e[7] = -mx - my;  e[8] = mx - my;    // :600-603
  const float h = 0.5f * rinv * csq_inv;                        // "0.5f*densinv*ic_sq" of :638-646
  const float q0 = w0 * (rho - h * msq);                        // :638
  o[0] = t[0] + omega * (q0 - t[0]);                            // :658
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const float a = e[k] * csq_inv;                             // :610-617
    const float b = a * e[k];                                   // :624-631
    const float wk = (k < 5) ? w1 : w2;
    const float q = wk * (rho + a + h * (b - msq));             // :639-646
    o[k] = t[k] + omega * (q - t[k]);                           // :659-666
  }
  msq_out = msq;
  rinv_out = rinv;
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

// Everything after the pull for the 4 cells at (y, x0..x0+3), partition-local cell index c:
// relaxation / bounce-back select, next step's accelerate_flow on row ny-2, stores, outgoing halo
// rows.  p[k] = streamed-in population k of the four cells.  Returns their sum|u| contribution.
template <bool NT>
__device__ __forceinline__ double finish_quad(const StepArgs& a, int c, int y, int x0, const f4 (&p)[9], uint32_t mbits)
{
  const size_t ps = a.ps;
  f4 out[9];
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < kCellsPerLane; ++j) {
    float t[9], o[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = p[k][j];
    const double term = relax_cell(t, a.omega, o);
    const bool blocked = (mbits >> j) & 1u;
    // bounce-back (d2q9-bgk.c:687-695): out[opposite(k)] = t[k]
    out[0][j] = blocked ? t[0] : o[0];
    out[1][j] = blocked ? t[3] : o[1];
    out[2][j] = blocked ? t[4] : o[2];
    out[3][j] = blocked ? t[1] : o[3];
    out[4][j] = blocked ? t[2] : o[4];
    out[5][j] = blocked ? t[7] : o[5];
    out[6][j] = blocked ? t[8] : o[6];
    out[7][j] = blocked ? t[5] : o[7];
    out[8][j] = blocked ? t[6] : o[8];
    acc += blocked ? 0.0 : term;
  }

  // accelerate_flow for the NEXT step, applied to the freshly written row ny-2 (d2q9-bgk.c:457-469)
  if (y == a.accel_row) {
#pragma unroll
    for (int j = 0; j < kCellsPerLane; ++j) {
      const bool blocked = (mbits >> j) & 1u;
      if (!blocked && out[3][j] - a.accel_w1 > 0.0f && out[6][j] - a.accel_w2 > 0.0f &&
          out[7][j] - a.accel_w2 > 0.0f) {
        out[1][j] += a.accel_w1; out[5][j] += a.accel_w2; out[8][j] += a.accel_w2;
        out[3][j] -= a.accel_w1; out[6][j] -= a.accel_w2; out[7][j] -= a.accel_w2;
      }
    }
  }

  float* d = a.dst + c;
#pragma unroll
  for (int k = 0; k < 9; ++k) store4<NT>(d + k * ps, out[k]);

  // next step's outgoing halo rows (row-partitioned runs only)
  if (a.send_south != nullptr && y == 0) {
    float* s = a.send_south + kHaloGuard + x0;
    store4<false>(s, out[4]); store4<false>(s + a.nxp, out[7]); store4<false>(s + 2 * a.nxp, out[8]);
  }
  if (a.send_north != nullptr && y == a.nyl - 1) {
    float* s = a.send_north + kHaloGuard + x0;
    store4<false>(s, out[2]); store4<false>(s + a.nxp, out[5]); store4<false>(s + 2 * a.nxp, out[6]);
  }
  return acc;
}

// Direct-load form: processes the 4 cells starting at partition-local cell index 4*quad
// (nx % 4 == 0, so the four share a row).
template <bool NT>
__device__ __forceinline__ double step_quad(const StepArgs& a, int quad)
{
  const int c = quad * kCellsPerLane;
  const int y = c / a.nx;
  const int x0 = c - y * a.nx;
  const size_t ps = a.ps;
  const int nx = a.nx;
  const RowPtrs r = source_rows(a, y);

  // pull (d2q9-bgk.c:530-538): aligned for x, dword-shifted for x-1 / x+1
  f4 p[9];
  p[0] = load4(r.here + x0);
  p[2] = load4(r.s2 + x0);
  p[4] = load4(r.n4 + x0);
  p[1] = load4u(r.here + ps + x0 - 1);
  p[5] = load4u(r.s5 + x0 - 1);
  p[8] = load4u(r.n8 + x0 - 1);
  p[3] = load4u(r.here + 3 * ps + x0 + 1);
  p[6] = load4u(r.s6 + x0 + 1);
  p[7] = load4u(r.n7 + x0 + 1);
  const uint32_t mword = a.mask[c >> 5];
  if (x0 == 0) {                       // x_w wraps to nx-1 (:529)
    p[1].x = r.here[ps + nx - 1];
    p[5].x = r.s5[nx - 1];
    p[8].x = r.n8[nx - 1];
  }
  if (x0 == nx - kCellsPerLane) {      // x_e wraps to 0 (:527-528)
    p[3].w = r.here[3 * ps];
    p[6].w = r.s6[0];
    p[7].w = r.n7[0];
  }
  const uint32_t mbits = (mword >> (c & 31)) & 0xFu;
  return finish_quad<NT>(a, c, y, x0, p, mbits);
}

// One-cell-per-lane form: used for grids so small that a step is bound by the latency of one lane's
// dependent instruction chain rather than by bandwidth (4x more lanes, each with a quarter of the
// chain), and for row lengths that are not a multiple of 4.  `cell` = partition-local cell index.
template <bool NT>
__device__ __forceinline__ double step_cell(const StepArgs& a, int cell)
{
  const int y = cell / a.nx;
  const int x = cell - y * a.nx;
  const size_t ps = a.ps;
  const int nx = a.nx;
  const RowPtrs r = source_rows(a, y);
  const int xe = (x + 1 >= nx) ? x + 1 - nx : x + 1;                   // :527-528
  const int xw = (x == 0) ? nx - 1 : x - 1;                             // :529
  float t[9], o[9];
  t[0] = r.here[x];            t[1] = r.here[ps + xw];      t[2] = r.s2[x];      // :530-532
  t[3] = r.here[3 * ps + xe];  t[4] = r.n4[x];              t[5] = r.s5[xw];     // :533-535
  t[6] = r.s6[xe];             t[7] = r.n7[xe];             t[8] = r.n8[xw];     // :536-538
  const bool blocked = (a.mask[cell >> 5] >> (cell & 31)) & 1u;
  const double term = relax_cell(t, a.omega, o);
  float out[9];
  out[0] = blocked ? t[0] : o[0];                                       // bounce-back :687-695
  out[1] = blocked ? t[3] : o[1];
  out[2] = blocked ? t[4] : o[2];
  out[3] = blocked ? t[1] : o[3];
  out[4] = blocked ? t[2] : o[4];
  out[5] = blocked ? t[7] : o[5];
  out[6] = blocked ? t[8] : o[6];
  out[7] = blocked ? t[5] : o[7];
  out[8] = blocked ? t[6] : o[8];
  if (y == a.accel_row && !blocked && out[3] - a.accel_w1 > 0.0f && out[6] - a.accel_w2 > 0.0f &&
      out[7] - a.accel_w2 > 0.0f) {                                     // next step's accelerate_flow :457-469
    out[1] += a.accel_w1; out[5] += a.accel_w2; out[8] += a.accel_w2;
    out[3] -= a.accel_w1; out[6] -= a.accel_w2; out[7] -= a.accel_w2;
  }
  float* d = a.dst + cell;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    if (NT) __builtin_nontemporal_store(out[k], d + k * ps);
    else d[k * ps] = out[k];
  }
  if (a.send_south != nullptr && y == 0) {
    float* s = a.send_south + kHaloGuard + x;
    s[0] = out[4]; s[a.nxp] = out[7]; s[2 * a.nxp] = out[8];
  }
  if (a.send_north != nullptr && y == a.nyl - 1) {
    float* s = a.send_north + kHaloGuard + x;
    s[0] = out[2]; s[a.nxp] = out[5]; s[2 * a.nxp] = out[6];
  }
  return blocked ? 0.0 : term;
}

// LDS-staged form (LBM_FLAG_KERNEL_LDS), the tiling BASELINE.json's north_star sentence describes:
// every global load is 16-byte aligned; the x-1 / x+1 values a lane needs from its neighbours'
// vectors travel through an LDS tile row with one halo column per side (filled from global memory
// by the first / last lane of the block), and the chunk's 1024 obstacle bits sit in LDS as a
// bitfield.  Same arithmetic, same results; kept as a measured alternative.
struct LdsTile {
  float w[3][kBlock + 2];      // [k][1+lane] = .w of populations 1,5,8 of that lane: the x-1 source of lane+1; [0] = halo
  float e[3][kBlock + 2];      // [k][1+lane] = .x of populations 3,6,7: the x+1 source of lane-1; [kBlock+1] = halo
  uint32_t mask[kBlock / 8 + 1];   // the (up to) 33 words holding the chunk's 1024 obstacle bits
};

template <bool NT>
__device__ __forceinline__ double step_quad_lds(const StepArgs& a, int quad, bool active, int chunk_first_cell, LdsTile& tile)
{
  const int tid = threadIdx.x;
  const int c = quad * kCellsPerLane;
  const int y = active ? c / a.nx : 0;
  const int x0 = c - y * a.nx;
  const size_t ps = a.ps;
  const int nx = a.nx;
  const int word0 = chunk_first_cell >> 5;
  f4 p[9];
  float hw[3] = {0.f, 0.f, 0.f}, he[3] = {0.f, 0.f, 0.f};
  const bool row_start = active && x0 == 0;                       // x_w wraps to nx-1 (:529)
  const bool row_end = active && x0 == nx - kCellsPerLane;        // x_e wraps to 0   (:527-528)
  if (tid <= kBlock / 8 && word0 + tid < a.mask_words) tile.mask[tid] = a.mask[word0 + tid];
  if (active) {
    const RowPtrs r = source_rows(a, y);
    p[0] = load4(r.here + x0);
    p[1] = load4(r.here + ps + x0);
    p[2] = load4(r.s2 + x0);
    p[3] = load4(r.here + 3 * ps + x0);
    p[4] = load4(r.n4 + x0);
    p[5] = load4(r.s5 + x0);
    p[6] = load4(r.s6 + x0);
    p[7] = load4(r.n7 + x0);
    p[8] = load4(r.n8 + x0);
    tile.w[0][tid + 1] = p[1].w; tile.w[1][tid + 1] = p[5].w; tile.w[2][tid + 1] = p[8].w;
    tile.e[0][tid + 1] = p[3].x; tile.e[1][tid + 1] = p[6].x; tile.e[2][tid + 1] = p[7].x;
    // halo columns of the tile row (only the block's first / last lane have no neighbour lane) and
    // the periodic wrap for lanes sitting on a row edge inside the block
    if (tid == 0 || row_start) {
      const int xw = row_start ? nx - 1 : x0 - 1;
      hw[0] = r.here[ps + xw]; hw[1] = r.s5[xw]; hw[2] = r.n8[xw];
      if (tid == 0) { tile.w[0][0] = hw[0]; tile.w[1][0] = hw[1]; tile.w[2][0] = hw[2]; }
    }
    if (tid == kBlock - 1 || row_end) {
      const int xe = row_end ? 0 : x0 + kCellsPerLane;
      he[0] = r.here[3 * ps + xe]; he[1] = r.s6[xe]; he[2] = r.n7[xe];
      if (tid == kBlock - 1) { tile.e[0][kBlock + 1] = he[0]; tile.e[1][kBlock + 1] = he[1]; tile.e[2][kBlock + 1] = he[2]; }
    }
  }
  __syncthreads();
  double acc = 0.0;
  if (active) {
    const float w1 = row_start ? hw[0] : tile.w[0][tid], w5 = row_start ? hw[1] : tile.w[1][tid],
                w8 = row_start ? hw[2] : tile.w[2][tid];
    const float e3 = row_end ? he[0] : tile.e[0][tid + 2], e6 = row_end ? he[1] : tile.e[1][tid + 2],
                e7 = row_end ? he[2] : tile.e[2][tid + 2];
    const f4 c1 = p[1], c5 = p[5], c8 = p[8], c3 = p[3], c6 = p[6], c7 = p[7];
    p[1] = f4{w1, c1.x, c1.y, c1.z};
    p[5] = f4{w5, c5.x, c5.y, c5.z};
    p[8] = f4{w8, c8.x, c8.y, c8.z};
    p[3] = f4{c3.y, c3.z, c3.w, e3};
    p[6] = f4{c6.y, c6.z, c6.w, e6};
    p[7] = f4{c7.y, c7.z, c7.w, e7};
    const uint32_t mbits = (tile.mask[(c >> 5) - word0] >> (c & 31)) & 0xFu;
    acc = finish_quad<NT>(a, c, y, x0, p, mbits);
  }
  __syncthreads();   // the tile is rewritten by the next chunk
  return acc;
}

// Same grid / chunk mapping as lbm_step_kernel, single contiguous quad range only (the two-row
// boundary launch of a row-partitioned run always uses the direct form).
template <bool NT>
__global__ void __launch_bounds__(kBlock) lbm_step_kernel_lds(const StepArgs a)
{
  __shared__ double red[kBlock / 64];
  __shared__ LdsTile tile;
  if (blockIdx.x == 0) { fold_previous(a, red); return; }
  const int wblock = blockIdx.x - 1;   // work block index
  double acc = 0.0;
  const int n1 = a.quad_end - a.quad_begin;
  for (int i = 0; i < a.iters; ++i) {
    const int r0 = (wblock * a.iters + i) * kBlock;          // block-uniform: every lane reaches the barriers
    if (r0 >= n1) break;
    const int r = r0 + threadIdx.x;
    acc += step_quad_lds<NT>(a, a.quad_begin + r, r < n1, (a.quad_begin + r0) * kCellsPerLane, tile);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) a.partials_out[wblock] = acc;
}

// The fused streaming-pull step.  Grid: ceil(#quads / (256*iters)) work blocks of 256 lanes (block b
// owns `iters` consecutive 1024-cell chunks) after one fold block (block 0, dispatched first).
template <bool NT>
__global__ void __launch_bounds__(kBlock) lbm_step_kernel(const StepArgs a)
{
  __shared__ double red[kBlock / 64];
  if (blockIdx.x == 0) { fold_previous(a, red); return; }
  const int wblock = blockIdx.x - 1;   // work block index
  double acc = 0.0;
  const int n1 = a.quad_end - a.quad_begin;
  const int n2 = a.quad_end2 > a.quad_begin2 ? a.quad_end2 - a.quad_begin2 : 0;
  const int base = wblock * a.iters * kBlock + threadIdx.x;
  for (int i = 0; i < a.iters; ++i) {
    const int r = base + i * kBlock;
    if (r < n1 + n2) acc += step_quad<NT>(a, r < n1 ? a.quad_begin + r : a.quad_begin2 + (r - n1));
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) a.partials_out[wblock] = acc;
}

// One cell per lane; the unit ranges of StepArgs are cell ranges here.
template <bool NT>
__global__ void __launch_bounds__(kBlock) lbm_step_kernel_narrow(const StepArgs a)
{
  __shared__ double red[kBlock / 64];
  if (blockIdx.x == 0) { fold_previous(a, red); return; }
  const int wblock = blockIdx.x - 1;   // work block index
  double acc = 0.0;
  const int n1 = a.quad_end - a.quad_begin;
  const int n2 = a.quad_end2 > a.quad_begin2 ? a.quad_end2 - a.quad_begin2 : 0;
  const int base = wblock * a.iters * kBlock + threadIdx.x;
  for (int i = 0; i < a.iters; ++i) {
    const int r = base + i * kBlock;
    if (r < n1 + n2) acc += step_cell<NT>(a, r < n1 ? a.quad_begin + r : a.quad_begin2 + (r - n1));
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) a.partials_out[wblock] = acc;
}

__device__ __forceinline__ void fold_previous(const StepArgs& a, double* red)
{
  if (a.n_prev <= 0) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < a.n_prev; i += kBlock) s += a.prev_partials[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const int t = *a.counter;
    a.sums[t] = s;
    *a.counter = t + 1;
  }
}

// ------------------------------------------------------------------------------------------------
// Temporally blocked form for launch-latency-bound grids (the three small shipped decks).
//
// A step of a <= 256 K-cell grid takes less time to compute than a kernel boundary costs, so one
// launch here advances the lattice by up to H steps (e.g. T = 16, H = 8: a 1024-lane block loads a 32x32
// region (a 16x16 owned tile + an 8-cell ghost ring, periodic in x and y) into LDS, every lane keeps
// ONE region cell for the whole launch, and sub-step s recomputes the region shrunk by s cells from
// the LDS copy of sub-step s-1 (double-buffered, one barrier per sub-step).  Ghost cells are computed
// redundantly by neighbouring blocks with the same arithmetic, so no block ever waits for another.
// Per-step sum|u| is taken over owned cells only; accelerate_flow is applied to row ny-2 (ghost
// copies too) between sub-steps exactly as between launches of the one-step kernels.  Results are
// bit-identical to the one-step kernels (same relax_cell, same order of steps).
// ------------------------------------------------------------------------------------------------
// Geometry is a template parameter pair: T = owned tile edge, H = ghost ring = max steps per launch;
// the region edge is R = T + 2H and the block has R*R lanes (<= 1024).
constexpr int kMaxTileSteps = 8;

template <int T, int H>
struct TileGeom {
  static constexpr int R = T + 2 * H;
  static constexpr int lanes = R * R;
  static constexpr int waves = (lanes + 63) / 64;
  static constexpr size_t lds_bytes = sizeof(float) * 2 * 9 * lanes + sizeof(double) * H * waves;
  static_assert(lanes <= 1024 && H <= kMaxTileSteps, "block too large");
};

struct TileArgs {
  const float* src;
  float* dst;
  const uint32_t* mask;
  size_t ps;
  int nx, ny;
  int tiles_x;                 // nx / T
  int ksteps;                  // 1..H steps in this launch
  float omega, accel_w1, accel_w2;
  int accel_row;               // ny-2
  int accel_last;              // apply accelerate_flow after the LAST sub-step too (another step follows)
  double* partials_out;        // [ksteps][ntiles]
  const double* prev_partials; // previous launch: [n_prev_vecs][n_prev]
  int n_prev, n_prev_vecs;
  double* sums;
  int* counter;
};

template <int T, int H>
__global__ void __launch_bounds__((T + 2 * H) * (T + 2 * H)) lbm_tile_kernel(const TileArgs a)
{
  using G = TileGeom<T, H>;
  constexpr int R = G::R, kLanes = G::lanes, kWaves = G::waves;
  extern __shared__ __attribute__((aligned(16))) float lds[];        // [2][9][lanes] floats, then reduction scratch
  double* red = reinterpret_cast<double*>(lds + 2 * 9 * kLanes);    // [H][kWaves]
  const int tid = threadIdx.x;

  if (blockIdx.x == 0) {
    // fold block: the previous launch's per-tile sums, one vector per step, into sums[counter..]
    for (int v = 0; v < a.n_prev_vecs; ++v) {
      double s = 0.0;
      for (int i = tid; i < a.n_prev; i += kLanes) s += a.prev_partials[static_cast<size_t>(v) * a.n_prev + i];
      s = wave_sum(s);
      __syncthreads();
      if ((tid & 63) == 0) red[tid >> 6] = s;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kWaves; ++w) t += red[w];
        a.sums[*a.counter + v] = t;
      }
    }
    __syncthreads();
    if (tid == 0 && a.n_prev_vecs > 0) *a.counter += a.n_prev_vecs;
    return;
  }

  const int tile = blockIdx.x - 1;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int ry = tid / R, rx = tid - ry * R;
  // global cell of this lane, periodic (d2q9-bgk.c:527-529 in x; :245-247 one-rank ring in y)
  int gx = (tx * T - H + rx) % a.nx; if (gx < 0) gx += a.nx;
  int gy = (ty * T - H + ry) % a.ny; if (gy < 0) gy += a.ny;
  const int cell = gy * a.nx + gx;
  const bool blocked = (a.mask[cell >> 5] >> (cell & 31)) & 1u;
  const bool owned = rx >= H && rx < H + T && ry >= H && ry < H + T;
  const bool on_accel_row = gy == a.accel_row;

  float* bufA = lds;
  float* bufB = lds + 9 * kLanes;
#pragma unroll
  for (int k = 0; k < 9; ++k) bufA[k * kLanes + tid] = a.src[k * a.ps + cell];
  __syncthreads();

  double acc[H];
#pragma unroll
  for (int i = 0; i < H; ++i) acc[i] = 0.0;
  float out[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) out[k] = 0.0f;

  const int k_total = a.ksteps;
#pragma unroll 1
  for (int s = 1; s <= k_total; ++s) {
    // cells still needed after this sub-step: the owned tile expanded by (k_total - s)
    const int e = k_total - s;
    const bool active = rx >= H - e && rx < H + T + e && ry >= H - e && ry < H + T + e;
    if (active) {
      float t[9], o[9];
      const int here = tid, south = tid - R, north = tid + R;                   // d2q9-bgk.c:530-538
      t[0] = bufA[0 * kLanes + here];
      t[1] = bufA[1 * kLanes + here - 1];
      t[2] = bufA[2 * kLanes + south];
      t[3] = bufA[3 * kLanes + here + 1];
      t[4] = bufA[4 * kLanes + north];
      t[5] = bufA[5 * kLanes + south - 1];
      t[6] = bufA[6 * kLanes + south + 1];
      t[7] = bufA[7 * kLanes + north + 1];
      t[8] = bufA[8 * kLanes + north - 1];
      const double term = relax_cell(t, a.omega, o);
      out[0] = blocked ? t[0] : o[0];                                          // bounce-back :687-695
      out[1] = blocked ? t[3] : o[1];
      out[2] = blocked ? t[4] : o[2];
      out[3] = blocked ? t[1] : o[3];
      out[4] = blocked ? t[2] : o[4];
      out[5] = blocked ? t[7] : o[5];
      out[6] = blocked ? t[8] : o[6];
      out[7] = blocked ? t[5] : o[7];
      out[8] = blocked ? t[6] : o[8];
      if (owned && !blocked) {
#pragma unroll
        for (int i = 0; i < H; ++i)
          if (i == s - 1) acc[i] = term;
      }
      // accelerate_flow of the following step (d2q9-bgk.c:457-469)
      if (on_accel_row && !blocked && (s < k_total || a.accel_last) && out[3] - a.accel_w1 > 0.0f &&
          out[6] - a.accel_w2 > 0.0f && out[7] - a.accel_w2 > 0.0f) {
        out[1] += a.accel_w1; out[5] += a.accel_w2; out[8] += a.accel_w2;
        out[3] -= a.accel_w1; out[6] -= a.accel_w2; out[7] -= a.accel_w2;
      }
      if (s < k_total) {
#pragma unroll
        for (int k = 0; k < 9; ++k) bufB[k * kLanes + tid] = out[k];
      }
    }
    __syncthreads();
    float* sw = bufA; bufA = bufB; bufB = sw;
  }

  if (owned) {
#pragma unroll
    for (int k = 0; k < 9; ++k) a.dst[k * a.ps + cell] = out[k];
  }

  // per-step sums over the owned cells of this tile: wave trees, then one lane per step over the waves
  const int ntiles = gridDim.x - 1;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const double w = wave_sum(acc[i]);
    if ((tid & 63) == 0) red[i * kWaves + (tid >> 6)] = w;
  }
  __syncthreads();
  if (tid < k_total) {
    double t = 0.0;
    for (int w = 0; w < kWaves; ++w) t += red[tid * kWaves + w];
    a.partials_out[static_cast<size_t>(tid) * ntiles + tile] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// K steps per pass over HBM for bandwidth-bound grids: lbm_multi_kernel<K>.
//
// The one-step kernel moves 72 B per cell-step and sits at ~90 % of what HBM delivers; the only way
// further up is to touch memory less often.  Here a 512-lane block owns a 64x16 tile and advances it
// by up to K steps per launch: sub-step 1 pulls straight from the source grid (as the one-step kernel
// does) for the tile plus a (K-1)-cell ring and keeps the result in LDS; sub-steps 2..K update that
// LDS frame in place (neighbours read into registers, barrier, results written back), each on a
// region one cell smaller; the last sub-step covers exactly the owned tile and writes the
// destination grid.  The ring is recomputed redundantly by the neighbouring blocks with the same
// arithmetic, so no block ever waits for another, and the results are bit-identical to K launches of
// the one-step kernel.  HBM traffic per K steps: (64+2K)(16+2K)/1024 x 36 B read + 36 B written
// (K = 2: 84 B instead of 144 B; K = 4: 97 B instead of 288 B).
//
// Rows outside the partition: `y_periodic` wraps (self-contained domain); otherwise the storage has
// `ghost` extra rows below and above the owned rows, filled by the neighbours before the launch.
// ------------------------------------------------------------------------------------------------
constexpr int kMTX = 64, kMTY = 16, kMLanes = 512, kMaxMultiSteps = 4;

// Sub-step j of k (1-based) works on the owned tile grown by (k-j) rows and 2(k-j) columns on each
// side: columns grow twice as fast so that every region starts on an even x and a lane can own an
// x-PAIR of cells (8-byte accesses; the two cells' arithmetic is packed by the compiler into
// v_pk_*_f32, which halves the instruction count - the one-cell form of this kernel was VALU-bound).
template <int K>
struct MultiGeom {
  static constexpr int EY = K - 1, EX = 2 * (K - 1);                // growth of the first sub-step
  static constexpr int W = kMTX + 2 * EX, H = kMTY + 2 * EY;        // LDS frame
  static constexpr int cells = W * H;
  static constexpr int pairs2 = K >= 2 ? ((kMTX + 4 * (K - 2)) / 2) * (kMTY + 2 * (K - 2)) : 0;   // largest in-LDS region
  static constexpr int passes = (pairs2 + kMLanes - 1) / kMLanes;
  static constexpr size_t lds_bytes = sizeof(float) * 9 * cells + sizeof(double) * K * (kMLanes / 64);
};

struct MultiArgs {
  const float* src;
  float* dst;
  const uint32_t* mask;        // bit per STORAGE cell (ghost rows included)
  size_t ps;
  int nx;
  int rows_owned;              // owned rows
  int ghost;                   // storage rows before the first owned row (0 when y_periodic)
  int y_periodic;
  int y0_global, ny_global;    // global row of the first owned row; global grid height
  int tiles_x;
  int tile_begin, tile_count, tile_begin2, tile_count2;   // tile ranges of this launch (second may be empty)
  int ntiles_total;            // stride of partials_out
  int ksteps;                  // 1..K steps in this launch
  int xcd_remap;               // tile order: contiguous eighth per XCD (needs (tile_count+tile_count2) % 8 == 0)
  float omega, accel_w1, accel_w2;
  int accel_row;               // GLOBAL row ny-2
  int accel_last;
  double* partials_out;        // [ksteps][ntiles_total]
  const double* prev_partials; // previous launch: [n_prev_vecs][n_prev]
  int n_prev, n_prev_vecs;
  double* sums;
  int* counter;
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

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

// Two x-adjacent cells: relaxation / bounce-back select, next step's accelerate_flow, sum|u| terms.
// p[k] = streamed-in population k of the pair; mbits = their two obstacle bits.
// Returns the pair's sum|u| contribution (0 unless want_term: ghost-ring cells do not count, and the
// double-precision sqrt is a tenth of the cell's instructions).
__device__ __forceinline__ double finish_pair(const f2 (&p)[9], uint32_t mbits, float omega, bool accel, float w1, float w2,
                                              bool want_term, f2 (&out)[9])
{
  float msq[2], rinv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float t[9], o[9], r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = p[k][j];
    relax_cell_core(t, omega, o, msq[j], rinv[j]);
    const bool blocked = (mbits >> j) & 1u;
    bounce_or_relax(t, o, blocked, r);
    if (accel && !blocked) accelerate_cell(r, w1, w2);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[k][j] = r[k];
  }
  double term = 0.0;
  if (want_term) {
    const double t0 = sqrt(static_cast<double>(msq[0])) * static_cast<double>(rinv[0]);   // :667
    const double t1 = sqrt(static_cast<double>(msq[1])) * static_cast<double>(rinv[1]);
    term = ((mbits & 1u) ? 0.0 : t0) + ((mbits & 2u) ? 0.0 : t1);
  }
  return term;
}

template <int K, bool FULL>   // FULL: this launch does exactly K steps (all region sizes are compile-time constants)
__global__ void __launch_bounds__(kMLanes) lbm_multi_kernel(const MultiArgs a)
{
  using G = MultiGeom<K>;
  constexpr int EX = G::EX, EY = G::EY, W = G::W, kCells = G::cells, kWaves = kMLanes / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [9][kCells], then [K][kWaves] doubles
  double* red = reinterpret_cast<double*>(lds + 9 * kCells);
  const int tid = threadIdx.x;

  if (blockIdx.x == 0) {
    // fold block: the previous launch's per-tile sums, one vector per step, into sums[counter..]
    for (int v = 0; v < a.n_prev_vecs; ++v) {
      double s = 0.0;
      for (int i = tid; i < a.n_prev; i += kMLanes) s += a.prev_partials[static_cast<size_t>(v) * a.n_prev + i];
      s = wave_sum(s);
      __syncthreads();
      if ((tid & 63) == 0) red[tid >> 6] = s;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kWaves; ++w) t += red[w];
        a.sums[*a.counter + v] = t;
      }
    }
    __syncthreads();
    if (tid == 0 && a.n_prev_vecs > 0) *a.counter += a.n_prev_vecs;
    return;
  }

  int b = blockIdx.x - 1;
  if (a.xcd_remap) {
    // blocks b, b+8, ... share an XCD (round-robin dispatch): give each XCD one contiguous eighth of
    // the launch so that tiles which overlap (x and y neighbours) meet in the same L2
    const int nb = gridDim.x - 1, per = nb >> 3;
    b = (b & 7) * per + (b >> 3);
  }
  const int tile = b < a.tile_count ? a.tile_begin + b : a.tile_begin2 + (b - a.tile_count);
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int x0 = tx * kMTX;
  const int sy0 = a.ghost + ty * kMTY;              // storage row of the tile's first owned row
  const size_t ps = a.ps;
  const int nx = a.nx;
  const int rows_storage = a.rows_owned + 2 * a.ghost;
  const int ksteps = FULL ? K : a.ksteps;
  double acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) acc[i] = 0.0;

  // storage row -> does it hold the global accelerate row ny-2 ?
  auto on_accel_row = [&](int sr) {
    int g = a.y0_global + sr - a.ghost;
    if (g < 0) g += a.ny_global; else if (g >= a.ny_global) g -= a.ny_global;
    return g == a.accel_row;
  };

  // ---- sub-step 1: pull from the source grid; region = owned tile grown by (ksteps-1) rows / 2(ksteps-1) columns
  {
    const int ey = ksteps - 1, ex = 2 * ey;
    const int wp = (kMTX + 2 * ex) / 2;                                 // pairs per region row
    const int np = wp * (kMTY + 2 * ey);
#pragma unroll 1
    for (int i = tid; i < np; i += kMLanes) {
      const int ry = i / wp, rp = i - ry * wp;
      const int fx = EX - ex + 2 * rp, fy = EY - ey + ry;               // LDS frame coordinates (fx even)
      int gx = x0 + fx - EX; if (gx < 0) gx += nx; else if (gx >= nx) gx -= nx;            // periodic (:527-529)
      int sr = sy0 + fy - EY;
      int ys = sr - 1, yn = sr + 1;
      if (a.y_periodic) {                                                                 // periodic (:245-247)
        if (sr < 0) sr += rows_storage; else if (sr >= rows_storage) sr -= rows_storage;
        ys = (sr == 0) ? rows_storage - 1 : sr - 1;
        yn = (sr + 1 >= rows_storage) ? 0 : sr + 1;
      }
      const float* here = a.src + static_cast<size_t>(sr) * nx + gx;
      const float* south = a.src + static_cast<size_t>(ys) * nx + gx;
      const float* north = a.src + static_cast<size_t>(yn) * nx + gx;
      f2 p[9];
      p[0] = *reinterpret_cast<const f2*>(here);                                           // :530
      p[2] = *reinterpret_cast<const f2*>(south + 2 * ps);                                 // :532
      p[4] = *reinterpret_cast<const f2*>(north + 4 * ps);                                 // :534
      p[1] = *reinterpret_cast<const f2u*>(here + ps - 1);                                 // :531
      p[5] = *reinterpret_cast<const f2u*>(south + 5 * ps - 1);                            // :535
      p[8] = *reinterpret_cast<const f2u*>(north + 8 * ps - 1);                            // :538
      p[3] = *reinterpret_cast<const f2u*>(here + 3 * ps + 1);                             // :533
      p[6] = *reinterpret_cast<const f2u*>(south + 6 * ps + 1);                            // :536
      p[7] = *reinterpret_cast<const f2u*>(north + 7 * ps + 1);                            // :537
      if (gx == 0) {                          // x_w wraps to nx-1 (:529)
        p[1].x = here[ps + nx - 1]; p[5].x = south[5 * ps + nx - 1]; p[8].x = north[8 * ps + nx - 1];
      }
      if (gx == nx - 2) {                     // x_e wraps to 0 (:527-528)
        p[3].y = here[3 * ps + 2 - nx]; p[6].y = south[6 * ps + 2 - nx]; p[7].y = north[7 * ps + 2 - nx];
      }
      const int cell = sr * nx + gx;
      const uint32_t mbits = (a.mask[cell >> 5] >> (cell & 31)) & 3u;
      f2 out[9];
      const bool owned = fx >= EX && fx < EX + kMTX && fy >= EY && fy < EY + kMTY;
      acc[0] += finish_pair(p, mbits, a.omega, (ksteps > 1 || a.accel_last) && on_accel_row(sr), a.accel_w1, a.accel_w2, owned, out);
      if (ksteps > 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(lds + k * kCells + fy * W + fx) = out[k];
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(out[k], reinterpret_cast<f2*>(a.dst + k * ps + cell));
      }
    }
  }
  if constexpr (K >= 2) {
    __syncthreads();
    // ---- sub-steps 2..ksteps: in place in the LDS frame, each on a region one row / two columns smaller
#pragma unroll 1
    for (int j = 2; j <= ksteps; ++j) {
      const int ey = ksteps - j, ex = 2 * ey;
      const int wp = (kMTX + 2 * ex) / 2;
      const int np = wp * (kMTY + 2 * ey);
      const bool last = j == ksteps;
      f2 outs[G::passes][9];
      int slot[G::passes];
#pragma unroll
      for (int q = 0; q < G::passes; ++q) {
        const int i = tid + q * kMLanes;
        slot[q] = -1;
        if (i < np) {
          const int ry = i / wp, rp = i - ry * wp;
          const int fx = EX - ex + 2 * rp, fy = EY - ey + ry;
          const int c = fy * W + fx;
          f2 p[9];
          p[0] = *reinterpret_cast<const f2*>(lds + 0 * kCells + c);
          p[2] = *reinterpret_cast<const f2*>(lds + 2 * kCells + c - W);
          p[4] = *reinterpret_cast<const f2*>(lds + 4 * kCells + c + W);
          p[1] = f2{lds[1 * kCells + c - 1], lds[1 * kCells + c]};
          p[5] = f2{lds[5 * kCells + c - W - 1], lds[5 * kCells + c - W]};
          p[8] = f2{lds[8 * kCells + c + W - 1], lds[8 * kCells + c + W]};
          p[3] = f2{lds[3 * kCells + c + 1], lds[3 * kCells + c + 2]};
          p[6] = f2{lds[6 * kCells + c - W + 1], lds[6 * kCells + c - W + 2]};
          p[7] = f2{lds[7 * kCells + c + W + 1], lds[7 * kCells + c + W + 2]};
          int gx = x0 + fx - EX; if (gx < 0) gx += nx; else if (gx >= nx) gx -= nx;
          int sr = sy0 + fy - EY;
          if (a.y_periodic) { if (sr < 0) sr += rows_storage; else if (sr >= rows_storage) sr -= rows_storage; }
          const int cell = sr * nx + gx;
          const uint32_t mbits = (a.mask[cell >> 5] >> (cell & 31)) & 3u;
          const bool owned = fx >= EX && fx < EX + kMTX && fy >= EY && fy < EY + kMTY;
          const double term = finish_pair(p, mbits, a.omega, (!last || a.accel_last) && on_accel_row(sr), a.accel_w1, a.accel_w2,
                                          owned, outs[q]);
#pragma unroll
          for (int m = 1; m < K; ++m)
            if (m == j - 1) acc[m] += term;
          slot[q] = last ? cell : c;
        }
      }
      if (!last) {
        __syncthreads();                       // every lane has read its neighbours
#pragma unroll
        for (int q = 0; q < G::passes; ++q)
          if (slot[q] >= 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(lds + k * kCells + slot[q]) = outs[q][k];
          }
        __syncthreads();
      } else {
#pragma unroll
        for (int q = 0; q < G::passes; ++q)
          if (slot[q] >= 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(outs[q][k], reinterpret_cast<f2*>(a.dst + k * ps + slot[q]));
          }
      }
    }
  }

  // per-step sums over the owned cells of this tile
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const double w = wave_sum(acc[q]);
    if ((tid & 63) == 0) red[q * kWaves + (tid >> 6)] = w;
  }
  __syncthreads();
  if (tid < ksteps) {
    double t = 0.0;
    for (int w = 0; w < kWaves; ++w) t += red[tid * kWaves + w];
    a.partials_out[static_cast<size_t>(tid) * a.ntiles_total + tile] = t;
  }
}

// Folds the last step's partials after the loop.
__global__ void __launch_bounds__(kBlock) lbm_fold_kernel(const double* partials, int n, int nvecs, double* sums, int* counter)
{
  __shared__ double red[kBlock / 64];
  for (int v = 0; v < nvecs; ++v) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) s += partials[static_cast<size_t>(v) * n + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sums[*counter + v] = s;
    __syncthreads();
  }
  if (threadIdx.x == 0) *counter += nvecs;
}

// accelerate_flow (d2q9-bgk.c:442-478) in place on one row: only needed before the first step of a run.
__global__ void lbm_accelerate_kernel(float* grid, size_t ps, const uint32_t* mask, int nx, int row, float w1, float w2)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  const size_t c = static_cast<size_t>(row) * nx + x;
  if ((mask[c >> 5] >> (c & 31)) & 1u) return;
  float* f = grid + c;
  const float f3 = f[3 * ps], f6 = f[6 * ps], f7 = f[7 * ps];
  if (f3 - w1 > 0.0f && f6 - w2 > 0.0f && f7 - w2 > 0.0f) {
    f[1 * ps] += w1; f[5 * ps] += w2; f[8 * ps] += w2;
    f[3 * ps] = f3 - w1; f[6 * ps] = f6 - w2; f[7 * ps] = f7 - w2;
  }
}

// Initial state (d2q9-bgk.c:880-902).
__global__ void lbm_init_kernel(float* grid, size_t ps, size_t ncells, float w0, float w1, float w2)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells) return;
  grid[i] = w0;
#pragma unroll
  for (int k = 1; k < 5; ++k) grid[k * ps + i] = w1;
#pragma unroll
  for (int k = 5; k < 9; ++k) grid[k * ps + i] = w2;
}

// AoS (reference t_speed) <-> SoA planes.
__global__ void lbm_aos_to_soa_kernel(const float* aos, float* grid, size_t ps, size_t ncells)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells * 9) return;
  const size_t c = i / 9;
  const int k = static_cast<int>(i - c * 9);
  grid[k * ps + c] = aos[i];
}

__global__ void lbm_soa_to_aos_kernel(const float* grid, float* aos, size_t ps, size_t ncells)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells * 9) return;
  const size_t c = i / 9;
  const int k = static_cast<int>(i - c * 9);
  aos[i] = grid[k * ps + c];
}

// Outgoing halo rows of the CURRENT grid (before the first step of a row-partitioned run).
__global__ void lbm_pack_halo_kernel(const float* grid, size_t ps, int nx, int nyl, int nxp, float* send_south, float* send_north)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  const float* first = grid + x;
  const float* last = grid + static_cast<size_t>(nyl - 1) * nx + x;
  float* ss = send_south + kHaloGuard + x;
  float* sn = send_north + kHaloGuard + x;
  ss[0] = first[4 * ps]; ss[nxp] = first[7 * ps]; ss[2 * nxp] = first[8 * ps];
  sn[0] = last[2 * ps];  sn[nxp] = last[5 * ps];  sn[2 * nxp] = last[6 * ps];
}

// av_velocity (d2q9-bgk.c:716-751): per-cell float arithmetic as the reference, double accumulation.
__global__ void __launch_bounds__(kBlock) lbm_av_velocity_kernel(const float* grid, size_t ps, const uint32_t* mask, size_t ncells, double* partials)
{
  __shared__ double red[kBlock / 64];
  double acc = 0.0;
  for (size_t c = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; c < ncells; c += static_cast<size_t>(gridDim.x) * kBlock) {
    if ((mask[c >> 5] >> (c & 31)) & 1u) continue;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = grid[k * ps + c];
    float rho = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) rho += f[k];                                   // :724-729
    const float ux = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;        // :732-738
    const float uy = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;        // :740-746
    acc += sqrt(static_cast<double>((ux * ux) + (uy * uy)));                   // :748
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------------
// host side of the device ABI
// ------------------------------------------------------------------------------------------------

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      lbm_internal::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));            \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Plane stride: rows*nx floats + guard for the dword-shifted loads at both ends, rounded to 256 B,
// then skewed by an odd number of 256-B units so the 9 planes (and the two grids) do not all start
// on the same HBM channel when rows*nx is a large power of two.
int tune_env(const char* name, int dflt)
{
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

size_t plane_stride_floats(size_t ncells)
{
  size_t s = round_up(ncells + 64, 64);
  if ((s / 64) % 2 == 0) s += 64;
  if (ncells >= (1u << 20)) s += 64 * static_cast<size_t>(tune_env("LBM_TUNE_SKEW", 34));   // default ~8.5 KiB skew between planes for large grids
  return s;
}

}  // namespace

struct lbm_ctx {
  lbm_params p{};
  int free_cells = 0;
  float free_cells_inv = 0.f;
  int y0 = 0, nyl = 0, device = 0;
  unsigned flags = 0;
  bool self_periodic = true;
  int accel_row = -1;
  int ghost = 0;             // storage rows below / above the owned rows (K-step kernels of a row-partitioned run)
  bool nt_stores = false;
  size_t ncells = 0, ncells_storage = 0, ps = 0, grid_floats = 0;   // owned cells; cells incl. ghost rows; plane stride
  float* grid_alloc[2] = {nullptr, nullptr};
  float* grid[2] = {nullptr, nullptr};       // plane 0 row 0 (after the front guard)
  int cur = 0;
  uint32_t* mask = nullptr;
  int mask_words = 0;
  bool lds_kernel = false;   // LBM_FLAG_KERNEL_LDS
  int lane_cells = kCellsPerLane;   // cells per lane: 4 (vector form) or 1 (narrow form: tiny grids, nx % 4 != 0)
  int nxp = 0;
  float* halo_alloc = nullptr;
  float* send[2] = {nullptr, nullptr};
  float* recv[2] = {nullptr, nullptr};
  double* partials[2] = {nullptr, nullptr};
  int partials_cap = 0;
  int n_part_interior = 0, n_part_boundary = 0, n_part_full = 0;
  int iters_full = 1, iters_interior = 1;
  double* sums = nullptr;
  int sums_cap = 0;
  int* counter = nullptr;
  hipStream_t stream = nullptr;
  // launch-bound grids: kGraphSteps steps captured once into a hipGraph and replayed (one per
  // starting source grid); see lbm_run
  bool use_graph = false;
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // around the step kernels of the last run
  int ev_launches = 0;
  int ev_tile_launches = 0;
  bool ev_valid = false;
  // run state (split-phase and lbm_run)
  int run_steps = 0, run_done = 0;
  int parity = 0;            // partials buffer written by the current step
  int n_prev = 0;            // partial count of the previous step (0 = nothing to fold)
  int n_prev_vecs = 1;       // ... and how many step vectors of that length the previous launch left (tile kernel: up to 8)
  int multi_K = 0;           // > 0: bandwidth-bound grid advanced K steps per launch by lbm_multi_kernel<K>
  int multi_tiles_x = 0, multi_tiles = 0;
  bool tile_kernel = false;  // lbm_run advances several steps per launch with lbm_tile_kernel (small grids)
  int tile_T = 16, tile_H = 8;   // its geometry: owned tile edge, ghost ring = max steps per launch
  int n_tiles = 0;
  float accel_w1 = 0.f, accel_w2 = 0.f;
};

namespace {

int pick_iters(long long quads)
{
  // keep the grid at <= ~4096 blocks (16 per CU): fewer, longer blocks and a short partial vector
  long long chunks = (quads + kBlock - 1) / kBlock;
  const int max_blocks = tune_env("LBM_TUNE_MAXBLOCKS", 16384);   // measured on 8192x8192: 16384 blocks x 4 chunks ~7 % faster than 4096 x 16
  int iters = 1;
  while (chunks / iters > max_blocks && iters < 1024) iters *= 2;
  return iters;
}

int blocks_for(long long quads, int iters)
{
  const long long per_block = static_cast<long long>(kBlock) * iters;
  return static_cast<int>((quads + per_block - 1) / per_block);
}

hipStream_t pick_stream(lbm_ctx* c, void* stream) { return stream ? static_cast<hipStream_t>(stream) : c->stream; }

void drop_graphs(lbm_ctx* c)
{
  for (hipGraphExec_t& g : c->graph_exec) {
    if (g) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
}

int ensure_sums(lbm_ctx* c, int n)
{
  if (n <= c->sums_cap) return 0;
  drop_graphs(c);   // captured kernel arguments hold the old pointer
  if (c->sums) HIP_TRY(hipFree(c->sums));
  c->sums = nullptr;
  c->sums_cap = 0;
  HIP_TRY(hipMalloc(&c->sums, sizeof(double) * static_cast<size_t>(n)));
  c->sums_cap = n;
  return 0;
}

StepArgs base_args(lbm_ctx* c, bool accel_next)
{
  StepArgs a{};
  a.src = c->grid[c->cur];
  a.dst = c->grid[c->cur ^ 1];
  a.mask = c->mask;
  a.mask_words = c->mask_words;
  a.ps = c->ps;
  a.nx = c->p.nx;
  a.nyl = c->nyl;
  a.nxp = c->nxp;
  a.omega = c->p.omega;
  a.accel_w1 = c->accel_w1;
  a.accel_w2 = c->accel_w2;
  a.accel_row = accel_next ? c->accel_row : -1;
  a.sums = c->sums;
  a.counter = c->counter;
  return a;
}

void launch_step(lbm_ctx* c, const StepArgs& a, int blocks, hipStream_t s)
{
  const dim3 grid(blocks + 1), block(kBlock);   // + the fold block
  if (c->lane_cells == 1) {
    if (c->nt_stores) hipLaunchKernelGGL(lbm_step_kernel_narrow<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(lbm_step_kernel_narrow<false>, grid, block, 0, s, a);
  } else if (c->lds_kernel && a.quad_begin2 >= a.quad_end2) {
    if (c->nt_stores) hipLaunchKernelGGL(lbm_step_kernel_lds<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(lbm_step_kernel_lds<false>, grid, block, 0, s, a);
  } else {
    if (c->nt_stores) hipLaunchKernelGGL(lbm_step_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(lbm_step_kernel<false>, grid, block, 0, s, a);
  }
}

template <int K>
void launch_multi_k(int blocks, hipStream_t s, const MultiArgs& a)
{
  using G = MultiGeom<K>;
  if (a.ksteps == K) lbm_multi_kernel<K, true><<<dim3(blocks + 1), dim3(kMLanes), G::lds_bytes, s>>>(a);
  else lbm_multi_kernel<K, false><<<dim3(blocks + 1), dim3(kMLanes), G::lds_bytes, s>>>(a);
}

// One launch of lbm_multi_kernel over the tile ranges [t0, t0+n0) and [t1, t1+n1): `ksteps` steps.
void launch_multi(lbm_ctx* c, int ksteps, bool accel_last, int t0, int n0, int t1, int n1, bool fold, hipStream_t s)
{
  MultiArgs a{};
  a.src = c->grid[c->cur]; a.dst = c->grid[c->cur ^ 1];
  a.mask = c->mask; a.ps = c->ps; a.nx = c->p.nx;
  a.rows_owned = c->nyl; a.ghost = c->ghost; a.y_periodic = c->self_periodic ? 1 : 0;
  a.y0_global = c->y0; a.ny_global = c->p.ny;
  a.tiles_x = c->multi_tiles_x;
  a.tile_begin = t0; a.tile_count = n0; a.tile_begin2 = t1; a.tile_count2 = n1;
  a.ntiles_total = c->multi_tiles;
  a.ksteps = ksteps;
  a.omega = c->p.omega; a.accel_w1 = c->accel_w1; a.accel_w2 = c->accel_w2;
  a.accel_row = c->p.ny - 2; a.accel_last = accel_last ? 1 : 0;
  a.partials_out = c->partials[c->parity];
  a.prev_partials = c->partials[c->parity ^ 1];
  a.n_prev = fold ? c->n_prev : 0; a.n_prev_vecs = (fold && c->n_prev > 0) ? c->n_prev_vecs : 0;
  a.sums = c->sums; a.counter = c->counter;
  const int blocks = n0 + n1;
  // measured on 8192x8192, K=2: 515 us/step with the XCD-contiguous tile order, 549 without
  a.xcd_remap = (tune_env("LBM_TUNE_MULTI_REMAP", 1) && blocks % 8 == 0 && blocks >= 64) ? 1 : 0;
  switch (c->multi_K) {
    case 1: launch_multi_k<1>(blocks, s, a); break;
    case 2: launch_multi_k<2>(blocks, s, a); break;
    case 3: launch_multi_k<3>(blocks, s, a); break;
    default: launch_multi_k<4>(blocks, s, a); break;
  }
}

template <int T, int H>
void launch_tile(dim3 grid, hipStream_t s, const TileArgs& a)
{
  using G = TileGeom<T, H>;
  lbm_tile_kernel<T, H><<<grid, dim3(G::lanes), G::lds_bytes, s>>>(a);
}

int begin_run(lbm_ctx* c, int n_steps, hipStream_t s)
{
  if (ensure_sums(c, n_steps)) return 1;
  HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(int), s));
  c->run_steps = n_steps;
  c->run_done = 0;
  c->n_prev = 0;
  c->n_prev_vecs = 1;
  c->parity = 0;
  if (c->accel_row >= 0 && n_steps > 0) {
    // accelerate_flow of step 0 (d2q9-bgk.c:345-348); later steps get it from the kernel epilogue
    const int nx = c->p.nx;
    hipLaunchKernelGGL(lbm_accelerate_kernel, dim3((nx + 255) / 256), dim3(256), 0, s, c->grid[c->cur], c->ps,
                       c->mask, nx, c->ghost + c->accel_row, c->accel_w1, c->accel_w2);
    HIP_TRY(hipGetLastError());
  }
  c->ev_valid = false;
  c->ev_launches = 0;
  c->ev_tile_launches = 0;
  HIP_TRY(hipEventRecord(c->ev_begin, s));
  return 0;
}

constexpr int kGraphSteps = 64;   // even: the source/destination roles and the partial-sum parity return to their start

// One whole-grid step of a self-contained domain: launch + state flip (d2q9-bgk.c:345-378).
void full_step(lbm_ctx* c, bool accel_next, hipStream_t s)
{
  const long long quads = static_cast<long long>(c->p.nx / c->lane_cells) * c->nyl;
  StepArgs a = base_args(c, accel_next);
  a.quad_begin = 0; a.quad_end = static_cast<int>(quads);
  a.quad_begin2 = a.quad_end2 = 0;
  a.iters = c->iters_full;
  a.partials_out = c->partials[c->parity];
  a.prev_partials = c->partials[c->parity ^ 1];
  a.n_prev = c->n_prev;
  launch_step(c, a, c->n_part_full, s);
  c->n_prev = c->n_part_full;
  c->n_prev_vecs = 1;
  c->parity ^= 1;
  c->cur ^= 1;
}

// Captures kGraphSteps mid-run steps (previous partials to fold, accelerate epilogue on) starting
// from the current source grid.  Every per-step quantity that changes from step to step lives on
// the device (sums[counter++]), so the same graph replays anywhere inside a run.
int ensure_graph(lbm_ctx* c, hipStream_t s)
{
  if (c->graph_exec[c->cur]) return 0;
  const int cur = c->cur, parity = c->parity, n_prev = c->n_prev;
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < kGraphSteps; ++i) full_step(c, true, s);
  hipError_t e = hipStreamEndCapture(s, &graph);
  c->cur = cur; c->parity = parity; c->n_prev = n_prev;   // nothing ran
  HIP_TRY(e);
  e = hipGraphInstantiate(&c->graph_exec[cur], graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  HIP_TRY(e);
  return 0;
}

}  // namespace

// lbm_create / lbm_create_global.  obstacles_global (ny*nx, may be null) additionally gives the
// obstacle flags of the rows around the partition, which the K-step kernels need for their ghost rows.
static int create_impl(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacles_rows,
                       const int* obstacles_global, int y0, int ny_local, int device, unsigned flags)
{
  if (!out || !p || !obstacles_rows) { lbm_internal::set_error("lbm_create: null argument"); return 1; }
  *out = nullptr;
  if (p->nx < 1) { lbm_internal::set_error("lbm_create: nx must be positive"); return 1; }
  if (p->ny < 3) { lbm_internal::set_error("lbm_create: ny must be >= 3 (accelerate_flow works on row ny-2, d2q9-bgk.c:449)"); return 1; }
  if (ny_local < 1 || y0 < 0 || y0 + ny_local > p->ny) { lbm_internal::set_error("lbm_create: partition rows out of range"); return 1; }
  if (free_cells <= 0) { lbm_internal::set_error("lbm_create: free_cells must be positive"); return 1; }
  const bool self_periodic = (ny_local == p->ny) && !(flags & LBM_FLAG_FORCE_HALO);
  const int accel_global = p->ny - 2;
  int accel_row = -1;
  if (accel_global >= y0 && accel_global < y0 + ny_local) accel_row = accel_global - y0;
  if (!self_periodic && accel_row >= 0 && (accel_row == 0 || accel_row == ny_local - 1)) {
    lbm_internal::set_error("lbm_create: the partition holding row ny-2 needs >= 3 rows (d2q9-bgk.c:848-849)");
    return 1;
  }
  if (static_cast<long long>(p->nx) * ny_local > (1LL << 31) - 4096) { lbm_internal::set_error("lbm_create: partition too large for 32-bit cell indices"); return 1; }

  HIP_TRY(hipSetDevice(device));
  lbm_ctx* c = new lbm_ctx();
  c->p = *p;
  c->free_cells = free_cells;
  c->free_cells_inv = 1.0f / free_cells;                                    // d2q9-bgk.c:950
  c->y0 = y0; c->nyl = ny_local; c->device = device; c->flags = flags;
  c->self_periodic = self_periodic;
  c->accel_row = accel_row;
  c->accel_w1 = p->density * p->accel * 0.111111111111111111111111f;        // d2q9-bgk.c:445
  c->accel_w2 = p->density * p->accel * 0.0277777777777777777777778f;       // d2q9-bgk.c:446
  c->ncells = static_cast<size_t>(p->nx) * ny_local;
  // K-step mode of a row-partitioned run: K ghost rows on each side of the owned rows, refreshed by the
  // neighbours every K steps, all steps done by lbm_multi_kernel (lbm_macro_* calls)
  if (!self_periodic && obstacles_global && !(flags & LBM_FLAG_ONE_STEP) && p->nx % kMTX == 0 && ny_local % kMTY == 0 &&
      ny_local >= 2 * kMTY) {
    // measured on a 1-rank ring (us/step; one-step loop 116 / 37): 8192x1024 rows K=2 108, K=3 71, K=4 74;
    // 1024x128 rows K=2 44, K=3 30, K=4 25 -- the exchange (36 messages) costs ~50 us per macro-step
    const int k = tune_env("LBM_TUNE_MACRO_K", c->ncells < (1u << 21) ? 4 : 3);
    if (k > 0) { c->multi_K = std::min(k, kMaxMultiSteps); c->ghost = c->multi_K; }
  }
  c->ncells_storage = static_cast<size_t>(p->nx) * (ny_local + 2 * c->ghost);
  c->ps = plane_stride_floats(c->ncells_storage);
  c->grid_floats = 9 * c->ps + 128;
  // non-temporal output stores once the two grids no longer fit the 256 MiB Infinity Cache
  const size_t state_bytes = 2 * 9 * c->ncells * sizeof(float);
  c->nt_stores = state_bytes > (192u << 20);
  if (flags & LBM_FLAG_NT_STORES) c->nt_stores = true;
  if (flags & LBM_FLAG_NO_NT_STORES) c->nt_stores = false;
  // narrow form for latency-bound grids (measured cross-over) and for nx % 4 != 0
  // (128x128: 3.5 vs 4.4 us/step, 256x256: 4.1 vs 4.6, 512x512: 7.3 vs 6.2 -> cross-over at 64 K cells)
  const size_t narrow_max = static_cast<size_t>(tune_env("LBM_TUNE_NARROW_MAX", 65536));
  c->lane_cells = (p->nx % kCellsPerLane != 0 || c->ncells <= narrow_max) ? 1 : kCellsPerLane;
  c->lds_kernel = (flags & LBM_FLAG_KERNEL_LDS) != 0 && c->lane_cells == kCellsPerLane;
  // hipGraph replay of 64-step blocks is opt-in: measured on MI355X it changes nothing (128x128:
  // 4.47 vs 4.33 us/step) because even the smallest grids are bound by the device-side kernel
  // boundary + kernel latency, not by the host's launch rate
  c->use_graph = self_periodic && (flags & LBM_FLAG_GRAPH);

  auto fail = [&](void) { lbm_destroy(c); return 1; };
#define HIP_TRY_C(expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      lbm_internal::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));            \
      return fail();                                                                         \
    }                                                                                        \
  } while (0)

  HIP_TRY_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY_C(hipEventCreate(&c->ev_begin));
  HIP_TRY_C(hipEventCreate(&c->ev_end));
  if (tune_env("LBM_TUNE_ONEALLOC", 0)) {
    // both grids in one allocation: fixed relative placement (tuning experiment)
    const size_t gap = 64 * static_cast<size_t>(tune_env("LBM_TUNE_GRIDGAP", 0));
    HIP_TRY_C(hipMalloc(&c->grid_alloc[0], sizeof(float) * (2 * c->grid_floats + gap)));
    HIP_TRY_C(hipMemsetAsync(c->grid_alloc[0], 0, sizeof(float) * (2 * c->grid_floats + gap), c->stream));
    c->grid[0] = c->grid_alloc[0] + 64;
    c->grid[1] = c->grid_alloc[0] + c->grid_floats + gap + 64;
  } else {
    for (int g = 0; g < 2; ++g) {
      HIP_TRY_C(hipMalloc(&c->grid_alloc[g], sizeof(float) * c->grid_floats));
      HIP_TRY_C(hipMemsetAsync(c->grid_alloc[g], 0, sizeof(float) * c->grid_floats, c->stream));
      c->grid[g] = c->grid_alloc[g] + 64;
    }
  }
  // obstacle bitfield
  const size_t mwords = (c->ncells_storage + 31) / 32 + 4;
  std::vector<uint32_t> bits(mwords, 0u);
  if (c->ghost == 0) {
    for (size_t i = 0; i < c->ncells; ++i)
      if (obstacles_rows[i]) bits[i >> 5] |= 1u << (i & 31);
  } else {
    for (int r = 0; r < ny_local + 2 * c->ghost; ++r) {          // storage row -> global row, periodic
      int g = (y0 + r - c->ghost) % p->ny;
      if (g < 0) g += p->ny;
      const int* row = obstacles_global + static_cast<size_t>(g) * p->nx;
      for (int x = 0; x < p->nx; ++x)
        if (row[x]) { const size_t i = static_cast<size_t>(r) * p->nx + x; bits[i >> 5] |= 1u << (i & 31); }
    }
  }
  c->mask_words = static_cast<int>(mwords);
  HIP_TRY_C(hipMalloc(&c->mask, sizeof(uint32_t) * mwords));
  HIP_TRY_C(hipMemcpy(c->mask, bits.data(), sizeof(uint32_t) * mwords, hipMemcpyHostToDevice));
  // halo buffers: 2 send + 2 recv, each [3][nxp]
  c->nxp = p->nx + 2 * kHaloGuard;
  const size_t hb = static_cast<size_t>(3) * c->nxp;
  HIP_TRY_C(hipMalloc(&c->halo_alloc, sizeof(float) * hb * 4));
  HIP_TRY_C(hipMemsetAsync(c->halo_alloc, 0, sizeof(float) * hb * 4, c->stream));
  c->send[0] = c->halo_alloc; c->send[1] = c->halo_alloc + hb;
  c->recv[0] = c->halo_alloc + 2 * hb; c->recv[1] = c->halo_alloc + 3 * hb;
  // launch geometry + partial buffers
  const long long qrow = p->nx / c->lane_cells;
  const long long qfull = qrow * ny_local;
  c->iters_full = pick_iters(qfull);
  c->n_part_full = blocks_for(qfull, c->iters_full);
  const long long qint = ny_local > 2 ? qrow * (ny_local - 2) : 0;
  c->iters_interior = pick_iters(qint > 0 ? qint : 1);
  c->n_part_interior = qint > 0 ? blocks_for(qint, c->iters_interior) : 0;
  c->n_part_boundary = blocks_for(ny_local > 1 ? 2 * qrow : qrow, 1);
  c->partials_cap = std::max(c->n_part_full, c->n_part_interior + c->n_part_boundary) + 1;
  // temporally blocked form: whole periodic grids whose edges are multiples of the tile edge and
  // that are small enough to be launch-latency-bound (measured cross-over)
  {
    // geometry by size, measured on MI355X (us/step; one-step kernels 3.4 / 3.5 / 4.0 / 6.3):
    //   128x128: <8,4> 1.6, <16,8> 1.9 | 128x256: <16,8> 2.0, <8,4> 2.1 | 256x256: <16,4> 2.5, <8,4> 2.9
    //   512x512: <16,4> 5.2 | 1024x1024: slower than the one-step kernel (13.6)
    const int by_size = c->ncells <= 16384 ? 84 : (c->ncells <= 32768 ? 168 : 164);
    const int geom = tune_env("LBM_TUNE_TILE_GEOM", by_size);   // T*10 + H
    c->tile_T = geom / 10; c->tile_H = geom % 10;
    if (!((c->tile_T == 16 || c->tile_T == 8) && (c->tile_H == 8 || c->tile_H == 4))) { c->tile_T = 16; c->tile_H = 4; }
  }
  c->n_tiles = (p->nx % c->tile_T == 0 && ny_local % c->tile_T == 0) ? (p->nx / c->tile_T) * (ny_local / c->tile_T) : 0;
  c->tile_kernel = self_periodic && c->n_tiles > 0 &&
                   c->ncells <= static_cast<size_t>(tune_env("LBM_TUNE_TILE_MAX", 65536));   // 512x512: lbm_multi_kernel<3> 3.5 us/step vs 5.2 here
  if (c->ghost > 0) {
    c->tile_kernel = false;
    c->multi_tiles_x = p->nx / kMTX;
    c->multi_tiles = c->multi_tiles_x * (ny_local / kMTY);
    c->partials_cap = std::max(c->partials_cap, kMaxMultiSteps * c->multi_tiles + 1);
  } else if (!c->tile_kernel && self_periodic && p->nx % kMTX == 0 && ny_local % kMTY == 0) {
    // K steps per pass over HBM (lbm_multi_kernel), measured us/step for K = 2 / 3 / 4 (one-step kernel):
    //   8192x8192 515 / 532 / 556 (853-917)   2048x2048 34.0 / 35.0 / 35.7 (59)
    //   1024x1024 11.4 / 10.0 / 10.2 (13.5)   512x512 3.9 / 3.5 / 3.6 (6.3; lbm_tile_kernel 5.2)
    c->multi_K = std::min(std::max(tune_env("LBM_TUNE_MULTI_K", c->ncells <= (1u << 21) ? 3 : 2), 0), kMaxMultiSteps);
    c->multi_tiles_x = p->nx / kMTX;
    c->multi_tiles = c->multi_tiles_x * (ny_local / kMTY);
    if (c->multi_K > 0) c->partials_cap = std::max(c->partials_cap, kMaxMultiSteps * c->multi_tiles + 1);
  } else if (c->tile_kernel) {
    c->partials_cap = std::max(c->partials_cap, kMaxTileSteps * c->n_tiles + 1);
    // up to 74 KB of dynamic LDS per block (two 9 x R x R float buffers): above the 64 KB default limit
    {
      using G168 = TileGeom<16, 8>;
      auto* k168 = &lbm_tile_kernel<16, 8>;
      HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k168), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(G168::lds_bytes)));
    }
  }
  for (int i = 0; i < 2; ++i) HIP_TRY_C(hipMalloc(&c->partials[i], sizeof(double) * c->partials_cap));
  HIP_TRY_C(hipMalloc(&c->counter, sizeof(int)));
  HIP_TRY_C(hipMemsetAsync(c->counter, 0, sizeof(int), c->stream));
  // initial state (d2q9-bgk.c:880-902)
  {
    const float w0 = p->density * 4.0f / 9.0f, w1 = p->density / 9.0f, w2 = p->density / 36.0f;
    const int blocks = static_cast<int>((c->ncells_storage + 255) / 256);
    hipLaunchKernelGGL(lbm_init_kernel, dim3(blocks), dim3(256), 0, c->stream, c->grid[0], c->ps, c->ncells_storage, w0, w1, w2);
    HIP_TRY_C(hipGetLastError());
  }
  HIP_TRY_C(hipStreamSynchronize(c->stream));
#undef HIP_TRY_C
  *out = c;
  return 0;
}

extern "C" {

int lbm_create(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacles_rows, int y0,
               int ny_local, int device, unsigned flags)
{
  return create_impl(out, p, free_cells, obstacles_rows, nullptr, y0, ny_local, device, flags);
}

int lbm_create_global(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacles_all, int y0,
                      int ny_local, int device, unsigned flags)
{
  if (!obstacles_all || !p || y0 < 0) { lbm_internal::set_error("lbm_create_global: bad argument"); return 1; }
  return create_impl(out, p, free_cells, obstacles_all + static_cast<size_t>(y0) * p->nx, obstacles_all, y0, ny_local, device, flags);
}

int lbm_destroy(lbm_ctx* c)
{
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drop_graphs(c);
  for (int g = 0; g < 2; ++g) if (c->grid_alloc[g]) (void)hipFree(c->grid_alloc[g]);
  if (c->mask) (void)hipFree(c->mask);
  if (c->halo_alloc) (void)hipFree(c->halo_alloc);
  for (int i = 0; i < 2; ++i) if (c->partials[i]) (void)hipFree(c->partials[i]);
  if (c->sums) (void)hipFree(c->sums);
  if (c->counter) (void)hipFree(c->counter);
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  if (c->ev_end) (void)hipEventDestroy(c->ev_end);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int lbm_run(lbm_ctx* c, int n_steps, float* av_vels)
{
  if (!c) { lbm_internal::set_error("lbm_run: null context"); return 1; }
  if (!c->self_periodic) { lbm_internal::set_error("lbm_run: partition is not a self-contained domain; use the lbm_step_* calls"); return 1; }
  if (n_steps < 0) { lbm_internal::set_error("lbm_run: negative step count"); return 1; }
  if (n_steps == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  if (begin_run(c, n_steps, s)) return 1;
  int tile_launches = 0;
  const bool multi = c->multi_K > 0 && c->self_periodic;
  for (int t = 0; multi && t < n_steps;) {
    // up to multi_K steps per pass over HBM (lbm_multi_kernel)
    const int k = std::min(c->multi_K, n_steps - t);
    launch_multi(c, k, /*accel_last=*/t + k < n_steps, 0, c->multi_tiles, 0, 0, /*fold=*/true, s);
    c->n_prev = c->multi_tiles; c->n_prev_vecs = k;
    c->parity ^= 1;
    c->cur ^= 1;
    t += k;
    ++tile_launches;
    if (t >= n_steps) c->ev_tile_launches = tile_launches;
  }
  for (int t = 0; !multi && c->tile_kernel && t < n_steps;) {
    // up to tile_H steps per launch (lbm_tile_kernel); every launch of such a run has this form
    const int k = std::min(c->tile_H, n_steps - t);
    TileArgs a{};
    a.src = c->grid[c->cur]; a.dst = c->grid[c->cur ^ 1];
    a.mask = c->mask; a.ps = c->ps; a.nx = c->p.nx; a.ny = c->nyl; a.tiles_x = c->p.nx / c->tile_T;
    a.ksteps = k;
    a.omega = c->p.omega; a.accel_w1 = c->accel_w1; a.accel_w2 = c->accel_w2;
    a.accel_row = c->accel_row; a.accel_last = (t + k < n_steps) ? 1 : 0;
    a.partials_out = c->partials[c->parity];
    a.prev_partials = c->partials[c->parity ^ 1];
    a.n_prev = c->n_prev; a.n_prev_vecs = c->n_prev > 0 ? c->n_prev_vecs : 0;
    a.sums = c->sums; a.counter = c->counter;
    const dim3 grid(c->n_tiles + 1);
    if (c->tile_T == 16 && c->tile_H == 8) launch_tile<16, 8>(grid, s, a);
    else if (c->tile_T == 16) launch_tile<16, 4>(grid, s, a);
    else if (c->tile_H == 8) launch_tile<8, 8>(grid, s, a);
    else launch_tile<8, 4>(grid, s, a);
    ++tile_launches;
    c->n_prev = c->n_tiles; c->n_prev_vecs = k;
    c->parity ^= 1;
    c->cur ^= 1;
    t += k;
    if (t >= n_steps) c->ev_tile_launches = tile_launches;
  }
  for (int t = 0; !multi && !c->tile_kernel && t < n_steps;) {
    // launch-bound grids: replay a captured block of kGraphSteps steps while at least one more
    // step follows it (the last step of a run is launched directly: it must not accelerate)
    if (c->use_graph && c->n_prev > 0 && n_steps - t > kGraphSteps) {
      if (ensure_graph(c, s)) return 1;
      HIP_TRY(hipGraphLaunch(c->graph_exec[c->cur], s));
      t += kGraphSteps;
    } else {
      full_step(c, /*accel_next=*/t + 1 < n_steps, s);
      t += 1;
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_end, s));
  c->ev_launches = (multi || c->tile_kernel) ? c->ev_tile_launches : n_steps;
  c->ev_valid = true;
  hipLaunchKernelGGL(lbm_fold_kernel, dim3(1), dim3(kBlock), 0, s, c->partials[c->parity ^ 1], c->n_prev, c->n_prev_vecs, c->sums, c->counter);
  HIP_TRY(hipGetLastError());
  c->n_prev = 0;
  c->n_prev_vecs = 1;
  c->run_done = n_steps;
  if (av_vels) {
    std::vector<double> host(static_cast<size_t>(n_steps));
    HIP_TRY(hipMemcpyAsync(host.data(), c->sums, sizeof(double) * n_steps, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const double inv = static_cast<double>(c->free_cells_inv);
    for (int t = 0; t < n_steps; ++t) av_vels[t] = static_cast<float>(host[t] * inv);   // d2q9-bgk.c:367
  } else {
    HIP_TRY(hipStreamSynchronize(s));
  }
  return 0;
}

int lbm_get_cells(lbm_ctx* c, float* cells_aos)
{
  if (!c || !cells_aos) { lbm_internal::set_error("lbm_get_cells: null argument"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  float* tmp = nullptr;
  const size_t n = c->ncells * 9;
  HIP_TRY(hipMalloc(&tmp, sizeof(float) * n));
  const int blocks = static_cast<int>((n + 255) / 256);
  hipLaunchKernelGGL(lbm_soa_to_aos_kernel, dim3(blocks), dim3(256), 0, c->stream,
                     c->grid[c->cur] + static_cast<size_t>(c->ghost) * c->p.nx, tmp, c->ps, c->ncells);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(cells_aos, tmp, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  HIP_TRY(e);
  return 0;
}

int lbm_set_cells(lbm_ctx* c, const float* cells_aos)
{
  if (!c || !cells_aos) { lbm_internal::set_error("lbm_set_cells: null argument"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  float* tmp = nullptr;
  const size_t n = c->ncells * 9;
  HIP_TRY(hipMalloc(&tmp, sizeof(float) * n));
  hipError_t e = hipMemcpyAsync(tmp, cells_aos, sizeof(float) * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    const int blocks = static_cast<int>((n + 255) / 256);
    hipLaunchKernelGGL(lbm_aos_to_soa_kernel, dim3(blocks), dim3(256), 0, c->stream, tmp,
                       c->grid[c->cur] + static_cast<size_t>(c->ghost) * c->p.nx, c->ps, c->ncells);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  HIP_TRY(e);
  return 0;
}

int lbm_av_velocity_sum(lbm_ctx* c, double* tot_u)
{
  if (!c || !tot_u) { lbm_internal::set_error("lbm_av_velocity_sum: null argument"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  const int blocks = static_cast<int>(std::min<size_t>((c->ncells + kBlock - 1) / kBlock, 1024));
  double* part = nullptr;
  HIP_TRY(hipMalloc(&part, sizeof(double) * blocks));
  // owned rows only; in K-step mode they start ghost rows in (ghost*nx is a multiple of 64 cells there)
  hipLaunchKernelGGL(lbm_av_velocity_kernel, dim3(blocks), dim3(kBlock), 0, c->stream,
                     c->grid[c->cur] + static_cast<size_t>(c->ghost) * c->p.nx, c->ps,
                     c->mask + static_cast<size_t>(c->ghost) * c->p.nx / 32, c->ncells, part);
  std::vector<double> host(blocks);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(host.data(), part, sizeof(double) * blocks, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(part);
  HIP_TRY(e);
  double s = 0.0;
  for (double v : host) s += v;
  *tot_u = s;
  return 0;
}

size_t lbm_halo_floats(const lbm_ctx* c) { return c ? static_cast<size_t>(3) * c->nxp : 0; }
void* lbm_halo_send_ptr(lbm_ctx* c, int dir) { return (c && (dir == 0 || dir == 1)) ? c->send[dir] : nullptr; }
void* lbm_halo_recv_ptr(lbm_ctx* c, int dir) { return (c && (dir == 0 || dir == 1)) ? c->recv[dir] : nullptr; }

int lbm_bind_halo_buffers(lbm_ctx* c, void* send_south, void* send_north, void* recv_south, void* recv_north)
{
  if (!c || !send_south || !send_north || !recv_south || !recv_north) { lbm_internal::set_error("lbm_bind_halo_buffers: null argument"); return 1; }
  void* ptrs[4] = {send_south, send_north, recv_south, recv_north};
  for (void* q : ptrs)
    if (reinterpret_cast<uintptr_t>(q) % 16 != 0) { lbm_internal::set_error("lbm_bind_halo_buffers: buffers must be 16-byte aligned"); return 1; }
  c->send[0] = static_cast<float*>(send_south); c->send[1] = static_cast<float*>(send_north);
  c->recv[0] = static_cast<float*>(recv_south); c->recv[1] = static_cast<float*>(recv_north);
  return 0;
}

int lbm_step_prepare(lbm_ctx* c, int n_steps, void* stream)
{
  if (!c || n_steps < 0) { lbm_internal::set_error("lbm_step_prepare: bad argument"); return 1; }
  if (c->ghost > 0) { lbm_internal::set_error("lbm_step_prepare: the context runs in K-step mode; use the lbm_macro_* calls"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = pick_stream(c, stream);
  if (begin_run(c, n_steps, s)) return 1;
  const int nx = c->p.nx;
  hipLaunchKernelGGL(lbm_pack_halo_kernel, dim3((nx + 255) / 256), dim3(256), 0, s, c->grid[c->cur], c->ps, nx, c->nyl, c->nxp,
                     c->send[0], c->send[1]);
  HIP_TRY(hipGetLastError());
  return 0;
}

int lbm_step_interior(lbm_ctx* c, void* stream)
{
  if (!c) { lbm_internal::set_error("lbm_step_interior: null context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_step_interior: no steps left; call lbm_step_prepare"); return 1; }
  hipStream_t s = pick_stream(c, stream);
  const int qrow = c->p.nx / c->lane_cells;
  StepArgs a = base_args(c, c->run_done + 1 < c->run_steps);
  a.quad_begin = qrow; a.quad_end = qrow * (c->nyl - 1);
  a.quad_begin2 = a.quad_end2 = 0;
  a.iters = c->iters_interior;
  a.partials_out = c->partials[c->parity];
  a.prev_partials = c->partials[c->parity ^ 1];
  a.n_prev = c->n_prev;
  if (c->n_part_interior > 0) {
    launch_step(c, a, c->n_part_interior, s);
    HIP_TRY(hipGetLastError());
    c->n_prev = 0;   // folded (by block 0 of this launch)
  }
  return 0;
}

int lbm_step_boundary(lbm_ctx* c, void* stream)
{
  if (!c) { lbm_internal::set_error("lbm_step_boundary: null context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_step_boundary: no steps left; call lbm_step_prepare"); return 1; }
  hipStream_t s = pick_stream(c, stream);
  const int qrow = c->p.nx / c->lane_cells;
  StepArgs a = base_args(c, c->run_done + 1 < c->run_steps);
  a.quad_begin = 0; a.quad_end = qrow;
  if (c->nyl > 1) { a.quad_begin2 = qrow * (c->nyl - 1); a.quad_end2 = qrow * c->nyl; }
  a.iters = 1;
  a.south_halo = c->recv[0];
  a.north_halo = c->recv[1];
  a.send_south = c->send[0];
  a.send_north = c->send[1];
  a.partials_out = c->partials[c->parity] + c->n_part_interior;
  a.prev_partials = c->partials[c->parity ^ 1];
  a.n_prev = c->n_prev;   // non-zero only when there was no interior launch to fold it
  launch_step(c, a, c->n_part_boundary, s);
  HIP_TRY(hipGetLastError());
  return 0;
}

int lbm_step_finish(lbm_ctx* c, void* stream)
{
  if (!c) { lbm_internal::set_error("lbm_step_finish: null context"); return 1; }
  hipStream_t s = pick_stream(c, stream);
  c->n_prev = c->n_part_interior + c->n_part_boundary;
  c->n_prev_vecs = 1;
  c->parity ^= 1;
  c->cur ^= 1;                                                              // d2q9-bgk.c:376-378
  c->run_done += 1;
  if (c->run_done == c->run_steps) {
    HIP_TRY(hipEventRecord(c->ev_end, s));
    c->ev_launches = c->run_steps * ((c->n_part_interior > 0 ? 1 : 0) + 1);
    c->ev_valid = true;
    hipLaunchKernelGGL(lbm_fold_kernel, dim3(1), dim3(kBlock), 0, s, c->partials[c->parity ^ 1], c->n_prev, c->n_prev_vecs, c->sums, c->counter);
    HIP_TRY(hipGetLastError());
    c->n_prev = 0;
  }
  return 0;
}

// ---- K-step ("macro-step") stepping of a row-partitioned run ------------------------------------

int lbm_macro_steps(const lbm_ctx* c) { return (c && c->ghost > 0) ? c->multi_K : 0; }

size_t lbm_macro_halo_floats(const lbm_ctx* c) { return (c && c->ghost > 0) ? static_cast<size_t>(c->ghost) * c->p.nx : 0; }

void* lbm_macro_send_ptr(lbm_ctx* c, int dir, int plane)
{
  if (!c || c->ghost == 0 || plane < 0 || plane >= 9 || (dir != 0 && dir != 1)) return nullptr;
  // first K owned rows go south, last K owned rows go north
  const size_t row = dir == 0 ? static_cast<size_t>(c->ghost) : static_cast<size_t>(c->nyl);
  return c->grid[c->cur] + plane * c->ps + row * c->p.nx;
}

void* lbm_macro_recv_ptr(lbm_ctx* c, int dir, int plane)
{
  if (!c || c->ghost == 0 || plane < 0 || plane >= 9 || (dir != 0 && dir != 1)) return nullptr;
  // ghost rows below the first owned row come from the south, those above the last one from the north
  const size_t row = dir == 0 ? 0 : static_cast<size_t>(c->ghost + c->nyl);
  return c->grid[c->cur] + plane * c->ps + row * c->p.nx;
}

int lbm_macro_prepare(lbm_ctx* c, int n_steps, void* stream)
{
  if (!c || n_steps < 0 || c->ghost == 0) { lbm_internal::set_error("lbm_macro_prepare: not a K-step context"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  return begin_run(c, n_steps, pick_stream(c, stream));
}

static int macro_k(const lbm_ctx* c) { return std::min(c->multi_K, c->run_steps - c->run_done); }

int lbm_macro_interior(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_interior: not a K-step context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_macro_interior: no steps left; call lbm_macro_prepare"); return 1; }
  const int k = macro_k(c), nty = c->nyl / kMTY;
  if (nty >= 3) {   // tile rows whose K-ring stays inside the owned rows
    launch_multi(c, k, c->run_done + k < c->run_steps, c->multi_tiles_x, c->multi_tiles_x * (nty - 2), 0, 0, /*fold=*/true,
                 pick_stream(c, stream));
    HIP_TRY(hipGetLastError());
    c->n_prev = 0;   // folded by this launch's block 0
  }
  return 0;
}

int lbm_macro_edge(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_edge: not a K-step context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_macro_edge: no steps left; call lbm_macro_prepare"); return 1; }
  const int k = macro_k(c), nty = c->nyl / kMTY;
  launch_multi(c, k, c->run_done + k < c->run_steps, 0, c->multi_tiles_x, (nty - 1) * c->multi_tiles_x, c->multi_tiles_x,
               /*fold=*/c->n_prev > 0, pick_stream(c, stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

int lbm_macro_finish(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_finish: not a K-step context"); return 1; }
  hipStream_t s = pick_stream(c, stream);
  const int k = macro_k(c);
  c->n_prev = c->multi_tiles;
  c->n_prev_vecs = k;
  c->parity ^= 1;
  c->cur ^= 1;
  c->run_done += k;
  c->ev_tile_launches += 2;
  if (c->run_done == c->run_steps) {
    HIP_TRY(hipEventRecord(c->ev_end, s));
    c->ev_launches = c->ev_tile_launches;
    c->ev_valid = true;
    hipLaunchKernelGGL(lbm_fold_kernel, dim3(1), dim3(kBlock), 0, s, c->partials[c->parity ^ 1], c->n_prev, c->n_prev_vecs, c->sums, c->counter);
    HIP_TRY(hipGetLastError());
    c->n_prev = 0;
    c->n_prev_vecs = 1;
  }
  return 0;
}

int lbm_macro_exchange_local(lbm_ctx* dst, lbm_ctx* src, int dir, void* stream)
{
  if (!dst || !src || dst->ghost == 0 || src->ghost != dst->ghost || src->p.nx != dst->p.nx || (dir != 0 && dir != 1)) {
    lbm_internal::set_error("lbm_macro_exchange_local: incompatible contexts");
    return 1;
  }
  // src's rows travelling in direction `dir` land in dst's ghost rows on the opposite side
  hipStream_t s = pick_stream(dst, stream);
  const size_t bytes = sizeof(float) * lbm_macro_halo_floats(src);
  for (int k = 0; k < 9; ++k)
    HIP_TRY(hipMemcpyAsync(lbm_macro_recv_ptr(dst, dir ^ 1, k), lbm_macro_send_ptr(src, dir, k), bytes, hipMemcpyDeviceToDevice, s));
  return 0;
}

int lbm_step_collect(lbm_ctx* c, void* stream, double* tot_u_per_step, int n_steps)
{
  if (!c || !tot_u_per_step || n_steps > c->run_done) { lbm_internal::set_error("lbm_step_collect: bad argument"); return 1; }
  hipStream_t s = pick_stream(c, stream);
  HIP_TRY(hipMemcpyAsync(tot_u_per_step, c->sums, sizeof(double) * n_steps, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

void* lbm_step_sums_device_ptr(lbm_ctx* c) { return c ? c->sums : nullptr; }

int lbm_last_run_kernel_ms(lbm_ctx* c, double* ms, int* launches)
{
  if (!c || !ms) { lbm_internal::set_error("lbm_last_run_kernel_ms: null argument"); return 1; }
  if (!c->ev_valid) { lbm_internal::set_error("lbm_last_run_kernel_ms: no completed run"); return 1; }
  float t = 0.f;
  HIP_TRY(hipEventSynchronize(c->ev_end));
  HIP_TRY(hipEventElapsedTime(&t, c->ev_begin, c->ev_end));
  *ms = t;
  if (launches) *launches = c->ev_launches;
  return 0;
}

int lbm_device(const lbm_ctx* c) { return c ? c->device : -1; }
void* lbm_stream(lbm_ctx* c) { return c ? c->stream : nullptr; }

int lbm_describe(const lbm_ctx* c, char* kernel_name, size_t len, long long* cells_per_launch, long long* state_bytes)
{
  if (!c) { lbm_internal::set_error("lbm_describe: null context"); return 1; }
  if (kernel_name && len) {
    if (c->multi_K > 0 && (c->self_periodic || c->ghost > 0)) std::snprintf(kernel_name, len, "lbm_multi_kernel<%d>", c->multi_K);
    else if (c->tile_kernel && c->self_periodic) std::snprintf(kernel_name, len, "lbm_tile_kernel<%d, %d>", c->tile_T, c->tile_H);
    else if (c->lane_cells == 1) std::snprintf(kernel_name, len, "lbm_step_kernel_narrow<%s>", c->nt_stores ? "true" : "false");
    else if (c->lds_kernel) std::snprintf(kernel_name, len, "lbm_step_kernel_lds<%s>", c->nt_stores ? "true" : "false");
    else std::snprintf(kernel_name, len, "lbm_step_kernel<%s>", c->nt_stores ? "true" : "false");
  }
  if (cells_per_launch) *cells_per_launch = static_cast<long long>(c->ncells);
  if (state_bytes) *state_bytes = static_cast<long long>(2 * 9 * c->ncells * sizeof(float) + c->ncells / 8);
  return 0;
}

}  // extern "C"
