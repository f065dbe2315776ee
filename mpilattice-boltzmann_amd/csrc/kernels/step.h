// step.h — one step per launch: lbm_step_kernel (4 cells per lane), lbm_step_kernel_narrow (1 cell per lane), lbm_step_kernel_lds (LDS-staged variant)
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

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

#if LBM_EXPERIMENTS   // (not in liblbm_d2q9.so as shipped: never faster than the direct-load form — no value is reused)
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
#endif   // LBM_EXPERIMENTS

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
  // boundary launch of a peer-to-peer run: the outgoing halo rows were stored straight into the neighbours' windows;
  // a full barrier drains every wave's stores (block_sum's barriers order LDS only), then one lane writes the XCD's L2
  // back towards the peers
  if (a.release_sends) __syncthreads();
  if (threadIdx.x == 0) {
    a.partials_out[wblock] = acc;
    if (a.release_sends) __atomic_thread_fence(__ATOMIC_RELEASE);
  }
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
  if (a.release_sends) __syncthreads();                                  // see lbm_step_kernel
  if (threadIdx.x == 0) {
    a.partials_out[wblock] = acc;
    if (a.release_sends) __atomic_thread_fence(__ATOMIC_RELEASE);
  }
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

}  // namespace
