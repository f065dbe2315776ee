// tile.h — lbm_tile_kernel<T,H>: up to H steps per launch for the launch-latency-bound small grids
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

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

}  // namespace
