// tile.h — lbm_tile_kernel<T,H>: up to H steps per launch for the launch-latency-bound small grids
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Temporally blocked form for launch-latency-bound grids (the three small shipped decks).
//
// A step of a <= 64 K-cell grid takes less time to compute than a kernel boundary costs, so one
// launch here advances the lattice by up to H steps (e.g. T = 16, H = 8: a 512-lane block loads a 32x32
// region (a 16x16 owned tile + an 8-cell ghost ring, periodic in x and y) into LDS, every lane keeps
// ONE x-PAIR of region cells for the whole launch, and sub-step s recomputes the region shrunk by s
// cells from the LDS copy of sub-step s-1 (double-buffered, one barrier per sub-step).  Ghost cells are
// computed redundantly by neighbouring blocks with the same arithmetic, so no block ever waits for
// another.  Per-step sum|u| is taken over owned cells only; accelerate_flow is applied to row ny-2
// (ghost copies too) between sub-steps exactly as between launches of the one-step kernels.  Results
// are bit-identical to the one-step kernels (same relax_core, same order of steps).
//
// Pairs: with one cell per lane a <16,8> block was 16 waves on the 4 SIMDs of its CU and a sub-step
// cost four waves' worth of instructions per SIMD; a lane that owns two x-adjacent cells runs the
// packed arithmetic of lbm_multi_kernel (finish_pair), so the same block is 8 waves.  A pair that
// straddles the edge of the shrinking region is computed whole; its outer cell is never read by a
// cell that is still needed.
// ------------------------------------------------------------------------------------------------
// Geometry is a template parameter pair: T = owned tile edge, H = ghost ring = max steps per launch
// (both even); the region edge is R = T + 2H and the block has R*R/2 lanes.
constexpr int kMaxTileSteps = 8;

template <int T, int H>
struct TileGeom {
  static constexpr int R = T + 2 * H;
  static constexpr int RP = R / 2;                  // pairs per region row
  static constexpr int cells = R * R;
  static constexpr int lanes = RP * R;
  static constexpr int waves = (lanes + 63) / 64;
  static constexpr size_t lds_bytes = sizeof(float) * 2 * 9 * cells + sizeof(double) * H * waves;
  static_assert(lanes <= 1024 && H <= kMaxTileSteps && T % 2 == 0 && H % 2 == 0, "unsupported tile geometry");
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
__global__ void __launch_bounds__((T + 2 * H) * (T + 2 * H) / 2) lbm_tile_kernel(const TileArgs a)
{
  using G = TileGeom<T, H>;
  constexpr int R = G::R, RP = G::RP, kCells = G::cells, kLanes = G::lanes, kWaves = G::waves;
  extern __shared__ __attribute__((aligned(16))) float lds[];        // [2][9][R*R] floats, then reduction scratch
  double* red = reinterpret_cast<double*>(lds + 2 * 9 * kCells);    // [H][kWaves]
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
  const int ry = tid / RP, rx = 2 * (tid - ry * RP);                  // this lane's pair: region cells (rx, ry), (rx+1, ry)
  // global cells of this lane, periodic (d2q9-bgk.c:527-529 in x; :245-247 one-rank ring in y); nx, T and H
  // are even, so a pair never straddles the wrap and its first cell has an even index
  int gx = (tx * T - H + rx) % a.nx; if (gx < 0) gx += a.nx;
  int gy = (ty * T - H + ry) % a.ny; if (gy < 0) gy += a.ny;
  const int cell = gy * a.nx + gx;
  const uint32_t mbits = (a.mask[cell >> 5] >> (cell & 31)) & 3u;
  const bool owned = rx >= H && rx < H + T && ry >= H && ry < H + T;
  const bool on_accel_row = gy == a.accel_row;
  // does the region of this tile meet row ny-2 at all ?  (block-uniform; the others skip accelerate_flow)
  bool tile_accel;
  {
    int d = (a.accel_row - (ty * T - H)) % a.ny;
    if (d < 0) d += a.ny;
    tile_accel = d < R || a.ny < R;
  }

  float* bufA = lds;
  float* bufB = lds + 9 * kCells;
  const int c = ry * R + rx;
#pragma unroll
  for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(bufA + k * kCells + c) = *reinterpret_cast<const f2*>(a.src + k * a.ps + cell);
  __syncthreads();

  double acc[H];
#pragma unroll
  for (int i = 0; i < H; ++i) acc[i] = 0.0;
  f2 out[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) out[k] = f2{0.0f, 0.0f};

  const int k_total = a.ksteps;
#pragma unroll 1
  for (int s = 1; s <= k_total; ++s) {
    // cells still needed after this sub-step: the owned tile expanded by (k_total - s); a pair with one
    // such cell is computed whole
    const int e = k_total - s;
    const bool active = rx + 1 >= H - e && rx < H + T + e && ry >= H - e && ry < H + T + e;
    if (active) {
      f2 p[9];                                                                   // d2q9-bgk.c:530-538
      p[0] = *reinterpret_cast<const f2*>(bufA + 0 * kCells + c);
      p[2] = *reinterpret_cast<const f2*>(bufA + 2 * kCells + c - R);
      p[4] = *reinterpret_cast<const f2*>(bufA + 4 * kCells + c + R);
      p[1] = f2{bufA[1 * kCells + c - 1], bufA[1 * kCells + c]};
      p[5] = f2{bufA[5 * kCells + c - R - 1], bufA[5 * kCells + c - R]};
      p[8] = f2{bufA[8 * kCells + c + R - 1], bufA[8 * kCells + c + R]};
      p[3] = f2{bufA[3 * kCells + c + 1], bufA[3 * kCells + c + 2]};
      p[6] = f2{bufA[6 * kCells + c - R + 1], bufA[6 * kCells + c - R + 2]};
      p[7] = f2{bufA[7 * kCells + c + R + 1], bufA[7 * kCells + c + R + 2]};
      // relaxation, bounce-back (:687-695), accelerate_flow of the following step (:457-469), sum|u| terms
      const double term = finish_pair(p, mbits, a.omega, tile_accel, on_accel_row && (s < k_total || a.accel_last),
                                      a.accel_w1, a.accel_w2, owned, out);
      if (owned) {
#pragma unroll
        for (int i = 0; i < H; ++i)
          if (i == s - 1) acc[i] = term;
      }
      if (s < k_total) {
#pragma unroll
        for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(bufB + k * kCells + c) = out[k];
      }
    }
    __syncthreads();
    float* sw = bufA; bufA = bufB; bufB = sw;
  }

  if (owned) {
#pragma unroll
    for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(a.dst + k * a.ps + cell) = out[k];
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
