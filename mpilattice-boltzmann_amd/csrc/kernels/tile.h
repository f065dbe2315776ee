// tile.h — lbm_tile_kernel<T,H>: up to H steps per launch for the launch-latency-bound small grids
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Temporally blocked form for launch-latency-bound grids (the three small shipped decks).
//
// A step of a <= 64 K-cell grid takes less time to compute than a kernel boundary costs, so one
// launch here advances the lattice by up to H steps: a block loads an R x R region (R = T + 2H: a T x T
// owned tile + an H-cell ghost ring, periodic in x and y) into LDS, and sub-step s recomputes the
// region shrunk by s cells from the LDS copy of sub-step s-1 (two buffers, one barrier per sub-step).
// Ghost cells are computed redundantly by neighbouring blocks with the same arithmetic, so no block
// ever waits for another.  Per-step sum|u| is taken over owned cells only; accelerate_flow is applied
// to row ny-2 (ghost copies too) between sub-steps exactly as between launches of the one-step
// kernels.  Results are bit-identical to the one-step kernels (same relax_core, same order of steps).
//
// Work per sub-step is what bounds this kernel (one block = one CU: its 4 SIMDs issue every wave
// that holds a live lane), so the lanes are dealt anew in every sub-step: lane i takes the i-th
// x-pair of the CURRENT region (packed arithmetic: finish_pair), live lanes fill whole waves from
// wave 0 up, and the rest of the block skips the sub-step.  <16,8>: 39 wave-passes per 8 steps instead
// of 52 with a fixed lane -> cell assignment (and 104 with one cell per lane).  The region's left edge
// alternates between even and odd x, so LDS accesses are pairs of dwords rather than 8-byte words.
// What a lane needs to know about its cells it reads from a flag byte per region cell.
// ------------------------------------------------------------------------------------------------
// Geometry is a template parameter pair: T = owned tile edge, H = ghost ring = max steps per launch
// (both even); the block has R*R/2 lanes (the load phase: one aligned x-pair per lane).
constexpr int kMaxTileSteps = 8;

// Diagnostic builds only (scripts/build_variant.sh tile_stamps -DLBM_TILE_STAMPS=1, scripts/tile_stamps.py): shader-clock
// stamps of block 1, lane 0 at the phase boundaries of a launch.  No stamp executes in the product build.
#ifndef LBM_TILE_STAMPS
#define LBM_TILE_STAMPS 0
#endif
#ifndef LBM_TILE88_BLOCK
#define LBM_TILE88_BLOCK 512      // lanes of a <8,8> block: 320 would do for its 288 load-phase lanes (1.045 vs 1.009 us/step at 128 x 128; 448: 1.016)
#endif
#if LBM_TILE_STAMPS
__device__ unsigned long long g_tile_stamps[16];
#define LBM_STAMP(i) do { if (blockIdx.x == 1 && threadIdx.x == 0) g_tile_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LBM_STAMP(i) do { } while (0)
#endif

template <int T, int H>
struct TileGeom {
  static constexpr int R = T + 2 * H;
  static constexpr int RP = R / 2;                  // aligned pairs per region row
  static constexpr int cells = R * R;
  // (a lane per region CELL, R*R lanes, so that every sub-step can deal one cell per lane, was measured and dropped:
  // 256x256 deck 0.142 -> 0.204 s, 128x128 0.0529 -> 0.0554 — sixteen-wave blocks pay more at the barriers than the
  // shorter chains of the early sub-steps save)
  static constexpr int lanes = RP * R;
  // launched size: whole waves, so that the wave-level sums see 64 active lanes; <8,8> gets 512 lanes rather than the
  // 320 its load phase needs, so that ALL its sub-steps (regions of 484 cells and less) deal one cell per lane.
  static constexpr int block = (T == 8 && H == 8 && LBM_TILE88_BLOCK > 64 * ((lanes + 63) / 64)) ? LBM_TILE88_BLOCK : 64 * ((lanes + 63) / 64);
  static constexpr int waves = block / 64;
  static constexpr size_t lds_bytes = sizeof(float) * 2 * 9 * cells + sizeof(double) * H * waves + cells;   // + flag bytes
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
  int single_max;              // sub-steps whose region holds at most this many cells deal ONE cell per lane (0: always x-pairs)
  float omega, accel_w1, accel_w2;
  int accel_row;               // ny-2
  int accel_last;              // apply accelerate_flow after the LAST sub-step too (another step follows)
  double* partials_out;        // [ksteps][ntiles]
  const double* prev_partials; // previous launch: [n_prev_vecs][n_prev]
  int n_prev, n_prev_vecs;
  double* sums;
  int* counter;
};

template <int T, int H, bool FULL, bool FAST>   // FULL: this launch does exactly H steps (region sizes are compile-time constants); FAST: float sum|u| terms
__global__ void __launch_bounds__((TileGeom<T, H>::block)) lbm_tile_kernel(const TileArgs a)
{
  using G = TileGeom<T, H>;
  constexpr int R = G::R, RP = G::RP, kCells = G::cells, kWaves = G::waves, kBlock = G::block;
  extern __shared__ __attribute__((aligned(16))) float lds[];        // [2][9][R*R] floats, reduction scratch, flag bytes
  double* red = reinterpret_cast<double*>(lds + 2 * 9 * kCells);    // [H][kWaves]
  uint8_t* cell_flags = reinterpret_cast<uint8_t*>(red + H * kWaves);   // per region cell: bit 0 obstacle, 1 owned, 2 on row ny-2
  const int tid = threadIdx.x;

  if (blockIdx.x == 0) {
    // fold block: the previous launch's per-tile sums, one vector per step, into sums[counter..]
    for (int v = 0; v < a.n_prev_vecs; ++v) {
      double s = 0.0;
      for (int i = tid; i < a.n_prev; i += kBlock) s += a.prev_partials[static_cast<size_t>(v) * a.n_prev + i];
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

  LBM_STAMP(0);
  const int tile = blockIdx.x - 1;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  // does the region of this tile meet row ny-2 at all ?  (block-uniform; the others skip accelerate_flow)
  bool tile_accel;
  {
    int d = (a.accel_row - (ty * T - H)) % a.ny;
    if (d < 0) d += a.ny;
    tile_accel = d < R || a.ny < R;
  }

  // ---- load: lane = one aligned x-pair of the region, periodic (d2q9-bgk.c:527-529 in x; :245-247 one-rank ring
  // in y); nx, T and H are even, so a pair never straddles the wrap and its first cell has an even index
  if (tid < RP * R) {
    const int ry = tid / RP, rx = 2 * (tid - ry * RP);
    int gx = (tx * T - H + rx) % a.nx; if (gx < 0) gx += a.nx;
    int gy = (ty * T - H + ry) % a.ny; if (gy < 0) gy += a.ny;
    const int cell = gy * a.nx + gx;
    const uint32_t mbits = (a.mask[cell >> 5] >> (cell & 31)) & 3u;
    const uint32_t common = ((rx >= H && rx < H + T && ry >= H && ry < H + T) ? 2u : 0u) | (gy == a.accel_row ? 4u : 0u);
    const int c = ry * R + rx;
#pragma unroll
    for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(lds + k * kCells + c) = *reinterpret_cast<const f2*>(a.src + k * a.ps + cell);
    *reinterpret_cast<uint16_t*>(cell_flags + c) = static_cast<uint16_t>(((mbits & 1u) | common) | ((((mbits >> 1) & 1u) | common) << 8));
  }
  __syncthreads();
  LBM_STAMP(1);

  double acc[H];
#pragma unroll
  for (int i = 0; i < H; ++i) acc[i] = 0.0;

  const int k_total = FULL ? H : a.ksteps;
  // One sub-step; S is a compile-time constant (the sub-steps are unrolled: buffer roles and the accumulator slot fold).
  auto substep = [&](auto sc) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    const float* bufA = lds + ((S & 1) ? 0 : 9 * kCells);
    float* bufB = lds + ((S & 1) ? 9 * kCells : 0);
    // cells still needed after this sub-step: the owned tile expanded by e = k_total - S
    const int e = k_total - S;
    const int side = T + 2 * e;
    if (side * side <= a.single_max && side * side <= kBlock) {   // block-uniform
      // Late sub-steps: the region fits the block's SIMDs with one CELL per lane — a wave alone on its SIMD issues an
      // instruction every 4+ cycles whether it is packed or not, so what counts is the length of one lane's chain:
      // ~190 instructions for a cell against ~330 for an x-pair (division, double sqrt and the unaligned LDS
      // accesses are per cell either way).  Same relax_core, same order of operations: the same bits.
      if (tid < side * side) {
        const int ry = small_div(tid, side), rx = tid - ry * side;
        const int x = H - e + rx, y = H - e + ry;
        const int c = y * R + x;
        const uint32_t fl = cell_flags[c];
        float t[9], o[9], out[9];                                                  // d2q9-bgk.c:530-538
        t[0] = bufA[0 * kCells + c];         t[1] = bufA[1 * kCells + c - 1];     t[2] = bufA[2 * kCells + c - R];
        t[3] = bufA[3 * kCells + c + 1];     t[4] = bufA[4 * kCells + c + R];     t[5] = bufA[5 * kCells + c - R - 1];
        t[6] = bufA[6 * kCells + c - R + 1]; t[7] = bufA[7 * kCells + c + R + 1]; t[8] = bufA[8 * kCells + c + R - 1];
        float msq, rinv;
        relax_core<float>(t, a.omega, o, msq, rinv);
        const bool blocked = fl & 1u;
        bounce_or_relax(t, o, blocked, out);                                      // :687-695
        if (tile_accel && (fl & 4u) && !blocked && (S < k_total || a.accel_last)) accelerate_cell(out, a.accel_w1, a.accel_w2);   // :457-469
        double term = 0.0;
        if ((fl & 3u) == 2u)                                                          // owned fluid cell (:667)
          term = FAST ? static_cast<double>(__builtin_amdgcn_sqrtf(msq) * rinv) : sqrt_of_float(msq) * static_cast<double>(rinv);
        acc[S - 1] = term;
        if (S < k_total) {
#pragma unroll
          for (int k = 0; k < 9; ++k) bufB[k * kCells + c] = out[k];
        } else {
          const int cell = (ty * T + ry) * a.nx + tx * T + rx;
#pragma unroll
          for (int k = 0; k < 9; ++k) a.dst[k * a.ps + cell] = out[k];
        }
      }
    } else {
      const int w = T / 2 + e;                          // pairs per row of that region
      if (tid < w * (T + 2 * e)) {
        const int ry = small_div(tid, w), rp = tid - ry * w;
        const int x = H - e + 2 * rp, y = H - e + ry;   // region cells (x, y), (x+1, y)
        const int c = y * R + x;
        const uint32_t fl0 = cell_flags[c], fl1 = cell_flags[c + 1];
        const uint32_t mbits = (fl0 & 1u) | ((fl1 & 1u) << 1);
        // a pair that starts on an odd x may straddle the edge of the owned tile: per-cell skip bits
        const uint32_t skip = (((fl0 & 1u) | ((fl0 & 2u) ? 0u : 1u))) | (((fl1 & 1u) | ((fl1 & 2u) ? 0u : 1u)) << 1);
        f2 p[9], out[9];                                                           // d2q9-bgk.c:530-538
        p[0] = f2{bufA[0 * kCells + c], bufA[0 * kCells + c + 1]};
        p[2] = f2{bufA[2 * kCells + c - R], bufA[2 * kCells + c - R + 1]};
        p[4] = f2{bufA[4 * kCells + c + R], bufA[4 * kCells + c + R + 1]};
        p[1] = f2{bufA[1 * kCells + c - 1], bufA[1 * kCells + c]};
        p[5] = f2{bufA[5 * kCells + c - R - 1], bufA[5 * kCells + c - R]};
        p[8] = f2{bufA[8 * kCells + c + R - 1], bufA[8 * kCells + c + R]};
        p[3] = f2{bufA[3 * kCells + c + 1], bufA[3 * kCells + c + 2]};
        p[6] = f2{bufA[6 * kCells + c - R + 1], bufA[6 * kCells + c - R + 2]};
        p[7] = f2{bufA[7 * kCells + c + R + 1], bufA[7 * kCells + c + R + 2]};
        // relaxation, bounce-back (:687-695), accelerate_flow of the following step (:457-469), sum|u| terms
        acc[S - 1] = finish_pair<FAST>(p, mbits, a.omega, tile_accel, (fl0 & 4u) && (S < k_total || a.accel_last), a.accel_w1, a.accel_w2,
                                 skip, out);
        if (S < k_total) {
#pragma unroll
          for (int k = 0; k < 9; ++k) { bufB[k * kCells + c] = out[k].x; bufB[k * kCells + c + 1] = out[k].y; }
        } else {
          // last sub-step: the region is the owned tile (x = H + 2 rp, y = H + ry), inside the grid
          const int cell = (ty * T + ry) * a.nx + tx * T + 2 * rp;
#pragma unroll
          for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(a.dst + k * a.ps + cell) = out[k];
        }
      }
    }
    if (S < k_total) __syncthreads();
    LBM_STAMP(1 + S);
  };
  if (1 <= k_total) substep(std::integral_constant<int, 1>{});
  if (2 <= k_total) substep(std::integral_constant<int, 2>{});
  if (3 <= k_total) substep(std::integral_constant<int, 3>{});
  if (4 <= k_total) substep(std::integral_constant<int, 4>{});
  if constexpr (H >= 8) {
    if (5 <= k_total) substep(std::integral_constant<int, 5>{});
    if (6 <= k_total) substep(std::integral_constant<int, 6>{});
    if (7 <= k_total) substep(std::integral_constant<int, 7>{});
    if (8 <= k_total) substep(std::integral_constant<int, 8>{});
  }
  static_assert(H == 4 || H == 8, "sub-steps are unrolled for H = 4 and 8");

  // per-step sums over the owned cells of this tile: one halving butterfly per wave (common.h), then one lane per step
  // over the waves
  const int ntiles = gridDim.x - 1;
  if constexpr (H == 8) {
    const double w = wave_sum_vec8(acc);
    if ((tid & 7) == 0) red[wave_sum_slot<8>(tid & 63) * kWaves + (tid >> 6)] = w;
  } else {
    const double w = wave_sum_vec4(acc[0], acc[1], acc[2], acc[3]);
    if ((tid & 15) == 0) red[wave_sum_slot<4>(tid & 63) * kWaves + (tid >> 6)] = w;
  }
  lds_barrier();                 // LDS only: no wait for the tile's global stores
  if (tid < k_total) {
    double t = 0.0;
    for (int w = 0; w < kWaves; ++w) t += red[tid * kWaves + w];
    a.partials_out[static_cast<size_t>(tid) * ntiles + tile] = t;
  }
  LBM_STAMP(10);
}

}  // namespace
