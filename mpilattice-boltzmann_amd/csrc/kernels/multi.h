// multi.h — lbm_multi_kernel<K>: K steps per pass over HBM for the bandwidth-bound grids and for K-step row partitions
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

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
#ifndef LBM_MTY            // experiment builds: -DLBM_MTY=24 -DLBM_MLANES=768 (taller tiles, two blocks per CU)
#define LBM_MTY 16
#define LBM_MLANES 512
#endif
#ifndef LBM_MWAVES         // most waves per SIMD the kernels are compiled for (register budget 512 / LBM_MWAVES)
#define LBM_MWAVES 6
#endif
#ifndef LBM_MTY4           // tile height of the 4-step instantiation, standard and narrow geometry (512 lanes)
#define LBM_MTY4 13
#endif
#ifndef LBM_MTY4T          // tile height and block size of the 4-step instantiation, tall geometry
#define LBM_MTY4T 23
#define LBM_MLANES4T 768
#endif
constexpr int kMTX = 64, kMTXNarrow = 32, kMTY = LBM_MTY, kMTY4 = LBM_MTY4, kMTY4Tall = LBM_MTY4T, kMLanes = LBM_MLANES, kMLanes4Tall = LBM_MLANES4T,
              kMaxMultiSteps = 4,
              kMaxGhost = 32,      // most ghost rows / columns a K-step partition keeps per side: the steps of a group of launches between two halo exchanges (32: column blocks)
              kMaxGroup = 8;       // most launches of such a group
constexpr int kMinMultiTY = kMTY < kMTY4 ? (kMTY < kMTY4Tall ? kMTY : kMTY4Tall) : (kMTY4 < kMTY4Tall ? kMTY4 : kMTY4Tall);
// Geometry of a launch: tile width, and by steps per launch tile height and block size.
//   kGeomStd     64-wide tiles.  K <= 3: 64 x 16, 512 lanes (K = 3: 72 x 20 frame, 51.8 KB, three blocks per CU).  K = 4 on the same tiles
//                (rounds 1-2) needs a 76 x 22 frame = 60 KB, two blocks per CU: 353 - 363 us/step at 8192 x 8192 against 341 - 347 for
//                K = 3.  On 64 x 13 tiles (round 3) its frame is 76 x 19 = 52.0 KB, three blocks per CU again: a tile recomputes 1.36 x
//                its cells per step instead of 1.31 x but the launch moves 21.5 B per cell-step instead of 26.9 — 8192 x 8192 346.6
//                (K = 3) -> 324.0, 4096 x 4096 87.2 -> 78.4, 1024 x 1024 8.16 -> 7.07, 8192 x 1024 53.0 -> 43.5 (profiles/r03/ab_k3_k4.txt).
//   kGeomTall    K = 4 on 64 x 23 tiles with 768-lane blocks: 76 x 29 frame = 79.3 KB, TWO blocks of twelve waves per CU (the same 24
//                waves): ring work 1.24 x, 60 wave-passes for 1.77 x the cells of 37.  With double-precision sum|u| terms this was a
//                draw (310.8 against 311.9 - 317.3, ab_k4_big_blocks_8192.txt); the launch runs at the socket power limit, and with
//                the compensated float terms the form that does less work per cell wins wherever a launch is rounds of blocks — us/step
//                64 x 13 / 64 x 21 / 64 x 22 / 64 x 23 (768 lanes): 8192 x 8192 315.0 & 324.7 / 304.6 / 307.7 / 303.9, 4096 x 4096 81.0 & 85.4 /
//                77.5 / 75.6 / 74.0, 8192 x 1024 43.6 & 44.1 / 40.5 / 40.2 / 39.9, 2048 x 2048 22.3 & 22.8 / 22.3 / 21.5 / 21.6, 1024 x 1024
//                6.98 & 7.25 / 7.58 / 6.93 / 6.81 (ab_big_blocks_matrix.txt; 832-, 896- and 1024-lane blocks spill: 500 us/step) — and
//                loses where it is one round or less (512 slots instead of 768): 768 x 768 4.91 against 4.56, 1024 x 768 6.20 / 5.57,
//                1536 x 1536 13.1 / 13.1 (ab_k3_k4_threshold_768lanes.txt): the host picks it from 2^20 cells up.  K <= 3: as kGeomStd.
//   kGeomNarrow  32-wide tiles, heights as kGeomStd: partitions so small that a launch is one round of blocks (twice the tiles, each
//                with half the dependent work — a 1024 x 128-row partition keeps 256 CUs busy instead of 128).
constexpr int kGeomStd = 0, kGeomNarrow = 1, kGeomTall = 2;
constexpr int geom_tx(int g) { return g == kGeomNarrow ? kMTXNarrow : kMTX; }
constexpr int multi_ty(int k, int g) { return k >= 4 ? (g == kGeomTall ? kMTY4Tall : kMTY4) : kMTY; }
constexpr int multi_lanes(int k, int g) { return (k >= 4 && g == kGeomTall) ? kMLanes4Tall : kMLanes; }
constexpr int geom_for(int k, int g) { return (g == kGeomTall && k < 4) ? kGeomStd : g; }      // the instantiation a launch of k steps uses

// Sub-step j of k (1-based) works on the owned tile grown by (k-j) rows and 2(k-j) columns on each
// side: columns grow twice as fast so that every region starts on an even x and a lane can own an
// x-PAIR of cells (8-byte accesses; the two cells' arithmetic is packed by the compiler into
// v_pk_*_f32, which halves the instruction count - the one-cell form of this kernel was VALU-bound).
#if LBM_TILE_STAMPS      // diagnostic builds (tile.h, scripts/tile_stamps.py): stamps of one block in the middle of the launch, lane 0
#define LBM_MSTAMP(i) do { if (blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) g_tile_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LBM_MSTAMP(i) do { } while (0)
#endif

template <int K, int GEOM = kGeomStd>
struct MultiGeom {
  static constexpr int TX = geom_tx(GEOM);                          // owned columns of a tile
  static constexpr int TY = multi_ty(K, GEOM);                      // owned rows of a tile
  static constexpr int LANES = multi_lanes(K, GEOM);                // block size
  static constexpr int EY = K - 1, EX = 2 * (K - 1);                // growth of the first sub-step
  static constexpr int W = TX + 2 * EX, H = TY + 2 * EY;            // LDS frame
  static constexpr int cells = W * H;
  static constexpr size_t lds_bytes = sizeof(float) * 9 * cells + sizeof(double) * K * (LANES / 64) + (K >= 2 ? cells / 2 : 0);   // + a flag byte per x-pair
  // waves per SIMD the kernel is compiled for: what its LDS frame lets a CU hold, LBM_MWAVES at most
  static constexpr int blocks_per_cu = lds_bytes * 8 <= 160 * 1024 ? 8 : static_cast<int>(160 * 1024 / lds_bytes);
  static constexpr int waves_per_simd = blocks_per_cu * (LANES / 64) / 4 < LBM_MWAVES ? (blocks_per_cu * (LANES / 64) / 4 < 1 ? 1 : blocks_per_cu * (LANES / 64) / 4) : LBM_MWAVES;
};

struct MultiArgs {
  const float* src;
  float* dst;
  const float* srck[9];        // src + k*ps, dst + k*ps: plane bases as kernel arguments, so that they are scalar
  float* dstk[9];              // loads at the point of use and never 64-bit vector arithmetic
  const uint32_t* mask;        // bit per STORAGE cell (ghost rows included)
  size_t ps;
  int nx;
  // Rows, all in STORAGE coordinates (row 0 = the first ghost row, or the first grid row of a whole periodic grid).  A launch
  // computes the rows [row_first, row_first + rows_compute): the owned rows, or — the first launches of a group that makes several
  // launches per halo exchange (lbm_p2p_impl.h) — the owned rows and `ext` ghost rows on each side, which the later launches of the
  // group then read instead of exchanged rows.  Only cells of the rows [count_first, count_end) — the rows the partition OWNS — enter the
  // per-step sums (the neighbours count theirs).
  int row_first, rows_compute;
  int rows_storage;            // owned rows + 2 x ghost rows
  int count_first, count_end;
  int y_periodic;
  int y0_global, ny_global;    // global row of storage row `row_first` (may be negative: a rank's bottom ghost rows wrap); global grid height
  int tiles_x;
  int tile_begin, tile_count, tile_begin2, tile_count2;   // tile ranges of this launch (second may be empty)
  int ntiles_total;            // stride of partials_out
  int xcd_remap;               // tile order: contiguous eighth per XCD (needs (tile_count+tile_count2) % 8 == 0)
  float omega, accel_w1, accel_w2;
  int accel_row;               // GLOBAL row ny-2
  int accel_last;
  double* partials_out;        // [ksteps][ntiles_total]
  const double* prev_partials; // previous launch: [n_prev_vecs][n_prev]
  int n_prev, n_prev_vecs;
  double* sums;
  int* counter;
  // peer-to-peer loop, the LAST launch of a group of several: "ready for epoch `ready_epoch`" to both neighbours (kernels/p2p.h
  // P2PWindowHeader::halo_ack), said by the fold block as the launch starts — this launch reads ghost rows of its source grid only and
  // writes owned rows, so the ghost rows of its destination grid, the next push's target, are free from here on.  0: nothing to say.
  unsigned long long* ready[2];  // [0] south, [1] north
  unsigned long long ready_epoch;
  // ... and, once the fold is done, waits until both neighbours have said the same to this rank (bounded; err is the transport's
  // host-mapped error word): the push kernel that follows this launch in the stream then stores without asking.  Every rank speaks
  // before it waits, so a ring of ranks cannot dead-lock here; the wait overlaps the launch's own tiles.
  const unsigned long long* wait_ready;  // this rank's halo_ack[2] (kPartTile: [4]), or null
  long long timeout_ticks;
  int* err;
  // Tile decomposition (kPartTile): the storage also has ghost COLUMNS on each side of the owned ones.  The launch computes every
  // column of its rows (the row wraps at the storage width: what the wrap feeds in is wrong by one more column per step and never
  // reaches a kept one before the next exchange), KEEPS — writes to the destination grid — the columns [keep_x0, keep_x1) only (owned +
  // the ghost columns the later launches of the group read; both even), and counts the owned columns [cx0, cx1).
  int keep_x0, keep_x1, cx0, cx1;
  unsigned long long* ready_x[2]; // ... and its ready words for the west and the east neighbour (this rank's: wait_ready[2], [3])
  // ... and the tiles of a launch as up to four RECTANGLES of the tile grid instead of two tile ranges (nrect > 0): the part of a group's first
  // launch that reads no exchanged row or column is one rectangle (tile rows x tile columns inside the rim), the rim the four around it.
  int nrect;
  struct Rect { int ty0, tx0, ntx, count; } rect[4];       // tile rows from ty0, tile columns [tx0, tx0 + ntx); count = rows x ntx blocks
  int nblocks;                     // kPartTile: tiles of this launch; the grid is padded to a multiple of 8 blocks so that the XCD-contiguous order
                                   // (xcd_remap) applies to any tile count — a tile rank's storage rows are never a multiple of 8 tiles wide
};

// A pair (x, x+1), x even, of population k into row `row` (a dword index) of an LDS frame of row stride W: interleaved
// for the populations pulled without an x shift, odd-x cells first then even-x cells for the others (see the reads).
template <int W>
__device__ __forceinline__ void store_pair(float* plane, int k, int row, int fx, f2 v)
{
  if (k == 0 || k == 2 || k == 4) {
    *reinterpret_cast<f2*>(plane + row + fx) = v;
  } else {
    plane[row + (fx >> 1)] = v.y;
    plane[row + (fx >> 1) + W / 2] = v.x;
  }
}

// One launch = exactly K steps (every region size, pass count and accumulator slot is a compile-time
// constant).  A run whose step count K does not divide ends with a launch of the smaller instantiation
// lbm_multi_kernel<k>, k < K: its frame needs k-1 <= ghost rows around the tile, so it runs on the same storage.
// Launch forms (PART).  The first round-4 build had ONE kernel for grids and partitions — row-range arguments, a `counted` bit beside
// `owned` in every pair, the peer-to-peer loop's ready words in the fold block — and ran lbm_multi_kernel<4> on 64 x 23 tiles 3 % slower at
// 8192 x 8192 than round 3's library in the same process (294 against 285.5 us/step); without the counted test and the ready words still
// 1.2 % slower, with an instruction count equal to the old kernel's to five in 1 851: hipcc had scheduled the two bodies differently (110
// differing lines in the opcode sequence), and a launch at the socket power limit notices.  Hence: whole grids, and the launches of a
// partition that compute its owned rows only, compile without either and with their row arithmetic spelled as rounds 1 - 3 spelled it;
// a launch that also computes ghost rows takes the counted form only in the tiles that hold such rows (its first and last tile rows),
// chosen by a block-uniform branch; the ready words live in a third instantiation.  Whole grids: back to round 3's time; the 8192 x 1024-row
// ring 43.4 -> 42.9 us/step beside 40.3 - 41.0 for the same rows as one periodic grid (profiles/r04/ab_part_template.txt).
// PART = kPartPlain (whole periodic grids, and the launches of a partition that compute its owned rows only), kPartGhost (a launch that
// also computes ghost rows: the counted test), kPartReady (owned rows only + the ready words in the fold block), kPartTile (every launch
// of a rank of the 2-D decomposition: ghost rows AND ghost columns — the counted and kept tests in x as well, in the tiles on the rim of
// the owned block only — and four ready words; the forms above do not change for it).
constexpr int kPartPlain = 0, kPartGhost = 1, kPartReady = 2, kPartTile = 3;
template <int K, int TERMS, int GEOM, int PART>   // TERMS: form of the sum|u| terms (kTermsCompensated by default), see finish_pair_lo; GEOM: kGeomStd / Narrow / Tall
__global__ void __launch_bounds__((MultiGeom<K, GEOM>::LANES), (MultiGeom<K, GEOM>::waves_per_simd)) lbm_multi_kernel(const MultiArgs a)
{
  using G = MultiGeom<K, GEOM>;
  constexpr int TX = G::TX, EX = G::EX, EY = G::EY, W = G::W, WH = G::W / 2, kCells = G::cells, kLanes = G::LANES, kWaves = kLanes / 64, TY = G::TY;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [9][kCells], then [K][kWaves] doubles
  double* red = reinterpret_cast<double*>(lds + 9 * kCells);
  // per x-pair of the frame, written by sub-step 1 and read by the in-LDS sub-steps (which then need no
  // grid coordinates at all): bits 0-1 obstacle bits, 2 owned (kept: written to the destination grid), 3 on the accelerate row,
  // 4 computed, 5 counted (owned by this partition: enters the per-step sums)
  uint8_t* pair_flags = reinterpret_cast<uint8_t*>(red + K * kWaves);
  const int tid = threadIdx.x;

  if (blockIdx.x == 0) {
    if constexpr (PART == kPartReady) {
      if (a.ready_epoch != 0ull && tid == 0) {
        for (int d = 0; d < 2; ++d) __hip_atomic_store(a.ready[d], a.ready_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if constexpr (PART == kPartTile) {
      if (a.ready_epoch != 0ull && tid == 0) {
        for (int d = 0; d < 2; ++d) __hip_atomic_store(a.ready[d], a.ready_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int d = 0; d < 2; ++d) __hip_atomic_store(a.ready_x[d], a.ready_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
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
    if constexpr (PART == kPartReady) {
      if (a.wait_ready && tid == 0) p2p_wait_flags(a.wait_ready, nullptr, 2, a.ready_epoch, 0ull, a.timeout_ticks, a.err, /*acquire=*/false);
    }
    if constexpr (PART == kPartTile) {
      if (a.wait_ready && tid == 0) p2p_wait_flags(a.wait_ready, nullptr, 4, a.ready_epoch, 0ull, a.timeout_ticks, a.err, /*acquire=*/false);
    }
    return;
  }

  LBM_MSTAMP(0);
  int b = blockIdx.x - 1;
  if (a.xcd_remap) {
    // blocks b, b+8, ... share an XCD (round-robin dispatch): give each XCD one contiguous eighth of
    // the launch so that tiles which overlap (x and y neighbours) meet in the same L2
    const int nb = gridDim.x - 1, per = nb >> 3;
    b = (b & 7) * per + (b >> 3);
  }
  if constexpr (PART == kPartTile) {
    if (b >= a.nblocks) return;                      // padding of the grid (block-uniform)
  }
  int tile_of_block = b < a.tile_count ? a.tile_begin + b : a.tile_begin2 + (b - a.tile_count);
  if constexpr (PART == kPartTile) {
    if (a.nrect > 0) {                               // block -> rectangle -> tile (block-uniform scalar work; constant indices: the arguments stay in SGPRs)
      int q = b, ty0 = a.rect[0].ty0, tx0 = a.rect[0].tx0, ntx = a.rect[0].ntx;
      if (a.nrect > 1 && q >= a.rect[0].count) {
        q -= a.rect[0].count; ty0 = a.rect[1].ty0; tx0 = a.rect[1].tx0; ntx = a.rect[1].ntx;
        if (a.nrect > 2 && q >= a.rect[1].count) {
          q -= a.rect[1].count; ty0 = a.rect[2].ty0; tx0 = a.rect[2].tx0; ntx = a.rect[2].ntx;
          if (a.nrect > 3 && q >= a.rect[2].count) { q -= a.rect[2].count; ty0 = a.rect[3].ty0; tx0 = a.rect[3].tx0; ntx = a.rect[3].ntx; }
        }
      }
      const int rr = q / ntx;
      tile_of_block = (ty0 + rr) * a.tiles_x + tx0 + (q - rr * ntx);
    }
  }
  const int tile = tile_of_block;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int x0 = tx * TX;
  const int sy0 = a.row_first + ty * TY;          // storage row of the tile's first row
  const int nx = a.nx;
  constexpr bool XR = PART == kPartTile;            // ghost columns: kept / counted column ranges
  const int rows_storage = (PART == kPartGhost || PART == kPartTile) ? a.rows_storage : a.rows_compute + 2 * a.row_first;   // (owned rows only: row_first ghost rows on each side)
  [[maybe_unused]] const int row_end = a.row_first + a.rows_compute;   // first storage row past the rows this launch computes
  const int tile_row_base = sy0 * nx;               // block-uniform: a scalar multiply
  const int grid_cells = rows_storage * nx;
  constexpr int ksteps = K;
  double acc[K];
  float acc_lo[K];             // kTermsCompensated: the low parts of a lane's terms, widened once at the end
#pragma unroll
  for (int i = 0; i < K; ++i) { acc[i] = 0.0; acc_lo[i] = 0.0f; }

  // does the LDS frame of this tile meet the global accelerate row ny-2 at all ?  (block-uniform)
  bool tile_accel;
  {
    int d = (a.accel_row - (a.y0_global + ty * TY - EY)) % a.ny_global;     // frame row 0 is storage row row_first + ty*TY - EY
    if (d < 0) d += a.ny_global;
    tile_accel = d < G::H || a.ny_global < G::H;
  }
  // storage row -> does it hold the global accelerate row ny-2 ?
  auto on_accel_row = [&](int sr) {
    int g = a.y0_global + sr - a.row_first;
    if (g < 0) g += a.ny_global; else if (g >= a.ny_global) g -= a.ny_global;
    return g == a.accel_row;
  };

  // The K sub-steps, in two forms: COUNT = the tile holds rows that are computed but not counted (a launch that also advances ghost rows:
  // its first and last tile rows) — every pair then carries a `counted` bit beside `owned`; all other tiles, and every tile of the other
  // launches, take the form without it, whose schedule is the one rounds 1 - 3 measured (see PART above: the test costs 2 - 3 % when every
  // tile carries it).  The choice is block-uniform.
  auto k_substeps = [&](auto count_c) __attribute__((always_inline)) {
    constexpr bool COUNT = decltype(count_c)::value;
    // ---- sub-step 1: pull from the source grid; region = owned tile grown by (ksteps-1) rows / 2(ksteps-1) columns
    {
      const int ey = ksteps - 1, ex = 2 * ey;
      const int wp = (TX + 2 * ex) / 2;                                 // pairs per region row
      const int np = wp * (TY + 2 * ey);
      // tiles whose frame (and its x -+ 1, y -+ 1 reads) lies inside the grid need none of the periodic
      // wraps and none of the partial-tile tests: block-uniform fast path for all but the edge tiles
      const bool inner = x0 - EX >= 2 && x0 + TX + EX + 2 <= nx && sy0 - EY >= 1 && sy0 + TY + EY + 1 <= rows_storage &&
                         sy0 + TY <= a.row_first + a.rows_compute;
  #pragma unroll 1
      for (int i = tid; i < np; i += kLanes) {
        const int ry = i / wp, rp = i - ry * wp;
        const int fx = EX - ex + 2 * rp, fy = EY - ey + ry;               // LDS frame coordinates (fx even)
        int gx = x0 + fx - EX;
        int sr = sy0 + fy - EY;
        // One multiply per pair, and a 24-bit one (v_mul_lo_u32 and v_mad_u64_u32 issue at quarter rate: the three
        // products sr * nx, ys * nx, yn * nx were 48 cycles of a pass): the tile's first row is a scalar product, the row
        // inside the frame a small factor (nx < 2^23 is part of the kernel's eligibility), the rows above and below and
        // the periodic wraps are additions.
        int cell = tile_row_base + __mul24(fy - EY, nx) + gx;
        int d_south = -nx, d_north = nx;
        if (!inner) {
          if (gx < 0) { gx += nx; cell += nx; } else if (gx >= nx) { gx -= nx; cell -= nx; }      // periodic (:527-529)
          // row partition whose rows the tile height does not divide: the last tile row sticks out past the
          // ghost rows; those cells lie outside every owned cell's dependency cone and are skipped
          if (!a.y_periodic && sr + 1 >= rows_storage) {
            if (ksteps > 1) pair_flags[(fy * W + fx) >> 1] = 0;
            continue;
          }
          if (a.y_periodic) {                                                               // periodic (:245-247)
            if (sr < 0) { sr += rows_storage; cell += grid_cells; } else if (sr >= rows_storage) { sr -= rows_storage; cell -= grid_cells; }
            if (sr == 0) d_south = grid_cells - nx;
            if (sr + 1 >= rows_storage) d_north = nx - grid_cells;
          }
        }
        // three 32-bit byte offsets per lane on block-uniform plane bases (scalar base + vector offset
        // addressing: no 64-bit address arithmetic in the vector unit; the x -+ 1 shifts live in the bases)
        const uint32_t o_here = 4u * static_cast<uint32_t>(cell);
        const uint32_t o_south = 4u * static_cast<uint32_t>(cell + d_south);
        const uint32_t o_north = 4u * static_cast<uint32_t>(cell + d_north);
        f2 p[9];
        p[0] = at_byte<f2>(a.srck[0], o_here);                                                   // :530
        p[2] = at_byte<f2>(a.srck[2], o_south);                                         // :532
        p[4] = at_byte<f2>(a.srck[4], o_north);                                         // :534
        p[1] = at_byte<f2u>(a.srck[1] - 1, o_here);                                         // :531
        p[5] = at_byte<f2u>(a.srck[5] - 1, o_south);                                    // :535
        p[8] = at_byte<f2u>(a.srck[8] - 1, o_north);                                    // :538
        p[3] = at_byte<f2u>(a.srck[3] + 1, o_here);                                     // :533
        p[6] = at_byte<f2u>(a.srck[6] + 1, o_south);                                    // :536
        p[7] = at_byte<f2u>(a.srck[7] + 1, o_north);                                    // :537
        if (!inner) {
          if (gx == 0) {                        // x_w wraps to nx-1 (:529)
            p[1].x = at_byte<float>(a.srck[1] + nx - 1, o_here); p[5].x = at_byte<float>(a.srck[5] + nx - 1, o_south);
            p[8].x = at_byte<float>(a.srck[8] + nx - 1, o_north);
          }
          if (gx == nx - 2) {                   // x_e wraps to 0 (:527-528)
            p[3].y = at_byte<float>(a.srck[3] + 2 - nx, o_here); p[6].y = at_byte<float>(a.srck[6] + 2 - nx, o_south);
            p[7].y = at_byte<float>(a.srck[7] + 2 - nx, o_north);
          }
        }
        const uint32_t mbits = (at_byte<uint32_t>(a.mask, 4u * (static_cast<uint32_t>(cell) >> 5)) >> (cell & 31)) & 3u;
        f2 out[9];
        // owned = inside the tile AND inside the grid (the last tile column / row may stick out of a grid
        // whose edges are not multiples of the tile: those cells are periodic images, computed but not kept)
        const int srow = sy0 + fy - EY;                                   // the pair's storage row before any periodic wrap
        const bool owned_rows = fx >= EX && fx < EX + TX && fy >= EY && fy < EY + TY &&
                                (inner || (x0 + fx - EX < nx && sy0 + fy - EY < a.row_first + a.rows_compute));
        const bool counted_rows = COUNT ? (owned_rows && srow >= a.count_first && srow < a.count_end) : owned_rows;
        bool owned = owned_rows, counted = counted_rows;
        if constexpr (XR && COUNT) {                                      // ghost columns: kept and counted column ranges (even bounds: whole pairs)
          const int gxo = x0 + fx - EX;                                   // the pair's storage column before any periodic wrap
          owned = owned_rows && gxo >= a.keep_x0 && gxo < a.keep_x1;
          counted = counted_rows && owned && gxo >= a.cx0 && gxo < a.cx1;
        }
        bool accel_row_here = false;
        if (tile_accel) accel_row_here = on_accel_row(sr);
        acc[0] += finish_pair_lo<TERMS>(p, mbits, a.omega, tile_accel, (ksteps > 1 || a.accel_last) && accel_row_here, a.accel_w1, a.accel_w2, counted ? mbits : 3u, out, acc_lo[0]);
        if (ksteps > 1) {
  #pragma unroll
          for (int k = 0; k < 9; ++k) store_pair<W>(lds + k * kCells, k, fy * W, fx, out[k]);
          pair_flags[(fy * W + fx) >> 1] = static_cast<uint8_t>(mbits | (owned ? 4u : 0u) | (accel_row_here ? 8u : 0u) | 16u | ((COUNT && counted) ? 32u : 0u));
        } else if (owned) {
  #pragma unroll
          for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(out[k], &at_byte<f2>(a.dstk[k], o_here));
        }
      }
    }
    LBM_MSTAMP(1);
    if constexpr (K >= 2) {
      __syncthreads();
      LBM_MSTAMP(2);
      // ---- sub-steps 2..ksteps: in place in the LDS frame, each on a region one row / two columns smaller.
      // In place without holding a whole region in registers: sub-step j writes its row r where the
      // frame it read kept row r-1 (the frame creeps down one storage row per sub-step), and a region
      // of more than 512 pairs goes in passes of whole rows, bottom to top.  A pass reads, meets at a
      // barrier, then writes; what it overwrites (old rows up to its last row - 1) no later pass reads.
      auto in_lds_substep = [&](const int j) __attribute__((always_inline)) {
        const int ey = ksteps - j, ex = 2 * ey;
        const int wp = (TX + 2 * ex) / 2;                                // pairs per region row
        const int rows = TY + 2 * ey;
        const int rpp = kLanes / wp;                                       // whole rows per pass
        const bool last = j == ksteps;
        const int rd = (j - 2) * W, wr = (j - 1) * W;                      // storage shift of the frame read / written
        // lane t -> row t / wp of the pass.  (Round 4 tried rows dealt to ALIGNED 32-lane groups — LDS reads are served in 32-lane groups, and
        // nine groups of ten straddle two region rows here, a two-way bank conflict each: SQ_LDS_BANK_CONFLICT 90.5 M -> 29.0 M cycles per
        // launch, LDS-array cycles 240 M -> 179 M, and the launch time unchanged: profiles/r04/ab_lds_rowmap.txt.  Code not kept.)
        const int ry = tid / wp, rp = tid - ry * wp;
        const int fx = EX - ex + 2 * rp;
        for (int r0 = 0; r0 < rows; r0 += rpp) {        // one or two passes (compile-time count: unrolled)
          f2 outs[9];
          int slot = -1;
          const bool in_region = ry < rpp && r0 + ry < rows;
          // whole waves without work skip the pass; in the others every lane computes (an idle lane on the
          // region's first row) and only the write is predicated: no per-lane state to merge at the barrier
          if (__builtin_amdgcn_ballot_w64(in_region) != 0ull) {
            const int fy = EY - ey + (in_region ? r0 + ry : 0);
            const int cf = fy * W + fx;                                    // frame position; stored rd (read) / wr (written) lower
            const uint32_t fl = pair_flags[cf >> 1];
            const bool lane_on = in_region && (fl & 16u);
            // Frame layout, per population: the three that are pulled without an x shift (0, 2, 4) keep their rows
            // interleaved — a pair is one aligned ds_read_b64.  The six that are pulled from x -+ 1 have their rows stored
            // DE-INTERLEAVED, odd-x cells first: a row is O[0..WH), E[0..WH) (WH = W/2).  For a pair (x, x+1), x = 2i, the
            // west pulls are O[i-1], E[i] and the east pulls O[i], E[i+1]: one ds_read2_b32 each, ascending addresses in
            // lane order (no register swap), lanes on consecutive dwords (interleaved, they were dword pairs at odd
            // addresses with lane stride 2, two-way bank conflicts in both passes of the instruction).
            const int ci = cf - rd;                                        // interleaved planes: cell (fx, fy)
            const int cs = fy * W + (fx >> 1) - rd;                        // split planes: O[i] of row fy; E[i] is WH further
            f2 p[9];
            p[0] = *reinterpret_cast<const f2*>(lds + 0 * kCells + ci);
            p[2] = *reinterpret_cast<const f2*>(lds + 2 * kCells + ci - W);
            p[4] = *reinterpret_cast<const f2*>(lds + 4 * kCells + ci + W);
            p[1] = f2{lds[1 * kCells + cs - 1], lds[1 * kCells + cs + WH]};
            p[5] = f2{lds[5 * kCells + cs - W - 1], lds[5 * kCells + cs - W + WH]};
            p[8] = f2{lds[8 * kCells + cs + W - 1], lds[8 * kCells + cs + W + WH]};
            p[3] = f2{lds[3 * kCells + cs], lds[3 * kCells + cs + WH + 1]};
            p[6] = f2{lds[6 * kCells + cs - W], lds[6 * kCells + cs - W + WH + 1]};
            p[7] = f2{lds[7 * kCells + cs + W], lds[7 * kCells + cs + W + WH + 1]};
            const bool owned = lane_on && (fl & 4u);
            const bool counted = COUNT ? (lane_on && (fl & 32u)) : owned;
            float term_lo = 0.0f;
            const double term = finish_pair_lo<TERMS>(p, fl & 3u, a.omega, tile_accel, (!last || a.accel_last) && (fl & 8u), a.accel_w1, a.accel_w2,
                                                      counted ? (fl & 3u) : 3u, outs, term_lo);
  #pragma unroll
            for (int m = 1; m < K; ++m)
              if (m == j - 1) { acc[m] += term; acc_lo[m] += term_lo; }
            // an owned pair lies inside the grid: its cell index needs no periodic wrap
            slot = !lane_on ? -1 : last ? (owned ? tile_row_base + __mul24(fy - EY, nx) + x0 + fx - EX : -1) : fy * W - wr;   // in LDS: the row; fx is added below
          }
          if (!last) {
            __syncthreads();                     // every lane of the pass has read its neighbours
            if (slot >= 0) {
  #pragma unroll
              for (int k = 0; k < 9; ++k) store_pair<W>(lds + k * kCells, k, slot, fx, outs[k]);
            }
          } else if (slot >= 0) {
  #pragma unroll
            for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(outs[k], &at_byte<f2>(a.dstk[k], 4u * static_cast<uint32_t>(slot)));
          }
        }
        if (!last) __syncthreads();
        LBM_MSTAMP(1 + j);
      };
      // compile-time sub-step index: region sizes, `last`, accumulator slot all fold
  #pragma unroll
      for (int j = 2; j <= K; ++j) in_lds_substep(j);
    }

  };
  if constexpr (PART == kPartGhost) {
    if (sy0 >= a.count_first && sy0 + TY <= a.count_end) k_substeps(std::false_type{});       // every row of the tile counts
    else k_substeps(std::true_type{});
  } else if constexpr (PART == kPartTile) {
    if (sy0 >= a.count_first && sy0 + TY <= a.count_end && x0 >= a.cx0 && x0 + TX <= a.cx1) k_substeps(std::false_type{});   // ... and every column
    else k_substeps(std::true_type{});
  } else {
    k_substeps(std::false_type{});
  }

  // per-step sums over the owned cells of this tile
  if constexpr (TERMS == kTermsCompensated) {
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] += static_cast<double>(acc_lo[i]);
  }
  if constexpr (K == 1) {
    const double w = wave_sum(acc[0]);
    if ((tid & 63) == 0) red[tid >> 6] = w;
  } else {
    static_assert(K <= 4, "wave_sum_vec4 carries four steps");
    const double w = wave_sum_vec4(acc[0], acc[1], K > 2 ? acc[2] : 0.0, K > 3 ? acc[K > 3 ? 3 : 0] : 0.0);
    const int q = wave_sum_slot<4>(tid & 63);
    if ((tid & 15) == 0 && q < K) red[q * kWaves + (tid >> 6)] = w;
  }
  lds_barrier();                 // LDS only: the block does not wait for its own global stores to complete
  if (tid < ksteps) {
    double t = 0.0;
    for (int w = 0; w < kWaves; ++w) t += red[tid * kWaves + w];
    a.partials_out[static_cast<size_t>(tid) * a.ntiles_total + tile] = t;
  }
  LBM_MSTAMP(2 + K);
}

}  // namespace
