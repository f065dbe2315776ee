// sweep.h — lbm_sweep_kernel<R>: three steps per pass over HBM by STREAMING temporal blocking (whole periodic grids)
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// lbm_multi_kernel<3> recomputes a ring around every 64 x 16 tile: 72x20 + 68x18 + 64x16 cells for 3 x 64x16 owned
// cell-steps (x 1.20; x 1.25 in wave-passes) and reads 72x20 / 64x16 = 1.41 x the tile from L2 / HBM.  Here a block owns
// a STRIP of 64 columns and sweeps it upwards, R rows per tick, with the three time levels in flight at once — a
// software pipeline in y instead of a ring in y:
//
//   tick t:   level 1 (t+1): rows  i1 in [R t, R t + R)            pulled from the source grid, result -> LDS ring 1
//             level 2 (t+2): rows  i2 in [R (t-1) - 2, R t - 2)    from ring 1 (rows i2-1 .. i2+1 of level 1), -> LDS ring 2
//             level 3 (t+3): rows  i3 in [R (t-2) - 4, R (t-1) - 4) from ring 2, -> destination grid
//   barrier, next tick.
//
// Every level only reads rows that EARLIER ticks wrote, so the lanes of a block work on all three levels at once
// (R = 5: 36 + 34 + 32 = 102 x-pairs per row triple, 510 of a 512-lane block's lanes busy) and a tick needs ONE barrier.
// Redundant work is left in x only: 72 + 68 + 64 columns for 3 x 64 (x 1.06), the source is read 72 / 64 = 1.125 times,
// and a segment of S rows pays 2R + 4 rows of pipeline fill.  Rings hold 2R + 2 rows each (a tick's new rows overwrite
// rows no later tick reads); level-1 lanes issue the NEXT tick's pulls before they relax this tick's.
// Same relax_core / finish_pair, same step order: results are bit-identical to three launches of the one-step kernel.
// Whole periodic grids with nx a multiple of 64 only (lbm_run); everything else stays with lbm_multi_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kSTX = 64;                 // owned columns of a strip
constexpr int kSW = kSTX + 8;            // level-1 frame width (ring of 4 columns per side: 2 per later level)

// Three forms of the pipeline's storage (MODE):
//   0  rings of 2R + 2 rows: a tick's new rows land in slots nobody reads this tick (ONE barrier per tick) and the level-1
//      lanes hold the next tick's pulls in registers while they relax this tick's: 62 KB of LDS and 97 VGPRs, two blocks per CU
//   1  rings of R + 2 rows, written in place: read, barrier, relax, write, barrier; pulls issued and awaited inside the
//      tick: 36 KB and 76 VGPRs, three blocks per CU — but every tick waits a memory round trip
//   2  as 1, and the level-1 pulls come through LDS: after the tick's reads, loader waves issue direct global -> LDS loads
//      (global_load_lds_dwordx4: no VGPRs) of the NEXT tick's source rows into a staging area — per population the R rows it
//      is pulled from, 80 columns (aligned quads around the frame: the x -+ 1 shifts become LDS offsets, the periodic wrap
//      a whole-quad matter) — which land while the block relaxes and writes; 51 KB, three blocks per CU.  Loader waves are
//      the ones that issue no global stores (levels 1 and 2), so their vmcnt(0) waits for loads only.
template <int R, int MODE>
struct SweepGeom {
  static constexpr int n1 = R * (kSW / 2), n2 = R * (kSW / 2 - 2), n3 = R * (kSW / 2 - 4);   // x-pairs per tick and level
  static constexpr int lanes = ((n1 + n2 + n3 + 63) / 64) * 64;
  static constexpr int ring = MODE == 0 ? 2 * R + 2 : R + 2;                                  // rows per LDS ring
  static constexpr int stage_w = kSW + 8;                                                     // staged source row: columns x0-8 .. x0+71
  static constexpr int stage_floats = MODE == 2 ? 9 * R * stage_w : 0;
  static constexpr int stage_quads = 9 * R * (stage_w / 4);
  static constexpr int loader_waves = (n1 + n2) / 64;                                         // waves below the first level-3 lane
  static constexpr int loader_iters = (stage_quads + loader_waves * 64 - 1) / (loader_waves * 64);
  static constexpr size_t lds_bytes = sizeof(float) * (2 * 9 * ring * kSW + stage_floats) + sizeof(double) * 3 * (lanes / 64);
  static constexpr int waves_per_simd = MODE == 0 ? 4 : 6;                                    // register budget: 2 / 3 blocks per CU
};

struct SweepArgs {
  const float* srck[9];        // plane bases of the source / destination grid (scalar kernel arguments)
  float* dstk[9];
  uint32_t ps;                 // plane stride in floats (MODE 2 addresses the nine planes from srck[0])
  const uint32_t* mask;        // bit per cell
  int nx, ny;
  int strips_x, nseg, seg_rows;   // blocks = strips_x * nseg; segment s owns rows [s * seg_rows, min(ny, (s + 1) * seg_rows))
  float omega, accel_w1, accel_w2;
  int accel_row;               // global row ny - 2
  int accel_last;              // accelerate_flow for the step AFTER this launch (0: the run ends here)
  double* partials_out;        // [3][nblocks]
  const double* prev_partials; // previous launch: [n_prev_vecs][n_prev]
  int n_prev, n_prev_vecs;
  double* sums;
  int* counter;
};

template <int R, bool FAST, int MODE>
__global__ void __launch_bounds__((SweepGeom<R, MODE>::lanes), (SweepGeom<R, MODE>::waves_per_simd)) lbm_sweep_kernel(const SweepArgs a)
{
  using G = SweepGeom<R, MODE>;
  [[maybe_unused]] constexpr bool INPLACE = MODE != 0;
  constexpr int N = G::ring, W = kSW, kWaves = G::lanes / 64, kPlane = N * W, SW = G::stage_w;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // ring 1 [9][N][W], ring 2 [9][N][W], stage [9][R][SW], then [3][kWaves] doubles
  float* ring1 = lds;
  float* ring2 = lds + 9 * kPlane;
  float* stage = lds + 2 * 9 * kPlane;
  double* red = reinterpret_cast<double*>(lds + 2 * 9 * kPlane + G::stage_floats);
  const int tid = threadIdx.x;

  if (blockIdx.x == 0) {
    // fold block: the previous launch's per-block sums, one vector per step, into sums[counter..]
    for (int v = 0; v < a.n_prev_vecs; ++v) {
      double s = 0.0;
      for (int i = tid; i < a.n_prev; i += G::lanes) s += a.prev_partials[static_cast<size_t>(v) * a.n_prev + i];
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

  // blocks b, b + 8, ... share an XCD: give each XCD a contiguous run of strips (neighbouring strips share 8 columns)
  const int nblocks = a.strips_x * a.nseg;
  int b = blockIdx.x - 1;
  if ((nblocks & 7) == 0) b = (b & 7) * (nblocks >> 3) + (b >> 3);
  const int seg = b / a.strips_x, strip = b - seg * a.strips_x;
  const int x0 = strip * kSTX;
  const int nx = a.nx, ny = a.ny;
  const int ys = seg * a.seg_rows, S = min(a.seg_rows, ny - ys);
  const int grid_cells = nx * ny;
  const bool edge_x = strip == 0 || strip == a.strips_x - 1;      // block-uniform: only these strips wrap in x

  // ---- this lane's role, fixed for the whole sweep: level, row within the tick's R rows, x-pair within the row — kept
  // as ONE packed register (level | r << 2 | fx << 5) and unpacked at the top of every tick behind an opaque asm
  // statement: left to itself hipcc hoists a dozen derived lane constants into registers, and MODE 2 must stay
  // within 80 VGPRs (three blocks per CU) without a single spill — a scratch reload drags a vmcnt(0) behind it, which
  // would make every direct load wait for the one before it.
  uint32_t lane_code;
  int cell, slot;
  {
    int level, r, pr;
    if (tid < G::n1) { level = 1; r = tid / (W / 2); pr = tid - r * (W / 2); }
    else if (tid < G::n1 + G::n2) { const int l = tid - G::n1; level = 2; r = l / (W / 2 - 2); pr = l - r * (W / 2 - 2); }
    else if (tid < G::n1 + G::n2 + G::n3) { const int l = tid - G::n1 - G::n2; level = 3; r = l / (W / 2 - 4); pr = l - r * (W / 2 - 4); }
    else { level = 0; r = 0; pr = 0; }
    const int fx = 2 * pr + 2 * (level > 1 ? level - 1 : 0);        // frame column of the pair: level 1 from 0, 2 from 2, 3 from 4
    lane_code = static_cast<uint32_t>(level | (r << 2) | (fx << 5));
    // row index within the level's own row sequence (level L row index i is global row ys - (3 - L) + i):
    //   level 1: i = R t + r;  level 2: i = R (t - 1) - 2 + r;  level 3: i = R (t - 2) - 4 + r   =   R t + r - (L - 1)(R + 2)
    const int i0 = r - (level > 1 ? (level - 1) * (R + 2) : 0);
    // global row of index i0, wrapped periodically (d2q9-bgk.c:245-247 with one rank); from here on only the cell index is kept
    int y = (ys - (3 - level) + i0) % ny;
    if (y < 0) y += ny;
    int gx = x0 - 4 + fx;                                           // global column of the pair's first cell (even)
    if (gx < 0) gx += nx; else if (gx >= nx) gx -= nx;               // periodic (:527-529): edge strips only
    cell = y * nx + gx;
    // LDS slots.  A level-L row of index i lives in slot i mod N of ring L (L = 1, 2).  Level 2 reads level-1 rows of
    // index i, i+1, i+2 (global rows y-1, y, y+1) and writes ring 2; level 3 reads ring-2 rows i, i+1, i+2.
    slot = i0 % N;
    if (slot < 0) slot += N;
  }
  const int ticks = (S + 4 + R - 1) / R + 2;
  const int accel_lo = a.accel_row * nx;                             // cells of global row ny-2: [accel_lo, accel_lo + nx)

  double acc = 0.0;
  // does accelerate_flow's row ny-2 come anywhere near this segment (its rows and the 3-row ring) ?  block-uniform
  bool seg_accel;
  {
    int d = (a.accel_row - (ys - 3)) % ny;
    if (d < 0) d += ny;
    seg_accel = d < S + 6 || ny < S + 6;
  }

  // ---- level-1 pulls (d2q9-bgk.c:526-538) of the pair at cell `cl` of frame column fx (MODES 0 and 1)
  auto pull = [&](int cl, int fx, f2 (&p)[9], uint32_t& mword) {
    int d_south = -nx, d_north = nx;
    if (cl < nx) d_south = grid_cells - nx;                         // periodic in y
    if (cl >= grid_cells - nx) d_north = nx - grid_cells;
    const uint32_t o_here = 4u * static_cast<uint32_t>(cl);
    const uint32_t o_south = 4u * static_cast<uint32_t>(cl + d_south);
    const uint32_t o_north = 4u * static_cast<uint32_t>(cl + d_north);
    p[0] = at_byte<f2>(a.srck[0], o_here);                          // :530
    p[2] = at_byte<f2>(a.srck[2], o_south);                         // :532
    p[4] = at_byte<f2>(a.srck[4], o_north);                         // :534
    p[1] = at_byte<f2u>(a.srck[1] - 1, o_here);                     // :531
    p[5] = at_byte<f2u>(a.srck[5] - 1, o_south);                    // :535
    p[8] = at_byte<f2u>(a.srck[8] - 1, o_north);                    // :538
    p[3] = at_byte<f2u>(a.srck[3] + 1, o_here);                     // :533
    p[6] = at_byte<f2u>(a.srck[6] + 1, o_south);                    // :536
    p[7] = at_byte<f2u>(a.srck[7] + 1, o_north);                    // :537
    if (edge_x) {
      int gx = x0 - 4 + fx;
      if (gx < 0) gx += nx; else if (gx >= nx) gx -= nx;
      if (gx == 0) {                        // x_w wraps to nx-1 (:529)
        p[1].x = at_byte<float>(a.srck[1] + nx - 1, o_here); p[5].x = at_byte<float>(a.srck[5] + nx - 1, o_south);
        p[8].x = at_byte<float>(a.srck[8] + nx - 1, o_north);
      }
      if (gx == nx - 2) {                   // x_e wraps to 0 (:527-528)
        p[3].y = at_byte<float>(a.srck[3] + 2 - nx, o_here); p[6].y = at_byte<float>(a.srck[6] + 2 - nx, o_south);
        p[7].y = at_byte<float>(a.srck[7] + 2 - nx, o_north);
      }
    }
    mword = at_byte<uint32_t>(a.mask, 4u * (static_cast<uint32_t>(cl) >> 5));
  };

  // ---- MODE 2: the staging area's quads, dealt to the loader waves.  Quad q = (population k, row rr of the tick, quad j of
  // the 80-column row): source row = (level-1 row of index R t + rr) + dy_k, columns x0 - 8 + 4 j .. + 3 (whole quads wrap)
  // One register per quad: the byte offset of (population, column) from srck[0] — a multiple of 16 — with c = rr + dy + 1
  // (0 .. R + 1) in its low four bits; the row comes from the block-uniform row of the tick being fetched.
  uint32_t q_code[G::loader_iters];
  const bool loader = MODE == 2 && (tid >> 6) < G::loader_waves;
  if constexpr (MODE == 2) {
#pragma unroll
    for (int it = 0; it < G::loader_iters; ++it) {
      const int q = tid + it * G::loader_waves * 64;
      const int k = q / (R * (SW / 4)), rem = q - k * (R * (SW / 4)), rr = rem / (SW / 4), j = rem - rr * (SW / 4);
      const int dy = (k == 2 || k == 5 || k == 6) ? -1 : (k == 4 || k == 7 || k == 8) ? 1 : 0;      // :530-538
      int gq = x0 - 8 + 4 * j;
      if (gq < 0) gq += nx; else if (gq >= nx) gq -= nx;
      q_code[it] = 4u * (static_cast<uint32_t>(k) * a.ps + static_cast<uint32_t>(gq)) + static_cast<uint32_t>(rr + dy + 1);
    }
  }
  int stage_row = (ys - 3) % ny;                                     // global row of (level-1 index R t) - 1 for the tick being fetched: scalar
  if (stage_row < 0) stage_row += ny;
  auto stage_next = [&]() {                                          // issue the direct loads of the next tick's source rows, advance the row
    if constexpr (MODE == 2) {
#pragma unroll
      for (int it = 0; it < G::loader_iters; ++it) {
        const int q = tid + it * G::loader_waves * 64;
        if (loader && q < G::stage_quads) {
          uint32_t code = q_code[it];
          asm volatile("" : "+v"(code));          // keep ONE register per quad: hipcc otherwise hoists both halves (and a 64-bit copy) out of the loop
          int yq = stage_row + static_cast<int>(code & 15u);
          if (yq >= ny) yq -= ny;
          const uint32_t off = static_cast<uint32_t>(__umul24(static_cast<uint32_t>(yq), 4u * static_cast<uint32_t>(nx))) + (code & ~15u);   // nx < 2^22
          // LDS destination: wave-uniform base + lane x 16 B — the staging area is quad-linear in q
          float* dst = stage + 4 * (it * G::loader_waves * 64 + (tid & ~63));
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(a.srck[0]) + off),
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
      stage_row += R;
      if (stage_row >= ny) stage_row -= ny;
    }
  };

  f2 pre[9];
  uint32_t pre_mask = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) pre[k] = f2{0.f, 0.f};
  if (MODE == 0 && (lane_code & 3u) == 1u) pull(cell, static_cast<int>(lane_code >> 5), pre, pre_mask);    // tick 0's rows
  if constexpr (MODE == 2) {
    // the obstacle word of every lane's cell is fetched ONE TICK AHEAD, before the direct loads of that tick are issued:
    // hipcc waits vmcnt(0) for an ordinary load's result while direct loads are in flight, so a mask load consumed inside
    // the tick would drag the next tick's staging into this tick's critical path
    pre_mask = at_byte<uint32_t>(a.mask, 4u * (static_cast<uint32_t>(cell) >> 5));
    stage_next();                                                    // tick 0's source rows
  }

#pragma unroll 1
  for (int t = 0; t < ticks; ++t) {
    uint32_t lc = lane_code;
    asm volatile("" : "+v"(lc));                                     // see lane_code
    const int level = static_cast<int>(lc & 3u), r = static_cast<int>((lc >> 2) & 7u), fx = static_cast<int>(lc >> 5);
    const int i = R * t + r - (level > 1 ? (level - 1) * (R + 2) : 0);          // row index in the level's own sequence
    const int count = S + 6 - 2 * level;                                         // rows of that sequence: S + 4, S + 2, S
    const bool active = level != 0 && i >= 0 && i < count;
    f2 p[9], out[9];
    uint32_t mword;
    const int cell_now = cell;
    if constexpr (MODE == 2) {
      // the loader waves' direct loads for this tick have landed (they issue no stores: vmcnt counts loads only) ...
      if (loader) __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0), expcnt / lgkmcnt untouched
      lds_barrier();                                                 // ... and everybody sees them, and last tick's ring rows
    }
    if (level == 1 && MODE != 2) {
      if constexpr (MODE == 1) {
        pull(cell_now, fx, p, mword);
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) p[k] = pre[k];
        mword = pre_mask;
        // the next tick's rows: R rows up, wrapped
        int cn = cell + R * nx;
        if (cn >= grid_cells) cn -= grid_cells;
        pull(cn, fx, pre, pre_mask);
      }
    } else if (level == 1) {
      // MODE 2: the staged source rows — population k's row r IS the row it is pulled from; column fx + 4 is the cell's own
      const int c = r * SW + fx + 4;
      p[0] = *reinterpret_cast<const f2*>(stage + (0 * R) * SW + c);
      p[2] = *reinterpret_cast<const f2*>(stage + (2 * R) * SW + c);
      p[4] = *reinterpret_cast<const f2*>(stage + (4 * R) * SW + c);
      p[1] = *reinterpret_cast<const f2u*>(stage + (1 * R) * SW + c - 1);
      p[5] = *reinterpret_cast<const f2u*>(stage + (5 * R) * SW + c - 1);
      p[8] = *reinterpret_cast<const f2u*>(stage + (8 * R) * SW + c - 1);
      p[3] = *reinterpret_cast<const f2u*>(stage + (3 * R) * SW + c + 1);
      p[6] = *reinterpret_cast<const f2u*>(stage + (6 * R) * SW + c + 1);
      p[7] = *reinterpret_cast<const f2u*>(stage + (7 * R) * SW + c + 1);
      mword = pre_mask;
    } else {
      // levels 2, 3: the nine pulls from the previous level's ring — rows i, i+1, i+2 of that level are global rows y-1, y, y+1
      const float* src = level == 2 ? ring1 : ring2;
      // level 2's own index i2 reads level-1 indices i2, i2+1, i2+2; its ring-2 slot is i2 mod N and the source slot is
      // ALSO i2 mod N (same index, other ring) — `slot` serves both; likewise for level 3 / ring 2
      const int s0 = slot;
      int s1 = s0 + 1; if (s1 >= N) s1 -= N;
      int s2 = s1 + 1; if (s2 >= N) s2 -= N;
      const int c_s = s0 * W + fx, c_h = s1 * W + fx, c_n = s2 * W + fx;      // south (y-1), here (y), north (y+1)
      p[0] = *reinterpret_cast<const f2*>(src + 0 * kPlane + c_h);
      p[2] = *reinterpret_cast<const f2*>(src + 2 * kPlane + c_s);
      p[4] = *reinterpret_cast<const f2*>(src + 4 * kPlane + c_n);
      p[1] = *reinterpret_cast<const f2u*>(src + 1 * kPlane + c_h - 1);
      p[5] = *reinterpret_cast<const f2u*>(src + 5 * kPlane + c_s - 1);
      p[8] = *reinterpret_cast<const f2u*>(src + 8 * kPlane + c_n - 1);
      p[3] = *reinterpret_cast<const f2u*>(src + 3 * kPlane + c_h + 1);
      p[6] = *reinterpret_cast<const f2u*>(src + 6 * kPlane + c_s + 1);
      p[7] = *reinterpret_cast<const f2u*>(src + 7 * kPlane + c_n + 1);
      if constexpr (MODE == 2) mword = pre_mask;
      else mword = at_byte<uint32_t>(a.mask, 4u * (static_cast<uint32_t>(cell_now) >> 5));
    }
    if constexpr (MODE == 2) {
      lds_barrier();                             // every lane of the tick has read: the staging area and the oldest ring rows are free
      if (t + 1 < ticks) {                       // (no direct load may be in flight when the block ends: its LDS is handed on)
        int cn = cell + R * nx;
        if (cn >= grid_cells) cn -= grid_cells;
        pre_mask = at_byte<uint32_t>(a.mask, 4u * (static_cast<uint32_t>(cn) >> 5));       // next tick's obstacle word, issued BEFORE the direct loads
        stage_next();
      }
    }
    const uint32_t mbits = (mword >> (cell_now & 31)) & 3u;
    // owned: a cell of this block's strip and segment (ring columns / rows are other blocks' cells, recomputed here)
    const int own_lo = 3 - level;                                    // level 1: i in [2, S+2); 2: [1, S+1); 3: [0, S)
    const bool owned = active && static_cast<unsigned>(fx - 4) < static_cast<unsigned>(kSTX) && i >= own_lo && i < own_lo + S;
    const bool accel_here = seg_accel && cell_now >= accel_lo && cell_now < accel_lo + nx && (level < 3 || a.accel_last);
    const double term = finish_pair<FAST>(p, mbits, a.omega, seg_accel, accel_here, a.accel_w1, a.accel_w2, owned ? mbits : 3u, out);
    acc += term;
    if constexpr (MODE == 1) lds_barrier();      // every lane of the tick has read its three source rows
    if (active) {
      if (level == 3) {
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(out[k], &at_byte<f2>(a.dstk[k], 4u * static_cast<uint32_t>(cell_now)));
      } else {
        float* dst = level == 1 ? ring1 : ring2;
        const int c = slot * W + fx;
#pragma unroll
        for (int k = 0; k < 9; ++k) *reinterpret_cast<f2*>(dst + k * kPlane + c) = out[k];
      }
    }
    // next tick: R rows further up
    cell += R * nx;
    if (cell >= grid_cells) cell -= grid_cells;
    slot += R; if (slot >= N) slot -= N;
    if constexpr (MODE != 2) lds_barrier();      // (MODE 2 meets at the top of the next tick, behind the loaders' wait)
  }
  if constexpr (MODE == 2) lds_barrier();
  const int level = static_cast<int>(lane_code & 3u);

  // per-step sums over the owned cells of this block: a lane contributes to the step of its level
  {
    const double w = wave_sum_vec4(level == 1 ? acc : 0.0, level == 2 ? acc : 0.0, level == 3 ? acc : 0.0, 0.0);
    const int q = wave_sum_slot<4>(tid & 63);
    if ((tid & 15) == 0 && q < 3) red[q * kWaves + (tid >> 6)] = w;
  }
  lds_barrier();
  if (tid < 3) {
    double t = 0.0;
    for (int w = 0; w < kWaves; ++w) t += red[tid * kWaves + w];
    a.partials_out[static_cast<size_t>(tid) * nblocks + b] = t;
  }
}

}  // namespace
