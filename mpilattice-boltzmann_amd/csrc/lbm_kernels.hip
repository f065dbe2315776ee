// lbm_kernels.hip — the translation unit of liblbm_d2q9.so's device half: it includes the hand-written
// gfx950 (MI355X / CDNA4) kernels (kernels/*.h) and implements the device entry points of the C ABI
// (include/lbm_d2q9.h) for the D2Q9-BGK timestep path of ag14774/MPILattice-Boltzmann.
//
// What one lattice step does, whichever kernel performs it (reference lines, relative to its tree):
//   pull-stream            d2q9-bgk.c:526-538      9 populations from the 3x3 neighbourhood
//   moments + equilibrium  d2q9-bgk.c:546-646
//   BGK relaxation         d2q9-bgk.c:658-666      fluid cells
//   bounce-back            d2q9-bgk.c:687-695      obstacle cells
//   sum |u|                d2q9-bgk.c:667,684      -> per-block partial, double
//   accelerate_flow        d2q9-bgk.c:442-478      fused as an EPILOGUE on global row ny-2: the row is
//                                                  written already accelerated for the next step
//   av_vels[tt-1]          d2q9-bgk.c:367          block 0 of the next launch folds the partials
//
// Kernels (all produce the same bits; lbm_run / the partitioned loops pick by grid):
//   kernels/multi.h  lbm_multi_kernel<K>                K steps per pass over HBM, 64x16 tiles, intermediate
//                                                       states in LDS — large grids and K-step row partitions
//   kernels/tile.h   lbm_tile_kernel<T,H>               up to H steps per launch, launch-latency-bound small grids
//   kernels/step.h   lbm_step_kernel / _narrow / _lds   one step per launch (4 cells or 1 cell per lane; the
//                                                       LDS-staged variant) — everything else
//   kernels/aux.h    fold, accelerate pre-pass, initial state, AoS<->SoA, halo pack/unpack, av_velocity
//
// Layout in HBM: struct-of-arrays, 9 planes of rows*nx floats (plane stride padded, see
// plane_stride_floats()), two grids (source / destination, swapped per launch like :376-378), the
// obstacle map as a bitfield (1 bit per cell); K-step row partitions carry K ghost rows per side.
// No MFMA anywhere: nothing on this path is a contraction.
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
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "lbm_d2q9.h"
#include "lbm_internal.h"
#include "kernels/common.h"
#include "kernels/step.h"
#include "kernels/tile.h"
#include "kernels/multi.h"
#ifndef LBM_EXPERIMENTS      // -DLBM_EXPERIMENTS=1 (scripts/build_variant.sh experiments): the forms that measured slower and are kept for the
#define LBM_EXPERIMENTS 0    // record — lbm_sweep_kernel (LBM_TUNE_SWEEP), lbm_step_kernel_lds (LBM_FLAG_KERNEL_LDS) — with their parity tests
#endif                       // (tests/experiments_suite.py); liblbm_d2q9.so as shipped does not carry them
#if LBM_EXPERIMENTS
#include "kernels/sweep.h"
#endif
#include "kernels/aux.h"
#include "kernels/p2p.h"

namespace {

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
  // Skew between planes for large grids (planes of 8192 x 8192 floats are 2^28 bytes: nine streams on the same channels without one).
  // 24 x 256 B = 6 KiB.  Rounds 1-2 used 34 (tuned on the K = 3 launch, 64 x 16 tiles); scanned again on the K = 4 launch on 64 x 23 tiles,
  // 8192 x 8192, us/step for skews 0 / 8 / 16 / 24 / 33 / 34 / 40 / 48 / 68: K = 4 286.7 / 284.6 / 285.8 / 284.7 / 314.3 / 307.8 & 295.3 / 289.0 /
  // 285.1 / 286.6; K = 3 (the tails) 384.6 & 329.5 / 341.1 / 324.0 / 317.9 / 339.8 / 339.7 & 348.5 / 323.5 / 335.4 / 333.7 — 33 and 34 are the
  // two bad values for the tall tiles, 24 is best for both; other sizes (4096 x 4096 ... 16384 x 4096, partitions) do not care
  // (profiles/r03/ab_skew_scan.txt, ab_skew_sizes.txt).
  if (ncells >= (1u << 20)) s += 64 * static_cast<size_t>(tune_env("LBM_TUNE_SKEW", 24));
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
  int group_max = 1;         // most launches a partitioned run makes per halo exchange (a group: their steps add up to <= ghost)
  // Tile (2-D) decomposition, lbm_create_tile: the rank owns the columns [x0, x0 + nxl) of its rows as well.  Its storage rows hold
  // ghost_x ghost columns on each side and p.nx is THEIR width (nxl + 2 * ghost_x): kernels, masks and row arithmetic all work on
  // storage rows; nx_global is the grid's.  Everywhere else ghost_x = 0, nxl = nx_global = p.nx.
  int ghost_x = 0, x0 = 0, nxl = 0, nx_global = 0;
  // Storage rows kept below / above the owned rows: `ghost` for every K-step partition — except a tile rank that owns ALL rows of the grid
  // (py = 1: a column block), which keeps none: its launches wrap in y as a whole grid's do, only columns are exchanged.  `ghost` stays the
  // number of steps between two exchanges either way.
  int ghost_rows = 0;
  int tiles_px = 1, tiles_py = 1, tile_rx = 0, tile_ry = 0;
  unsigned long long* ready_ptr[4] = {nullptr, nullptr, nullptr, nullptr};   // peer-to-peer loop: the next launch_multi says "ready for epoch ready_epoch" to the
  unsigned long long ready_epoch = 0;                       // neighbours (MultiArgs::ready) and waits for theirs; cleared by that launch
  const unsigned long long* ready_wait = nullptr;
  long long ready_timeout_ticks = 0;
  int* ready_err = nullptr;
  bool nt_stores = false;
  bool fast_avvels = false;  // LBM_FLAG_FAST_AVVELS: float sum|u| terms in lbm_multi_kernel / lbm_tile_kernel
  int multi_terms = kTermsCompensated;   // lbm_multi_kernel's form of the terms (kernels/common.h): LBM_FLAG_FAST_AVVELS / LBM_FLAG_EXACT_AVVELS
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
  float* macro_pack[2] = {nullptr, nullptr};   // K-step mode: packed outgoing / incoming messages, [dir][plane][K*nx] each
  float* send[2] = {nullptr, nullptr};
  bool release_sends = false;   // send[] point into peers' windows (one-step peer-to-peer loop)
  float* recv[2] = {nullptr, nullptr};
  double* partials[2] = {nullptr, nullptr};
  int partials_cap = 0;
  int n_part_interior = 0, n_part_boundary = 0, n_part_full = 0;
  int iters_full = 1, iters_interior = 1;
  double* fold_scratch = nullptr;   // kFoldSlices x 8 slice sums of the end-of-run fold (long partial vectors)
  double* sums = nullptr;
  int sums_cap = 0;
  double* sums_host = nullptr;   // pinned, sums_cap doubles
  int* counter = nullptr;
  bool counter_clean = false;   // the device counter is 0 (left so by the last fold of the previous run, enqueued on counter_clean_stream): a run that
  hipStream_t counter_clean_stream = nullptr;   // starts on THAT stream needs no memset; on another stream nothing would order its first fold behind the reset
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
  int multi_tiles_x = 0;
  int multi_geom = kGeomStd; // geometry of lbm_multi_kernel's launches (kernels/multi.h): standard, narrow (32-wide tiles), tall (K = 4 on 64 x 23)
  int multi_tx = kMTX;       // its tile width: 64, or 32 for partitions of one round of blocks
  bool multi_tail4 = true;   // lbm_run at K = 3: 4-step launches instead of a 1- or 2-step tail (LBM_TUNE_MULTI_TAIL4)
  int sweep_R = 0;           // > 0: lbm_run's 3-step launches are lbm_sweep_kernel<R> (streaming temporal blocking, kernels/sweep.h)
  int sweep_nseg = 0, sweep_seg_rows = 0;
  int sweep_mode = 2;        // storage form of the pipeline (kernels/sweep.h SweepGeom; LBM_TUNE_SWEEP_MODE)
  bool tile_kernel = false;  // lbm_run advances several steps per launch with lbm_tile_kernel (small grids)
  int tile_T = 16, tile_H = 8;   // its geometry: owned tile edge, ghost ring = max steps per launch
  int tile_single_max = 0;       // sub-steps with regions of at most this many cells deal one cell per lane
  int n_tiles = 0;
  float accel_w1 = 0.f, accel_w2 = 0.f;
  // lbm_set_profile: timing events around every step-kernel launch of lbm_run (pool grown on demand, reused)
  bool profile = false;
  std::vector<hipEvent_t> prof_pool;
  size_t prof_used = 0;
  struct ProfLaunch { int steps; hipEvent_t begin, end; };
  std::vector<ProfLaunch> prof_launches;
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

// Every entry point that launches or copies goes through here: the launch must happen with the context's
// device current (a caller driving several GPUs from one thread leaves another one current).
hipStream_t pick_stream(lbm_ctx* c, void* stream)
{
  (void)hipSetDevice(c->device);   // cheap when already current
  return stream ? static_cast<hipStream_t>(stream) : c->stream;
}

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
  // room for the deck's own run length from the start: growing means hipFree, which waits for the whole DEVICE — harmless for one rank per
  // device, a dead-lock until the time-out where several ranks of one process share a device and the others already wait for this one's
  // rows (seen in the randomised tile cases: a second run one step longer than the first)
  n = std::max(n, std::max(c->p.max_iters, 4096));
  drop_graphs(c);   // captured kernel arguments hold the old pointer
  if (c->sums) HIP_TRY(hipFree(c->sums));
  if (c->sums_host) HIP_TRY(hipHostFree(c->sums_host));
  c->sums = nullptr;
  c->sums_host = nullptr;
  c->sums_cap = 0;
  HIP_TRY(hipMalloc(&c->sums, sizeof(double) * static_cast<size_t>(n)));
  // pinned landing zone of lbm_run's one device-to-host copy per run (a pageable destination is staged: ~2x the time)
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->sums_host), sizeof(double) * static_cast<size_t>(n), hipHostMallocDefault));
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
#if LBM_EXPERIMENTS
  } else if (c->lds_kernel && a.quad_begin2 >= a.quad_end2) {
    if (c->nt_stores) hipLaunchKernelGGL(lbm_step_kernel_lds<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(lbm_step_kernel_lds<false>, grid, block, 0, s, a);
#endif
  } else {
    if (c->nt_stores) hipLaunchKernelGGL(lbm_step_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(lbm_step_kernel<false>, grid, block, 0, s, a);
  }
}

template <int K, int GEOM, int TERMS>
void launch_multi_kgt(int blocks, hipStream_t s, const MultiArgs& a, int part)
{
  using G = MultiGeom<K, GEOM>;
  if (part == kPartGhost) lbm_multi_kernel<K, TERMS, GEOM, kPartGhost><<<dim3(blocks + 1), dim3(G::LANES), G::lds_bytes, s>>>(a);
  else if (part == kPartReady) lbm_multi_kernel<K, TERMS, GEOM, kPartReady><<<dim3(blocks + 1), dim3(G::LANES), G::lds_bytes, s>>>(a);
  else if (part == kPartTile) lbm_multi_kernel<K, TERMS, GEOM, kPartTile><<<dim3(blocks + 1), dim3(G::LANES), G::lds_bytes, s>>>(a);
  else lbm_multi_kernel<K, TERMS, GEOM, kPartPlain><<<dim3(blocks + 1), dim3(G::LANES), G::lds_bytes, s>>>(a);
}

// Frames above the default limit of dynamic LDS (the tall geometry: 79 KB) need the limit raised — per DEVICE (a function attribute
// belongs to the device's copy of the code object): called from lbm_create on the context's device, not from the first launch of a process.
template <int K, int GEOM>
hipError_t raise_multi_lds_limit()
{
  using G = MultiGeom<K, GEOM>;
  if constexpr (G::lds_bytes > 65536) {
    const void* fns[12] = {reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsDouble, GEOM, kPartPlain>), reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsFloat, GEOM, kPartPlain>),
                          reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsCompensated, GEOM, kPartPlain>),
                          reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsDouble, GEOM, kPartGhost>), reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsFloat, GEOM, kPartGhost>),
                          reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsCompensated, GEOM, kPartGhost>),
                          reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsDouble, GEOM, kPartReady>), reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsFloat, GEOM, kPartReady>),
                          reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsCompensated, GEOM, kPartReady>),
                          reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsDouble, GEOM, kPartTile>), reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsFloat, GEOM, kPartTile>),
                          reinterpret_cast<const void*>(&lbm_multi_kernel<K, kTermsCompensated, GEOM, kPartTile>)};
    for (const void* f : fns) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(G::lds_bytes));
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

template <int K, int GEOM>
void launch_multi_kg(int blocks, hipStream_t s, const MultiArgs& a, int terms, int part)
{
  if (terms == kTermsFloat) launch_multi_kgt<K, GEOM, kTermsFloat>(blocks, s, a, part);
  else if (terms == kTermsDouble) launch_multi_kgt<K, GEOM, kTermsDouble>(blocks, s, a, part);
  else launch_multi_kgt<K, GEOM, kTermsCompensated>(blocks, s, a, part);
}

template <int GEOM>
hipError_t raise_multi_lds_limits()
{
  hipError_t e = raise_multi_lds_limit<1, GEOM>();
  if (e == hipSuccess) e = raise_multi_lds_limit<2, GEOM>();
  if (e == hipSuccess) e = raise_multi_lds_limit<3, GEOM>();
  if (e == hipSuccess) e = raise_multi_lds_limit<4, GEOM>();
  return e;
}
hipError_t raise_multi_lds_limits_for(int geom)          // every instantiation a context of this geometry may launch (K = 3 tails of the tall one: standard)
{
  if (geom == kGeomNarrow) return raise_multi_lds_limits<kGeomNarrow>();
  hipError_t e = raise_multi_lds_limits<kGeomStd>();
  if (e == hipSuccess && geom == kGeomTall) e = raise_multi_lds_limit<4, kGeomTall>();
  return e;
}

template <int K>
void launch_multi_k(int blocks, hipStream_t s, const MultiArgs& a, int terms, int geom, int part)
{
  if (geom == kGeomNarrow) launch_multi_kg<K, kGeomNarrow>(blocks, s, a, terms, part);
  else if (geom_for(K, geom) == kGeomTall) launch_multi_kg<K, geom_for(K, kGeomTall)>(blocks, s, a, terms, part);
  else launch_multi_kg<K, kGeomStd>(blocks, s, a, terms, part);
}

// Tiles of a launch that makes `k` steps on the owned rows and `ext` more rows on each side (ext > 0: a launch of a partitioned
// run that is followed by `ext` more steps before the next halo exchange): the tile height depends on k (kernels/multi.h multi_ty).
int ext_rows(const lbm_ctx* c, int ext) { return c->ghost_rows > 0 ? ext : 0; }
int multi_tile_rows(const lbm_ctx* c, int k, int ext = 0) { return (c->nyl + 2 * ext_rows(c, ext) + multi_ty(k, c->multi_geom) - 1) / multi_ty(k, c->multi_geom); }
int multi_tiles_for(const lbm_ctx* c, int k, int ext = 0) { return c->multi_tiles_x * multi_tile_rows(c, k, ext); }

// One launch of lbm_multi_kernel over the tile ranges [t0, t0+n0) and [t1, t1+n1): `ksteps` steps of the owned rows and `ext`
// ghost rows on each side (tile row 0 starts at storage row ghost - ext).
void launch_multi(lbm_ctx* c, int ksteps, int ext, bool accel_last, int t0, int n0, int t1, int n1, bool fold, hipStream_t s,
                  const MultiArgs::Rect* rects = nullptr, int nrect = 0)    // rects: the launch's tiles as rectangles of the tile grid (tile ranks) instead of t0 .. n1
{
  MultiArgs a{};
  a.src = c->grid[c->cur]; a.dst = c->grid[c->cur ^ 1];
  for (int k = 0; k < 9; ++k) { a.srck[k] = a.src + k * c->ps; a.dstk[k] = a.dst + k * c->ps; }
  a.mask = c->mask; a.ps = c->ps; a.nx = c->p.nx;
  const int ext_y = ext_rows(c, ext);                           // ghost rows this launch advances (none where the rows wrap)
  a.row_first = c->ghost_rows - ext_y; a.rows_compute = c->nyl + 2 * ext_y; a.rows_storage = c->nyl + 2 * c->ghost_rows;
  a.count_first = c->ghost_rows; a.count_end = c->ghost_rows + c->nyl;
  a.cx0 = c->ghost_x; a.cx1 = c->ghost_x + c->nxl;
  a.keep_x0 = std::max(0, (c->ghost_x - ext) & ~1); a.keep_x1 = std::min(c->p.nx, (c->ghost_x + c->nxl + ext + 1) & ~1);
  a.y_periodic = (c->self_periodic || (c->ghost > 0 && c->ghost_rows == 0)) ? 1 : 0;
  a.y0_global = c->y0 - ext_y; a.ny_global = c->p.ny;        // global row of storage row row_first
  a.tiles_x = c->multi_tiles_x;
  a.tile_begin = t0; a.tile_count = n0; a.tile_begin2 = t1; a.tile_count2 = n1;
  a.ntiles_total = multi_tiles_for(c, ksteps, ext);
  a.omega = c->p.omega; a.accel_w1 = c->accel_w1; a.accel_w2 = c->accel_w2;
  a.accel_row = c->p.ny - 2; a.accel_last = accel_last ? 1 : 0;
  a.partials_out = c->partials[c->parity];
  a.prev_partials = c->partials[c->parity ^ 1];
  a.n_prev = fold ? c->n_prev : 0; a.n_prev_vecs = (fold && c->n_prev > 0) ? c->n_prev_vecs : 0;
  a.sums = c->sums; a.counter = c->counter;
  a.ready[0] = c->ready_ptr[0]; a.ready[1] = c->ready_ptr[1]; a.ready_x[0] = c->ready_ptr[2]; a.ready_x[1] = c->ready_ptr[3];
  a.ready_epoch = c->ready_epoch;
  a.wait_ready = c->ready_epoch ? c->ready_wait : nullptr; a.timeout_ticks = c->ready_timeout_ticks; a.err = c->ready_err;
  c->ready_epoch = 0;
  int blocks = n0 + n1;
  if (nrect > 0) {
    blocks = 0;
    a.nrect = nrect; a.tile_count = 0; a.tile_count2 = 0;
    for (int i = 0; i < nrect; ++i) { a.rect[i] = rects[i]; blocks += rects[i].count; }
  }
  // measured on 8192x8192, K=2: 515 us/step with the XCD-contiguous tile order, 549 without
  a.xcd_remap = (tune_env("LBM_TUNE_MULTI_REMAP", 1) && blocks % 8 == 0 && blocks >= 64) ? 1 : 0;
  a.nblocks = blocks;
  if (c->ghost_x > 0 && tune_env("LBM_TUNE_MULTI_REMAP", 1) && blocks >= 64 && blocks % 8 != 0 && tune_env("LBM_TUNE_TILE_PAD_GRID", 1)) {
    blocks = (blocks + 7) / 8 * 8;                              // tile ranks: pad the grid (the form drops the extra blocks) and keep the XCD-contiguous order
    a.xcd_remap = 1;
  }
  // the instantiation that does exactly `ksteps` steps: the tail of a run whose step count multi_K does not
  // divide is a launch of a smaller frame, not a run-time loop bound (which cost scratch and ~10 % speed)
  // the instantiation (kernels/multi.h PART): ghost rows computed too -> the counted test; ready words to say -> the fold block carries them
  // (a rank of the tile decomposition: ghost columns in every launch)
  const int part = c->ghost_x > 0 ? kPartTile : ext > 0 ? kPartGhost : a.ready_epoch != 0ull ? kPartReady : kPartPlain;
  switch (ksteps) {                                            // <= multi_K, or 4 in the tail of a K = 3 run (lbm_run)
    case 1: launch_multi_k<1>(blocks, s, a, c->multi_terms, c->multi_geom, part); break;
    case 2: launch_multi_k<2>(blocks, s, a, c->multi_terms, c->multi_geom, part); break;
    case 3: launch_multi_k<3>(blocks, s, a, c->multi_terms, c->multi_geom, part); break;
    default: launch_multi_k<4>(blocks, s, a, c->multi_terms, c->multi_geom, part); break;
  }
}

#if LBM_EXPERIMENTS
// One launch of lbm_sweep_kernel: three steps of a whole periodic grid, strips of 64 columns swept upwards.
template <int R, int MODE>
void launch_sweep_r(lbm_ctx* c, bool accel_last, hipStream_t s)
{
  SweepArgs a{};
  const float* src = c->grid[c->cur];
  float* dst = c->grid[c->cur ^ 1];
  for (int k = 0; k < 9; ++k) { a.srck[k] = src + k * c->ps; a.dstk[k] = dst + k * c->ps; }
  a.ps = static_cast<uint32_t>(c->ps);
  a.mask = c->mask; a.nx = c->p.nx; a.ny = c->nyl;
  a.strips_x = c->p.nx / kSTX; a.nseg = c->sweep_nseg; a.seg_rows = c->sweep_seg_rows;
  a.omega = c->p.omega; a.accel_w1 = c->accel_w1; a.accel_w2 = c->accel_w2;
  a.accel_row = c->p.ny - 2; a.accel_last = accel_last ? 1 : 0;
  a.partials_out = c->partials[c->parity];
  a.prev_partials = c->partials[c->parity ^ 1];
  a.n_prev = c->n_prev; a.n_prev_vecs = c->n_prev > 0 ? c->n_prev_vecs : 0;
  a.sums = c->sums; a.counter = c->counter;
  const int blocks = a.strips_x * a.nseg;
  using G = SweepGeom<R, MODE>;
  if (c->fast_avvels) lbm_sweep_kernel<R, true, MODE><<<dim3(blocks + 1), dim3(G::lanes), G::lds_bytes, s>>>(a);
  else lbm_sweep_kernel<R, false, MODE><<<dim3(blocks + 1), dim3(G::lanes), G::lds_bytes, s>>>(a);
}

template <int MODE>
void launch_sweep_m(lbm_ctx* c, bool accel_last, hipStream_t s)
{
  if (c->sweep_R == 4) launch_sweep_r<4, MODE>(c, accel_last, s);      // (R = 3 was measured too: 488 us/step at 8192 x 8192; not kept)
  else launch_sweep_r<5, MODE>(c, accel_last, s);
}

void launch_sweep(lbm_ctx* c, bool accel_last, hipStream_t s)
{
  if (c->sweep_mode == 0) launch_sweep_m<0>(c, accel_last, s);
  else if (c->sweep_mode == 1) launch_sweep_m<1>(c, accel_last, s);
  else launch_sweep_m<2>(c, accel_last, s);
}

#endif   // LBM_EXPERIMENTS

template <int T, int H>
void launch_tile(dim3 grid, hipStream_t s, const TileArgs& a, bool fast)
{
  using G = TileGeom<T, H>;
  if (fast) {
    if (a.ksteps == H) lbm_tile_kernel<T, H, true, true><<<grid, dim3(G::block), G::lds_bytes, s>>>(a);
    else lbm_tile_kernel<T, H, false, true><<<grid, dim3(G::block), G::lds_bytes, s>>>(a);
  } else {
    if (a.ksteps == H) lbm_tile_kernel<T, H, true, false><<<grid, dim3(G::block), G::lds_bytes, s>>>(a);
    else lbm_tile_kernel<T, H, false, false><<<grid, dim3(G::block), G::lds_bytes, s>>>(a);
  }
}

// Profile mode of lbm_run (lbm_set_profile): a pooled timing event recorded on `s`; nullptr when off.
hipEvent_t prof_stamp(lbm_ctx* c, hipStream_t s)
{
  if (!c->profile) return nullptr;
  if (c->prof_used == c->prof_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    c->prof_pool.push_back(e);
  }
  hipEvent_t e = c->prof_pool[c->prof_used++];
  if (hipEventRecord(e, s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}

int begin_run(lbm_ctx* c, int n_steps, hipStream_t s)
{
  if (ensure_sums(c, n_steps)) return 1;
  // the per-step sums of a run go to sums[counter++]: the counter starts at 0.  The last fold of a run leaves it there
  // (fold_last(final)); a memset launch — a kernel boundary on the critical path of a short run — only when it did not
  if (!c->counter_clean || c->counter_clean_stream != s) HIP_TRY(hipMemsetAsync(c->counter, 0, sizeof(int), s));
  c->counter_clean = false;
  c->run_steps = n_steps;
  c->run_done = 0;
  c->n_prev = 0;
  c->n_prev_vecs = 1;
  c->parity = 0;
  if (c->accel_row >= 0 && n_steps > 0) {
    // accelerate_flow of step 0 (d2q9-bgk.c:345-348); later steps get it from the kernel epilogue
    const int nx = c->p.nx;
    hipLaunchKernelGGL(lbm_accelerate_kernel, dim3((c->nxl + 255) / 256), dim3(256), 0, s, c->grid[c->cur], c->ps,
                       c->mask, nx, c->ghost_rows + c->accel_row, c->accel_w1, c->accel_w2, c->ghost_x, c->nxl);
    HIP_TRY(hipGetLastError());
  }
  c->ev_valid = false;
  c->ev_launches = 0;
  c->ev_tile_launches = 0;
  HIP_TRY(hipEventRecord(c->ev_begin, s));
  return 0;
}

// d2q9-bgk.c:367 for the LAST launch of a (macro-)step sequence: its per-block sums have no following launch to fold them.
// final: the run ends here — the fold also resets the counter for the next run.
int fold_last(lbm_ctx* c, hipStream_t s, bool final = false)
{
  if (c->n_prev == 0) return 0;          // already folded (lbm_step_fold)
  const double* part = c->partials[c->parity ^ 1];
  // one block folds 23 K partials (4 vectors of an 8192 x 1024-row rank's 5760 tiles) in 31 us — at the end of EVERY run, on the critical
  // path of a 20-step region; sliced over 64 blocks per vector and folded again it is two launches of a few us (threshold was 8192)
  if (c->n_prev >= tune_env("LBM_TUNE_FOLD_SLICED_MIN", 1024)) {
    hipLaunchKernelGGL(lbm_fold_slices_kernel, dim3(kFoldSlices, c->n_prev_vecs), dim3(kBlock), 0, s, part, c->n_prev, c->fold_scratch);
    hipLaunchKernelGGL(lbm_fold_kernel, dim3(1), dim3(kBlock), 0, s, c->fold_scratch, kFoldSlices, c->n_prev_vecs, c->sums, c->counter, final ? 1 : 0);
  } else {
    hipLaunchKernelGGL(lbm_fold_kernel, dim3(1), dim3(kBlock), 0, s, part, c->n_prev, c->n_prev_vecs, c->sums, c->counter, final ? 1 : 0);
  }
  HIP_TRY(hipGetLastError());
  c->n_prev = 0;
  c->n_prev_vecs = 1;
  c->counter_clean = final;
  c->counter_clean_stream = s;
  return 0;
}

// Wait for the stream's work: hipStreamSynchronize, optionally preceded by LBM_SPIN_WAIT_US microseconds of polling
// (hipStreamQuery).  A blocked host thread is woken some microseconds after the last kernel ends, which a 1 ms run of 20
// steps could notice; measured in one process on a 1-rank ring of 8192 x 1024 rows (scripts/ab_ring.py, 60 rounds of
// 20-step runs): 52.70 us/step blocking, 52.63 polling — no difference, so the default is 0 (no core kept spinning).
hipError_t stream_wait(hipStream_t s)
{
  const int spin_us = tune_env("LBM_SPIN_WAIT_US", 0);          // read per call: scripts/ab_ring.py alternates it in one process
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e != hipErrorNotReady) return e;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
    (void)hipGetLastError();             // hipErrorNotReady is not an error
  }
  return hipStreamSynchronize(s);
}

// Steps of the next launch of lbm_multi_kernel when `left` steps remain: multi_K, except that a count multi_K does not
// divide is split into 3s and 4s where that avoids a K = 2 / K = 1 launch at the end (8192 x 8192, us per launch: K = 1 870,
// K = 2 1000, K = 3 1050, K = 4 1290) — at K = 3: n = 3a + 4 or 3a + 8; at K = 4: n = 4a + 3, 4a + 6 or 4a + 9.  Whole periodic
// grids (the frame wraps) and row partitions that keep four ghost rows.  A function of (K, ghost, left) only, so every
// rank of a partitioned run makes the same sequence of macro-steps.
int next_multi_k(const lbm_ctx* c, int left)
{
  return lbm_plan_next(c->multi_K, (c->self_periodic || c->ghost >= 4) ? 1 : 0, c->multi_tail4 ? 1 : 0, left);
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

// Is a row partition of `rows` rows eligible for K-step mode (lbm_multi_kernel with ghost rows) ?
static bool macro_eligible(const lbm_params* p, int rows, unsigned flags)
{
  // the multi kernel addresses a plane with 32-bit byte offsets: < 2^30 storage cells
  const bool fits_u32 = static_cast<size_t>(p->nx) * (rows + 2 * kMaxGhost) < (size_t(1) << 30);
  return !(flags & LBM_FLAG_ONE_STEP) && rows >= 2 * kMTY && fits_u32 && p->nx < (1 << 23) && (p->nx % kMTX == 0 || (p->nx % 2 == 0 && p->nx >= 2 * kMTX));
}

// K for partitions of at most `max_cells` cells.  Measured on a 1-rank ring with the packed exchange,
// us/step for K = 2 / 3 / 4 (one-step loop):
//   8192x4096 rows 243 / 182 / 199   8192x2048 rows 122 / 92.6 / 103   8192x1024 rows 66.2 / 52.5 / 55.9 (116)
//   1024x128 rows 26.1 / 18.8 / 14.7 (37)
// Round 3: with the 4-step launch on 64 x 13 tiles (three blocks per CU, kernels/multi.h) K = 4 wins at every size — 1-rank
// p2p rings, us/step for K = 3 / K = 4: 8192x4096 178.9 / 166.2, 8192x1024 51.3 / 48.5, 1024x128 4.40 / 4.04.
static int macro_k_for(size_t max_cells)
{
  (void)max_cells;
  return std::min(std::max(tune_env("LBM_TUNE_MACRO_K", 4), 0), kMaxMultiSteps);
}

// Ghost rows kept on each side of a K-step partition, and with them how often it exchanges: the launches between two exchanges (a
// GROUP) make at most `ghost` steps together.  Round 4: the first launch of a group also advances `ext` = (steps of the later ones)
// ghost rows on each side from the exchanged rows, so the later ones are launches over all tiles that read no exchanged row: no
// interior / edge split, no push, no wait, no join.  Rounds 1-3 kept K rows (4 at K = 3) and exchanged before every launch.
//   partitions that run the edge-stream schedule (>= 2 M cells): 2 K rows (8 at K = 3: 3 + 4, 4 + 4, 3 + 3), two launches per exchange —
//     1-rank ring of 8192 x 1024 rows, us/step at 20 / 200 steps per run for K, 8, 12, 16 rows: 48.1 / 45.7, 46.8 / 43.8, 46.5 / 43.7, 47.2 / 43.9
//     (profiles/r04/rings_p2p_final_build.txt: past two launches per exchange nothing more is gained, so the fewest ghost rows stay);
//   smaller ones (everything on one stream: each exchange is an exposed push + wait): as deep as their rows carry — 16 rows (four
//     launches per exchange) from 128 rows per rank, 8 from 64, K below (a 32-row rank would compute 56 rows in a group's first launch) —
//     1024 x 128 rows: 4.85 (K rows), 4.23 (8), 3.85 (12), 3.84 (16) us/step at 200 steps, 6.40 / 6.12 / 5.87 / 5.65 at 20
//     (profiles/r04/rings_p2p_small.txt; with the neighbours' "ready" awaited inside the push kernel, by every block or by block 0 with a
//     go word for the rest, 8 rows were no faster than K: 4.95 - 5.76 — the wait now sits in the fold block of the group's last launch).
// One answer for all ranks: from nx and the smallest / largest row count of the run.  LBM_TUNE_MACRO_GHOST overrides (0 or anything
// below K: K rows, one launch per exchange).  The exchange moves the rows the NEXT group needs (peer-to-peer loop) or all `ghost`
// rows (RCCL loop).
static int macro_ghost_for(int k, int nx, int rows_min, int rows_max, bool row_blocks = true)
{
  if (k <= 0) return 0;
  const int classic = k == 3 ? 4 : k, two = k == 3 ? 8 : 2 * k;
  int by_size = two;
  if (static_cast<size_t>(nx) * rows_max < (size_t(1) << 21)) by_size = rows_min >= 128 ? std::max(16 / k * k, two) : rows_min >= 64 ? two : classic;
  // ... and deeper still for the smallest ranks (round 4, last: kMaxGhost 16 -> 32), whose launches are bound by latency, not by the rows they
  // compute: us/step for 16 / 24 / 32 ghost rows — 1024 x 128 rows 3.30 / 3.14 / 3.10, 1024 x 256 3.97 / 3.82 / 3.76, 512 x 512 3.86 / 3.70 / 3.63,
  // 2048 x 256 5.50 / 5.40 / 5.33; not for wider or larger ones: 4096 x 128 6.03 / 6.18 / 6.22, 8192 x 128 10.2 / 10.5 / 10.6, 2048 x 512 7.75 / 7.70 / 7.97,
  // 1024 x 1024 7.4 / 7.3 / 7.4, and 1024 x 192 3.40 / 3.43 / 3.55 (profiles/r04/ab_row_block_ghost_depth.txt): 24 rows from 128 rows per rank, 32 from 256,
  // for ranks of at most 2^19 cells in rows of at most 2048 cells
  if (row_blocks && nx <= 2048 && static_cast<size_t>(nx) * rows_max <= (size_t(1) << 19) && rows_min >= 128)     // (tile ranks: lbm_tile_layout_of has its own rule)
    by_size = std::max(by_size, std::min((rows_min >= 256 ? 32 : 24) / k * k, static_cast<int>(kMaxGhost)));
  return std::min(std::max(tune_env("LBM_TUNE_MACRO_GHOST", by_size), k), kMaxGhost);
}

// Most launches per exchange: what the ghost rows allow (LBM_TUNE_MACRO_GROUP caps it; 1 = rounds 1-3's loop on any number of ghost rows).
static int macro_group_for(int k, int ghost)
{
  if (k <= 0) return 1;
  return std::min(std::max(tune_env("LBM_TUNE_MACRO_GROUP", std::max(ghost / k, 1)), 1), kMaxGroup);
}

// Geometry of lbm_multi_kernel's launches by partition size.  Width: 64 x 16 tiles for the bandwidth-bound grids; 32 x 16 where 64 x 16
// tiles would not even fill the chip once (256 CUs x 3 blocks), so that the launch is bound by one block's chain of
// sub-steps: half the work per block, twice the blocks.  Measured us/step for 64 / 32 wide tiles (K = 3, one GPU):
// 1024x128 3.22 / 2.53, 512x256 3.17 / 2.50, 512x512 3.42 / 3.44, 2048x256 4.94 / 5.05, 1024x1024 8.39 / 9.01.
// LBM_TUNE_MULTI_TILE = 64 / 32 overrides the width; LBM_TUNE_MULTI_GEOM = 0 / 1 / 2 the whole choice.
// The tall geometry (K = 4 on 64 x 23 tiles, 768-lane blocks, two per CU) from 2^20 cells up: where a launch is several rounds of
// blocks it is 4 - 10 % faster, at one round or less its 512 slots lose to 768 (kernels/multi.h).
// Row partitions (interior + edge launch per macro-step) follow the same rule.  Measured one ring per PROCESS, as ranks run (two rings in
// one process can share a hardware queue, which made the tall geometry look 20 % worse in a same-process A/B): standard / tall, us/step
// at 200 and 20 steps per run: 8192 x 1024 rows 46.5 / 45.5 and 48.2 / 48.1, 8192 x 2048 rows 86.9 / 82.4 and 88.9 / 84.2
// (profiles/r03/ab_fused_schedule.txt).
static int pick_geom(size_t ncells)
{
  const int by_size = ncells <= static_cast<size_t>(tune_env("LBM_TUNE_NARROW_TILE_MAX", 1 << 17)) ? kMTXNarrow : kMTX;
  const int t = tune_env("LBM_TUNE_MULTI_TILE", by_size);
  int g = t == kMTXNarrow ? kGeomNarrow : ncells >= static_cast<size_t>(tune_env("LBM_TUNE_TALL_TILE_MIN", 1 << 20)) ? kGeomTall : kGeomStd;
  const int forced = tune_env("LBM_TUNE_MULTI_GEOM", -1);
  if (forced >= kGeomStd && forced <= kGeomTall) g = forced;
  return g;
}

// Obstacle bitfield of the storage rows: bit i of the linear storage cell index, row r of the storage taken
// from row_ptr(r).  Word ranges are packed by several host threads (67 M cells at 8192x8192).
template <typename RowPtr>
static void pack_obstacle_bits(std::vector<uint32_t>& bits, int rows, int nx, RowPtr row_ptr)
{
  const size_t ncells = static_cast<size_t>(rows) * nx, nwords = (ncells + 31) / 32;
  auto pack = [&](size_t w0, size_t w1) {
    size_t i = w0 * 32;
    int r = static_cast<int>(i / nx), x = static_cast<int>(i - static_cast<size_t>(r) * nx);
    const int* row = r < rows ? row_ptr(r) : nullptr;
    for (size_t w = w0; w < w1; ++w) {
      uint32_t v = 0;
      for (int b = 0; b < 32 && i < ncells; ++b, ++i) {
        if (row[x]) v |= 1u << b;
        if (++x == nx) { x = 0; ++r; row = r < rows ? row_ptr(r) : nullptr; }
      }
      bits[w] = v;
    }
  };
  unsigned workers = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (nwords < (1u << 16)) workers = 1;
  if (workers == 1) { pack(0, nwords); return; }
  std::vector<std::thread> pool;
  const size_t per = (nwords + workers - 1) / workers;
  for (unsigned t = 0; t < workers; ++t) {
    const size_t w0 = std::min(nwords, t * per), w1 = std::min(nwords, w0 + per);
    if (w0 < w1) pool.emplace_back(pack, w0, w1);
  }
  for (std::thread& t : pool) t.join();
}

// lbm_create / lbm_create_global / lbm_create_rank.  The obstacle flags of the ghost rows that the K-step
// kernels need come either from obstacles_global (ny*nx, lbm_create_global) or from obstacles_window
// ((ny_local + 2*forced_k)*nx: the rows around the partition only, lbm_create_rank); both null = no ghost
// rows possible.  forced_k < 0: K-step mode and K decided from this partition's own shape
// (lbm_create_global); >= 0: decided by the caller for the whole run (lbm_rank_layout).
struct TileSpec { int px, py, rx, ry, x0, nxl, ghost_x, nx_global, ghost_rows; };    // lbm_create_tile: `p->nx` is then the storage row width nxl + 2 ghost_x

static int create_impl(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacles_rows,
                       const int* obstacles_global, const int* obstacles_window, int forced_k, int forced_ghost, int y0, int ny_local,
                       int device, unsigned flags, const TileSpec* tile = nullptr)
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
  c->nxl = c->nx_global = p->nx;
  if (tile) {
    c->ghost_x = tile->ghost_x; c->x0 = tile->x0; c->nxl = tile->nxl; c->nx_global = tile->nx_global;
    c->tiles_px = tile->px; c->tiles_py = tile->py; c->tile_rx = tile->rx; c->tile_ry = tile->ry;
  }
  c->self_periodic = self_periodic;
  c->fast_avvels = (flags & LBM_FLAG_FAST_AVVELS) != 0;
  c->multi_terms = c->fast_avvels ? kTermsFloat : (flags & LBM_FLAG_EXACT_AVVELS) ? kTermsDouble : kTermsCompensated;
  {
    const int t = tune_env("LBM_TUNE_TERMS", -1);            // 0 double, 1 float, 2 compensated (A/B runs of the DEFAULT form:
    if (t >= 0 && t <= 2 && !(flags & (LBM_FLAG_EXACT_AVVELS | LBM_FLAG_FAST_AVVELS))) c->multi_terms = t;   // a form asked for by flag stays)
  }
  c->accel_row = accel_row;
  c->accel_w1 = p->density * p->accel * 0.111111111111111111111111f;        // d2q9-bgk.c:445
  c->accel_w2 = p->density * p->accel * 0.0277777777777777777777778f;       // d2q9-bgk.c:446
  c->ncells = static_cast<size_t>(p->nx) * ny_local;
  // K-step mode of a row-partitioned run: K ghost rows on each side of the owned rows, refreshed by the
  // neighbours every K steps, all steps done by lbm_multi_kernel (lbm_macro_* calls)
  const bool fits_u32 = static_cast<size_t>(p->nx) * (ny_local + 2 * kMaxGhost) < (size_t(1) << 30);
  if (forced_k > 0) {
    if (self_periodic || !obstacles_window || forced_k > kMaxMultiSteps || forced_ghost < forced_k || forced_ghost > kMaxGhost ||
        !macro_eligible(p, ny_local, flags)) {
      lbm_internal::set_error("lbm_create_rank: partition cannot run the K-step mode its layout asks for");
      delete c;
      return 1;
    }
    c->multi_K = forced_k; c->ghost = forced_ghost;
    c->ghost_rows = (tile && !tile->ghost_rows) ? 0 : forced_ghost;
    c->group_max = macro_group_for(forced_k, forced_ghost);
  } else if (forced_k < 0 && !self_periodic && obstacles_global && macro_eligible(p, ny_local, flags)) {
    const int k = macro_k_for(c->ncells);
    if (k > 0) { c->multi_K = k; c->ghost = c->ghost_rows = macro_ghost_for(k, p->nx, ny_local, ny_local); c->group_max = macro_group_for(k, c->ghost); }
  }
  c->multi_tail4 = tune_env("LBM_TUNE_MULTI_TAIL4", 1) != 0;
  c->ncells_storage = static_cast<size_t>(p->nx) * (ny_local + 2 * c->ghost_rows);
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
#if LBM_EXPERIMENTS
  c->lds_kernel = (flags & LBM_FLAG_KERNEL_LDS) != 0 && c->lane_cells == kCellsPerLane;
#else
  if (flags & LBM_FLAG_KERNEL_LDS) {
    lbm_internal::set_error("lbm_create: LBM_FLAG_KERNEL_LDS needs a library built with -DLBM_EXPERIMENTS=1 (the LDS-staged one-step kernel is never faster and is not shipped)");
    delete c;
    return 1;
  }
#endif
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
  for (int g = 0; g < 2; ++g) {
    HIP_TRY_C(hipMalloc(&c->grid_alloc[g], sizeof(float) * c->grid_floats));
    HIP_TRY_C(hipMemsetAsync(c->grid_alloc[g], 0, sizeof(float) * c->grid_floats, c->stream));
    c->grid[g] = c->grid_alloc[g] + 64;
  }
  if (tune_env("LBM_DEBUG_ADDR", 0))      // placement experiments (scripts/experiments/alloc_order.py)
    std::fprintf(stderr, "lbm_create: grids at %p %p (%zu bytes each, plane stride %zu floats)\n", static_cast<void*>(c->grid_alloc[0]),
                 static_cast<void*>(c->grid_alloc[1]), sizeof(float) * c->grid_floats, c->ps);
  // obstacle bitfield
  const size_t mwords = (c->ncells_storage + 31) / 32 + 4;
  std::vector<uint32_t> bits(mwords, 0u);
  {
    const int nx = p->nx, ny = p->ny, ghost = c->ghost_rows;
    const int rows = ny_local + 2 * ghost;
    if (ghost == 0) pack_obstacle_bits(bits, rows, nx, [&](int r) { return obstacles_rows + static_cast<size_t>(r) * nx; });
    else if (obstacles_window) pack_obstacle_bits(bits, rows, nx, [&](int r) { return obstacles_window + static_cast<size_t>(r) * nx; });
    else pack_obstacle_bits(bits, rows, nx, [&](int r) {          // storage row -> global row, periodic
      int g = (y0 + r - ghost) % ny;
      if (g < 0) g += ny;
      return obstacles_global + static_cast<size_t>(g) * nx;
    });
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
    // geometry by size, measured on MI355X (us/step for <8,4> / <8,8> / <16,4> / <16,8>; one-step kernels
    // 3.4 / 3.5 / 4.0):  128x128 1.7 / 1.4 / 1.8 / 1.6 | 128x256 2.2 / 1.8 / 1.9 / 1.7 | 256x256 3.4 / 3.0 / 2.2 / 1.9
    const int by_size = c->ncells <= 16384 ? 88 : 168;
    const int geom = tune_env("LBM_TUNE_TILE_GEOM", by_size);   // T*10 + H
    c->tile_T = geom / 10; c->tile_H = geom % 10;
    if (!((c->tile_T == 16 || c->tile_T == 8) && (c->tile_H == 8 || c->tile_H == 4))) { c->tile_T = 16; c->tile_H = 4; }
  }
  // measured whole-deck times (s) for 0 / 256 / 512: 128x128 0.0648 / 0.0576 / 0.0565, 128x256 0.0746 / 0.0747 / 0.0712,
  // 256x256 0.1598 / 0.1570 / 0.1532
  c->tile_single_max = tune_env("LBM_TUNE_TILE_SINGLE_MAX", 512);
  c->n_tiles = (p->nx % c->tile_T == 0 && ny_local % c->tile_T == 0) ? (p->nx / c->tile_T) * (ny_local / c->tile_T) : 0;
  c->tile_kernel = self_periodic && c->n_tiles > 0 &&
                   c->ncells <= static_cast<size_t>(tune_env("LBM_TUNE_TILE_MAX", 131072));  // us/step here vs lbm_multi_kernel<3>: 256x256 1.9 / 3.1, 512x256 2.8 / 3.2, 384x384 3.6 / 3.2, 512x512 4.5 / 3.3
  if (c->ghost > 0) {
    const size_t pack_floats = static_cast<size_t>(2) * 9 * std::max(c->ghost_rows, 1) * p->nx;
    for (int i = 0; i < 2; ++i) HIP_TRY_C(hipMalloc(&c->macro_pack[i], sizeof(float) * pack_floats));
    c->tile_kernel = false;
    c->multi_geom = pick_geom(c->ncells);
    if (tile && c->ghost_rows == 0 && c->multi_geom == kGeomTall && tune_env("LBM_TUNE_MULTI_GEOM", -1) < 0) {
      // A column block's launches all cover exactly its ny rows: where 23-row tiles fit them badly the last tile row is mostly waste — 256 rows: 12
      // tile rows cover 276 (7.8 % over) against 260 on 13-row tiles; 128 rows: 138 against 130 — and the standard geometry wins by 10 % (us/step
      // tall / standard: 4096 x 256 8.96 / 8.11, 8192 x 128 9.40 / 8.37; 512 and 1024 rows fit: 2048 x 512 7.82 / 8.25, 1024 x 1024 8.20 / 8.60, 2048 x 1024
      // 13.8 / 14.1; profiles/r04/ab_column_block_geometry.txt).  Row blocks and whole grids compute different row counts from launch to launch (no rule).
      auto over = [&](int ty) { return static_cast<double>((ny_local + ty - 1) / ty * ty) / ny_local; };
      if (over(kMTY4Tall) - over(kMTY4) > 0.03) c->multi_geom = kGeomStd;
    }
    c->multi_tx = geom_tx(c->multi_geom);
    HIP_TRY_C(raise_multi_lds_limits_for(c->multi_geom));
    c->multi_tiles_x = (p->nx + c->multi_tx - 1) / c->multi_tx;
    c->partials_cap = std::max(c->partials_cap, kMaxMultiSteps * c->multi_tiles_x * ((ny_local + 2 * c->ghost_rows + kMinMultiTY - 1) / kMinMultiTY) + 1);
  } else if (!c->tile_kernel && self_periodic && fits_u32 && p->nx < (1 << 23) &&      // (24-bit row multiplies in lbm_multi_kernel)
             ((p->nx % kMTX == 0 && ny_local % kMTY == 0) || (p->nx % 2 == 0 && p->nx >= 2 * kMTX && ny_local >= 2 * kMTY))) {
    // grids tiled exactly by 64x16, or any even nx >= 128 with ny >= 32, where the last tile column / row
    // sticks out of the grid (periodic images: computed, not kept)
    // K steps per pass over HBM (lbm_multi_kernel), measured us/step for K = 2 / 3 / 4 (one-step kernel):
    //   8192x8192 500 / 360 / 405 (853-917)   2048x2048 33.5 / 25.5 / 27.4 (59)
    //   1024x1024 11.2 / 8.3 / 8.5 (13.5)   512x512 3.7 / 3.4 / 3.3 (6.3; lbm_tile_kernel 5.2)
    // K = 2 is HBM-bound, K = 4 instruction-bound at 2 blocks per CU (60 KB frames); K = 3 sits at both limits
    // round 3, 4-step launch on 64 x 13 tiles: K = 3 / K = 4 8192x8192 346.6 / 324.0, 4096x4096 87.2 / 78.4, 2048x2048 24.0 / 21.6,
    // 1536x1536 14.2 / 13.8, 1024x1024 8.16 / 7.07, 768x768 5.39 / 4.63, 1024x512 4.51 / 4.72, 512x1024 4.39 / 4.62, 640x640 4.01 / 4.13,
    // 512x512 3.11 / 3.45 -> K = 4 from 768 x 768 cells up (profiles/r03/ab_k3_k4.txt, ab_k3_k4_threshold.txt)
    c->multi_K = std::min(std::max(tune_env("LBM_TUNE_MULTI_K", c->ncells >= size_t(768) * 768 ? 4 : 3), 0), kMaxMultiSteps);
    c->multi_geom = pick_geom(c->ncells);
    c->multi_tx = geom_tx(c->multi_geom);
    HIP_TRY_C(raise_multi_lds_limits_for(c->multi_geom));
    c->multi_tiles_x = (p->nx + c->multi_tx - 1) / c->multi_tx;
    if (c->multi_K > 0) c->partials_cap = std::max(c->partials_cap, kMaxMultiSteps * c->multi_tiles_x * ((ny_local + kMinMultiTY - 1) / kMinMultiTY) + 1);
#if LBM_EXPERIMENTS
    // streaming form of the 3-step launch (kernels/sweep.h): strips of 64 columns, segments of rows so that the launch is
    // about one round of two blocks per CU (8192 x 8192: 128 strips x 4 segments of 2048 rows = 512 blocks).
    // LBM_TUNE_SWEEP = R (rows per tick: 4 or 5); 0 = off (the default: measured 10-20 % slower than lbm_multi_kernel<3>)
    const int want = tune_env("LBM_TUNE_SWEEP", 0);
    if (want > 0 && c->multi_K >= 3 && p->nx % kSTX == 0 && ny_local >= 64) {
      c->multi_K = 3;                                        // the sweep makes 3-step launches; tails are lbm_multi_kernel's
      const int strips = p->nx / kSTX;
      c->sweep_mode = std::min(std::max(tune_env("LBM_TUNE_SWEEP_MODE", 2), 0), 2);
      const int target_blocks = tune_env("LBM_TUNE_SWEEP_BLOCKS", c->sweep_mode == 0 ? 512 : 768);
      int nseg = std::max(1, (target_blocks + strips / 2) / strips);
      int seg_rows = (ny_local + nseg - 1) / nseg;
      seg_rows = std::max(seg_rows, 32);
      nseg = (ny_local + seg_rows - 1) / seg_rows;
      c->sweep_R = want == 4 ? 4 : 5;
      c->sweep_nseg = nseg; c->sweep_seg_rows = seg_rows;
      c->partials_cap = std::max(c->partials_cap, 3 * strips * nseg + 1);
    }
#endif   // LBM_EXPERIMENTS
  } else if (c->tile_kernel) {
    c->partials_cap = std::max(c->partials_cap, kMaxTileSteps * c->n_tiles + 1);
    // up to 74 KB of dynamic LDS per block (two 9 x R x R float buffers): above the 64 KB default limit
    {
      using G168 = TileGeom<16, 8>;
      const void* big[4] = {reinterpret_cast<const void*>(&lbm_tile_kernel<16, 8, true, false>), reinterpret_cast<const void*>(&lbm_tile_kernel<16, 8, false, false>),
                            reinterpret_cast<const void*>(&lbm_tile_kernel<16, 8, true, true>), reinterpret_cast<const void*>(&lbm_tile_kernel<16, 8, false, true>)};
      for (const void* k : big) HIP_TRY_C(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(G168::lds_bytes)));
    }
  }
  for (int i = 0; i < 2; ++i) HIP_TRY_C(hipMalloc(&c->partials[i], sizeof(double) * c->partials_cap));
  HIP_TRY_C(hipMalloc(&c->fold_scratch, sizeof(double) * kFoldSlices * 8));
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

// The owned cells of a context as the I/O entry points see them: all columns of the owned rows, or — a rank of the tile
// decomposition — the columns [ghost_x, ghost_x + nxl) of its storage rows.
static ColWindow col_window(const lbm_ctx* c) { return ColWindow{static_cast<unsigned>(c->p.nx), static_cast<unsigned>(c->ghost_x), static_cast<unsigned>(c->nxl)}; }
static size_t owned_cells(const lbm_ctx* c) { return static_cast<size_t>(c->nxl) * c->nyl; }

extern "C" {

int lbm_create(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacles_rows, int y0,
               int ny_local, int device, unsigned flags)
{
  return create_impl(out, p, free_cells, obstacles_rows, nullptr, nullptr, 0, 0, y0, ny_local, device, flags);
}

int lbm_create_global(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacles_all, int y0,
                      int ny_local, int device, unsigned flags)
{
  if (!obstacles_all || !p || y0 < 0) { lbm_internal::set_error("lbm_create_global: bad argument"); return 1; }
  return create_impl(out, p, free_cells, obstacles_all + static_cast<size_t>(y0) * p->nx, obstacles_all, nullptr, -1, 0, y0, ny_local, device, flags);
}

// One mode and one K for every rank of a run, from global quantities only (see the header).
int lbm_rank_layout(const lbm_params* p, int nranks, int rank, unsigned flags, lbm_layout* out)
{
  if (!p || !out || nranks < 1 || rank < 0 || rank >= nranks) { lbm_internal::set_error("lbm_rank_layout: bad argument"); return 1; }
  if (p->nx < 1 || p->ny < 3 || p->ny < nranks) { lbm_internal::set_error("lbm_rank_layout: grid too small for this many ranks"); return 1; }
  std::vector<int> nyl(nranks), dis(nranks);
  if (lbm_decompose(p->ny, nranks, nyl.data(), dis.data())) return 1;                 // d2q9-bgk.c:834-862
  const int lo = *std::min_element(nyl.begin(), nyl.end()), hi = *std::max_element(nyl.begin(), nyl.end());
  if (lo < 1) { lbm_internal::set_error("lbm_rank_layout: a rank would own no rows"); return 1; }
  out->y0 = dis[rank];
  out->ny_local = nyl[rank];
  out->macro_k = 0;
  const bool partitioned = nranks > 1 || (flags & LBM_FLAG_FORCE_HALO);
  if (partitioned && macro_eligible(p, lo, flags) && macro_eligible(p, hi, flags))
    out->macro_k = macro_k_for(static_cast<size_t>(p->nx) * hi);
  out->ghost = macro_ghost_for(out->macro_k, p->nx, lo, hi);
  out->group = macro_group_for(out->macro_k, out->ghost);
  return 0;
}

int lbm_create_rank(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacle_window, int nranks,
                    int rank, int device, unsigned flags)
{
  lbm_layout lay;
  if (!obstacle_window) { lbm_internal::set_error("lbm_create_rank: null argument"); return 1; }
  if (lbm_rank_layout(p, nranks, rank, flags, &lay)) return 1;
  const int* rows = obstacle_window + static_cast<size_t>(lay.ghost) * p->nx;          // the owned rows inside the window
  return create_impl(out, p, free_cells, rows, nullptr, obstacle_window, lay.macro_k, lay.ghost, lay.y0, lay.ny_local, device, flags);
}

// ---- tile (2-D) decomposition: px x py ranks, rank = ry * px + rx ------------------------------------------------------------
// Rows by the reference's rule over py (d2q9-bgk.c:834-862), columns by lbm_decompose_columns over px.  Always K-step mode: ghost rows
// as a row partition of the same cells would keep, ghost columns the same number rounded up to even (x-pairs).
int lbm_tile_layout_of(const lbm_params* p, int px, int py, int rank, unsigned flags, lbm_tile_layout* out)
{
  if (!p || !out || px < 1 || py < 1 || rank < 0 || rank >= px * py) { lbm_internal::set_error("lbm_tile_layout_of: bad argument"); return 1; }
  if (p->nx < 1 || p->ny < 3 || p->ny < py) { lbm_internal::set_error("lbm_tile_layout_of: grid too small for this many ranks"); return 1; }
  std::vector<int> nyl(py), ydis(py), nxl(px), xdis(px);
  if (lbm_decompose(p->ny, py, nyl.data(), ydis.data())) return 1;
  if (lbm_decompose_columns(p->nx, px, nxl.data(), xdis.data())) return 1;
  const int rlo = *std::min_element(nyl.begin(), nyl.end()), rhi = *std::max_element(nyl.begin(), nyl.end());
  const int clo = *std::min_element(nxl.begin(), nxl.end()), chi = *std::max_element(nxl.begin(), nxl.end());
  if (rlo < 1) { lbm_internal::set_error("lbm_tile_layout_of: a rank would own no rows"); return 1; }
  std::memset(out, 0, sizeof *out);
  out->px = px; out->py = py; out->rx = rank % px; out->ry = rank / px;
  out->x0 = xdis[out->rx]; out->nx_local = nxl[out->rx];
  out->y0 = ydis[out->ry]; out->ny_local = nyl[out->ry];
  const int k = macro_k_for(static_cast<size_t>(chi) * rhi);
  int ghost = macro_ghost_for(k, chi, rlo, rhi, /*row_blocks=*/false);
  // Column blocks (py = 1) below the edge-stream size: 32 ghost columns, eight launches per exchange.  Their ghost depth costs columns only (no
  // launch advances ghost rows), and their launches are bound by latency, not by the cells they compute: us/step for 16 / 24 / 32 ghost columns
  // 2048 x 512 8.05 / 8.00 / 7.85, 4096 x 256 8.40 / 8.21 / 8.18, 8192 x 128 8.77 / 8.58 / 8.46, 512 x 512 4.06 / 3.96 / 3.87, 256 x 512 3.32 / 3.20 / 3.14
  // (profiles/r04/ab_column_block_ghost_depth.txt).  LBM_TUNE_MACRO_GHOST still overrides.
  if (k > 0 && py == 1 && clo >= 256 && !tune_env("LBM_TUNE_TILE_GHOST_ROWS", 0) && static_cast<size_t>(chi) * rhi < (size_t(1) << 21) && tune_env("LBM_TUNE_MACRO_GHOST", -1) < 0)   // (blocks of >= 256 columns: the measured range)
    ghost = std::max(ghost, std::min(32 / k * k, static_cast<int>(kMaxGhost)));
  int ghost_x = (ghost + 1) & ~1;
  ghost_x = std::min(std::max(tune_env("LBM_TUNE_TILE_GHOST_X", ghost_x) & ~1, ghost_x), kMaxGhost);
  // every rank's storage rows (owned + ghost columns) must be ones the K-step kernels take, and its own columns at least the ghost
  // columns its neighbours need from it
  lbm_params narrow = *p, wide = *p;
  narrow.nx = clo + 2 * ghost_x; wide.nx = chi + 2 * ghost_x;
  if (k <= 0 || !macro_eligible(&narrow, rlo, flags) || !macro_eligible(&wide, rhi, flags) || !macro_eligible(&narrow, rhi, flags) || !macro_eligible(&wide, rlo, flags) ||
      clo < ghost_x || (py > 1 && rlo < ghost)) {
    lbm_internal::set_error("lbm_tile_layout_of: the tile decomposition runs in K-step mode only: every rank needs >= 32 rows, an even number of columns with "
                            "at least 128 storage columns (owned + ghost) and LBM_FLAG_ONE_STEP clear — use the row decomposition (lbm_rank_layout)");
    return 1;
  }
  out->macro_k = k; out->ghost = ghost; out->ghost_x = ghost_x;
  out->group = macro_group_for(k, ghost);
  // column blocks (py = 1: every rank owns all rows) keep no ghost rows: their launches wrap in y like a whole grid's, and an exchange is the
  // column push alone.  LBM_TUNE_TILE_GHOST_ROWS=1 keeps them (a 1 x 1 ring then stands for a block of ANY tiling: the rank is its own south
  // and north neighbour through the row push, as it is its own west and east one)
  out->ghost_y = (py == 1 && !tune_env("LBM_TUNE_TILE_GHOST_ROWS", 0)) ? 0 : ghost;
  return 0;
}

// Row blocks or tiles, and which tiles, for `nranks` ranks: the decomposition whose ranks recompute the smallest share of cells they do not own.
// A rank of R rows and C columns that keeps g ghost rows / columns advances, averaged over a group of launches, about 3/8 g ghost rows per
// side (the first launch of a group g - k of them, the last none) and — tiles — all 2 g ghost columns in every launch:
//     row blocks     0.75 g / R              + 0.1 below 128 rows (an exchange every 8 steps), + 0.2 below 64 (every 4), + 1 below 32 (one-step loop)
//     column blocks  2 g / C + 0.02          (px x 1 tilings: no ghost rows, one exchange kernel; the margin: 128-row blocks against 320 / 384 /
//                                            448 / 512-column blocks came out 4.58 / 5.06, 5.39 / 5.87, 5.80 / 5.61, 4.36 / 4.01 us/step)
//     other tiles    0.75 g / R + 2 g / C + 0.05 (the second exchange kernel), thin blocks charged as thin row blocks are
// The rule orders the 25 pairs measured on 1-rank rings as they came out (profiles/r04/{wide,tile,auto,column}_*.json), us/step
// rows / tiles of the same cells: 8192 x 1024 43.8 / 46.0 as 1024 x 8192 column blocks and 45.5 as 2048 x 4096; 1024 x 128 3.28 / 4.05 as 128 x 1024,
// 4.13 as 256 x 512; 1024 x 256 3.99 / 4.93; 2048 x 256 5.5 / 6.2; 1024 x 64 3.25 / 3.51 as 512 x 128 (rows: a second exchange kernel and ghost
// columns cost more than the ghost rows of blocks this tall, or than a small block's frequent exchanges) — and 2048 x 128 4.36 / 4.01 as 512 x 512
// column blocks, 4096 x 128 6.15 / 5.37, 2048 x 64 4.12 / 3.31, 4096 x 64 5.38 / 4.03, 8192 x 64 7.45 / 5.64, 16384 x 64 13.2 / 8.7, 32768 x 32 18.3 / 9.4,
// 65536 x 16 21.8 / 9.5 (tiles: wide column blocks beat row blocks of up to 128 rows).  Every BASELINE.json config comes out as row blocks.
// A function of p, nranks and flags only.  *px == 1 means ROW BLOCKS (lbm_create_rank: no ghost columns), anything else lbm_create_tile on *px x *py.
int lbm_choose_rank_grid(const lbm_params* p, int nranks, unsigned flags, int* px, int* py)
{
  if (!p || !px || !py || nranks < 1) { lbm_internal::set_error("lbm_choose_rank_grid: bad argument"); return 1; }
  *px = 1; *py = nranks;
  lbm_layout rows;
  if (lbm_rank_layout(p, nranks, nranks - 1, flags, &rows)) return 1;
  if (nranks == 1) return 0;
  std::vector<int> nyl(nranks), dis(nranks);
  if (lbm_decompose(p->ny, nranks, nyl.data(), dis.data())) return 1;
  const int rmin = *std::min_element(nyl.begin(), nyl.end());
  auto thin = [](int r) { return (r < 128 ? 0.1 : 0.0) + (r < 64 ? 0.2 : 0.0); };
  double best = rows.macro_k > 0 ? 0.75 * rows.ghost / rmin + thin(rmin) : 1.0 + thin(rmin);
  for (int qx = 2; qx <= nranks; ++qx) {
    if (nranks % qx != 0) continue;
    lbm_tile_layout t;
    if (lbm_tile_layout_of(p, qx, nranks / qx, nranks - 1, flags, &t)) continue;      // a rank would fall out of K-step mode: not a candidate
    // (the last rank holds the smallest column block and, by the reference's rule, not the largest row block)
    // (a column block's 32 ghost columns count as 16 here: the rule was measured at 16, and the deeper halo only made the column blocks faster)
    const double cost = t.ghost_y > 0 ? 0.75 * t.ghost_y / t.ny_local + 2.0 * t.ghost_x / t.nx_local + 0.05 + thin(t.ny_local)
                                      : 2.0 * std::min(t.ghost_x, 16) / t.nx_local + 0.02;
    if (cost < best) { best = cost; *px = qx; *py = nranks / qx; }
  }
  (void)lbm_last_error();
  return 0;
}

int lbm_create_tile(lbm_ctx** out, const lbm_params* p, int free_cells, const int* obstacle_window, int px, int py, int rank, int device, unsigned flags)
{
  lbm_tile_layout lay;
  if (!obstacle_window) { lbm_internal::set_error("lbm_create_tile: null argument"); return 1; }
  if (lbm_tile_layout_of(p, px, py, rank, flags, &lay)) return 1;
  lbm_params local = *p;
  local.nx = lay.nx_local + 2 * lay.ghost_x;                                           // the storage row: what every kernel works on
  const TileSpec tile{px, py, lay.rx, lay.ry, lay.x0, lay.nx_local, lay.ghost_x, p->nx, lay.ghost_y};
  const int* rows = obstacle_window + static_cast<size_t>(lay.ghost_y) * local.nx;    // the owned rows inside the window
  return create_impl(out, &local, free_cells, rows, nullptr, obstacle_window, lay.macro_k, lay.ghost, lay.y0, lay.ny_local, device,
                     flags | LBM_FLAG_FORCE_HALO, &tile);
}

int lbm_tile_info(const lbm_ctx* c, lbm_tile_layout* out)
{
  if (!c || !out) { lbm_internal::set_error("lbm_tile_info: null argument"); return 1; }
  std::memset(out, 0, sizeof *out);
  out->px = c->tiles_px; out->py = c->tiles_py; out->rx = c->tile_rx; out->ry = c->tile_ry;
  out->x0 = c->x0; out->nx_local = c->nxl; out->y0 = c->y0; out->ny_local = c->nyl;
  out->macro_k = c->ghost > 0 ? c->multi_K : 0; out->ghost = c->ghost; out->ghost_x = c->ghost_x; out->group = c->group_max;
  out->ghost_y = c->ghost_rows;
  return 0;
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
  for (float* b : c->macro_pack) if (b) (void)hipFree(b);
  for (int i = 0; i < 2; ++i) if (c->partials[i]) (void)hipFree(c->partials[i]);
  if (c->sums) (void)hipFree(c->sums);
  if (c->sums_host) (void)hipHostFree(c->sums_host);
  if (c->fold_scratch) (void)hipFree(c->fold_scratch);
  if (c->counter) (void)hipFree(c->counter);
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  if (c->ev_end) (void)hipEventDestroy(c->ev_end);
  for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
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
  c->prof_used = 0;
  c->prof_launches.clear();
  if (begin_run(c, n_steps, s)) return 1;
  int tile_launches = 0;
  const bool multi = c->multi_K > 0 && c->self_periodic;
  for (int t = 0; multi && t < n_steps;) {
    // up to multi_K steps per pass over HBM (lbm_multi_kernel).  A step count that 3 does not divide is split into 3s
    // and 4s where that avoids the K = 2 / K = 1 launch at the end (8192 x 8192, us per launch: K = 1 870, K = 2 1000,
    // K = 3 1050, K = 4 1460): n = 3a + 4 for n mod 3 = 1, n = 3a + 8 for n mod 3 = 2 (next_multi_k; row partitions
    // with four ghost rows split the same way).
    const int k = next_multi_k(c, n_steps - t);
    hipEvent_t pb = prof_stamp(c, s);
#if LBM_EXPERIMENTS
    const bool sweep = c->sweep_R > 0 && k == 3;
    if (sweep) launch_sweep(c, /*accel_last=*/t + k < n_steps, s);
    else
#endif
    launch_multi(c, k, 0, /*accel_last=*/t + k < n_steps, 0, multi_tiles_for(c, k), 0, 0, /*fold=*/true, s);
    if (c->profile) c->prof_launches.push_back({k, pb, prof_stamp(c, s)});
#if LBM_EXPERIMENTS
    c->n_prev = sweep ? (c->p.nx / kSTX) * c->sweep_nseg : multi_tiles_for(c, k);
#else
    c->n_prev = multi_tiles_for(c, k);
#endif
    c->n_prev_vecs = k;
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
    a.single_max = c->tile_single_max;
    a.omega = c->p.omega; a.accel_w1 = c->accel_w1; a.accel_w2 = c->accel_w2;
    a.accel_row = c->accel_row; a.accel_last = (t + k < n_steps) ? 1 : 0;
    a.partials_out = c->partials[c->parity];
    a.prev_partials = c->partials[c->parity ^ 1];
    a.n_prev = c->n_prev; a.n_prev_vecs = c->n_prev > 0 ? c->n_prev_vecs : 0;
    a.sums = c->sums; a.counter = c->counter;
    const dim3 grid(c->n_tiles + 1);
    hipEvent_t pb = prof_stamp(c, s);
    if (c->tile_T == 16 && c->tile_H == 8) launch_tile<16, 8>(grid, s, a, c->fast_avvels);
    else if (c->tile_T == 16) launch_tile<16, 4>(grid, s, a, c->fast_avvels);
    else if (c->tile_H == 8) launch_tile<8, 8>(grid, s, a, c->fast_avvels);
    else launch_tile<8, 4>(grid, s, a, c->fast_avvels);
    if (c->profile) c->prof_launches.push_back({k, pb, prof_stamp(c, s)});
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
      hipEvent_t pb = prof_stamp(c, s);
      full_step(c, /*accel_next=*/t + 1 < n_steps, s);
      if (c->profile) c->prof_launches.push_back({1, pb, prof_stamp(c, s)});
      t += 1;
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_end, s));
  c->ev_launches = (multi || c->tile_kernel) ? c->ev_tile_launches : n_steps;
  c->ev_valid = true;
  if (fold_last(c, s, /*final=*/true)) return 1;
  c->run_done = n_steps;
  if (av_vels) {
    const double* host = c->sums_host;
    HIP_TRY(hipMemcpyAsync(c->sums_host, c->sums, sizeof(double) * n_steps, hipMemcpyDeviceToHost, s));
    HIP_TRY(stream_wait(s));
    const double inv = static_cast<double>(c->free_cells_inv);
    for (int t = 0; t < n_steps; ++t) av_vels[t] = static_cast<float>(host[t] * inv);   // d2q9-bgk.c:367
  } else {
    HIP_TRY(stream_wait(s));
  }
  return 0;
}

int lbm_get_cells(lbm_ctx* c, float* cells_aos)
{
  if (!c || !cells_aos) { lbm_internal::set_error("lbm_get_cells: null argument"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  float* tmp = nullptr;
  const size_t n = owned_cells(c) * 9;
  HIP_TRY(hipMalloc(&tmp, sizeof(float) * n));
  const int blocks = static_cast<int>((n + 255) / 256);
  hipLaunchKernelGGL(lbm_soa_to_aos_kernel, dim3(blocks), dim3(256), 0, c->stream,
                     c->grid[c->cur] + static_cast<size_t>(c->ghost_rows) * c->p.nx, tmp, c->ps, owned_cells(c), col_window(c));
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
  const size_t n = owned_cells(c) * 9;
  HIP_TRY(hipMalloc(&tmp, sizeof(float) * n));
  hipError_t e = hipMemcpyAsync(tmp, cells_aos, sizeof(float) * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    const int blocks = static_cast<int>((n + 255) / 256);
    hipLaunchKernelGGL(lbm_aos_to_soa_kernel, dim3(blocks), dim3(256), 0, c->stream, tmp,
                       c->grid[c->cur] + static_cast<size_t>(c->ghost_rows) * c->p.nx, c->ps, owned_cells(c), col_window(c));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  HIP_TRY(e);
  return 0;
}

int lbm_get_observables(lbm_ctx* c, float* obs)
{
  if (!c || !obs) { lbm_internal::set_error("lbm_get_observables: null argument"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  // row blocks of at most 16 M cells through a 256 MB device buffer: no second copy of the state
  // (whole rows per block, so that a block of a tile rank's column window starts on a row)
  const size_t total = owned_cells(c);
  const size_t chunk = std::min<size_t>(total, std::max<size_t>(1, static_cast<size_t>(tune_env("LBM_TUNE_OBS_CHUNK_CELLS", 16 << 20)) / c->nxl) * c->nxl);
  float* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, sizeof(float) * 4 * chunk));
  hipError_t e = hipSuccess;
  const float* owned = c->grid[c->cur] + static_cast<size_t>(c->ghost_rows) * c->p.nx;
  for (size_t c0 = 0; c0 < total && e == hipSuccess; c0 += chunk) {
    const size_t n = std::min(chunk, total - c0);
    hipLaunchKernelGGL(lbm_observables_kernel, dim3(static_cast<unsigned>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                       owned + c0 / c->nxl * c->p.nx, c->ps, n, tmp, col_window(c));
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(obs + 4 * c0, tmp, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  (void)hipFree(tmp);
  HIP_TRY(e);
  return 0;
}

int lbm_state_checksum(lbm_ctx* c, int y_begin, int y_end, unsigned long long* digest)
{
  if (!c || !digest) { lbm_internal::set_error("lbm_state_checksum: null argument"); return 1; }
  if (y_begin < c->y0 || y_end > c->y0 + c->nyl || y_begin > y_end) { lbm_internal::set_error("lbm_state_checksum: rows outside the partition"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  unsigned long long* dev = nullptr;
  HIP_TRY(hipMalloc(&dev, sizeof *dev));
  const size_t nx = static_cast<size_t>(c->p.nx);
  const size_t n = static_cast<size_t>(y_end - y_begin) * c->nxl;
  const size_t c0 = (static_cast<size_t>(c->ghost_rows) + static_cast<size_t>(y_begin - c->y0)) * nx;
  hipError_t e = hipMemsetAsync(dev, 0, sizeof *dev, c->stream);
  if (e == hipSuccess && n > 0) {
    const int blocks = static_cast<int>(std::min<size_t>((n + kBlock - 1) / kBlock, 4096));
    hipLaunchKernelGGL(lbm_checksum_kernel, dim3(blocks), dim3(kBlock), 0, c->stream, c->grid[c->cur] + c0, c->ps, n,
                       static_cast<unsigned long long>(y_begin) * c->nx_global + c->x0, dev, col_window(c), static_cast<unsigned>(c->nx_global));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(digest, dev, sizeof *dev, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dev);
  HIP_TRY(e);
  return 0;
}

int lbm_av_velocity_sum(lbm_ctx* c, double* tot_u)
{
  if (!c || !tot_u) { lbm_internal::set_error("lbm_av_velocity_sum: null argument"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  const int blocks = static_cast<int>(std::min<size_t>((owned_cells(c) + kBlock - 1) / kBlock, 1024));
  double* part = nullptr;
  HIP_TRY(hipMalloc(&part, sizeof(double) * blocks));
  // owned rows only; in K-step mode they start ghost rows in: bit offset ghost*nx of the bitfield (any value:
  // nx = 130, K = 3 gives 390)
  hipLaunchKernelGGL(lbm_av_velocity_kernel, dim3(blocks), dim3(kBlock), 0, c->stream,
                     c->grid[c->cur] + static_cast<size_t>(c->ghost_rows) * c->p.nx, c->ps,
                     c->mask, static_cast<size_t>(c->ghost_rows) * c->p.nx, owned_cells(c), part, col_window(c));
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
                     c->send[0], c->send[1], c->release_sends ? 1 : 0);
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
  a.release_sends = c->release_sends ? 1 : 0;
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
    if (fold_last(c, s, /*final=*/true)) return 1;
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

// Packed form of the exchange: 2 messages per direction instead of 18.
size_t lbm_macro_pack_floats(const lbm_ctx* c) { return (c && c->ghost > 0) ? static_cast<size_t>(9) * c->ghost * c->p.nx : 0; }

void* lbm_macro_pack_ptr(lbm_ctx* c, int dir, int incoming)
{
  if (!c || c->ghost == 0 || (dir != 0 && dir != 1)) return nullptr;
  return c->macro_pack[incoming ? 1 : 0] + static_cast<size_t>(dir) * lbm_macro_pack_floats(c);
}

static int macro_pack_launch(lbm_ctx* c, bool unpack, hipStream_t s)
{
  const int nfloats = c->ghost * c->p.nx;
  const dim3 grid((nfloats / 2 + 255) / 256, 9, 2);
  // outgoing: first K owned rows (dir 0, south) and last K owned rows (dir 1, north);
  // incoming: ghost rows below (from the south, dir 0) and above (from the north, dir 1)
  const size_t row_a = unpack ? 0 : static_cast<size_t>(c->ghost);
  const size_t row_b = unpack ? static_cast<size_t>(c->ghost + c->nyl) : static_cast<size_t>(c->nyl);
  hipLaunchKernelGGL(lbm_macro_pack_kernel, grid, dim3(256), 0, s, c->grid[c->cur], c->macro_pack[unpack ? 1 : 0], c->ps, nfloats,
                     row_a, row_b, c->p.nx, unpack ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return 0;
}

int lbm_macro_pack(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_pack: not a K-step context"); return 1; }
  return macro_pack_launch(c, false, pick_stream(c, stream));
}

int lbm_macro_unpack(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_unpack: not a K-step context"); return 1; }
  return macro_pack_launch(c, true, pick_stream(c, stream));
}

int lbm_macro_prepare(lbm_ctx* c, int n_steps, void* stream)
{
  if (!c || n_steps < 0 || c->ghost == 0) { lbm_internal::set_error("lbm_macro_prepare: not a K-step context"); return 1; }
  if (c->ghost_x > 0) { lbm_internal::set_error("lbm_macro_prepare: ranks of the tile decomposition are stepped by the peer-to-peer loop (lbm_p2p_run) only"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  return begin_run(c, n_steps, pick_stream(c, stream));
}

// The launches between two halo exchanges of a partitioned run (a group): next_multi_k's launches for as long as their steps add
// up to at most the ghost rows, group_max at most.  Launch i of a group advances, besides the owned rows, ext(i) = the steps of the
// launches AFTER it in the group ghost rows on each side: what those launches read in place of exchanged rows.
struct GroupPlan {
  int n = 0, total = 0;
  int k[kMaxGroup] = {};
  int ext(int i) const { int e = 0; for (int j = i + 1; j < n; ++j) e += k[j]; return e; }
};
static GroupPlan plan_group(const lbm_ctx* c, int left)
{
  GroupPlan g;
  while (left > 0 && g.n < c->group_max && g.n < kMaxGroup) {
    const int k = next_multi_k(c, left);
    if (g.n > 0 && g.total + k > c->ghost) break;
    g.k[g.n++] = k; g.total += k; left -= k;
  }
  return g;
}
static GroupPlan macro_group(const lbm_ctx* c) { return plan_group(c, c->run_steps - c->run_done); }

int lbm_macro_next_steps(const lbm_ctx* c) { return (c && c->ghost > 0 && c->run_done < c->run_steps) ? macro_group(c).total : 0; }
int lbm_macro_next_launches(const lbm_ctx* c) { return (c && c->ghost > 0 && c->run_done < c->run_steps) ? macro_group(c).n : 0; }

// Tile rows of a launch of k steps that also advances `ext` ghost rows per side (tile row 0 starts at storage row ghost - ext): the
// first `bottom_edge_rows` and the last `top_edge_rows` tile rows read exchanged rows (edge launch, after the exchange); the
// `interior_rows` between them do not: the rows they need — their own, k below and k above — are owned rows.  (The last tile row
// may hold fewer rows than a launch makes steps: the ring of the row below then reaches the ghost rows, and the top edge is two rows.)
// (Tile ranks: the first `left_cols` and the last `right_cols` tile COLUMNS read exchanged columns as well — a tile's first sub-step reads
// 2 (k - 1) + 1 columns beyond its own on each side; the interior is then the rectangle inside all four.)
struct MacroRows { int bottom_edge_rows, interior_rows, top_edge_rows, left_cols, right_cols; };
static MacroRows macro_rows(const lbm_ctx* c, int k, int ext = 0)
{
  const int ty = multi_ty(k, c->multi_geom);
  const int ext_y = ext_rows(c, ext);
  const int first = c->ghost_rows - ext_y, rows = c->nyl + 2 * ext_y;
  const int nty = (rows + ty - 1) / ty;
  const int lo = c->ghost_rows, hi = c->ghost_rows + c->nyl;      // the owned rows [lo, hi)
  int b = 0, t = 0;
  if (c->ghost_rows > 0) {                                        // (a column block wraps in y: no tile row reads an exchanged row)
    while (b < nty && first + b * ty - k < lo) ++b;
    while (t < nty - b && std::min(first + (nty - t) * ty, first + rows) - 1 + k >= hi) ++t;
  }
  int l = 0, r = 0;
  if (c->ghost_x > 0) {
    const int tx = c->multi_tx, ntx = c->multi_tiles_x, reach = 2 * (k - 1) + 1;
    const int xlo = c->ghost_x, xhi = c->ghost_x + c->nxl;         // the owned columns [xlo, xhi)
    while (l < ntx && l * tx - reach < xlo) ++l;
    while (r < ntx - l && (ntx - r) * tx - 1 + reach >= xhi) ++r;
    if (ntx - l - r <= 0) return {nty, 0, 0, 0, 0};               // no tile column inside the rim: everything waits for the exchange
  }
  return {b, nty - b - t, t, l, r};
}

// The tiles of a tile rank's launch of k steps + ext as rectangles: the interior (at most one), or the rim around it (at most four).
static int macro_rects(const lbm_ctx* c, int k, int ext, bool interior, MultiArgs::Rect* out)
{
  const MacroRows m = macro_rows(c, k, ext);
  const int ntx = c->multi_tiles_x, mid = m.interior_rows;
  int n = 0;
  auto add = [&](int ty0, int nrows, int tx0, int cols) { if (nrows > 0 && cols > 0) out[n++] = MultiArgs::Rect{ty0, tx0, cols, nrows * cols}; };
  if (interior) {
    add(m.bottom_edge_rows, mid, m.left_cols, ntx - m.left_cols - m.right_cols);
  } else {
    add(0, m.bottom_edge_rows, 0, ntx);
    add(m.bottom_edge_rows + mid, m.top_edge_rows, 0, ntx);
    add(m.bottom_edge_rows, mid, 0, m.left_cols);
    add(m.bottom_edge_rows, mid, ntx - m.right_cols, m.right_cols);
  }
  return n;
}

// The launches of a group, called by both native loops and by the split-phase entry points below.
static void launch_group_interior(lbm_ctx* c, const GroupPlan& g, bool more_after_group, hipStream_t s)
{
  const int k = g.k[0], ext = g.ext(0);
  if (c->ghost_x > 0) {                                            // tile rank: the rectangle inside the rim
    MultiArgs::Rect rects[4];
    const int n = macro_rects(c, k, ext, true, rects);
    launch_multi(c, k, ext, /*accel_last=*/g.n > 1 || more_after_group, 0, 0, 0, 0, /*fold=*/c->n_prev > 0, s, rects, n);
    return;
  }
  const MacroRows r = macro_rows(c, k, ext);
  launch_multi(c, k, ext, /*accel_last=*/g.n > 1 || more_after_group, r.bottom_edge_rows * c->multi_tiles_x, r.interior_rows * c->multi_tiles_x, 0, 0,
               /*fold=*/c->n_prev > 0, s);
}
static void launch_group_edge(lbm_ctx* c, const GroupPlan& g, bool more_after_group, hipStream_t s)
{
  const int k = g.k[0], ext = g.ext(0);
  if (c->ghost_x > 0) {                                            // tile rank: the rim, four rectangles at most
    MultiArgs::Rect rects[4];
    const int n = macro_rects(c, k, ext, false, rects);
    launch_multi(c, k, ext, /*accel_last=*/g.n > 1 || more_after_group, 0, 0, 0, 0, /*fold=*/c->n_prev > 0, s, rects, n);
    return;
  }
  const MacroRows r = macro_rows(c, k, ext);
  launch_multi(c, k, ext, /*accel_last=*/g.n > 1 || more_after_group, 0, r.bottom_edge_rows * c->multi_tiles_x,
               (r.bottom_edge_rows + r.interior_rows) * c->multi_tiles_x, r.top_edge_rows * c->multi_tiles_x, /*fold=*/c->n_prev > 0, s);
}
static void launch_group_whole(lbm_ctx* c, const GroupPlan& g, int i, bool more_after_group, hipStream_t s)   // launch i of the group over all its tiles
{
  launch_multi(c, g.k[i], g.ext(i), /*accel_last=*/i + 1 < g.n || more_after_group, 0, multi_tiles_for(c, g.k[i], g.ext(i)), 0, 0, /*fold=*/c->n_prev > 0, s);
}
// state flip after launch i of a group (d2q9-bgk.c:376-378)
static void group_launch_done(lbm_ctx* c, const GroupPlan& g, int i, int launches)
{
  c->n_prev = multi_tiles_for(c, g.k[i], g.ext(i));
  c->n_prev_vecs = g.k[i];
  c->parity ^= 1;
  c->cur ^= 1;
  c->run_done += g.k[i];
  c->ev_tile_launches += launches;
}

int lbm_macro_interior(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_interior: not a K-step context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_macro_interior: no steps left; call lbm_macro_prepare"); return 1; }
  const GroupPlan g = macro_group(c);
  if (macro_rows(c, g.k[0], g.ext(0)).interior_rows > 0) {   // tile rows whose rings stay inside the owned rows
    launch_group_interior(c, g, c->run_done + g.total < c->run_steps, pick_stream(c, stream));
    HIP_TRY(hipGetLastError());
    c->n_prev = 0;   // folded by this launch's block 0
  }
  return 0;
}

int lbm_macro_edge(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_edge: not a K-step context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_macro_edge: no steps left; call lbm_macro_prepare"); return 1; }
  const GroupPlan g = macro_group(c);
  // whichever of the two launches of a macro-step comes first folds the previous launch's sums
  launch_group_edge(c, g, c->run_done + g.total < c->run_steps, pick_stream(c, stream));
  HIP_TRY(hipGetLastError());
  c->n_prev = 0;
  return 0;
}

int lbm_macro_all(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_all: not a K-step context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_macro_all: no steps left; call lbm_macro_prepare"); return 1; }
  const GroupPlan g = macro_group(c);
  launch_group_whole(c, g, 0, c->run_done + g.total < c->run_steps, pick_stream(c, stream));
  HIP_TRY(hipGetLastError());
  c->n_prev = 0;
  return 0;
}

int lbm_macro_finish(lbm_ctx* c, void* stream)
{
  if (!c || c->ghost == 0) { lbm_internal::set_error("lbm_macro_finish: not a K-step context"); return 1; }
  if (c->run_done >= c->run_steps) { lbm_internal::set_error("lbm_macro_finish: no steps left; call lbm_macro_prepare"); return 1; }
  hipStream_t s = pick_stream(c, stream);
  const GroupPlan g = macro_group(c);
  const bool more = c->run_done + g.total < c->run_steps;
  group_launch_done(c, g, 0, 2);
  // the rest of the group: launches over all tiles that read the ghost rows the first one advanced — no exchange in between
  for (int i = 1; i < g.n; ++i) {
    launch_group_whole(c, g, i, more, s);
    HIP_TRY(hipGetLastError());
    group_launch_done(c, g, i, 1);
  }
  if (c->run_done == c->run_steps) {
    HIP_TRY(hipEventRecord(c->ev_end, s));
    c->ev_launches = c->ev_tile_launches;
    c->ev_valid = true;
    if (fold_last(c, s, /*final=*/true)) return 1;
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

int lbm_step_fold(lbm_ctx* c, void* stream)
{
  if (!c) { lbm_internal::set_error("lbm_step_fold: null context"); return 1; }
  return fold_last(c, pick_stream(c, stream));
}

int lbm_step_collect(lbm_ctx* c, void* stream, double* tot_u_per_step, int n_steps)
{
  if (!c || !tot_u_per_step || n_steps > c->run_done) { lbm_internal::set_error("lbm_step_collect: bad argument"); return 1; }
  hipStream_t s = pick_stream(c, stream);
  HIP_TRY(hipMemcpyAsync(tot_u_per_step, c->sums, sizeof(double) * n_steps, hipMemcpyDeviceToHost, s));
  HIP_TRY(stream_wait(s));
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

int lbm_set_profile(lbm_ctx* c, int on)
{
  if (!c) { lbm_internal::set_error("lbm_set_profile: null context"); return 1; }
  c->profile = on != 0;
  if (!c->profile) c->prof_launches.clear();
  return 0;
}

int lbm_launch_profile(lbm_ctx* c, int cap, int* steps, double* us, int* n_launches)
{
  if (!c || !n_launches || cap < 0 || (cap > 0 && (!steps || !us))) { lbm_internal::set_error("lbm_launch_profile: bad argument"); return 1; }
  HIP_TRY(hipSetDevice(c->device));
  *n_launches = static_cast<int>(c->prof_launches.size());
  const int n = std::min(cap, *n_launches);
  for (int i = 0; i < n; ++i) {
    const lbm_ctx::ProfLaunch& l = c->prof_launches[i];
    float ms = 0.f;
    if (!l.begin || !l.end) { lbm_internal::set_error("lbm_launch_profile: a timing event could not be recorded"); return 1; }
    HIP_TRY(hipEventSynchronize(l.end));
    HIP_TRY(hipEventElapsedTime(&ms, l.begin, l.end));
    steps[i] = l.steps;
    us[i] = static_cast<double>(ms) * 1e3;
  }
  return 0;
}

int lbm_device(const lbm_ctx* c) { return c ? c->device : -1; }
void* lbm_stream(lbm_ctx* c) { return c ? c->stream : nullptr; }

int lbm_describe(const lbm_ctx* c, char* kernel_name, size_t len, long long* cells_per_launch, long long* state_bytes)
{
  if (!c) { lbm_internal::set_error("lbm_describe: null context"); return 1; }
  if (kernel_name && len) {
#if LBM_EXPERIMENTS
    if (c->sweep_R > 0 && c->self_periodic) std::snprintf(kernel_name, len, c->fast_avvels ? "lbm_sweep_kernel<%d, fast av_vels>" : "lbm_sweep_kernel<%d>", c->sweep_R);
    else if (c->lds_kernel && !(c->multi_K > 0 && (c->self_periodic || c->ghost > 0)) && !(c->tile_kernel && c->self_periodic) && c->lane_cells != 1)
      std::snprintf(kernel_name, len, "lbm_step_kernel_lds<%s>", c->nt_stores ? "true" : "false");
    else
#endif
    if (c->multi_K > 0 && (c->self_periodic || c->ghost > 0)) std::snprintf(kernel_name, len, c->multi_terms == kTermsFloat ? "lbm_multi_kernel<%d, fast av_vels>" : c->multi_terms == kTermsDouble ? "lbm_multi_kernel<%d, double-precision av_vels terms>" : "lbm_multi_kernel<%d>", c->multi_K);
    else if (c->tile_kernel && c->self_periodic) std::snprintf(kernel_name, len, c->fast_avvels ? "lbm_tile_kernel<%d, %d, fast av_vels>" : "lbm_tile_kernel<%d, %d>", c->tile_T, c->tile_H);
    else if (c->lane_cells == 1) std::snprintf(kernel_name, len, "lbm_step_kernel_narrow<%s>", c->nt_stores ? "true" : "false");
    else std::snprintf(kernel_name, len, "lbm_step_kernel<%s>", c->nt_stores ? "true" : "false");
  }
  if (cells_per_launch) *cells_per_launch = static_cast<long long>(c->ncells);
  if (state_bytes) *state_bytes = static_cast<long long>(2 * 9 * c->ncells * sizeof(float) + c->ncells / 8);
  return 0;
}

}  // extern "C"

#if LBM_TILE_STAMPS
extern "C" int lbm_debug_tile_stamps(unsigned long long* out16)   // diagnostic builds only
{
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tile_stamps), sizeof(unsigned long long) * 16));
  return 0;
}
#endif

#include "lbm_p2p_impl.h"   // peer-to-peer halo transport: drives the K-step launches above (include/lbm_d2q9_p2p.h)
