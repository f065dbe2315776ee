/*
 * lbm_d2q9.h — C ABI of the MI355X-native D2Q9-BGK timestep path (liblbm_d2q9.so).
 *
 * The reference (ag14774/MPILattice-Boltzmann, one C file) has no plugin/FFI surface: its hot path
 * is what main()'s loop calls between tic and toc (d2q9-bgk.c:278-398).  This header is the
 * boundary a maintainer binds instead of those calls: plain pointers and sizes, no C++ or torch
 * types.  Each entry point cites the reference lines it replaces (paths relative to the reference
 * tree).  INTEGRATION.md shows the patch to the reference's main().
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; lbm_last_error() then returns a
 *     message for the calling thread (HIP error text, or the reference's die() message for the
 *     file parsers) — the reference itself reports nothing from these calls (d2q9-bgk.c:439,477,703);
 *   - host arrays are caller-owned; device memory is library-owned (freed by lbm_destroy);
 *   - cells cross the boundary in the reference's AoS layout: t_speed = float[9] per cell
 *     (d2q9-bgk.c:95-98), row-major, x fastest, WITHOUT halo rows; obstacles as the reference's
 *     int map (d2q9-bgk.c:162), 0 = fluid, non-zero = blocked;
 *   - one context drives one partition (a block of consecutive rows, d2q9-bgk.c:834-862) on one
 *     GPU from one host thread, like one MPI rank of the reference.
 */
#ifndef LBM_D2Q9_H
#define LBM_D2Q9_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_ABI_VERSION 5
#define LBM_NSPEEDS 9               /* d2q9-bgk.c:62 */

/* Run constants as read from the parameter file: t_param (d2q9-bgk.c:79-90) minus free_cells_inv,
 * which lbm_create derives from free_cells exactly as d2q9-bgk.c:950. */
typedef struct lbm_params {
  int nx, ny;                       /* GLOBAL grid size */
  int max_iters;
  int reynolds_dim;
  float density, accel, omega;
} lbm_params;

typedef struct lbm_ctx lbm_ctx;     /* opaque: device state of one partition */

/* lbm_create flags */
#define LBM_FLAG_DEFAULT       0u
#define LBM_FLAG_NT_STORES     1u   /* force non-temporal stores of the output grid (default: auto by size) */
#define LBM_FLAG_NO_NT_STORES  2u   /* force plain stores */
#define LBM_FLAG_KERNEL_LDS    4u   /* use the LDS-staged row kernel instead of the direct-load kernel */
#define LBM_FLAG_GRAPH        16u   /* lbm_run: replay 64-step hipGraphs instead of launching every step (measured: no
                                       gain on MI355X, the small grids are bound by device-side launch latency) */
#define LBM_FLAG_ONE_STEP     32u   /* row-partitioned run: keep the one-step split-phase calls (lbm_step_*) even when
                                       the partition is eligible for K-step mode (lbm_macro_*) */
#define LBM_FLAG_FAST_AVVELS  64u   /* lbm_multi_kernel / lbm_tile_kernel form each cell's sum|u| term (d2q9-bgk.c:667) in float: populations
                                       unchanged bit for bit, av_vels equal to ~1e-7 (an ulp of the float it is); 2-5 % faster */
#define LBM_FLAG_EXACT_AVVELS 128u  /* lbm_multi_kernel forms the terms in double precision, each correctly rounded (as lbm_step_kernel and
                                       lbm_tile_kernel always do) instead of as compensated float sums of relative error ~2^-44 (its
                                       default: av_vels, a float, comes out the same; the launch runs at the socket power limit and the
                                       double-precision instructions cost 1-2.5 % of clock) */
#define LBM_FLAG_FORCE_HALO    8u   /* treat a whole-grid partition like any other rank: edge rows read the halo
                                       buffers (a 1-rank run that exchanges with itself, d2q9-bgk.c:245-247) */

int         lbm_abi_version(void);
const char* lbm_last_error(void);

/* ---- host-side setup (no GPU needed) ------------------------------------------------------- */

/* Replaces the parameter-file half of initialise(): d2q9-bgk.c:772-803.  On failure the message
 * is the reference's die() text ("could not read param file: nx", ...). */
int lbm_read_params(const char* paramfile, lbm_params* out);

/* Replaces the obstacle-file half of initialise(): d2q9-bgk.c:917-953.  obstacles = ny*nx ints
 * (zero-filled here); *free_cells = nx*ny minus the number of distinct blocked cells (:805,945-946). */
int lbm_read_obstacles(const char* obstaclefile, int nx, int ny, int* obstacles, int* free_cells);

/* Row decomposition rule of d2q9-bgk.c:834-862 (last rank keeps >= 3 rows). */
int lbm_decompose(int ny, int size, int* ny_local, int* displs);

/* How a run of n_steps iterations of d2q9-bgk.c:315-394 is cut into launches of the K-step kernels: steps[i] = the
 * steps launch (or macro-step) i makes.  K at a time; with `four_rows` (a whole periodic grid, or a partition that
 * keeps four ghost rows) a count K does not divide is split into 4s and 3s where that avoids a 1- or 2-step launch at
 * the end (K = 4: n = 4a + 3, 4a + 6, 4a + 9; K = 3: n = 3a + 4, 3a + 8).  A function of its arguments only — every rank
 * of a partitioned run plans the same sequence.  Returns the number of launches (at most `cap` entries are written),
 * or -1 on a bad argument. */
int lbm_plan_steps(int K, int four_rows, int n_steps, int* steps, int cap);
/* A row-partitioned run exchanges halo rows (d2q9-bgk.c:326-328,364) once per GROUP of launches: the launches lbm_plan_steps gives
 * (with four_rows = ghost >= 4), taken for as long as their steps add up to at most `ghost` — the rows kept on each side of the owned
 * rows (lbm_layout.ghost) — and at most `group_max` of them (lbm_layout.group).  steps[i] = the steps of launch i of the group that
 * starts when `left` steps of the run remain; returns the number of launches (>= 1 when left > 0), -1 on a bad argument.  Again a
 * function of its arguments only: every rank plans the same groups. */
int lbm_plan_group(int K, int ghost, int group_max, int left, int* steps, int cap);

/* ---- device state ---------------------------------------------------------------------------- */

/* Replaces the allocation + initial-state part of initialise() (d2q9-bgk.c:865-911) for the rows
 * [y0, y0+ny_local) of the global grid, on HIP device `device`.  obstacles_rows = this partition's
 * ny_local*nx ints.  free_cells is the GLOBAL count.  ny_local == p->ny makes a self-contained
 * periodic domain (the reference's 1-rank run, where top == bottom == self, :245-247).
 * Any nx >= 1, ny >= 3, ny_local >= 1 (>= 3 on the partition holding global row ny-2); the kernel
 * form is chosen from the shape (nx % 4 != 0 runs the one-cell-per-lane form). */
int lbm_create(lbm_ctx** ctx, const lbm_params* p, int free_cells, const int* obstacles_rows,
               int y0, int ny_local, int device, unsigned flags);
int lbm_destroy(lbm_ctx* ctx);

/* Same, given the GLOBAL obstacle map (ny*nx ints, as rank 0 of the reference holds it before the
 * Scatterv, d2q9-bgk.c:917-970).  Knowing the rows around the partition lets a row-partitioned
 * context run in K-step mode: K ghost rows on each side of the owned rows, refreshed by the
 * neighbours every K steps, all steps done K at a time by lbm_multi_kernel (lbm_macro_* below).
 * Chosen when ny_local >= 32, nx is a multiple of 64 or an even number >= 128, and LBM_FLAG_ONE_STEP is not set;
 * lbm_macro_steps() tells.  Self-contained domains (ny_local == ny) are unaffected. */
int lbm_create_global(lbm_ctx** ctx, const lbm_params* p, int free_cells, const int* obstacles_all,
                      int y0, int ny_local, int device, unsigned flags);

/* ---- one rank of a row-partitioned run: layout decided from GLOBAL quantities ------------------
 *
 * lbm_create_global picks K-step mode and K from the partition it is given; ranks of an uneven
 * decomposition (ny = 190 on 6 ranks: 32,32,32,32,31,31 rows) or ranks straddling the K = 3 / K = 4
 * size threshold would then disagree about message sizes and exchange cadence.  A multi-rank run
 * therefore asks lbm_rank_layout: it applies the reference's decomposition (d2q9-bgk.c:834-862) and
 * derives ONE mode and ONE K for all ranks from nx, ny, nranks and flags alone (every rank eligible:
 * min rows >= 32, nx a multiple of 64 or even >= 128, LBM_FLAG_ONE_STEP clear; K = 4 — four-step launches on
 * 64 x 13 tiles; LBM_TUNE_MACRO_K overrides) — the same answer on every rank by
 * construction.  macro_k == 0 means one-step mode (lbm_step_*). */
typedef struct lbm_layout {
  int y0, ny_local;                 /* rows [y0, y0+ny_local) of the global grid belong to the rank */
  int macro_k;                      /* K of K-step mode for the whole run, or 0 */
  int ghost;                        /* rows kept (and obstacle rows to supply) below and above the owned rows: 2 * macro_k (8 at macro_k = 3):
                                       the launches between two halo exchanges make at most that many steps together (LBM_TUNE_MACRO_GHOST) */
  int group;                        /* most launches per halo exchange: ghost / macro_k (LBM_TUNE_MACRO_GROUP) */
} lbm_layout;
int lbm_rank_layout(const lbm_params* p, int nranks, int rank, unsigned flags, lbm_layout* out);

/* Replaces, for rank `rank` of `nranks`, the allocation + initial state of initialise()
 * (d2q9-bgk.c:865-911) AND the receiving end of the obstacle scatter (:968-970): obstacle_window holds
 * (ny_local + 2*ghost) * nx ints — global rows y0-ghost .. y0+ny_local+ghost-1, wrapping periodically —
 * i.e. only what this rank needs; no rank but the one that parsed the file ever holds the whole map.
 * With nranks == 1 the context is a self-contained domain unless LBM_FLAG_FORCE_HALO is set. */
int lbm_create_rank(lbm_ctx** ctx, const lbm_params* p, int free_cells, const int* obstacle_window,
                    int nranks, int rank, int device, unsigned flags);

/* ---- one rank of a TILE (2-D) decomposition: px x py ranks, rank = ry * px + rx ---------------------
 *
 * The reference splits rows only (d2q9-bgk.c:834-862); its report discusses a 2-D split for grids wider than tall and never
 * built it (report.odt, "MPI Design").  Here: rows by the reference's rule over py, columns in whole x-pairs over px
 * (lbm_decompose_columns), every rank in K-step mode with `ghost` ghost rows AND `ghost_x` ghost columns around its block, refreshed
 * once per group of launches (lbm_plan_group) by the peer-to-peer loop (lbm_d2q9_p2p.h): columns from the west / east neighbours
 * first, then whole storage rows — ghost columns included, which carries the corners — from south / north.  px = 1 is allowed
 * (the rank is its own west and east neighbour); the layout fails where a rank would not be eligible for K-step mode (use the row
 * decomposition there).  The RCCL loop and the split-phase calls take row partitions only. */
typedef struct lbm_tile_layout {
  int px, py, rx, ry;               /* the rank grid and this rank's place in it */
  int x0, nx_local, y0, ny_local;   /* columns [x0, x0+nx_local) of rows [y0, y0+ny_local) belong to the rank */
  int macro_k, ghost, group;        /* as lbm_layout: `ghost` = the steps between two exchanges */
  int ghost_x;                      /* ghost columns kept on each side: ghost rounded up to even */
  int ghost_y;                      /* ghost ROWS kept on each side: ghost — or 0 for column blocks (py == 1: the rank owns every row, its
                                       launches wrap in y like a whole grid's and an exchange is the column push alone) */
} lbm_tile_layout;
int lbm_decompose_columns(int nx, int px, int* nx_local, int* displs);
int lbm_tile_layout_of(const lbm_params* p, int px, int py, int rank, unsigned flags, lbm_tile_layout* out);
/* Row blocks or tiles for `nranks` ranks: the decomposition whose ranks recompute the smallest share of cells they do not own (ghost rows
 * advanced by the first launches of a group, ghost columns in every launch; thin row blocks charged for their frequent exchanges, tiles
 * with ghost rows for their second exchange kernel) — a rule that orders every pair measured: the reference's row blocks
 * for the square decks, wide column blocks (px x 1) where row blocks would be 128 rows or thinner.  A function of p, nranks and flags
 * only.  *px == 1: row blocks (lbm_create_rank); otherwise lbm_create_tile on *px x *py. */
int lbm_choose_rank_grid(const lbm_params* p, int nranks, unsigned flags, int* px, int* py);
/* obstacle_window: (ny_local + 2*ghost_y) rows of (nx_local + 2*ghost_x) ints — global rows y0-ghost_y .., global columns x0-ghost_x ..,
 * both wrapping periodically.  lbm_get_cells / lbm_set_cells / lbm_get_observables of such a context move its ny_local x nx_local
 * block; lbm_state_checksum covers the rank's columns of the rows asked for (the digests of all ranks still add up to the grid's). */
int lbm_create_tile(lbm_ctx** ctx, const lbm_params* p, int free_cells, const int* obstacle_window,
                    int px, int py, int rank, int device, unsigned flags);
/* The block a context owns, whichever call created it (px = py = 1, ghost_x = 0 for everything but lbm_create_tile). */
int lbm_tile_info(const lbm_ctx* ctx, lbm_tile_layout* out);

/* Replaces the whole timestep loop d2q9-bgk.c:315-394 for a self-contained domain
 * (ny_local == ny): n_steps x { accelerate_flow (:442-478); timestep (:493-704); av_vels[tt]
 * (:367); swap (:376-378) }.  av_vels (host, n_steps floats, may be NULL) receives one value per
 * step.  May be called repeatedly; the state left behind is the reference's post-step state
 * (collided, NOT yet accelerated), so reading it back at any point matches the reference. */
int lbm_run(lbm_ctx* ctx, int n_steps, float* av_vels);

/* Read / overwrite the partition's cells in the reference's AoS layout (ny_local*nx*9 floats). */
int lbm_get_cells(lbm_ctx* ctx, float* cells_aos);
int lbm_set_cells(lbm_ctx* ctx, const float* cells_aos);

/* Device-side write_values() arithmetic (d2q9-bgk.c:1076-1111) for this partition's rows: obs receives
 * ny_local*nx*4 floats, per cell {u_x, u_y, u, pressure} exactly as the reference computes them for a
 * fluid cell (the obstacle override :1076-1080 is applied by the writer, which has the map).  4 floats
 * per cell cross PCIe instead of 9, and no second copy of the state is made on the device. */
int lbm_get_observables(lbm_ctx* ctx, float* obs);

/* 64-bit digest of the populations of the GLOBAL rows [y_begin, y_end) (which must belong to this partition),
 * computed on the device.  The digest is a wrap-around sum over cells of a hash of (value bits, global cell
 * index, population index): the digests of disjoint row ranges ADD UP to the digest of their union, whichever
 * context holds them.  Lets a partitioned run be compared bit for bit with a single-GPU run of the same deck
 * (or with itself on other hardware) without moving the state: 8 bytes cross PCIe. */
int lbm_state_checksum(lbm_ctx* ctx, int y_begin, int y_end, unsigned long long* digest);

/* Device-side av_velocity (d2q9-bgk.c:716-751) over this partition's rows: *tot_u = sum over fluid
 * cells of |u|, accumulated in double.  The caller applies free_cells_inv and sums partitions
 * (d2q9-bgk.c:753-755). */
int lbm_av_velocity_sum(lbm_ctx* ctx, double* tot_u);

/* ---- split-phase stepping for row-partitioned runs (one context per GPU / rank) -------------
 *
 * Mirrors one iteration of d2q9-bgk.c:315-378 on one rank:
 *     [caller starts the halo exchange: send buffers -> neighbours' recv buffers]   (:326-327)
 *     lbm_step_interior(ctx, stream)      rows that need no halo                    (:345-350)
 *     [caller waits for the exchange]                                                (:364)
 *     lbm_step_boundary(ctx, stream)      first and last owned row + next step's send buffers (:365-366)
 *     lbm_step_finish(ctx)                swap grids, advance the step counter      (:376-378)
 * Only the three populations that cross each cut travel: one message per direction of
 * lbm_halo_floats() floats (layout private to the library, identical on every rank of a run).
 *   dir 0 = south = towards row y-1 (the reference's `top` rank, rank-1)
 *   dir 1 = north = towards row y+1 (the reference's `bottom` rank, rank+1)
 * stream: a hipStream_t passed as void*; NULL = the context's own stream.
 * lbm_step_prepare() must be called once before the first step of a run of n_steps steps and
 * after any lbm_set_cells: it applies the step-0 accelerate_flow and fills the send buffers. */
size_t lbm_halo_floats(const lbm_ctx* ctx);
void*  lbm_halo_send_ptr(lbm_ctx* ctx, int dir);      /* device pointers */
void*  lbm_halo_recv_ptr(lbm_ctx* ctx, int dir);
/* Optional: make the library use caller-owned DEVICE buffers (each lbm_halo_floats() floats, 16-byte
 * aligned) for the four halo messages instead of its own — e.g. tensors of the framework that
 * also owns the communicator.  Index as dir above.  Call before lbm_step_prepare. */
int    lbm_bind_halo_buffers(lbm_ctx* ctx, void* send_south, void* send_north, void* recv_south, void* recv_north);
int    lbm_step_prepare(lbm_ctx* ctx, int n_steps, void* stream);
int    lbm_step_interior(lbm_ctx* ctx, void* stream);
int    lbm_step_boundary(lbm_ctx* ctx, void* stream);
int    lbm_step_finish(lbm_ctx* ctx, void* stream);
/* ---- K-step stepping of a row-partitioned run (contexts from lbm_create_rank / lbm_create_global) ----
 *
 * One macro-step = ONE halo exchange followed by a group of lbm_macro_next_launches() launches that together make
 * lbm_macro_next_steps() iterations of d2q9-bgk.c:315-378 — by default two launches of K = 4 steps on 8 ghost rows; fewer at
 * the end of a run, where a step count K does not divide is split into 4s and 3s; lbm_plan_group gives the sequence, which
 * depends on (K, ghost rows, launches per exchange, steps left) only, so every rank makes the same one:
 *     [caller: for each of the 9 planes, send lbm_macro_send_ptr(dir, plane) to the neighbour in
 *      direction dir and receive lbm_macro_recv_ptr(dir, plane) from it: lbm_macro_halo_floats()
 *      floats = `ghost` whole rows each; the pointers refer to the CURRENT grid and change every macro-step]
 *     lbm_macro_interior(ctx, stream)   first launch of the group, tiles that need no ghost row: overlaps the exchange
 *     [exchange complete]
 *     lbm_macro_edge(ctx, stream)       first launch of the group, first and last tile rows
 *     lbm_macro_finish(ctx, stream)     swap grids, advance the step counter; then the group's LATER launches, each over all tiles,
 *                                       on `stream` — they read the ghost rows the first launch advanced, no exchanged row: `stream`
 *                                       must be ordered behind both launches above when lbm_macro_next_launches() > 1
 * lbm_macro_prepare / lbm_step_collect / lbm_step_sums_device_ptr play the roles they have above.
 * lbm_macro_steps(): K, or 0 when the context is not in K-step mode (use lbm_step_* then).
 * lbm_macro_exchange_local(): the exchange between two contexts of one process (device copies):
 * src's rows travelling in direction dir become dst's ghost rows. */
int    lbm_macro_steps(const lbm_ctx* ctx);
int    lbm_macro_next_steps(const lbm_ctx* ctx);   /* steps of the macro-step about to be made (0: no run in progress) */
int    lbm_macro_next_launches(const lbm_ctx* ctx);/* launches of that macro-step */
size_t lbm_macro_halo_floats(const lbm_ctx* ctx);
void*  lbm_macro_send_ptr(lbm_ctx* ctx, int dir, int plane);
void*  lbm_macro_recv_ptr(lbm_ctx* ctx, int dir, int plane);
/* Packed exchange (2 messages per direction instead of 18): lbm_macro_pack gathers the outgoing rows of
 * all planes into lbm_macro_pack_ptr(dir, 0) (lbm_macro_pack_floats() floats per direction), the caller
 * moves them into the neighbours' lbm_macro_pack_ptr(dir, 1), lbm_macro_unpack scatters those into the
 * ghost rows.  dir of an incoming buffer = the side it came from (0 = from the south neighbour). */
size_t lbm_macro_pack_floats(const lbm_ctx* ctx);
void*  lbm_macro_pack_ptr(lbm_ctx* ctx, int dir, int incoming);
int    lbm_macro_pack(lbm_ctx* ctx, void* stream);
int    lbm_macro_unpack(lbm_ctx* ctx, void* stream);
int    lbm_macro_prepare(lbm_ctx* ctx, int n_steps, void* stream);
int    lbm_macro_interior(lbm_ctx* ctx, void* stream);
int    lbm_macro_edge(lbm_ctx* ctx, void* stream);
/* lbm_macro_interior + lbm_macro_edge as ONE launch over all tiles, for a caller whose exchange is complete before the group's first
 * launch starts (nothing left to overlap: small ranks on one stream). */
int    lbm_macro_all(lbm_ctx* ctx, void* stream);
int    lbm_macro_finish(lbm_ctx* ctx, void* stream);
int    lbm_macro_exchange_local(lbm_ctx* dst, lbm_ctx* src, int dir, void* stream);

/* Fold the per-block sums of the (macro-)step just finished into the per-step totals NOW, on `stream`,
 * instead of leaving them to block 0 of the next launch: for callers that need this step's total before the
 * next step starts (a per-step all-reduce, d2q9-bgk.c:367 as the first MPI version of the reference had it).
 * Call after lbm_step_finish / lbm_macro_finish, once every launch of the step is ordered before `stream`. */
int    lbm_step_fold(lbm_ctx* ctx, void* stream);

/* After the last lbm_step_finish of a run: this partition's per-step tot_u sums (double, device
 * resident until now) for the n_steps steps since lbm_step_prepare.  The caller reduces them over
 * partitions (the reference's MPI_Reduce, d2q9-bgk.c:396) and scales by free_cells_inv. */
int    lbm_step_collect(lbm_ctx* ctx, void* stream, double* tot_u_per_step, int n_steps);
/* Device pointer of the same per-step sums (for a device-side all-reduce), valid after
 * lbm_step_collect or a stream sync following the last lbm_step_finish. */
void*  lbm_step_sums_device_ptr(lbm_ctx* ctx);

/* Device time of the step kernels of the last completed run (lbm_run, or lbm_step_prepare ..
 * last lbm_step_finish), from HIP events recorded on the stream the kernels were launched on:
 * *ms = time from just before the first step kernel to just after the last one, *launches = the
 * number of step-kernel launches in between.  Valid once that stream has been synchronised. */
int lbm_last_run_kernel_ms(lbm_ctx* ctx, double* ms, int* launches);

/* Per-launch timing of lbm_run — what the measurement harness divides the per-launch HBM bytes by (the reference's
 * profiling region is MPI_Pcontrol(1/-1, "mainloop"), d2q9-bgk.c:275-277,404-406).  lbm_set_profile(ctx, 1): every later
 * lbm_run brackets each step-kernel launch with HIP timing events on its stream (a profiled run is for the breakdown:
 * the events cost a few microseconds per launch).  lbm_launch_profile: the last profiled run's launches in order —
 * steps[i] = lattice steps launch i advanced (lbm_multi_kernel<K>: K), us[i] = its duration; *n_launches = how many
 * there were (at most `cap` are written). */
int lbm_set_profile(lbm_ctx* ctx, int on);
int lbm_launch_profile(lbm_ctx* ctx, int cap, int* steps, double* us, int* n_launches);

/* The context's HIP device ordinal and its own stream (a hipStream_t as void*): what NULL means
 * for the `stream` arguments above. */
int   lbm_device(const lbm_ctx* ctx);
void* lbm_stream(lbm_ctx* ctx);

/* Kernel/launch facts for the measurement harness: name of the dominant kernel as rocprofv3
 * prints it, cells per launch, bytes of state in HBM. */
int lbm_describe(const lbm_ctx* ctx, char* kernel_name, size_t len, long long* cells_per_launch,
                 long long* state_bytes);

/* ---- host-side epilogue (no GPU needed) ------------------------------------------------------ */

/* av_velocity() for `rows` rows of AoS cells, reference order and precision (d2q9-bgk.c:716-751):
 * returns tot_u (float accumulator). */
float lbm_av_velocity_host(const lbm_params* p, const float* cells_aos, const int* obstacles, int rows);
/* The same value from lbm_get_observables() output: u_x, u_y are the floats of :732-746, the double
 * sqrt and the float accumulation (:748) happen here in the reference's cell order. */
float lbm_av_velocity_obs(const lbm_params* p, const float* obs, const int* obstacles, int rows);
/* calc_reynolds() given av_velocity()'s value (d2q9-bgk.c:1005-1007). */
float lbm_reynolds(const lbm_params* p, float av_velocity);
/* write_values(): final_state.dat rows (d2q9-bgk.c:1054-1120; displ = global y of row 0; append as
 * ranks > 0 do, :1057) and av_vels.dat (:1127-1139). */
int lbm_write_final_state(const char* path, const lbm_params* p, const float* cells_aos,
                          const int* obstacles, int rows, int displ, int append);
int lbm_write_final_state_obs(const char* path, const lbm_params* p, const float* obs,
                              const int* obstacles, int rows, int displ, int append);
int lbm_write_av_vels(const char* path, const float* av_vels, int n);

#ifdef __cplusplus
}
#endif
#endif
