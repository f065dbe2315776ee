/*
 * lbm_d2q9_p2p.h — C ABI of the row-partitioned step loop with DIRECT peer-to-peer halo stores over xGMI
 * (part of liblbm_d2q9.so: no RCCL, no MPI).
 *
 * Replaces, for one rank of a run with one context per GPU, the communication half of the reference's main
 * loop — the persistent halo requests (d2q9-bgk.c:295-313), MPI_Startall (:326-327), MPI_Waitall (:364) and
 * the end-of-run MPI_Reduce (:396) — without a communication library on the path:
 *
 *   - every rank maps its two ring neighbours' grids (hipIpcOpenMemHandle across processes, plain peer
 *     access inside one process) and, once per GROUP of launches (lbm_plan_group: the launches between two
 *     exchanges, r steps together — two 4-step launches on 8 ghost rows by default), a small kernel stores its
 *     first / last r rows straight into the neighbours' ghost rows (system-scope stores over the direct xGMI
 *     link), then raises an epoch flag in the neighbour's exported window (release, system scope);
 *   - the first launch of a group also advances the ghost rows the later launches read, so those are launches
 *     over all tiles with no exchange in between;
 *   - the consumer's edge launch is preceded on its stream by the same kernel's wait: one lane spins on the two
 *     flags (acquire, system scope) with a wall-clock bound: a missing peer ends the run with an error
 *     instead of a hang, and no workgroup ever waits for another workgroup of its own launch;
 *   - with one launch per exchange the two grids are the double buffer (rows for group m+1 land in the grid the
 *     consumer does not read during group m); a group of several launches returns to the grid it started from,
 *     so a rank tells both neighbours when its ghost rows may be written again — a "ready" word per neighbour,
 *     said and awaited (bounded) by the fold block of the group's last launch: flags only ever travel forward;
 *   - the end-of-run reduction is an all-gather of the per-step double sums into every rank's window and a
 *     local sum in rank order: bitwise the same vector on every rank, no collective library.
 *
 * Ranks of the tile (2-D) decomposition (contexts from lbm_create_tile, px x py of them, nranks = px * py, rank = ry * px + rx) run the
 * same loop with a second push in front of each exchange: my first / last r owned COLUMNS into the west / east neighbours' ghost columns
 * (their own flags and "ready" words), awaited, and then the row push above over whole storage rows — the ghost columns that have just
 * arrived included, which brings the corner blocks along.  The part of a group's first launch that runs beside the exchange is the
 * rectangle of tiles inside the rim of tile rows and tile columns that read exchanged cells (ranks of >= 2^25 cells; smaller ones run
 * everything on the compute stream).  Column blocks (py = 1: every rank owns all rows; lbm_tile_layout.ghost_y == 0) keep no ghost rows: their launches
 * wrap in y like a whole grid's and an exchange is the column push alone.
 *
 * Set-up is a two-phase handshake the caller carries by any means (this repo: torch.distributed
 * all_gather of LBM_P2P_HANDLE_BYTES per rank; the C CLI: an array in its own address space):
 *     lbm_p2p_create(&t, ctx, nranks, rank);  lbm_p2p_handle(t, my_blob);
 *     [all-gather the blobs in rank order]
 *     lbm_p2p_connect(t, all_blobs);          verifies that every rank runs the same K-step layout
 *     lbm_p2p_run(t, n_steps, tot_u);         any number of times, the same n_steps on every rank
 *     lbm_p2p_disconnect(t);  [barrier]  lbm_p2p_destroy(t);  lbm_destroy(ctx);
 * Contexts come from lbm_create_rank (or lbm_create_global / lbm_create with LBM_FLAG_FORCE_HALO).  In K-step mode
 * (lbm_macro_steps() > 0) the loop is the one above; runs that are not eligible for it (a rank with fewer than 32 rows,
 * odd or short rows) step one at a time: the boundary launch stores the three populations that cross each cut straight
 * into the neighbours' windows, one small kernel per step raises and awaits the flags.  Ranks of one run call
 * lbm_p2p_run concurrently (one host thread per rank when several ranks share a process).
 */
#ifndef LBM_D2Q9_P2P_H
#define LBM_D2Q9_P2P_H

#include "lbm_d2q9.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_P2P_HANDLE_BYTES 512

typedef struct lbm_p2p lbm_p2p;     /* opaque: exported window, mapped peers, streams and events of one rank */

int lbm_p2p_create(lbm_p2p** t, lbm_ctx* ctx, int nranks, int rank);
int lbm_p2p_handle(lbm_p2p* t, void* blob /* LBM_P2P_HANDLE_BYTES */);
int lbm_p2p_connect(lbm_p2p* t, const void* blobs /* nranks * LBM_P2P_HANDLE_BYTES, rank order */);
/* Unmap every peer's memory (the converse of lbm_p2p_connect).  Memory that another process has mapped must not be freed
 * before that process has unmapped it: ranks in separate processes call lbm_p2p_disconnect, meet at a barrier of the
 * caller's, and only then lbm_p2p_destroy / lbm_destroy (which free what the others had mapped). */
int lbm_p2p_disconnect(lbm_p2p* t);
int lbm_p2p_destroy(lbm_p2p* t);

/* n_steps iterations of d2q9-bgk.c:315-378 for this rank, then the reduction of :396: tot_u_per_step (host,
 * n_steps doubles) receives the GLOBAL per-step sum of |u|, bitwise identical on every rank;
 * av_vels[tt] = tot_u_per_step[tt] * free_cells_inv (:367).  Returns after the device work has completed;
 * non-zero if a neighbour's rows did not arrive within the time-out (LBM_P2P_TIMEOUT_MS, default 30 000): the
 * time-out bounds how far apart the ranks may ENTER a run (rank skew — a rank busy writing a multi-GB file while
 * the others start the next run), not only how long a failed peer is waited for.  After such an error the transport
 * is unusable (the neighbours' epochs no longer agree): every later lbm_p2p_run fails at once; destroy and re-create. */
int lbm_p2p_run(lbm_p2p* t, int n_steps, double* tot_u_per_step);

/* Where a run's time goes — the counterpart of the reference's profiling region around its main loop
 * (MPI_Pcontrol(1/-1, "mainloop"), d2q9-bgk.c:275-277,404-406).  With the profile switched on, lbm_p2p_run brackets its
 * launches with HIP events on the streams they run on (which perturbs the schedule by a few per cent: a profiled run is
 * for the breakdown, not for the headline time) and lbm_p2p_phases returns, in microseconds unless named otherwise:
 *    0 host_total          wall time of the lbm_p2p_run call
 *    1 host_enqueue        ... of which until the last launch was enqueued
 *    2 device_span         first event of the run -> end of the reduction, on the device
 *    3 setup               start of the run -> just before the first step kernel (counter reset, step-0 accelerate_flow)
 *    4 steps               first step kernel -> last step kernel done
 *    5 reduce              last step kernel done -> global sums in host memory (fold + all-gather + sum)
 *    6 macro_steps         COUNT of halo exchanges (groups of launches) of the run
 *    7 macro_step_avg      steps / macro_steps
 *    8 macro_step_steady   the same without the first and the last group (0 with fewer than three)
 *    9 interior_avg        average duration of the interior part of a group's first launch (serial schedule: that launch over all tiles)
 *   10 edge_avg            ... of its edge part (0 in the serial schedule)
 *   11 push_first          the push + wait kernel before the first group
 *   12 push_avg            the later push + wait kernels (each includes waiting for both neighbours' rows)
 *   13 host_overhead       host_total - device_span: launch latency before the first event + wake-up after the last
 *   14 launches            COUNT of step launches (a first launch's two parts count once)
 *   15 whole_avg           average duration of the later launches of the groups (all tiles, nothing exchanged)
 * One-step mode fills 0-7 only.  lbm_p2p_phase_name(i) returns the names above (NULL past the last). */
#define LBM_P2P_PHASES 16
int lbm_p2p_set_profile(lbm_p2p* t, int on);
int lbm_p2p_phases(const lbm_p2p* t, double* values /* LBM_P2P_PHASES */);
const char* lbm_p2p_phase_name(int i);

/* Facts for logs and the measurement harness: how the exported window was allocated ("uncached",
 * "fine-grained" or "coarse"), how the neighbours are reached ("ipc", "in-process" or "self"), and the
 * schedule ("edge stream" = edge rows beside the interior launch, or "serial"); tile ranks add "tiles PX x PY; ghost columns G". */
int lbm_p2p_describe(const lbm_p2p* t, char* text, size_t len);

#ifdef __cplusplus
}
#endif
#endif
