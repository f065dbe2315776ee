/*
 * lbm_d2q9_rccl.h — C ABI of the row-partitioned step loop with the halo exchange done by RCCL
 * (liblbm_d2q9_rccl.so; depends on liblbm_d2q9.so and librccl).
 *
 * Replaces, for one rank of a run with one process per GPU, the communication half of the
 * reference's main loop: the persistent halo requests (d2q9-bgk.c:295-313), MPI_Startall (:326-327),
 * MPI_Waitall (:364) and the end-of-run MPI_Reduce (:396).  The loop runs natively: per step one
 * RCCL group (2 sends + 2 receives over the direct xGMI links to the two ring neighbours) on a side
 * HIP stream, overlapped with the interior kernel on the compute stream; events order the two.
 * The communicator is bootstrapped from a 128-byte unique id that the caller distributes by any
 * means (this repo: torch.distributed broadcast).
 */
#ifndef LBM_D2Q9_RCCL_H
#define LBM_D2Q9_RCCL_H

#include "lbm_d2q9.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_COMM_ID_BYTES 128

typedef struct lbm_comm lbm_comm;   /* opaque: RCCL communicator + side stream + events of one rank */

/* Rank 0 calls this and ships the 128 bytes to every rank (ncclGetUniqueId). */
int lbm_comm_unique_id(char id[LBM_COMM_ID_BYTES]);

/* Collective over the nranks processes (ncclCommInitRank) on ctx's device.  Ring neighbours follow
 * d2q9-bgk.c:245-247: south = `top` = rank-1 (wrapping), north = `bottom` = (rank+1) % nranks.
 * With nranks == 1 the rank exchanges with itself (the reference's 1-rank behaviour); the context
 * must then have been created with LBM_FLAG_FORCE_HALO. */
int lbm_comm_create(lbm_comm** comm, lbm_ctx* ctx, const char id[LBM_COMM_ID_BYTES], int nranks, int rank);
int lbm_comm_destroy(lbm_comm* comm);

/* Number of ranks of the communicator as RCCL reports it (ncclCommCount): what actually runs, for logs. */
int lbm_comm_nranks(const lbm_comm* comm);

/* Reduction mode.  0 (default): the reference's final form — per-rank partial sums stay on the device and
 * ONE all-reduce of the whole per-step vector follows the loop (d2q9-bgk.c:367,396; report.odt §1 measured
 * 1.13-2.39x from hoisting the reduce out of the loop).  1: one ncclAllReduce per (macro-)step on the compute
 * stream, the next step ordered behind it — BASELINE.json's north_star wording and the reference's first MPI
 * version (newprofiles/firstMPI*.out).  Same av_vels either way (same double sums, same rank order inside
 * RCCL's ring for a given communicator); the mode exists so that its cost can be measured. */
int lbm_comm_set_step_allreduce(lbm_comm* comm, int on);

/* n_steps iterations of d2q9-bgk.c:315-378 for this rank, then the reduction of :396 as an
 * all-reduce: tot_u_per_step (host, n_steps doubles) receives the GLOBAL per-step sum of |u| on
 * every rank; av_vels[tt] = tot_u_per_step[tt] * free_cells_inv (:367).  Returns after the
 * device work has completed. */
int lbm_comm_run(lbm_comm* comm, int n_steps, double* tot_u_per_step);

#ifdef __cplusplus
}
#endif
#endif
