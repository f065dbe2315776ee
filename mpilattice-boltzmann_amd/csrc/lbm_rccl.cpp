// lbm_rccl.cpp — row-partitioned step loop with RCCL halo exchange (include/lbm_d2q9_rccl.h).
//
// A pure client of the core C ABI (lbm_step_* + halo buffer accessors): this file adds the
// communicator, two side streams and a few events.  Three queues per rank (reference lines d2q9-bgk.c):
//
//   compute stream                 comm stream (high priority)        edge stream (high priority)
//   ──────────────                 ───────────────────────────        ───────────────────────────
//   wait(edge_done[t-1])           wait(edge_done[t-1])               wait(halo[t]), wait(interior_done[t-1])
//   interior kernel t (:350)       group{send S, send N,              boundary kernel t (:365-366): reads the
//   record(interior_done[t])             recv N, recv S} (:327)        received rows, writes rows 0 / n-1 and the
//                                  record(halo[t])        (:364)       NEXT step's outgoing messages
//                                                                     record(edge_done[t])
//
// The exchange (2 x 96 KiB at nx = 8192: latency-bound, ~10-15 us) and the two edge rows (~6 us) form
// a short chain that runs beside the interior kernel (~110 us for 1024 rows of 8192), so a step
// costs the interior kernel plus one kernel boundary.  Dependencies are exactly the data ones:
// edge rows of step t need the interior of step t-1 (rows 1 and n-2 as pull sources, and their
// destination rows are that step's sources); the interior of step t+1 needs the edge rows of step t.
// xGMI is point-to-point: each of the two messages uses the direct link to its neighbour.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lbm_d2q9_rccl.h"
#include "lbm_internal.h"

struct lbm_comm {
  lbm_ctx* ctx = nullptr;
  ncclComm_t nccl = nullptr;
  int nranks = 1, rank = 0, south = 0, north = 0, device = 0;
  hipStream_t compute = nullptr;   // the context's own stream: interior kernels, folds, final collect
  hipStream_t side = nullptr;      // exchange stream
  hipStream_t edge = nullptr;      // boundary-row kernels
  hipEvent_t halo = nullptr, edge_done = nullptr, interior_done = nullptr;
  bool step_allreduce = false;     // one all-reduce per (macro-)step instead of one after the loop
  bool three_queues = true;        // edge rows on their own stream beside the interior kernel; LBM_RCCL_SCHEDULE=serial
                                   // puts them on the compute stream after the interior kernel instead
};

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      lbm_internal::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));        \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

#define NCCL_TRY(expr)                                                                   \
  do {                                                                                   \
    ncclResult_t r_ = (expr);                                                            \
    if (r_ != ncclSuccess) {                                                             \
      lbm_internal::set_error(std::string(#expr) + ": " + ncclGetErrorString(r_));       \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

#define LBM_TRY(expr)                                                                    \
  do {                                                                                   \
    if ((expr) != 0) return 1; /* message already set by the core library */             \
  } while (0)

extern "C" {

int lbm_comm_unique_id(char id[LBM_COMM_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) == LBM_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return 0;
}

int lbm_comm_create(lbm_comm** out, lbm_ctx* ctx, const char id[LBM_COMM_ID_BYTES], int nranks, int rank)
{
  if (!out || !ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) { lbm_internal::set_error("lbm_comm_create: bad argument"); return 1; }
  *out = nullptr;
  {
    lbm_tile_layout tile;
    if (lbm_tile_info(ctx, &tile) == 0 && tile.ghost_x > 0) {
      lbm_internal::set_error("lbm_comm_create: the context is a rank of the tile decomposition, which the peer-to-peer loop steps (lbm_p2p_run); the RCCL loop takes row partitions");
      return 1;
    }
  }
  lbm_comm* c = new lbm_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  c->south = (rank + nranks - 1) % nranks;   // `top`    d2q9-bgk.c:245-246
  c->north = (rank + 1) % nranks;            // `bottom` d2q9-bgk.c:247
  c->device = lbm_device(ctx);
  c->compute = static_cast<hipStream_t>(lbm_stream(ctx));
  auto fail = [&]() { lbm_comm_destroy(c); return 1; };
  if (hipSetDevice(c->device) != hipSuccess) { lbm_internal::set_error("lbm_comm_create: hipSetDevice failed"); return fail(); }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, u, rank);
  if (r != ncclSuccess) { lbm_internal::set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); c->nccl = nullptr; return fail(); }
  if (nranks > 1) {
    // every rank must run the same stepping mode with the same K and message size: ranks that disagree would
    // post sends and receives of different sizes at different cadences (hang, or corrupted ghost rows).
    // min and max over ranks of (K, message floats) must coincide.
    long long mine[4], *dev = nullptr;
    const long long k = lbm_macro_steps(ctx), n = k > 0 ? static_cast<long long>(lbm_macro_pack_floats(ctx)) : static_cast<long long>(lbm_halo_floats(ctx));
    mine[0] = k; mine[1] = -k; mine[2] = n; mine[3] = -n;
    hipStream_t s0 = c->compute;
    bool ok = hipMalloc(&dev, sizeof mine) == hipSuccess && hipMemcpyAsync(dev, mine, sizeof mine, hipMemcpyHostToDevice, s0) == hipSuccess;
    if (ok) {
      r = ncclAllReduce(dev, dev, 4, ncclInt64, ncclMax, c->nccl, s0);
      ok = r == ncclSuccess && hipMemcpyAsync(mine, dev, sizeof mine, hipMemcpyDeviceToHost, s0) == hipSuccess && hipStreamSynchronize(s0) == hipSuccess;
    }
    if (dev) (void)hipFree(dev);
    if (!ok) { lbm_internal::set_error("lbm_comm_create: layout check (ncclAllReduce) failed"); return fail(); }
    if (mine[0] != -mine[1] || mine[2] != -mine[3]) {
      lbm_internal::set_error("lbm_comm_create: ranks disagree about the stepping mode (K between " + std::to_string(-mine[1]) + " and " +
                              std::to_string(mine[0]) + ", halo message between " + std::to_string(-mine[3]) + " and " + std::to_string(mine[2]) +
                              " floats): create every rank with lbm_create_rank");
      return fail();
    }
  }
  const char* sched = std::getenv("LBM_RCCL_SCHEDULE");
  // default by size: three queues pay once the interior kernel is long enough to cover two extra
  // cross-queue waits (measured on MI355X: 8192x1024 rows 120 vs 133 us/step; 1024x128 rows 55 vs 34)
  long long cells = 0;
  (void)lbm_describe(ctx, nullptr, 0, &cells, nullptr);
  c->three_queues = cells >= (1LL << 21);
  if (sched && std::string(sched) == "serial") c->three_queues = false;
  if (sched && std::string(sched) == "edge") c->three_queues = true;
  // LBM_RCCL_PRIORITY=1: side/edge streams at the highest stream priority (measured: slower)
  const char* prio = std::getenv("LBM_RCCL_PRIORITY");
  int prio_low = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);   // numerically lower = higher priority
  const int use_prio = (prio && prio[0] == '1') ? prio_high : 0;   // 0 = the normal priority of hipStreamCreate
  if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, use_prio) != hipSuccess ||
      hipStreamCreateWithPriority(&c->edge, hipStreamNonBlocking, use_prio) != hipSuccess ||
      hipEventCreateWithFlags(&c->halo, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->edge_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->interior_done, hipEventDisableTiming) != hipSuccess) {
    lbm_internal::set_error("lbm_comm_create: stream/event creation failed");
    return fail();
  }
  *out = c;
  return 0;
}

int lbm_comm_nranks(const lbm_comm* c)
{
  int n = 0;
  if (!c || !c->nccl || ncclCommCount(c->nccl, &n) != ncclSuccess) return -1;
  return n;
}

int lbm_comm_set_step_allreduce(lbm_comm* c, int on)
{
  if (!c) { lbm_internal::set_error("lbm_comm_set_step_allreduce: null communicator"); return 1; }
  c->step_allreduce = on != 0;
  return 0;
}

int lbm_comm_destroy(lbm_comm* c)
{
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->side) (void)hipStreamSynchronize(c->side);
  if (c->edge) (void)hipStreamSynchronize(c->edge);
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  if (c->halo) (void)hipEventDestroy(c->halo);
  if (c->edge_done) (void)hipEventDestroy(c->edge_done);
  if (c->interior_done) (void)hipEventDestroy(c->interior_done);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->edge) (void)hipStreamDestroy(c->edge);
  delete c;
  return 0;
}

// K-step mode (contexts from lbm_create_rank / lbm_create_global, lbm_macro_steps() = K > 0): one exchange of the partition's ghost
// rows (`ghost` whole rows of each of the 9 planes: 2 K) per GROUP of launches (lbm_macro_next_launches(): two 4-step launches by
// default), same three-queue schedule for the first launch of the group; the later ones — launches over all tiles that read no
// exchanged row — are made by lbm_macro_finish on the compute stream.  By default the 18 row blocks of a direction pair are packed into
// one message per direction by a small kernel on the exchange stream (fewer, larger messages); LBM_RCCL_PACK=0 sends them straight
// from the edge rows into the neighbour's ghost rows as 18 + 18 messages.
static int run_macro(lbm_comm* c, int n_steps, double* tot_u_per_step)
{
  lbm_ctx* ctx = c->ctx;
  const size_t n = lbm_macro_halo_floats(ctx), np = lbm_macro_pack_floats(ctx);
  const char* pk = std::getenv("LBM_RCCL_PACK");
  const bool packed = !(pk && pk[0] == '0');              // default: 2 + 2 packed messages; 0 = 18 + 18 direct ones
  const bool three_queues = c->three_queues;
  hipStream_t edge_stream = c->edge;
  // small ranks (one-queue schedule): the exchange goes to the compute stream as well — every hand-off between queues costs 6 - 8 us of idle
  // queue, and there is no launch long enough to hide the exchange behind (a 1024 x 128-row ring: 26 of 98 us per exchange were hand-offs)
  hipStream_t xs = three_queues ? c->side : c->compute;
  LBM_TRY(lbm_macro_prepare(ctx, n_steps, c->compute));   // step-0 accelerate_flow
  HIP_TRY(hipEventRecord(c->edge_done, c->compute));
  HIP_TRY(hipEventRecord(c->interior_done, c->compute));
  int prev_launches = 1;
  for (int done = 0; done < n_steps;) {
    const int k = lbm_macro_next_steps(ctx);                 // steps until the next exchange: those of the group's launches together
    const int launches = lbm_macro_next_launches(ctx);
    if (k <= 0 || launches <= 0) { lbm_internal::set_error("lbm_comm_run: the context has no macro-step left"); return 1; }
    // the rows to send were written by the last launch of the previous group: its edge launch and whatever ran on the compute stream
    if (three_queues) {
      HIP_TRY(hipStreamWaitEvent(c->side, c->edge_done, 0));
      HIP_TRY(hipStreamWaitEvent(c->side, c->interior_done, 0));
    }
    if (packed) {
      // gather the 9 planes' rows into one message per direction, exchange 2 + 2 messages, scatter
      LBM_TRY(lbm_macro_pack(ctx, xs));
      NCCL_TRY(ncclGroupStart());                            // order as in the one-step loop: sends [S, N], receives [N, S]
      NCCL_TRY(ncclSend(lbm_macro_pack_ptr(ctx, 0, 0), np, ncclFloat, c->south, c->nccl, xs));
      NCCL_TRY(ncclSend(lbm_macro_pack_ptr(ctx, 1, 0), np, ncclFloat, c->north, c->nccl, xs));
      NCCL_TRY(ncclRecv(lbm_macro_pack_ptr(ctx, 1, 1), np, ncclFloat, c->north, c->nccl, xs));
      NCCL_TRY(ncclRecv(lbm_macro_pack_ptr(ctx, 0, 1), np, ncclFloat, c->south, c->nccl, xs));
      NCCL_TRY(ncclGroupEnd());
      LBM_TRY(lbm_macro_unpack(ctx, xs));
    } else {
      NCCL_TRY(ncclGroupStart());
      for (int q = 0; q < LBM_NSPEEDS; ++q) {                // rows straight into the neighbour's ghost rows
        NCCL_TRY(ncclSend(lbm_macro_send_ptr(ctx, 0, q), n, ncclFloat, c->south, c->nccl, xs));
        NCCL_TRY(ncclSend(lbm_macro_send_ptr(ctx, 1, q), n, ncclFloat, c->north, c->nccl, xs));
        NCCL_TRY(ncclRecv(lbm_macro_recv_ptr(ctx, 1, q), n, ncclFloat, c->north, c->nccl, xs));
        NCCL_TRY(ncclRecv(lbm_macro_recv_ptr(ctx, 0, q), n, ncclFloat, c->south, c->nccl, xs));
      }
      NCCL_TRY(ncclGroupEnd());
    }
    if (three_queues) HIP_TRY(hipEventRecord(c->halo, c->side));
    if (three_queues) {
      if (prev_launches == 1) HIP_TRY(hipStreamWaitEvent(c->compute, c->edge_done, 0));
      LBM_TRY(lbm_macro_interior(ctx, c->compute));
      HIP_TRY(hipStreamWaitEvent(edge_stream, c->halo, 0));
      HIP_TRY(hipStreamWaitEvent(c->edge, c->interior_done, 0));
      LBM_TRY(lbm_macro_edge(ctx, edge_stream));
      HIP_TRY(hipEventRecord(c->edge_done, edge_stream));
    } else {
      LBM_TRY(lbm_macro_all(ctx, c->compute));               // the exchange is complete in stream order: one launch over all tiles
    }
    // the later launches of the group (and the last fold of a run) follow on the compute stream and read the edge rows
    if (three_queues && (launches > 1 || done + k >= n_steps)) HIP_TRY(hipStreamWaitEvent(c->compute, c->edge_done, 0));
    LBM_TRY(lbm_macro_finish(ctx, c->compute));
    if (three_queues) HIP_TRY(hipEventRecord(c->interior_done, c->compute));
    if (c->step_allreduce) {
      // this group's totals now, and the next group behind their all-reduce (north_star wording)
      if (three_queues) HIP_TRY(hipStreamWaitEvent(c->compute, c->edge_done, 0));
      LBM_TRY(lbm_step_fold(ctx, c->compute));
      double* sums_now = static_cast<double*>(lbm_step_sums_device_ptr(ctx)) + done;
      NCCL_TRY(ncclAllReduce(sums_now, sums_now, static_cast<size_t>(k), ncclDouble, ncclSum, c->nccl, c->compute));
    }
    done += k;
    prev_launches = launches;
  }
  double* sums = static_cast<double*>(lbm_step_sums_device_ptr(ctx));
  if (c->nranks > 1 && !c->step_allreduce) NCCL_TRY(ncclAllReduce(sums, sums, static_cast<size_t>(n_steps), ncclDouble, ncclSum, c->nccl, c->compute));
  LBM_TRY(lbm_step_collect(ctx, nullptr, tot_u_per_step, n_steps));
  HIP_TRY(hipStreamSynchronize(c->side));
  HIP_TRY(hipStreamSynchronize(c->edge));
  return 0;
}

int lbm_comm_run(lbm_comm* c, int n_steps, double* tot_u_per_step)
{
  if (!c || n_steps < 0 || (n_steps > 0 && !tot_u_per_step)) { lbm_internal::set_error("lbm_comm_run: bad argument"); return 1; }
  if (n_steps == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  lbm_ctx* ctx = c->ctx;
  if (lbm_macro_steps(ctx) > 0) return run_macro(c, n_steps, tot_u_per_step);
  const size_t n = lbm_halo_floats(ctx);
  float* send_s = static_cast<float*>(lbm_halo_send_ptr(ctx, 0));
  float* send_n = static_cast<float*>(lbm_halo_send_ptr(ctx, 1));
  float* recv_s = static_cast<float*>(lbm_halo_recv_ptr(ctx, 0));
  float* recv_n = static_cast<float*>(lbm_halo_recv_ptr(ctx, 1));

  LBM_TRY(lbm_step_prepare(ctx, n_steps, nullptr));       // step-0 accelerate_flow + first outgoing rows (compute stream)
  HIP_TRY(hipEventRecord(c->edge_done, c->compute));      // "edge rows of step -1" = the prepared state
  HIP_TRY(hipEventRecord(c->interior_done, c->compute));
  const bool three_queues = c->three_queues;
  hipStream_t edge_stream = three_queues ? c->edge : c->compute;
  for (int t = 0; t < n_steps; ++t) {
    // exchange t: needs the outgoing rows written by the edge kernel of step t-1
    HIP_TRY(hipStreamWaitEvent(c->side, c->edge_done, 0));
    // sends [south, north] pair with receives [north, south]: the reference's request order
    // (d2q9-bgk.c:295-303), which also keeps the two messages apart when both neighbours are one rank
    NCCL_TRY(ncclGroupStart());
    NCCL_TRY(ncclSend(send_s, n, ncclFloat, c->south, c->nccl, c->side));
    NCCL_TRY(ncclSend(send_n, n, ncclFloat, c->north, c->nccl, c->side));
    NCCL_TRY(ncclRecv(recv_n, n, ncclFloat, c->north, c->nccl, c->side));
    NCCL_TRY(ncclRecv(recv_s, n, ncclFloat, c->south, c->nccl, c->side));
    NCCL_TRY(ncclGroupEnd());                              // MPI_Startall (:327)
    HIP_TRY(hipEventRecord(c->halo, c->side));
    // interior t: needs the edge rows of step t-1 (and, by stream order, the interior of step t-1)
    if (three_queues) HIP_TRY(hipStreamWaitEvent(c->compute, c->edge_done, 0));
    LBM_TRY(lbm_step_interior(ctx, c->compute));           // :350, overlaps the exchange
    // edge rows t: need the halos of step t and the interior of step t-1
    HIP_TRY(hipStreamWaitEvent(edge_stream, c->halo, 0));  // MPI_Waitall (:364), on the device
    if (three_queues) HIP_TRY(hipStreamWaitEvent(c->edge, c->interior_done, 0));
    LBM_TRY(lbm_step_boundary(ctx, edge_stream));          // :365-366
    HIP_TRY(hipEventRecord(c->edge_done, edge_stream));
    if (three_queues) {
      HIP_TRY(hipEventRecord(c->interior_done, c->compute));
      if (t + 1 == n_steps) HIP_TRY(hipStreamWaitEvent(c->compute, c->edge_done, 0));   // the final fold reads both
    }
    LBM_TRY(lbm_step_finish(ctx, c->compute));             // :376-378
    if (c->step_allreduce) {
      if (three_queues) HIP_TRY(hipStreamWaitEvent(c->compute, c->edge_done, 0));
      LBM_TRY(lbm_step_fold(ctx, c->compute));
      double* sum_now = static_cast<double*>(lbm_step_sums_device_ptr(ctx)) + t;
      NCCL_TRY(ncclAllReduce(sum_now, sum_now, 1, ncclDouble, ncclSum, c->nccl, c->compute));
    }
  }
  // MPI_Reduce of the per-step vector (:396), as an in-place all-reduce on the compute stream
  double* sums = static_cast<double*>(lbm_step_sums_device_ptr(ctx));
  if (c->nranks > 1 && !c->step_allreduce) NCCL_TRY(ncclAllReduce(sums, sums, static_cast<size_t>(n_steps), ncclDouble, ncclSum, c->nccl, c->compute));
  LBM_TRY(lbm_step_collect(ctx, nullptr, tot_u_per_step, n_steps));
  HIP_TRY(hipStreamSynchronize(c->side));
  HIP_TRY(hipStreamSynchronize(c->edge));
  return 0;
}

}  // extern "C"
