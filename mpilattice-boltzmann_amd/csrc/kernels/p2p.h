// p2p.h — kernels of the peer-to-peer halo transport (include/lbm_d2q9_p2p.h): push rows into a neighbour's
// ghost rows, raise / await epoch flags, all-gather + fold of the per-step sums.
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
//
// Memory-model notes (gfx950): a peer GPU's memory is reached over xGMI through this GPU's L2, so what is
// meant for the peer is followed by a system-scope release fence (L2 write-back) before the flag is raised;
// flags live in an uncached window and are read / written with system-scope atomics; the consumer's data loads
// happen in a LATER kernel of its stream (kernel boundaries invalidate the non-coherent caches, as they must
// between any two launches that hand data across XCDs).
// Round 4: a rank's OWN launches now write ghost rows too (the first launches of a group advance them), rows a peer overwrites one
// exchange later.  A peer's stores reach this GPU's memory without passing its L2s, so no dirty line of those rows may survive in an
// XCD's L2 until then: between this rank's last write and the peer's first there is always a kernel boundary of this rank — the
// "ready" word is said by a LATER launch than any that wrote them — and a kernel boundary writes the L2s back (it must: the next
// launch's blocks on the other seven XCDs read what this one wrote).
#pragma once
#include "common.h"

namespace {

struct P2PWindowHeader {                 // start of every rank's exported window (uncached device memory)
  unsigned long long halo_flag[2];       // [0]: epoch of the rows my SOUTH neighbour pushed into my bottom ghost rows; [1]: NORTH
  unsigned long long halo_parity[4];     // [2*dir + (epoch & 1)]: the pusher's grid parity for that epoch (lock-step check).  A pusher
                                         // may run ONE epoch ahead of the waiter, never two: a slot per epoch parity is not overwritten early
  unsigned long long reduce_flag[64];    // [r]: rank r's sums of reduce round N are in my slot r
  unsigned long long halo_ack[4];        // [0]: my SOUTH neighbour has finished every launch that reads or writes its ghost rows of the grid the push of
                                         // this epoch writes: they may be written; [1]: NORTH.  The two grids
                                         // double-buffered the ghost rows only as long as every exchange was followed by ONE launch (rounds 1-3): a
                                         // group of two launches returns to the grid it started from, and the rows of epoch e+1 would land in the
                                         // rows a neighbour still reads for epoch e.  Tile decomposition: [2] WEST, [3] EAST, for the ghost columns.
  unsigned long long halo_flag_x[2];     // tile decomposition: [0]: epoch of the columns my WEST neighbour pushed into my west ghost columns; [1]: EAST
  unsigned long long halo_parity_x[4];   // as halo_parity
};
constexpr size_t kP2PHeaderBytes = 4096;
static_assert(sizeof(P2PWindowHeader) <= kP2PHeaderBytes, "window header");

struct P2PPushArgs {
  const float* src;                      // my grid (plane 0, storage row 0)
  size_t ps;                             // my plane stride
  float* dst[2];                         // [0]: south neighbour's grid, plane 0, first of its TOP ghost rows; [1]: north neighbour's, row 0
  size_t dst_ps[2];                      // their plane strides
  size_t src_row[2];                     // my storage row where direction d's K rows start
  int nfloats;                           // K * nx
  unsigned long long* flag[2];           // [0]: south neighbour's halo_flag[1] (rows arrive from ITS north); [1]: north neighbour's halo_flag[0]
  unsigned long long* parity_word[2];
  unsigned long long epoch, parity;
  unsigned int* done;                    // block-done counter (my device memory), zero between launches
  // then wait (block 0, one lane) until BOTH neighbours' rows of the same epoch have arrived here: the push of a
  // macro-step and the wait before the next one are always adjacent in the stream, and one kernel boundary less
  // per macro-step is what a 1024 x 128-row partition notices (20 % of its step)
  const unsigned long long* wait_flags;  // my halo_flag[2], or null: push only
  const unsigned long long* wait_parity; // my halo_parity[4]
  long long timeout_ticks;
  int* err;
};

// My first K owned rows -> the south neighbour's top ghost rows, my last K owned rows -> the north
// neighbour's bottom ghost rows (all 9 planes), then the two epoch flags, raised by the last block to
// finish.  A few large blocks (grid-stride over 2 directions x 9 planes x K*nx/2 float2's): the release towards
// the peer is one L2 write-back per BLOCK — every wave's stores have left the CU at the barrier (vmcnt(0)), all
// waves of a block sit on one XCD, so thread 0's system-scope fence covers them — and a write-back per wave of
// an 864-block form of this kernel cost 50 us beside the interior launch instead of 8.
constexpr int kP2PPushBlocks = 64;

template <typename V>      // V = f4 (16-byte accesses: nx a multiple of 4) or f2
__global__ void __launch_bounds__(256) lbm_p2p_push_kernel(const P2PPushArgs a, int nx)
{
  // (Where the launches since the last exchange were more than one, both neighbours have said that their ghost rows may be written
  // before this kernel starts: the fold block of this rank's last launch waited for it, MultiArgs::wait_ready.)
  constexpr int kPer = sizeof(V) / sizeof(float);
  const int per_plane = a.nfloats / kPer;                      // vectors of one plane's rows
  const int total = per_plane * 18;
  // four independent loads in flight per lane, then their stores: the kernel is a chain of memory round trips
  // (a 1024 x 4-row message is 295 KB: one pass; a 16-row one three passes of 8-byte accesses, 12 us — 16-byte accesses halve them), not a
  // bandwidth problem
  constexpr int kUnroll = 4;
  for (int w0 = blockIdx.x * 256 + threadIdx.x; w0 < total; w0 += gridDim.x * 256 * kUnroll) {
    V v[kUnroll];
    float* q[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int w = w0 + u * gridDim.x * 256;
      q[u] = nullptr;
      if (w < total) {
        const int seg = w / per_plane, i = w - seg * per_plane;
        const int dir = seg / 9, plane = seg - dir * 9;
        v[u] = *reinterpret_cast<const V*>(a.src + plane * a.ps + a.src_row[dir] * nx + static_cast<size_t>(i) * kPer);
        q[u] = a.dst[dir] + plane * a.dst_ps[dir] + static_cast<size_t>(i) * kPer;
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
      if (q[u]) *reinterpret_cast<V*>(q[u]) = v[u];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // Release towards the peers, once per block: every wave's stores have left the CU at the barrier, this fence
    // writes this XCD's L2 back (system scope) and completes before the arrival counter moves.  The block that
    // sees every other block's arrival (acquire) therefore knows all rows are out, and raises the flags with plain
    // write-through (system-scope, relaxed) stores: no second and third L2 write-back on the critical path.
    __atomic_thread_fence(__ATOMIC_RELEASE);                   // system scope
    const unsigned int prev = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {                               // the last block raises the flags
      __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {   // (after a time-out the run is failing: raise nothing)
        for (int d = 0; d < 2; ++d) __hip_atomic_store(a.parity_word[d], a.parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_waitcnt(0);                         // parity words before flags (both write-through)
        for (int d = 0; d < 2; ++d) __hip_atomic_store(a.flag[d], a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // my own push does not depend on this wait, so a ring of ranks that all sit here cannot dead-lock: every
    // rank's flags are raised by blocks that never wait for rows
    if (blockIdx.x == 0 && a.wait_flags) p2p_wait_flags(a.wait_flags, a.wait_parity, 2, a.epoch, a.parity, a.timeout_ticks, a.err);
  }
}

// Tile decomposition, first half of an exchange: my first k owned COLUMNS -> the west neighbour's east ghost columns, my last k -> the
// east neighbour's west ghost columns, owned rows only, all 9 planes; flags; wait for both neighbours' columns.  The row push that
// follows in the stream then sends whole storage rows — the ghost columns that have just arrived included — which carries the corner
// blocks from the diagonal neighbours in two hops, with no third message.  Same structure as lbm_p2p_push_kernel (one release per block,
// the last block raises the flags, block 0 waits); a message is k-float row segments, a chain of short strided accesses.
struct P2PPushColsArgs {
  const float* src;                      // my grid (plane 0, storage row 0)
  size_t ps;
  int src_w;                             // my storage row width
  float* dst[2];                         // [0]: west neighbour's grid, plane 0, at (its first owned row, its first EAST ghost column); [1]: east neighbour's, at (first owned row, ghost_x - k)
  size_t dst_ps[2];
  int dst_w[2];                          // their storage row widths
  int src_col[2];                        // my column where direction d's k columns start
  int row0, nrows, k;                    // my first owned storage row, owned rows, columns per direction
  unsigned long long* flag[2];           // [0]: west neighbour's halo_flag_x[1] (columns arrive from ITS east); [1]: east neighbour's halo_flag_x[0]
  unsigned long long* parity_word[2];
  unsigned long long epoch, parity;
  unsigned int* done;
  const unsigned long long* wait_flags;  // my halo_flag_x[2]
  const unsigned long long* wait_parity; // my halo_parity_x[4]
  long long timeout_ticks;
  int* err;
};

template <typename V>      // V = f4, f2 or float: the widest access every row segment is aligned for
__global__ void __launch_bounds__(256) lbm_p2p_push_cols_kernel(const P2PPushColsArgs a)
{
  constexpr int kPer = sizeof(V) / sizeof(float);
  const int kv = a.k / kPer;                                   // vectors of one row segment
  const int per_plane = a.nrows * kv;
  const int total = per_plane * 18;
  constexpr int kUnroll = 4;
  for (int w0 = blockIdx.x * 256 + threadIdx.x; w0 < total; w0 += gridDim.x * 256 * kUnroll) {
    V v[kUnroll];
    float* q[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int w = w0 + u * gridDim.x * 256;
      q[u] = nullptr;
      if (w < total) {
        const int seg = w / per_plane, i = w - seg * per_plane;
        const int dir = seg / 9, plane = seg - dir * 9;
        const int r = i / kv, j = i - r * kv;
        v[u] = *reinterpret_cast<const V*>(a.src + plane * a.ps + static_cast<size_t>(a.row0 + r) * a.src_w + a.src_col[dir] + j * kPer);
        q[u] = a.dst[dir] + plane * a.dst_ps[dir] + static_cast<size_t>(r) * a.dst_w[dir] + j * kPer;
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
      if (q[u]) *reinterpret_cast<V*>(q[u]) = v[u];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);                   // system scope: this block's stores are out (see lbm_p2p_push_kernel)
    const unsigned int prev = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        for (int d = 0; d < 2; ++d) __hip_atomic_store(a.parity_word[d], a.parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_waitcnt(0);
        for (int d = 0; d < 2; ++d) __hip_atomic_store(a.flag[d], a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (blockIdx.x == 0 && a.wait_flags) p2p_wait_flags(a.wait_flags, a.wait_parity, 2, a.epoch, a.parity, a.timeout_ticks, a.err);
  }
}

// One-step loop: "my halo messages of epoch e are complete" to both neighbours, then wait for theirs.  The messages
// were stored into the neighbours' (uncached) windows by the kernel before this one in the stream.
__global__ void __launch_bounds__(64) lbm_p2p_signal_wait_kernel(unsigned long long* flag_s, unsigned long long* flag_n, unsigned long long* parity_s,
                                                                  unsigned long long* parity_n, const unsigned long long* wait_flags,
                                                                  const unsigned long long* wait_parity, unsigned long long epoch, unsigned long long parity,
                                                                  long long timeout_ticks, int* err)
{
  if (threadIdx.x != 0) return;
  __atomic_thread_fence(__ATOMIC_RELEASE);                     // system scope
  __hip_atomic_store(parity_s, parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(parity_n, parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_waitcnt(0);
  __hip_atomic_store(flag_s, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(flag_n, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  p2p_wait_flags(wait_flags, wait_parity, 2, epoch, parity, timeout_ticks, err);
}

// End-of-run reduction in ONE launch — the reference's MPI_Reduce(MPI_SUM) (d2q9-bgk.c:396) as an all-gather and a
// local sum in rank order, the same bits on every rank.  Block r copies my per-step sums into slot `my rank` of rank r's
// window and raises my flag there (stores, barrier, one system-scope release, flag).  Every block then waits until every
// rank's flag in MY window has reached this round (bounded, as every wait here) and folds its share of the nranks slots
// into `out`, a host-mapped buffer: no copy kernel, no memcpy.  No block waits for another block of its own launch:
// what a block waits for is raised by the OTHER ranks' launches and by its own launch's blocks before they wait.
struct P2PReduceArgs {
  const double* sums;                        // my per-step sums of this chunk
  int n, nranks;
  double* const* slots;                      // [nranks]: rank r's slot for me (this round's parity)
  unsigned long long* const* flags;          // [nranks]: my flag word in rank r's window
  unsigned long long round;
  const unsigned long long* my_flags;        // my window's reduce_flag[nranks]
  const double* my_slots;                    // my window's slots of this parity, slot_stride doubles apart
  size_t slot_stride;
  double* out;                               // host-mapped, n doubles
  long long timeout_ticks;
  int* err;
};

__global__ void __launch_bounds__(256) lbm_p2p_allreduce_kernel(const P2PReduceArgs a)
{
  const int r = blockIdx.x;
  double* dst = a.slots[r];
  for (int i = threadIdx.x; i < a.n; i += 256) dst[i] = a.sums[i];
  __syncthreads();                                             // every wave's stores have left the CU
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);                   // system scope
    __hip_atomic_store(a.flags[r], a.round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // every block then takes a share of the fold (a run of 80 000 steps reduces 80 000 sums): each waits for itself
  if (threadIdx.x == 0) p2p_wait_flags(a.my_flags, nullptr, a.nranks, a.round, 0ull, a.timeout_ticks, a.err);
  __syncthreads();
  for (int i = r * 256 + threadIdx.x; i < a.n; i += 256 * a.nranks) {
    double s = 0.0;
    for (int q = 0; q < a.nranks; ++q)
      s += __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(a.my_slots + q * a.slot_stride + i), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_SYSTEM));
    a.out[i] = s;
  }
}

}  // namespace
