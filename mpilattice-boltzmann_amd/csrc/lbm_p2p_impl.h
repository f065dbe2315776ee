// lbm_p2p_impl.h — host half of the peer-to-peer halo transport (include/lbm_d2q9_p2p.h).
// Included at the end of lbm_kernels.hip: it drives the K-step launches of a context directly
// (launch_multi, begin_run, fold_last) and owns the streams / events / mapped peer memory of one rank.
//
// Per GROUP of launches (plan_group: the launches next_multi_k plans — K steps each, 3s and 4s at the end — for as long as their steps
// add up to at most the ghost rows; two 4-step launches on 8 ghost rows by default) of one rank, reference lines d2q9-bgk.c:
//
//   compute stream                         edge stream
//   ──────────────                         ───────────
//   interior tiles of launch 0   (:350)    push kernel: "ready" to both neighbours, wait for theirs; my first / last r owned rows
//                                                       (r = the steps of the group) of the current state -> neighbours' ghost rows,
//                                                       flags := epoch                          MPI_Startall (:327)
//                                                       then wait: both neighbours' flags >= epoch   MPI_Waitall (:364)
//                                          edge tile rows of launch 0              (:365-366)
//   wait(edge rows)                        record(edge rows)
//   launch 1 ... over ALL tiles: reads
//     the ghost rows launch 0 advanced,
//     no exchanged row
//   record(group done)                     wait(group done); push for the next group
//
// Launch 0 computes, besides the owned rows, ext = (steps of the later launches) ghost rows on each side from the exchanged rows;
// the later launches are plain launches over all tiles.  With one launch per group (ghost rows = K: rounds 1-3's loop) the push
// follows the edge launch directly, as the rows it reads are that launch's.
//
// Small partitions (< 2 M cells) run everything on the compute stream — the launches of the group, each over all tiles, then
// the push + wait kernel: there the cross-queue waits cost more than the overlap hides.
// (Tried and dropped: the push fused into the step launch, the edge tiles storing their rows to the neighbours
// as well and the last of them raising the flags and waiting — 5.9 instead of 4.3 us/step on a 1024 x 128-row
// ring: every pushing tile's system-scope fence writes back an L2 that the whole launch is streaming through.)
#pragma once

#include <unistd.h>

#include <chrono>

#include "lbm_d2q9_p2p.h"

namespace {

constexpr uint32_t kP2PMagic = 0x4C424D50u;   // "LBMP"

struct P2PBlob {                       // what every rank tells every other rank (POD, <= LBM_P2P_HANDLE_BYTES)
  uint32_t magic, version;
  int32_t pid, device, nranks, rank;
  int32_t nx, ny, y0, nyl, ghost, K, cur, ipc_ok, group;      // nx: the GLOBAL grid width
  int32_t w, x0, nxl, ghost_x, px, py;                         // storage row width; tile decomposition: the rank's columns and the rank grid (1 x nranks otherwise)
  int32_t ghost_rows;                                          // storage rows below / above the owned ones (`ghost`; 0 for the column blocks of a px x 1 tiling)
  uint64_t ps, window_bytes, reduce_cap;
  uint64_t grid_ptr[2], window_ptr;    // raw device pointers: valid inside the exporting process
  hipIpcMemHandle_t grid_h[2], window_h;
  char host[64];
};
static_assert(sizeof(P2PBlob) <= LBM_P2P_HANDLE_BYTES, "blob fits the handle");

struct P2PPeer {
  bool mapped = false, ipc = false;
  float* grid_alloc[2] = {nullptr, nullptr};   // base of the peer's two grid allocations in MY address space
  char* window = nullptr;
  P2PBlob blob{};
};

}  // namespace

struct lbm_p2p {
  lbm_ctx* ctx = nullptr;
  int device = 0;                      // ctx's device, kept here: lbm_p2p_destroy must not depend on the context still existing
  int nranks = 1, rank = 0, south = 0, north = 0;
  bool tiles = false;                  // the context is a rank of the tile decomposition (lbm_create_tile): ghost columns, west / east neighbours
  int west = 0, east = 0;
  hipStream_t compute = nullptr, edge = nullptr;
  hipEvent_t edge_done = nullptr, interior_done = nullptr;
  bool edge_stream = true;             // edge rows on their own stream beside the interior launch
  int push_blocks_edge = 32;           // blocks of the push kernel when it runs beside the interior launch (LBM_P2P_PUSH_BLOCKS)
  char* window = nullptr;              // my exported window: header + reduce slots [2][nranks][cap]
  size_t window_bytes = 0, reduce_cap = 0, halo_bytes = 0;   // halo_bytes: one-step mode's incoming messages, [2 parities][2 dirs][3 * nxp] floats
  const char* window_kind = "coarse";
  int* err = nullptr;                  // host-mapped error word written by the wait kernels
  unsigned int* done = nullptr;        // block-done counter of the push kernel
  double* reduce_out = nullptr;        // folded global sums of one reduce round (host-mapped: the fold kernel writes, the host reads)
  double** d_slots = nullptr;          // device arrays of per-rank pointers, [2 parities][nranks]
  unsigned long long** d_flags = nullptr;
  std::vector<P2PPeer> peers;
  bool connected = false;
  unsigned long long epoch = 0, reduce_round = 0;
  long long timeout_ticks = 0;
  // lbm_p2p_set_profile: timing events around the launches of a run (a pool, grown on demand and reused)
  bool profile = false, phases_valid = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  hipEvent_t ev_reduce_end = nullptr;
  double phases[LBM_P2P_PHASES] = {};
};

namespace {

P2PWindowHeader* header_of(char* w) { return reinterpret_cast<P2PWindowHeader*>(w); }
double* slot_of(char* w, size_t halo_bytes, size_t cap, int nranks, int parity, int r)
{
  return reinterpret_cast<double*>(w + kP2PHeaderBytes + halo_bytes) + (static_cast<size_t>(parity) * nranks + r) * cap;
}
// one-step mode: where the halo message of epoch parity `parity` arriving from direction `dir` (0 = from the south
// neighbour) lies in a rank's window — three rows of nxp floats, as the step kernels read them (StepArgs::south_halo)
float* halo_slot(char* w, int nxp, int parity, int dir)
{
  return reinterpret_cast<float*>(w + kP2PHeaderBytes) + static_cast<size_t>(parity * 2 + dir) * 3 * nxp;
}

void p2p_unmap(lbm_p2p* t)
{
  for (P2PPeer& p : t->peers) {
    if (p.mapped && p.ipc) {
      for (float*& g : p.grid_alloc) { if (g) (void)hipIpcCloseMemHandle(g); g = nullptr; }
      if (p.window) (void)hipIpcCloseMemHandle(p.window);
    }
    p = P2PPeer{};
  }
}

// Profile mode: the next pooled timing event, recorded on `s` (nullptr when the profile is off or on failure).
hipEvent_t p2p_stamp(lbm_p2p* t, hipStream_t s)
{
  if (!t->profile) return nullptr;
  if (t->ev_used == t->ev_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    t->ev_pool.push_back(e);
  }
  hipEvent_t e = t->ev_pool[t->ev_used++];
  if (hipEventRecord(e, s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}

double p2p_us(hipEvent_t a, hipEvent_t b)
{
  float ms = 0.f;
  if (!a || !b || hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
  return static_cast<double>(ms) * 1e3;
}

struct P2PSpan { hipEvent_t begin = nullptr, end = nullptr; };

double p2p_avg_us(const std::vector<P2PSpan>& v, size_t from = 0)
{
  double s = 0.0;
  size_t n = 0;
  for (size_t i = from; i < v.size(); ++i)
    if (v[i].begin && v[i].end) { s += p2p_us(v[i].begin, v[i].end); ++n; }
  return n ? s / static_cast<double>(n) : 0.0;
}

// The k rows of the CURRENT grid that each neighbour needs for its next group of launches (k steps in all), into the k ghost rows
// next to its owned rows in the grid of the same parity — once both neighbours have said that they are done with the rows of the
// epoch before — flags := epoch; then (same kernel) wait for the neighbours' rows of that epoch to have arrived here.
// The next launch of this rank's context — the last of a group of several — tells both neighbours that this rank's ghost rows may be
// written for `epoch` (MultiArgs::ready): I am the south neighbour's NORTH neighbour and the north neighbour's SOUTH one.
void p2p_say_ready(lbm_p2p* t, unsigned long long epoch)
{
  lbm_ctx* c = t->ctx;
  c->ready_ptr[0] = &header_of(t->peers[t->south].window)->halo_ack[1];
  c->ready_ptr[1] = &header_of(t->peers[t->north].window)->halo_ack[0];
  if (t->tiles) {                      // ... and the west neighbour's EAST one, the east neighbour's WEST one (ghost columns)
    c->ready_ptr[2] = &header_of(t->peers[t->west].window)->halo_ack[3];
    c->ready_ptr[3] = &header_of(t->peers[t->east].window)->halo_ack[2];
  }
  c->ready_epoch = epoch;
  c->ready_wait = header_of(t->window)->halo_ack;
  c->ready_timeout_ticks = t->timeout_ticks;
  c->ready_err = t->err;
}

// (After a group of SEVERAL launches the push needs the neighbours' word that their ghost rows may be written — see
// P2PWindowHeader::halo_ack; it is said and awaited by the group's last launch, p2p_say_ready.  Never needed for the first push of a
// run: every rank's launches of the run before are complete when any rank leaves its reduction.)
// Tile decomposition: the k ghost columns on each side first (owned rows; push + flags + wait), so that the row push below sends them along.
int p2p_push_cols(lbm_p2p* t, unsigned long long epoch, int k, hipStream_t s)
{
  lbm_ctx* c = t->ctx;
  const P2PPeer& pw = t->peers[t->west];
  const P2PPeer& pe = t->peers[t->east];
  const int g = c->cur, gx = c->ghost_x;
  P2PPushColsArgs a{};
  a.src = c->grid[g]; a.ps = c->ps; a.src_w = c->p.nx;
  // west neighbour: the first k of its EAST ghost columns; east neighbour: the last k of its west ghost columns (its rows are mine:
  // the same y0, the same ghost rows)
  a.dst[0] = pw.grid_alloc[g] + 64 + static_cast<size_t>(pw.blob.ghost_rows) * pw.blob.w + (pw.blob.ghost_x + pw.blob.nxl);
  a.dst[1] = pe.grid_alloc[g] + 64 + static_cast<size_t>(pe.blob.ghost_rows) * pe.blob.w + (pe.blob.ghost_x - k);
  a.dst_ps[0] = pw.blob.ps; a.dst_ps[1] = pe.blob.ps;
  a.dst_w[0] = pw.blob.w; a.dst_w[1] = pe.blob.w;
  a.src_col[0] = gx;                    // my first k owned columns
  a.src_col[1] = gx + c->nxl - k;       // my last k owned columns
  a.row0 = c->ghost_rows; a.nrows = c->nyl; a.k = k;
  a.flag[0] = &header_of(pw.window)->halo_flag_x[1];      // my columns arrive from the west neighbour's EAST
  a.flag[1] = &header_of(pe.window)->halo_flag_x[0];
  a.parity_word[0] = &header_of(pw.window)->halo_parity_x[2 * 1 + (epoch & 1ull)];
  a.parity_word[1] = &header_of(pe.window)->halo_parity_x[2 * 0 + (epoch & 1ull)];
  a.epoch = epoch; a.parity = static_cast<unsigned long long>(g);
  a.done = t->done;
  a.wait_flags = header_of(t->window)->halo_flag_x;
  a.wait_parity = header_of(t->window)->halo_parity_x;
  a.timeout_ticks = t->timeout_ticks; a.err = t->err;
  auto all_mult = [&](int m) {
    return k % m == 0 && gx % m == 0 && c->nxl % m == 0 && c->p.nx % m == 0 && c->ps % m == 0 && pw.blob.w % m == 0 && pe.blob.w % m == 0 &&
           pw.blob.ps % m == 0 && pe.blob.ps % m == 0 && pw.blob.nxl % m == 0 && pw.blob.ghost_x % m == 0 && pe.blob.ghost_x % m == 0;
  };
  const int per = all_mult(4) ? 4 : all_mult(2) ? 2 : 1;
  const int work = 18 * c->nyl * (k / per);
  const dim3 grid(std::max(1, std::min(std::max(1, tune_env("LBM_P2P_PUSH_BLOCKS_SERIAL", kP2PPushBlocks)), (work + 1023) / 1024)));
  if (per == 4) hipLaunchKernelGGL(lbm_p2p_push_cols_kernel<f4>, grid, dim3(256), 0, s, a);
  else if (per == 2) hipLaunchKernelGGL(lbm_p2p_push_cols_kernel<f2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(lbm_p2p_push_cols_kernel<float>, grid, dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

int p2p_push(lbm_p2p* t, unsigned long long epoch, int k, hipStream_t s, bool exposed = false)
{
  lbm_ctx* c = t->ctx;
  if (t->tiles && p2p_push_cols(t, epoch, k, s)) return 1;
  if (c->ghost_rows == 0) return 0;      // a column block (px x 1 tiling): its rows wrap inside the launch, the column push is the whole exchange
  const P2PPeer& ps = t->peers[t->south];
  const P2PPeer& pn = t->peers[t->north];
  const int nx = c->p.nx, g = c->cur;
  P2PPushArgs a{};
  a.src = c->grid[g];
  a.ps = c->ps;
  // south neighbour: the first k of its TOP ghost rows (storage row ghost + nyl of ITS layout); north neighbour: the
  // last k of its bottom ghost rows (storage rows ghost - k .. ghost - 1)
  a.dst[0] = ps.grid_alloc[g] + 64 + static_cast<size_t>(ps.blob.ghost_rows + ps.blob.nyl) * nx;
  a.dst[1] = pn.grid_alloc[g] + 64 + static_cast<size_t>(pn.blob.ghost_rows - k) * nx;
  a.dst_ps[0] = ps.blob.ps;
  a.dst_ps[1] = pn.blob.ps;
  a.src_row[0] = static_cast<size_t>(c->ghost_rows);             // my first k owned rows
  a.src_row[1] = static_cast<size_t>(c->ghost_rows + c->nyl - k);   // my last k owned rows
  a.nfloats = k * nx;
  a.flag[0] = &header_of(ps.window)->halo_flag[1];       // my rows arrive from the south neighbour's NORTH
  a.flag[1] = &header_of(pn.window)->halo_flag[0];
  a.parity_word[0] = &header_of(ps.window)->halo_parity[2 * 1 + (epoch & 1ull)];
  a.parity_word[1] = &header_of(pn.window)->halo_parity[2 * 0 + (epoch & 1ull)];
  a.epoch = epoch;
  a.parity = static_cast<unsigned long long>(g);
  a.done = t->done;
  a.wait_flags = header_of(t->window)->halo_flag;
  a.wait_parity = header_of(t->window)->halo_parity;
  a.timeout_ticks = t->timeout_ticks;
  a.err = t->err;
  const bool wide = nx % 4 == 0 && ps.blob.ps % 4 == 0 && pn.blob.ps % 4 == 0 && c->ps % 4 == 0;    // 16-byte accesses (rows and planes 16-byte aligned on both sides)
  const int work = 18 * (a.nfloats / (wide ? 4 : 2));
  // at most 64 blocks (every block ends with an L2 write-back towards the peers), each lane moving up to four
  // float2's per pass
  // (`exposed`: the push before the first macro-step of a run in the serial schedule, which nothing overlaps)
  // (pushes of more than four rows, round 4: twice the blocks — 8 rows beside the interior launch of 8192 x 1024 rows, us/step at 20 / 200
  // steps per run for 16 / 32 / 64 / 128 blocks: 51.9 / 49.4, 46.9 / 43.5, 47.0 / 43.1, 48.6 / 44.0; profiles/r04/ring_push_blocks.txt)
  const int max_blocks = (t->edge_stream && !exposed) ? t->push_blocks_edge * (k > 4 ? 2 : 1) : std::max(1, tune_env("LBM_P2P_PUSH_BLOCKS_SERIAL", kP2PPushBlocks));
  // (exposed push of a 1024 x 128-row rank, 16 rows: us/step at 200 steps for 32 / 64 / 72 blocks of 1024 vectors 3.59 / 3.50 - 3.55 / 3.51, 144 blocks of
  // 512 vectors 3.61, 288 of 256 3.90 — the block count is not the lever; profiles/r04/ring_push_blocks_serial.txt)
  const int per_block = std::max(256, tune_env("LBM_P2P_PUSH_WORK_PER_BLOCK", 1024));
  const dim3 grid(std::max(1, std::min(max_blocks, (work + per_block - 1) / per_block)));
  if (wide) hipLaunchKernelGGL(lbm_p2p_push_kernel<f4>, grid, dim3(256), 0, s, a, nx);
  else hipLaunchKernelGGL(lbm_p2p_push_kernel<f2>, grid, dim3(256), 0, s, a, nx);
  HIP_TRY(hipGetLastError());
  return 0;
}

// MPI_Reduce (:396): all-gather of the per-step sums into every rank's window, local sum in rank order.
int p2p_reduce(lbm_p2p* t, int n_steps, double* tot_u_per_step)
{
  lbm_ctx* c = t->ctx;
  hipStream_t cs = t->compute;
  for (int t0 = 0; t0 < n_steps; t0 += static_cast<int>(t->reduce_cap)) {
    const int n = std::min(n_steps - t0, static_cast<int>(t->reduce_cap));
    const unsigned long long round = ++t->reduce_round;
    const int par = static_cast<int>(round & 1);
    P2PReduceArgs a{};
    a.sums = c->sums + t0; a.n = n; a.nranks = t->nranks;
    a.slots = t->d_slots + static_cast<size_t>(par) * t->nranks;
    a.flags = t->d_flags;
    a.round = round;
    a.my_flags = header_of(t->window)->reduce_flag;
    a.my_slots = slot_of(t->window, t->halo_bytes, t->reduce_cap, t->nranks, par, 0);
    a.slot_stride = t->reduce_cap;
    a.out = t->reduce_out;
    a.timeout_ticks = t->timeout_ticks;
    a.err = t->err;
    hipLaunchKernelGGL(lbm_p2p_allreduce_kernel, dim3(t->nranks), dim3(256), 0, cs, a);
    HIP_TRY(hipGetLastError());
    t->ev_reduce_end = p2p_stamp(t, cs);
    HIP_TRY(stream_wait(cs));
    std::memcpy(tot_u_per_step + t0, t->reduce_out, sizeof(double) * n);     // host-mapped: the kernel wrote it in place
  }
  return 0;
}

int p2p_check_error(lbm_p2p* t)
{
  if (*t->err == 0) return 0;
  const int e = *t->err;
  const char* who = e >= 100 ? "a neighbour's grids are out of step with this rank's (different number of steps run ?)"
                             : "a peer's data did not arrive in time (peer failed or never started the run ?)";
  // what this rank's window held when it gave up: the epochs of the rows / columns that have arrived and of the neighbours' "ready" words
  std::string words;
  P2PWindowHeader h{};
  if (hipMemcpy(&h, t->window, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
    words = "; epoch " + std::to_string(t->epoch) + ", rows from south / north " + std::to_string(h.halo_flag[0]) + " / " + std::to_string(h.halo_flag[1]) +
            ", ready south / north " + std::to_string(h.halo_ack[0]) + " / " + std::to_string(h.halo_ack[1]);
    if (t->tiles)
      words += ", columns from west / east " + std::to_string(h.halo_flag_x[0]) + " / " + std::to_string(h.halo_flag_x[1]) + ", ready west / east " +
               std::to_string(h.halo_ack[2]) + " / " + std::to_string(h.halo_ack[3]);
  }
  (void)hipGetLastError();
  lbm_internal::set_error("lbm_p2p_run: rank " + std::to_string(t->rank) + ": " + who + " [code " + std::to_string(e) + words + "]");
  return 1;
}

// One-step mode (contexts that are not eligible for K-step mode: fewer than 32 rows on some rank, odd or short rows):
// the three populations that cross each cut travel every step, stored by the boundary launch straight into the
// neighbours' windows (their `south_halo` / `north_halo` of the next step), two slots per direction by epoch parity.
//   [signal + wait kernel: my messages of step t are out; theirs have arrived]   MPI_Startall / MPI_Waitall (:327,364)
//   interior rows t (:350), boundary rows t (:365-366) -> writes the messages of step t+1
int p2p_run_one_step(lbm_p2p* t, int n_steps, double* tot_u_per_step)
{
  lbm_ctx* c = t->ctx;
  hipStream_t cs = t->compute;
  const int nxp = c->nxp;
  P2PPeer& ps = t->peers[t->south];
  P2PPeer& pn = t->peers[t->north];
  auto bind = [&](unsigned long long e_recv, unsigned long long e_send) {
    c->recv[0] = halo_slot(t->window, nxp, static_cast<int>(e_recv & 1), 0);
    c->recv[1] = halo_slot(t->window, nxp, static_cast<int>(e_recv & 1), 1);
    c->send[0] = halo_slot(ps.window, nxp, static_cast<int>(e_send & 1), 1);   // my southward rows are the south neighbour's NORTH halo
    c->send[1] = halo_slot(pn.window, nxp, static_cast<int>(e_send & 1), 0);
  };
  c->release_sends = true;
  unsigned long long epoch = t->epoch + 1;
  bind(epoch, epoch);
  const std::chrono::steady_clock::time_point h0 = std::chrono::steady_clock::now();
  hipEvent_t e_run0 = p2p_stamp(t, cs);
  if (lbm_step_prepare(c, n_steps, cs)) return 1;              // step-0 accelerate_flow + the messages of the first step
  hipEvent_t e_steps0 = p2p_stamp(t, cs);
  P2PWindowHeader* mine = header_of(t->window);
  for (int step = 0; step < n_steps; ++step, ++epoch) {
    hipLaunchKernelGGL(lbm_p2p_signal_wait_kernel, dim3(1), dim3(64), 0, cs, &header_of(ps.window)->halo_flag[1], &header_of(pn.window)->halo_flag[0],
                       &header_of(ps.window)->halo_parity[2 * 1 + (epoch & 1ull)], &header_of(pn.window)->halo_parity[2 * 0 + (epoch & 1ull)],
                       mine->halo_flag, mine->halo_parity, epoch, static_cast<unsigned long long>(c->cur), t->timeout_ticks, t->err);
    HIP_TRY(hipGetLastError());
    bind(epoch, epoch + 1);
    if (lbm_step_interior(c, cs) || lbm_step_boundary(c, cs) || lbm_step_finish(c, cs)) { (void)hipStreamSynchronize(cs); return 1; }
  }
  t->epoch = epoch - 1;
  hipEvent_t e_steps1 = p2p_stamp(t, cs);
  const std::chrono::steady_clock::time_point h_enq = std::chrono::steady_clock::now();
  if (p2p_reduce(t, n_steps, tot_u_per_step)) return 1;
  if (p2p_check_error(t)) return 1;
  if (t->profile) {
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    double* ph = t->phases;
    std::fill(ph, ph + LBM_P2P_PHASES, 0.0);
    ph[0] = us(h0, std::chrono::steady_clock::now());
    ph[1] = us(h0, h_enq);
    ph[2] = p2p_us(e_run0, t->ev_reduce_end);
    ph[3] = p2p_us(e_run0, e_steps0);
    ph[4] = p2p_us(e_steps0, e_steps1);
    ph[5] = p2p_us(e_steps1, t->ev_reduce_end);
    ph[6] = n_steps;
    ph[7] = ph[4] / n_steps;
    ph[13] = ph[0] - ph[2];
    t->phases_valid = true;
  }
  return 0;
}

}  // namespace

extern "C" {

int lbm_p2p_create(lbm_p2p** out, lbm_ctx* ctx, int nranks, int rank)
{
  if (!out || !ctx || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) { lbm_internal::set_error("lbm_p2p_create: bad argument (1..64 ranks)"); return 1; }
  *out = nullptr;
  if (ctx->self_periodic) {
    lbm_internal::set_error("lbm_p2p_create: the context is a self-contained periodic domain (create it with lbm_create_rank, or with LBM_FLAG_FORCE_HALO)");
    return 1;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  lbm_p2p* t = new lbm_p2p();
  t->ctx = ctx;
  t->device = ctx->device;
  t->nranks = nranks;
  t->rank = rank;
  t->south = (rank + nranks - 1) % nranks;   // `top`    d2q9-bgk.c:245-246
  t->north = (rank + 1) % nranks;            // `bottom` d2q9-bgk.c:247
  if (ctx->ghost_x > 0) {                    // tile decomposition: rank = ry * px + rx, periodic in both directions
    const int px = ctx->tiles_px, py = ctx->tiles_py, rx = ctx->tile_rx, ry = ctx->tile_ry;
    if (nranks != px * py || rank != ry * px + rx) { lbm_internal::set_error("lbm_p2p_create: the context is rank " + std::to_string(ry * px + rx) + " of " + std::to_string(px * py) + " tiles"); delete t; return 1; }
    t->tiles = true;
    t->south = ((ry + py - 1) % py) * px + rx;
    t->north = ((ry + 1) % py) * px + rx;
    t->west = ry * px + (rx + px - 1) % px;
    t->east = ry * px + (rx + 1) % px;
  }
  t->compute = ctx->stream;
  t->peers.resize(nranks);
  t->timeout_ticks = static_cast<long long>(tune_env("LBM_P2P_TIMEOUT_MS", 30000)) * 100000LL;   // wall_clock64: 100 MHz
  // default by size, as the RCCL loop: an own stream for the edge rows pays once the interior launch is long
  // enough to cover two cross-queue waits
  t->edge_stream = ctx->ncells >= (size_t(1) << 21);
  // Tile ranks: the interior is the rectangle of tiles inside the rim of tile rows AND tile columns that read exchanged cells (macro_rects).
  // That rim is a round of blocks by itself (a 2048 x 4096 block: 422 of its 5874 tiles, 46 us) where a row block's is two tile rows, and the
  // exchange it hides is short (31 us per 8 steps): 1-rank rings, us/step serial / edge stream: 2048 x 4096 45.5 / 46.5, 4096 x 4096 82.0 / 82.4,
  // 4096 x 8192 158.0 / 155.5 (profiles/r04/tile_ring_*_{serial,edge_stream}.json) — from 2^25 cells up
  if (t->tiles) t->edge_stream = ctx->ncells >= (size_t(1) << 25);
  if (const char* sched = std::getenv("LBM_P2P_SCHEDULE")) {
    if (std::string(sched) == "serial") t->edge_stream = false;
    if (std::string(sched) == "edge") t->edge_stream = true;
  }
  // beside the interior launch the push has a whole macro-step to finish, and every block of it takes a CU slot and
  // an L2 write-back away from that launch: us/step on a 1-rank ring of 8192 x 1024 rows for 8 / 12 / 16 / 32 / 64
  // blocks 60.1 / 52.8 / 51.7 / 52.1 / 53.6 (8192 x 2048: 94.8 / - / 95.2 / 95.1 / 96.7)
  // 32, not the 16 that measure best on a self-ring: there the push kernel takes ~110 of the ~147 us it has, and a
  // real link is slower than local memory — a push that outlasts the interior launch would be the critical path
  t->push_blocks_edge = std::max(1, tune_env("LBM_P2P_PUSH_BLOCKS", 32));
  auto fail = [&]() { lbm_p2p_destroy(t); return 1; };
#define P2P_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      lbm_internal::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));            \
      return fail();                                                                         \
    }                                                                                        \
  } while (0)
  P2P_TRY(hipEventCreateWithFlags(&t->edge_done, hipEventDisableTiming));
  P2P_TRY(hipEventCreateWithFlags(&t->interior_done, hipEventDisableTiming));
  // exported window: flags + reduce slots.  Uncached device memory, so that a flag raised by a peer is seen
  // by a kernel that is already running here; fine-grained, then ordinary memory as fall-backs.
  t->reduce_cap = static_cast<size_t>(std::max(ctx->p.max_iters, 4096));
  // (the one-step mode's halo slots; the reduce slots behind them must lie at the same offset in EVERY rank's window: tile ranks — always
  // K-step mode, storage rows of different widths where the column blocks are uneven — keep none)
  t->halo_bytes = t->tiles ? 0 : round_up(sizeof(float) * 2 * 2 * 3 * static_cast<size_t>(ctx->nxp), 256);
  t->window_bytes = kP2PHeaderBytes + t->halo_bytes + sizeof(double) * 2 * nranks * t->reduce_cap;
  // The protocol needs a window whose flags a running kernel sees change and whose one-step halo slots need no
  // cache maintenance on the reader's side: uncached, or fine-grained as the fall-back.  Ordinary (coarse-grained)
  // device memory guarantees neither — the polls could spin on a cached line — so when both allocations fail the
  // transport is NOT created and the caller falls back to the RCCL loop.  LBM_P2P_WINDOW=2 asks for the coarse
  // window explicitly (experiments on one GPU only).
  void* w = nullptr;
  const int want = tune_env("LBM_P2P_WINDOW", 0);
  if (want <= 0 && hipExtMallocWithFlags(&w, t->window_bytes, hipDeviceMallocUncached) == hipSuccess) t->window_kind = "uncached";
  else if (want <= 1 && hipExtMallocWithFlags(&w, t->window_bytes, hipDeviceMallocFinegrained) == hipSuccess) t->window_kind = "fine-grained";
  else if (want >= 2) { (void)hipGetLastError(); P2P_TRY(hipMalloc(&w, t->window_bytes)); t->window_kind = "coarse"; }
  else {
    (void)hipGetLastError();
    lbm_internal::set_error("lbm_p2p_create: neither an uncached nor a fine-grained device allocation is available for the flag window "
                            "(hipExtMallocWithFlags failed); the peer-to-peer transport does not run on ordinary device memory");
    return fail();
  }
  (void)hipGetLastError();
  t->window = static_cast<char*>(w);
  P2P_TRY(hipMemset(t->window, 0, t->window_bytes));
  P2P_TRY(hipHostMalloc(reinterpret_cast<void**>(&t->err), sizeof(int), hipHostMallocMapped));
  *t->err = 0;
  P2P_TRY(hipMalloc(&t->done, 4 * sizeof(unsigned int)));
  P2P_TRY(hipMemset(t->done, 0, 4 * sizeof(unsigned int)));
  P2P_TRY(hipHostMalloc(reinterpret_cast<void**>(&t->reduce_out), sizeof(double) * t->reduce_cap, hipHostMallocMapped));
  P2P_TRY(hipMalloc(&t->d_slots, sizeof(double*) * 2 * nranks));
  P2P_TRY(hipMalloc(&t->d_flags, sizeof(unsigned long long*) * nranks));
  P2P_TRY(hipDeviceSynchronize());
#undef P2P_TRY
  *out = t;
  return 0;
}

int lbm_p2p_handle(lbm_p2p* t, void* blob_out)
{
  if (!t || !blob_out) { lbm_internal::set_error("lbm_p2p_handle: null argument"); return 1; }
  lbm_ctx* c = t->ctx;
  HIP_TRY(hipSetDevice(c->device));
  P2PBlob b{};
  b.magic = kP2PMagic; b.version = LBM_ABI_VERSION;
  b.pid = static_cast<int32_t>(getpid()); b.device = c->device; b.nranks = t->nranks; b.rank = t->rank;
  b.nx = c->nx_global; b.ny = c->p.ny; b.y0 = c->y0; b.nyl = c->nyl; b.ghost = c->ghost; b.K = c->multi_K; b.cur = c->cur; b.group = c->group_max;
  b.ghost_rows = c->ghost_rows;
  b.w = c->p.nx; b.x0 = c->x0; b.nxl = c->nxl; b.ghost_x = c->ghost_x; b.px = t->tiles ? c->tiles_px : 1; b.py = t->tiles ? c->tiles_py : t->nranks;
  b.ps = c->ps; b.window_bytes = t->window_bytes; b.reduce_cap = t->reduce_cap;
  // IPC handles serve peers in OTHER processes; contexts of one process use the raw pointers, so a
  // runtime that cannot export a handle only rules out the multi-process form (checked at connect)
  b.ipc_ok = 1;
  for (int g = 0; g < 2; ++g) {
    b.grid_ptr[g] = reinterpret_cast<uint64_t>(c->grid_alloc[g]);
    if (hipIpcGetMemHandle(&b.grid_h[g], c->grid_alloc[g]) != hipSuccess) b.ipc_ok = 0;
  }
  b.window_ptr = reinterpret_cast<uint64_t>(t->window);
  if (hipIpcGetMemHandle(&b.window_h, t->window) != hipSuccess) b.ipc_ok = 0;
  (void)hipGetLastError();
  (void)gethostname(b.host, sizeof b.host - 1);
  std::memset(blob_out, 0, LBM_P2P_HANDLE_BYTES);
  std::memcpy(blob_out, &b, sizeof b);
  return 0;
}

int lbm_p2p_connect(lbm_p2p* t, const void* blobs)
{
  if (!t || !blobs) { lbm_internal::set_error("lbm_p2p_connect: null argument"); return 1; }
  lbm_ctx* c = t->ctx;
  HIP_TRY(hipSetDevice(c->device));
  p2p_unmap(t);
  t->connected = false;
  const char* base = static_cast<const char*>(blobs);
  const int32_t me = static_cast<int32_t>(getpid());
  char host[64] = {0};
  (void)gethostname(host, sizeof host - 1);
  long long cells = 0;
  for (int r = 0; r < t->nranks; ++r) {
    P2PPeer& p = t->peers[r];
    std::memcpy(&p.blob, base + static_cast<size_t>(r) * LBM_P2P_HANDLE_BYTES, sizeof(P2PBlob));
    const P2PBlob& b = p.blob;
    // every rank must run the same K-step layout: a mismatch would mean different message sizes and
    // exchange cadence, i.e. a hang or corrupted ghost rows (an error here instead)
    if (b.magic != kP2PMagic || b.version != LBM_ABI_VERSION || b.rank != r || b.nranks != t->nranks) {
      lbm_internal::set_error("lbm_p2p_connect: handle " + std::to_string(r) + " is not rank " + std::to_string(r) + " of this run");
      return 1;
    }
    if (b.nx != c->nx_global || b.ny != c->p.ny || b.K != c->multi_K || b.ghost != c->ghost || b.group != c->group_max || b.cur != c->cur || b.reduce_cap != t->reduce_cap ||
        b.ghost_x != c->ghost_x || b.ghost_rows != c->ghost_rows || b.px != (t->tiles ? c->tiles_px : 1) || b.py != (t->tiles ? c->tiles_py : t->nranks)) {
      lbm_internal::set_error("lbm_p2p_connect: rank " + std::to_string(r) + " runs a different layout (nx " + std::to_string(b.nx) + ", ny " +
                              std::to_string(b.ny) + ", K " + std::to_string(b.K) + ", " + std::to_string(b.ghost) + " ghost rows, " + std::to_string(b.group) +
                              " launches per exchange) than rank " + std::to_string(t->rank) + " (K " + std::to_string(c->multi_K) + ", " + std::to_string(c->ghost) +
                              ", " + std::to_string(c->group_max) + "): create every rank with lbm_create_rank");
      return 1;
    }
    if (std::strncmp(b.host, host, sizeof host) != 0) {
      lbm_internal::set_error("lbm_p2p_connect: rank " + std::to_string(r) + " runs on another host; peer-to-peer halos need one node");
      return 1;
    }
    cells += static_cast<long long>(b.nyl) * b.nxl;
  }
  if (cells != static_cast<long long>(c->p.ny) * c->nx_global) { lbm_internal::set_error("lbm_p2p_connect: the ranks' blocks do not add up to the grid"); return 1; }
  for (int r = 0; r < t->nranks; ++r) {
    P2PPeer& p = t->peers[r];
    const P2PBlob& b = p.blob;
    const bool neighbour = r == t->south || r == t->north || (t->tiles && (r == t->west || r == t->east));
    if (r == t->rank) {                                        // myself: a 1-rank ring, or my own reduce slot
      for (int g = 0; g < 2; ++g) p.grid_alloc[g] = c->grid_alloc[g];
      p.window = t->window;
    } else if (b.pid == me) {                                  // another context of this process: plain pointers
      // Ranks of one process on ONE device share that process's few hardware queues: a wait kernel could end
      // up in front of the very push it waits for.  One stream per rank keeps that from happening in practice
      // (the single-process host is meant for one rank per GPU, where the question does not arise).
      if (b.device == c->device) t->edge_stream = false;
      if (b.device != c->device) {
        hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
          lbm_internal::set_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
          return 1;
        }
        (void)hipGetLastError();
      }
      for (int g = 0; g < 2; ++g) p.grid_alloc[g] = reinterpret_cast<float*>(b.grid_ptr[g]);
      p.window = reinterpret_cast<char*>(b.window_ptr);
    } else {                                                   // another process: map its memory
      if (!b.ipc_ok) {
        lbm_internal::set_error("lbm_p2p_connect: rank " + std::to_string(r) + " could not export IPC handles for its memory (hipIpcGetMemHandle failed; is HSA_ENABLE_IPC_MODE_LEGACY=0 set ?)");
        return 1;
      }
      p.ipc = true;
      void* q = nullptr;
      HIP_TRY(hipIpcOpenMemHandle(&q, b.window_h, hipIpcMemLazyEnablePeerAccess));
      p.window = static_cast<char*>(q);
      p.mapped = true;
      if (neighbour) {
        for (int g = 0; g < 2; ++g) {
          HIP_TRY(hipIpcOpenMemHandle(&q, b.grid_h[g], hipIpcMemLazyEnablePeerAccess));
          p.grid_alloc[g] = static_cast<float*>(q);
        }
      }
    }
    p.mapped = true;
  }
  // the edge stream exists only where the schedule uses it: every stream takes a share of the process's few
  // hardware queues, and ranks of one process on one device must not share a queue (see above)
  if (c->ghost == 0) t->edge_stream = false;               // one-step mode runs on the compute stream
  if (t->edge_stream && !t->edge) HIP_TRY(hipStreamCreateWithFlags(&t->edge, hipStreamNonBlocking));
  // device tables for the reduce kernels: where my sums go in every rank's window, and my flag there
  std::vector<double*> slots(static_cast<size_t>(2) * t->nranks);
  std::vector<unsigned long long*> flags(t->nranks);
  for (int r = 0; r < t->nranks; ++r) {
    for (int par = 0; par < 2; ++par) slots[static_cast<size_t>(par) * t->nranks + r] = slot_of(t->peers[r].window, t->halo_bytes, t->reduce_cap, t->nranks, par, t->rank);
    flags[r] = &header_of(t->peers[r].window)->reduce_flag[t->rank];
  }
  HIP_TRY(hipMemcpy(t->d_slots, slots.data(), sizeof(double*) * slots.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(t->d_flags, flags.data(), sizeof(unsigned long long*) * flags.size(), hipMemcpyHostToDevice));
  t->connected = true;
  return 0;
}

int lbm_p2p_disconnect(lbm_p2p* t)
{
  if (!t) return 0;
  (void)hipSetDevice(t->device);
  if (t->edge) (void)hipStreamSynchronize(t->edge);
  p2p_unmap(t);
  t->connected = false;
  return 0;
}

int lbm_p2p_destroy(lbm_p2p* t)
{
  if (!t) return 0;
  // (every lbm_p2p_run returns with its streams drained; the compute stream belongs to the context, which may
  // already be gone, so only what this object owns is touched here)
  (void)hipSetDevice(t->device);
  if (t->edge) (void)hipStreamSynchronize(t->edge);
  p2p_unmap(t);
  if (t->window) (void)hipFree(t->window);
  if (t->err) (void)hipHostFree(t->err);
  if (t->done) (void)hipFree(t->done);
  if (t->reduce_out) (void)hipHostFree(t->reduce_out);
  if (t->d_slots) (void)hipFree(t->d_slots);
  if (t->d_flags) (void)hipFree(t->d_flags);
  if (t->edge_done) (void)hipEventDestroy(t->edge_done);
  if (t->interior_done) (void)hipEventDestroy(t->interior_done);
  for (hipEvent_t e : t->ev_pool) (void)hipEventDestroy(e);
  if (t->edge) (void)hipStreamDestroy(t->edge);
  delete t;
  return 0;
}

int lbm_p2p_run(lbm_p2p* t, int n_steps, double* tot_u_per_step)
{
  using clock = std::chrono::steady_clock;
  if (!t || n_steps < 0 || (n_steps > 0 && !tot_u_per_step)) { lbm_internal::set_error("lbm_p2p_run: bad argument"); return 1; }
  if (!t->connected) { lbm_internal::set_error("lbm_p2p_run: call lbm_p2p_connect first"); return 1; }
  if (n_steps == 0) return 0;
  lbm_ctx* c = t->ctx;
  HIP_TRY(hipSetDevice(c->device));
  // the error word is sticky: after a time-out the neighbours' epochs no longer agree with this rank's, and every
  // wait of a later run would return at once — fail here instead of stepping without the halo rows
  if (*t->err != 0) {
    lbm_internal::set_error("lbm_p2p_run: rank " + std::to_string(t->rank) + ": an earlier run on this transport failed [code " +
                            std::to_string(*t->err) + "]; destroy it and create a new one");
    return 1;
  }
  const clock::time_point h0 = clock::now();
  t->ev_used = 0;
  t->phases_valid = false;
  t->ev_reduce_end = nullptr;
  if (c->ghost == 0) return p2p_run_one_step(t, n_steps, tot_u_per_step);
  hipStream_t cs = t->compute, es = t->edge_stream ? t->edge : t->compute;
  // error exits leave both streams drained: events recorded on one may be pending on the other
  auto bail = [&]() {
    (void)hipStreamSynchronize(cs);
    if (t->edge_stream) (void)hipStreamSynchronize(es);
    return 1;
  };
#define P2P_RUN_TRY(expr)                                                                    \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      lbm_internal::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));            \
      return bail();                                                                         \
    }                                                                                        \
  } while (0)
  hipEvent_t e_run0 = p2p_stamp(t, cs);
  if (begin_run(c, n_steps, cs)) return bail();                // step-0 accelerate_flow (d2q9-bgk.c:345-348)
  // The rows my neighbours need for the first macro-step (the state may have been set since the last run, and the
  // step-0 accelerate_flow has just changed row ny-2).  Edge-stream schedule: the push goes to the edge stream behind
  // the accelerate kernel and runs BESIDE interior launch 0, which reads no ghost row — only the edge launch waits
  // for it, as in every later macro-step.  Serial schedule: right behind the accelerate kernel with the full
  // complement of blocks (nothing overlaps it).
  unsigned long long epoch = t->epoch + 1;
  std::vector<P2PSpan> sp_interior, sp_edge, sp_push, sp_whole;
  GroupPlan g = plan_group(c, n_steps);                        // the same sequence of groups on every rank
  {
    P2PSpan sp;
    if (t->edge_stream) {
      P2P_RUN_TRY(hipEventRecord(t->interior_done, cs));       // "the accelerated state is ready"
      P2P_RUN_TRY(hipStreamWaitEvent(es, t->interior_done, 0));
      sp.begin = p2p_stamp(t, es);
      if (p2p_push(t, epoch, g.total, es)) return bail();
      sp.end = p2p_stamp(t, es);
    } else {
      sp.begin = p2p_stamp(t, cs);
      if (p2p_push(t, epoch, g.total, cs, /*exposed=*/true)) return bail();
      sp.end = p2p_stamp(t, cs);
    }
    if (t->profile) sp_push.push_back(sp);
  }
  hipEvent_t e_steps0 = p2p_stamp(t, cs);
  int groups = 0, launches = 0, prev_n = 0;
  bool es_has_waited = true;                                   // the edge stream has waited for the compute stream's last launch (run start: above)
  for (int done = 0; done < n_steps; ++epoch, ++groups) {
    const bool more = done + g.total < n_steps;
    const MacroRows rows = macro_rows(c, g.k[0], g.ext(0));    // (the tile height follows the steps of the launch)
    if (t->edge_stream) {
      if (rows.interior_rows > 0) {                            // :350, beside the exchange
        // sources: the previous group's last launch — on this stream when that group had several launches, else its edge rows
        if (groups > 0 && prev_n == 1) P2P_RUN_TRY(hipStreamWaitEvent(cs, t->edge_done, 0));
        P2PSpan sp;
        sp.begin = p2p_stamp(t, cs);
        launch_group_interior(c, g, more, cs);
        sp.end = p2p_stamp(t, cs);
        if (t->profile) sp_interior.push_back(sp);
        c->n_prev = 0;
      }
      // MPI_Waitall (:364) happened on the device, at the end of the push kernel that precedes this launch
      if (!es_has_waited) P2P_RUN_TRY(hipStreamWaitEvent(es, t->interior_done, 0));   // the previous group's launches on the compute stream
      P2PSpan sp;
      sp.begin = p2p_stamp(t, es);
      launch_group_edge(c, g, more, es);                       // :365-366
      sp.end = p2p_stamp(t, es);
      if (t->profile) sp_edge.push_back(sp);
      c->n_prev = 0;
      P2P_RUN_TRY(hipGetLastError());
      P2P_RUN_TRY(hipEventRecord(t->edge_done, es));
      group_launch_done(c, g, 0, 2);
      for (int i = 1; i < g.n; ++i) {                          // over all tiles: nothing exchanged is read
        if (i == 1) P2P_RUN_TRY(hipStreamWaitEvent(cs, t->edge_done, 0));
        if (i == g.n - 1 && more) p2p_say_ready(t, epoch + 1);
        P2PSpan sw;
        sw.begin = p2p_stamp(t, cs);
        launch_group_whole(c, g, i, more, cs);
        sw.end = p2p_stamp(t, cs);
        if (t->profile) sp_whole.push_back(sw);
        P2P_RUN_TRY(hipGetLastError());
        group_launch_done(c, g, i, 1);
      }
      P2P_RUN_TRY(hipEventRecord(t->interior_done, cs));       // "the compute stream's launches of this group"
      es_has_waited = false;
    } else {
      for (int i = 0; i < g.n; ++i) {
        if (g.n > 1 && i == g.n - 1 && more) p2p_say_ready(t, epoch + 1);
        P2PSpan sp;
        sp.begin = p2p_stamp(t, cs);
        launch_group_whole(c, g, i, more, cs);
        sp.end = p2p_stamp(t, cs);
        if (t->profile) (i == 0 ? sp_interior : sp_whole).push_back(sp);
        P2P_RUN_TRY(hipGetLastError());
        group_launch_done(c, g, i, 1);
      }
    }
    launches += g.n;
    done += g.total;
    prev_n = g.n;
    if (more) {                                                // MPI_Startall (:327) for the next group
      const GroupPlan next = plan_group(c, n_steps - done);
      if (t->edge_stream) {
        // the rows to push are the last launch's: its edge rows when the group was one launch and those tile rows hold all
        // next.total rows of either side (then the push need not wait for the interior launch); else the compute stream's
        const bool edge_rows_suffice = g.n == 1 && !t->tiles && (c->ghost_rows - g.ext(0)) + rows.bottom_edge_rows * multi_ty(g.k[0], c->multi_geom) >= c->ghost_rows + next.total &&
                                       (c->ghost_rows - g.ext(0)) + (rows.bottom_edge_rows + rows.interior_rows) * multi_ty(g.k[0], c->multi_geom) <= c->ghost_rows + c->nyl - next.total;
        if (!edge_rows_suffice) { P2P_RUN_TRY(hipStreamWaitEvent(es, t->interior_done, 0)); es_has_waited = true; }
      }
      P2PSpan sp;
      sp.begin = p2p_stamp(t, es);
      if (p2p_push(t, epoch + 1, next.total, es)) return bail();
      sp.end = p2p_stamp(t, es);
      if (t->profile) sp_push.push_back(sp);
      g = next;
    }
  }
  t->epoch = epoch - 1;
  if (t->edge_stream) P2P_RUN_TRY(hipStreamWaitEvent(cs, t->edge_done, 0));
  P2P_RUN_TRY(hipEventRecord(c->ev_end, cs));
  hipEvent_t e_steps1 = p2p_stamp(t, cs);
  c->ev_launches = c->ev_tile_launches;
  c->ev_valid = true;
  if (fold_last(c, cs, /*final=*/true)) return bail();
  const clock::time_point h_enq = clock::now();
  if (p2p_reduce(t, n_steps, tot_u_per_step)) return bail();
  if (t->edge_stream) P2P_RUN_TRY(hipStreamSynchronize(es));
#undef P2P_RUN_TRY
  if (p2p_check_error(t)) return 1;
  if (t->profile) {
    const clock::time_point h1 = clock::now();
    auto us = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    double* ph = t->phases;
    std::fill(ph, ph + LBM_P2P_PHASES, 0.0);
    ph[0] = us(h0, h1);
    ph[1] = us(h0, h_enq);
    ph[2] = p2p_us(e_run0, t->ev_reduce_end);
    ph[3] = p2p_us(e_run0, e_steps0);
    ph[4] = p2p_us(e_steps0, e_steps1);
    ph[5] = p2p_us(e_steps1, t->ev_reduce_end);
    ph[6] = groups;
    ph[7] = groups ? ph[4] / groups : 0.0;
    if (sp_interior.size() >= 3) ph[8] = p2p_us(sp_interior[1].begin, sp_interior.back().begin) / static_cast<double>(sp_interior.size() - 2);
    ph[9] = p2p_avg_us(sp_interior);
    ph[10] = p2p_avg_us(sp_edge);
    ph[11] = sp_push.empty() ? 0.0 : p2p_us(sp_push[0].begin, sp_push[0].end);
    ph[12] = p2p_avg_us(sp_push, 1);
    ph[13] = ph[0] - ph[2];
    ph[14] = launches;
    ph[15] = p2p_avg_us(sp_whole);
    t->phases_valid = true;
  }
  return 0;
}

int lbm_p2p_set_profile(lbm_p2p* t, int on)
{
  if (!t) { lbm_internal::set_error("lbm_p2p_set_profile: null argument"); return 1; }
  t->profile = on != 0;
  if (!t->profile) t->phases_valid = false;
  return 0;
}

int lbm_p2p_phases(const lbm_p2p* t, double* values)
{
  if (!t || !values) { lbm_internal::set_error("lbm_p2p_phases: null argument"); return 1; }
  if (!t->phases_valid) { lbm_internal::set_error("lbm_p2p_phases: no profiled run (lbm_p2p_set_profile(t, 1), then lbm_p2p_run)"); return 1; }
  std::memcpy(values, t->phases, sizeof t->phases);
  return 0;
}

const char* lbm_p2p_phase_name(int i)
{
  static const char* const names[] = {"host_total", "host_enqueue", "device_span", "setup", "steps", "reduce", "macro_steps", "macro_step_avg",
                                      "macro_step_steady", "interior_avg", "edge_avg", "push_first", "push_avg", "host_overhead", "launches", "whole_avg"};
  return (i >= 0 && i < static_cast<int>(sizeof names / sizeof names[0])) ? names[i] : nullptr;
}

int lbm_p2p_describe(const lbm_p2p* t, char* text, size_t len)
{
  if (!t || !text || len == 0) { lbm_internal::set_error("lbm_p2p_describe: null argument"); return 1; }
  const char* reach = "self";
  if (t->nranks > 1) {
    bool ipc = t->connected && (t->peers[t->north].ipc || t->peers[t->south].ipc);
    if (t->tiles) ipc = ipc || (t->connected && (t->peers[t->west].ipc || t->peers[t->east].ipc));   // (a 2 x 1 grid: north and south are the rank itself)
    reach = ipc ? "ipc" : "in-process";
  }
  if (t->ctx->ghost == 0) {
    std::snprintf(text, len, "window %s; neighbours %s; one-step mode", t->window_kind, reach);
    return 0;
  }
  int n = std::snprintf(text, len, "window %s; neighbours %s; schedule %s; K %d; ghost rows %d; launches per exchange %d", t->window_kind, reach,
                        t->edge_stream ? "edge stream" : "serial", t->ctx->multi_K, t->ctx->ghost, t->ctx->group_max);
  if (t->tiles && n > 0 && static_cast<size_t>(n) < len)
    std::snprintf(text + n, len - n, "; tiles %d x %d; ghost columns %d%s", t->ctx->tiles_px, t->ctx->tiles_py, t->ctx->ghost_x,
                  t->ctx->ghost_rows == 0 ? "; rows wrap in the launch" : "");
  return 0;
}

}  // extern "C"
