// aux.h — small kernels: fold, accelerate pre-pass, initial state, AoS<->SoA, halo pack, av_velocity
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

// Folds the last launch's partials after the loop (every other launch's are folded by block 0 of the
// launch that follows it).  One block per step vector when the vectors are short; long vectors (8192x8192:
// 3 x 65 536 per-tile sums, which took one block 102 us) go through lbm_fold_slices_kernel first:
// kFoldSlices blocks per vector, then this kernel over the kFoldSlices slice sums.  Fixed order, no atomics.
constexpr int kFoldSlices = 64;

__global__ void __launch_bounds__(kBlock) lbm_fold_slices_kernel(const double* partials, int n, double* slice_sums)
{
  __shared__ double red[kBlock / 64];
  const int v = blockIdx.y, b = blockIdx.x;
  const int per = (n + kFoldSlices - 1) / kFoldSlices;
  const int lo = b * per, hi = min(n, lo + per);
  double s = 0.0;
  for (int i = lo + static_cast<int>(threadIdx.x); i < hi; i += kBlock) s += partials[static_cast<size_t>(v) * n + i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) slice_sums[v * kFoldSlices + b] = s;
}

// reset: this is the last fold of a run — leave the counter at 0 for the next run (saves that run a memset launch).
__global__ void __launch_bounds__(kBlock) lbm_fold_kernel(const double* partials, int n, int nvecs, double* sums, int* counter, int reset)
{
  __shared__ double red[kBlock / 64];
  for (int v = 0; v < nvecs; ++v) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) s += partials[static_cast<size_t>(v) * n + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sums[*counter + v] = s;
    __syncthreads();
  }
  if (threadIdx.x == 0) *counter = reset ? 0 : *counter + nvecs;
}

// accelerate_flow (d2q9-bgk.c:442-478) in place on one row: only needed before the first step of a run.
// Columns [col0, col0 + ncol) of the storage row (a rank of the tile decomposition accelerates its OWNED columns only: its ghost columns of
// that row arrive accelerated from their owners, possibly before this kernel runs).
__global__ void lbm_accelerate_kernel(float* grid, size_t ps, const uint32_t* mask, int nx, int row, float w1, float w2, int col0, int ncol)
{
  const int x = col0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= col0 + ncol) return;
  const size_t c = static_cast<size_t>(row) * nx + x;
  if ((mask[c >> 5] >> (c & 31)) & 1u) return;
  float* f = grid + c;
  const float f3 = f[3 * ps], f6 = f[6 * ps], f7 = f[7 * ps];
  if (f3 - w1 > 0.0f && f6 - w2 > 0.0f && f7 - w2 > 0.0f) {
    f[1 * ps] += w1; f[5 * ps] += w2; f[8 * ps] += w2;
    f[3 * ps] = f3 - w1; f[6 * ps] = f6 - w2; f[7 * ps] = f7 - w2;
  }
}

// Initial state (d2q9-bgk.c:880-902).
__global__ void lbm_init_kernel(float* grid, size_t ps, size_t ncells, float w0, float w1, float w2)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells) return;
  grid[i] = w0;
#pragma unroll
  for (int k = 1; k < 5; ++k) grid[k * ps + i] = w1;
#pragma unroll
  for (int k = 5; k < 9; ++k) grid[k * ps + i] = w2;
}

// A column window of the storage rows: cell c of a dense block of `ncol`-wide rows -> its place in rows of `row_w` floats, `col0`
// columns in (ranks of the tile decomposition own the columns [ghost_x, ghost_x + nx_local) of their rows; everywhere else
// row_w == ncol, col0 == 0 and the mapping is the identity).
struct ColWindow { unsigned row_w, col0, ncol; };
__device__ __forceinline__ size_t window_cell(const ColWindow w, size_t c)
{
  if (w.row_w == w.ncol) return c;
  const size_t r = c / w.ncol;
  return r * w.row_w + w.col0 + (c - r * w.ncol);
}

// AoS (reference t_speed) <-> SoA planes.
__global__ void lbm_aos_to_soa_kernel(const float* aos, float* grid, size_t ps, size_t ncells, ColWindow w)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells * 9) return;
  const size_t c = i / 9;
  const int k = static_cast<int>(i - c * 9);
  grid[k * ps + window_cell(w, c)] = aos[i];
}

__global__ void lbm_soa_to_aos_kernel(const float* grid, float* aos, size_t ps, size_t ncells, ColWindow w)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells * 9) return;
  const size_t c = i / 9;
  const int k = static_cast<int>(i - c * 9);
  aos[i] = grid[k * ps + window_cell(w, c)];
}

// Outgoing halo rows of the CURRENT grid (before the first step of a row-partitioned run).
// release: the send buffers are a peer's memory (one-step peer-to-peer loop) — every block ends with the same
// per-block system-scope release as the boundary step kernels (StepArgs::release_sends), so that the flag the
// one-wave signal kernel raises next cannot overtake rows still sitting in another XCD's L2.
__global__ void lbm_pack_halo_kernel(const float* grid, size_t ps, int nx, int nyl, int nxp, float* send_south, float* send_north, int release)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < nx) {
    const float* first = grid + x;
    const float* last = grid + static_cast<size_t>(nyl - 1) * nx + x;
    float* ss = send_south + kHaloGuard + x;
    float* sn = send_north + kHaloGuard + x;
    ss[0] = first[4 * ps]; ss[nxp] = first[7 * ps]; ss[2 * nxp] = first[8 * ps];
    sn[0] = last[2 * ps];  sn[nxp] = last[5 * ps];  sn[2 * nxp] = last[6 * ps];
  }
  if (release) {
    __syncthreads();                                             // every wave's stores have left the CU
    if (threadIdx.x == 0) __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope: this XCD's L2 written back
  }
}

// K-step row partitions: gather the K first / last owned rows of all 9 planes into one contiguous
// message per direction (pack), and scatter the two received messages into the ghost rows (unpack).
// buf layout: [dir][plane][K*nx].  rows_a / rows_b = storage row where direction 0 / 1's K rows start.
__global__ void lbm_macro_pack_kernel(const float* grid, float* buf, size_t ps, int nfloats /* K*nx */, size_t row_a, size_t row_b, int nx, int unpack)
{
  // float2 granularity: nx is even in K-step mode, so K*nx and every row offset are multiples of 2 floats
  // (float4 dropped / overran the last two floats of a message when K*nx % 4 == 2, e.g. K = 3, nx = 206)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // float2 index within one plane's K rows
  const int plane = blockIdx.y, dir = blockIdx.z;
  if (i * 2 >= nfloats) return;
  const size_t goff = plane * ps + (dir == 0 ? row_a : row_b) * nx + static_cast<size_t>(i) * 2;
  const size_t boff = (static_cast<size_t>(dir) * 9 + plane) * nfloats + static_cast<size_t>(i) * 2;
  if (unpack) *reinterpret_cast<f2*>(const_cast<float*>(grid) + goff) = *reinterpret_cast<const f2*>(buf + boff);
  else *reinterpret_cast<f2*>(buf + boff) = *reinterpret_cast<const f2*>(grid + goff);
}

// av_velocity (d2q9-bgk.c:716-751): per-cell float arithmetic as the reference, double accumulation.
// Cell c of the owned rows is bit c + bit0 of the obstacle bitfield (K-step partitions: bit0 = ghost*nx,
// any value, not only multiples of 32).
__global__ void __launch_bounds__(kBlock) lbm_av_velocity_kernel(const float* grid, size_t ps, const uint32_t* mask, size_t bit0, size_t ncells, double* partials, ColWindow w)
{
  __shared__ double red[kBlock / 64];
  double acc = 0.0;
  for (size_t cw = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; cw < ncells; cw += static_cast<size_t>(gridDim.x) * kBlock) {
    const size_t c = window_cell(w, cw);
    const size_t b = c + bit0;
    if ((mask[b >> 5] >> (b & 31)) & 1u) continue;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = grid[k * ps + c];
    float rho = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) rho += f[k];                                   // :724-729
    const float ux = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;        // :732-738
    const float uy = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;        // :740-746
    acc += sqrt(static_cast<double>((ux * ux) + (uy * uy)));                   // :748
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// write_values()' per-cell arithmetic for a fluid cell (d2q9-bgk.c:1084-1111) on cells [0, n) of `grid`
// (already offset to the first cell wanted): obs[4c..4c+3] = {u_x, u_y, u, pressure}.  Same float
// operations in the same order as the reference (division and sqrt correctly rounded), so the values are
// the bits the host writer computes from the 9 populations.  A NaN produced here from non-NaN inputs
// (rho = 0) gets the sign x86 gives a generated NaN, which is what the reference's fprintf shows ("-NAN").
__global__ void __launch_bounds__(kBlock) lbm_observables_kernel(const float* grid, size_t ps, size_t n, float* obs, ColWindow w)
{
  const size_t c = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (c >= n) return;
  const size_t cs = window_cell(w, c);
  float f[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) f[k] = grid[k * ps + cs];
  float rho = 0.0f;
  bool nan_in = false;
#pragma unroll
  for (int k = 0; k < 9; ++k) { rho += f[k]; nan_in |= (f[k] != f[k]); }         // :1084-1090
  f4 o;
  o.x = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;                       // :1093-1099
  o.y = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;                       // :1101-1107
  o.z = static_cast<float>(sqrt_of_float((o.x * o.x) + (o.y * o.y)));            // :1109
  o.w = rho * (1.0f / 3.0f);                                                     // :1111, c_sq of :1040
  if (!nan_in) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (o[j] != o[j]) o[j] = __uint_as_float(0xFFC00000u);
  }
  *reinterpret_cast<f4*>(obs + 4 * c) = o;
}

// Order-independent 64-bit digest of the populations of `n` cells starting at local cell c0 (all 9 planes):
// sum over (cell, k) of mix(bits(f) ^ golden * global_index), wrap-around.  Additive over disjoint row ranges,
// so the digests of the ranks of a partitioned run add up to the digest of the whole grid; equal states give
// equal digests whatever the partitioning, and a single differing bit changes it (with probability 1 - 2^-64).
__global__ void __launch_bounds__(kBlock) lbm_checksum_kernel(const float* grid, size_t ps, size_t n, unsigned long long global_cell0,
                                                              unsigned long long* out, ColWindow w, unsigned nx_global)
{
  // (global_cell0: the global index of the window's first cell; a window narrower than the grid — a rank of the tile decomposition —
  // steps nx_global cells per row)
  unsigned long long h = 0;
  for (size_t c = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; c < n; c += static_cast<size_t>(gridDim.x) * kBlock) {
    const size_t cs = window_cell(w, c);
    const unsigned long long gc = w.row_w == w.ncol ? global_cell0 + c : global_cell0 + (c / w.ncol) * nx_global + (c % w.ncol);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const unsigned long long idx = gc * 9ull + static_cast<unsigned long long>(k);
      unsigned long long z = static_cast<unsigned long long>(__float_as_uint(grid[k * ps + cs])) ^ (idx * 0x9E3779B97F4A7C15ull);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      h += z ^ (z >> 31);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

}  // namespace
