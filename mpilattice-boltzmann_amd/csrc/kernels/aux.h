// aux.h — small kernels: fold, accelerate pre-pass, initial state, AoS<->SoA, halo pack, av_velocity
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once
#include "common.h"

namespace {

// Folds the last step's partials after the loop.
__global__ void __launch_bounds__(kBlock) lbm_fold_kernel(const double* partials, int n, int nvecs, double* sums, int* counter)
{
  __shared__ double red[kBlock / 64];
  for (int v = 0; v < nvecs; ++v) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) s += partials[static_cast<size_t>(v) * n + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sums[*counter + v] = s;
    __syncthreads();
  }
  if (threadIdx.x == 0) *counter += nvecs;
}

// accelerate_flow (d2q9-bgk.c:442-478) in place on one row: only needed before the first step of a run.
__global__ void lbm_accelerate_kernel(float* grid, size_t ps, const uint32_t* mask, int nx, int row, float w1, float w2)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
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

// AoS (reference t_speed) <-> SoA planes.
__global__ void lbm_aos_to_soa_kernel(const float* aos, float* grid, size_t ps, size_t ncells)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells * 9) return;
  const size_t c = i / 9;
  const int k = static_cast<int>(i - c * 9);
  grid[k * ps + c] = aos[i];
}

__global__ void lbm_soa_to_aos_kernel(const float* grid, float* aos, size_t ps, size_t ncells)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= ncells * 9) return;
  const size_t c = i / 9;
  const int k = static_cast<int>(i - c * 9);
  aos[i] = grid[k * ps + c];
}

// Outgoing halo rows of the CURRENT grid (before the first step of a row-partitioned run).
__global__ void lbm_pack_halo_kernel(const float* grid, size_t ps, int nx, int nyl, int nxp, float* send_south, float* send_north)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  const float* first = grid + x;
  const float* last = grid + static_cast<size_t>(nyl - 1) * nx + x;
  float* ss = send_south + kHaloGuard + x;
  float* sn = send_north + kHaloGuard + x;
  ss[0] = first[4 * ps]; ss[nxp] = first[7 * ps]; ss[2 * nxp] = first[8 * ps];
  sn[0] = last[2 * ps];  sn[nxp] = last[5 * ps];  sn[2 * nxp] = last[6 * ps];
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
__global__ void __launch_bounds__(kBlock) lbm_av_velocity_kernel(const float* grid, size_t ps, const uint32_t* mask, size_t ncells, double* partials)
{
  __shared__ double red[kBlock / 64];
  double acc = 0.0;
  for (size_t c = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; c < ncells; c += static_cast<size_t>(gridDim.x) * kBlock) {
    if ((mask[c >> 5] >> (c & 31)) & 1u) continue;
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

}  // namespace
