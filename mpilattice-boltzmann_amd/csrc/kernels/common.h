// common.h — constants, argument block of the one-step kernels, loads/stores, reductions, the cell arithmetic (relax_cell), source-row selection
// Part of the single translation unit lbm_kernels.hip (device code of liblbm_d2q9.so, gfx950 only).
#pragma once

namespace {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kCellsPerLane = 4;     // one 16-byte access per population per lane
constexpr int kHaloGuard = 4;        // floats of guard on each side of a halo-buffer row

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access

struct StepArgs {
  const float* src;            // source grid, plane 0 row 0
  float* dst;                  // destination grid
  const uint32_t* mask;        // obstacle bitfield, bit c of the partition-local cell index
  int mask_words;              // words allocated for it
  size_t ps;                   // plane stride in floats
  int nx, nyl;                 // row length, rows owned by this partition
  int quad_begin, quad_end;    // 4-cell groups [begin,end) of the partition handled by this launch
  int quad_begin2, quad_end2;  // optional second range (boundary launch: last row), empty if begin2>=end2
  int iters;                   // 1024-cell chunks per block
  // sources outside the partition (row-partitioned runs); nullptr = periodic wrap inside the plane
  const float* south_halo;     // populations 2,5,6 of the row below row 0   [3][nxp], data at +kHaloGuard
  const float* north_halo;     // populations 4,7,8 of the row above row nyl-1
  float* send_south;           // row 0's populations 4,7,8 for the southern neighbour (next step)
  float* send_north;           // row nyl-1's populations 2,5,6 for the northern neighbour
  int nxp;                     // halo-buffer row pitch = nx + 2*kHaloGuard
  float omega;
  float accel_w1, accel_w2;    // d2q9-bgk.c:445-446
  int accel_row;               // local row that is global row ny-2, or -1: epilogue accelerate for the NEXT step
  double* partials_out;        // this launch's per-block sums
  const double* prev_partials; // previous step's per-block sums, folded by block 0 of this launch
  int n_prev;
  double* sums;                // per-step totals of this run
  int* counter;                // index of the next entry of sums
};

__device__ __forceinline__ f4 load4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 load4u(const float* p) { return *reinterpret_cast<const f4u*>(p); }

template <bool NT>
__device__ __forceinline__ void store4(float* p, f4 v)
{
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Deterministic block sum (fixed tree): every thread gets the total.
__device__ __forceinline__ double block_sum(double v, double* lds /* kBlock/64 doubles */)
{
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[wave] = v;
  __syncthreads();
  double t = lds[0];
#pragma unroll
  for (int w = 1; w < kBlock / 64; ++w) t += lds[w];
  return t;
}

// Block 0 of every step launch does no lattice work: it folds the PREVIOUS
// step's per-block sums into sums[counter++] (d2q9-bgk.c:367) while the other blocks stream, so the
// fold's latency (a dependent load + two barriers) is off the critical path of the tiny grids.
__device__ __forceinline__ void fold_previous(const StepArgs& a, double* red);

// One cell: moments, equilibrium, relaxation in the reference's operation order (d2q9-bgk.c:546-666).
// t[] = streamed-in populations, o[] = relaxed populations; returns sqrt(m^2)/rho in double (:667).
__device__ __forceinline__ void relax_cell_core(const float (&t)[9], float omega, float (&o)[9], float& msq_out, float& rinv_out);

__device__ __forceinline__ double relax_cell(const float (&t)[9], float omega, float (&o)[9])
{
  float msq, rinv;
  relax_cell_core(t, omega, o, msq, rinv);
  return sqrt(static_cast<double>(msq)) * static_cast<double>(rinv);   // :667
}

// The same without the sum|u| term: msq = m^2 (un-normalised momentum squared), rinv = 1/rho.
__device__ __forceinline__ void relax_cell_core(const float (&t)[9], float omega, float (&o)[9], float& msq_out, float& rinv_out)
{
  const float csq_inv = 3.0f;                                   // :497
  const float w0 = 4.0f / 9.0f, w1 = 1.0f / 9.0f, w2 = 1.0f / 36.0f;   // :499-501
  float rho = t[0];                                             // :546-554
  rho += t[1]; rho += t[2]; rho += t[3]; rho += t[4];
  rho += t[5]; rho += t[6]; rho += t[7]; rho += t[8];
  const float rinv = 1.0f / rho;                                // :561
  float mx = t[1] + t[5];                                       // :570-574
  mx += t[8]; mx -= t[3]; mx -= t[6]; mx -= t[7];
  float my = t[2] + t[5];                                       // :576-580
  my += t[6]; my -= t[4]; my -= t[7]; my -= t[8];
  const float msq = mx * mx + my * my;                          // :589
  float e[9];
  e[1] = mx;       e[2] = my;        e[3] = -mx;       e[4] = -my;        // :596-599
  e[5] = mx + my;  e[6] = -mx + my;  e[7] = -mx - my;  e[8] = mx - my;    // :600-603
  const float h = 0.5f * rinv * csq_inv;                        // "0.5f*densinv*ic_sq" of :638-646
  const float q0 = w0 * (rho - h * msq);                        // :638
  o[0] = t[0] + omega * (q0 - t[0]);                            // :658
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const float a = e[k] * csq_inv;                             // :610-617
    const float b = a * e[k];                                   // :624-631
    const float wk = (k < 5) ? w1 : w2;
    const float q = wk * (rho + a + h * (b - msq));             // :639-646
    o[k] = t[k] + omega * (q - t[k]);                           // :659-666
  }
  msq_out = msq;
  rinv_out = rinv;
}

// Row bases of the three source rows of destination row y, per population (d2q9-bgk.c:511-512,
// 526-538): here -> 0,1,3 (+k*ps); south row -> 2,5,6; north row -> 4,7,8.  Rows outside the
// partition come from the halo messages (row-partitioned runs) or wrap periodically.
struct RowPtrs {
  const float *here, *s2, *s5, *s6, *n4, *n7, *n8;
};

__device__ __forceinline__ RowPtrs source_rows(const StepArgs& a, int y)
{
  RowPtrs r;
  const size_t ps = a.ps;
  const int nx = a.nx;
  r.here = a.src + static_cast<size_t>(y) * nx;
  if (y > 0 || a.south_halo == nullptr) {
    const int ys = (y > 0) ? y - 1 : a.nyl - 1;                        // periodic wrap (:245-247 with one rank)
    const float* q = a.src + static_cast<size_t>(ys) * nx;
    r.s2 = q + 2 * ps; r.s5 = q + 5 * ps; r.s6 = q + 6 * ps;
  } else {
    const float* q = a.south_halo + kHaloGuard;
    r.s2 = q; r.s5 = q + a.nxp; r.s6 = q + 2 * a.nxp;
  }
  if (y < a.nyl - 1 || a.north_halo == nullptr) {
    const int yn = (y < a.nyl - 1) ? y + 1 : 0;
    const float* q = a.src + static_cast<size_t>(yn) * nx;
    r.n4 = q + 4 * ps; r.n7 = q + 7 * ps; r.n8 = q + 8 * ps;
  } else {
    const float* q = a.north_halo + kHaloGuard;
    r.n4 = q; r.n7 = q + a.nxp; r.n8 = q + 2 * a.nxp;
  }
  return r;
}


typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ void bounce_or_relax(const float (&t)[9], const float (&o)[9], bool blocked, float (&out)[9])
{
  out[0] = blocked ? t[0] : o[0];                                       // bounce-back d2q9-bgk.c:687-695
  out[1] = blocked ? t[3] : o[1];
  out[2] = blocked ? t[4] : o[2];
  out[3] = blocked ? t[1] : o[3];
  out[4] = blocked ? t[2] : o[4];
  out[5] = blocked ? t[7] : o[5];
  out[6] = blocked ? t[8] : o[6];
  out[7] = blocked ? t[5] : o[7];
  out[8] = blocked ? t[6] : o[8];
}

__device__ __forceinline__ void accelerate_cell(float (&out)[9], float w1, float w2)   // d2q9-bgk.c:457-469
{
  if (out[3] - w1 > 0.0f && out[6] - w2 > 0.0f && out[7] - w2 > 0.0f) {
    out[1] += w1; out[5] += w2; out[8] += w2;
    out[3] -= w1; out[6] -= w2; out[7] -= w2;
  }
}

}  // namespace
