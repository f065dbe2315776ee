// Every non-negative float: how many Newton corrections does sqrt((double)x) need after v_rsq_f64 to equal the correctly
// rounded double square root (d2q9-bgk.c:667 promotes u_sq to double before sqrt)?  kernels/exact_math.h sqrt_of_float
// (the SHIPPED function: the header common.h includes, included here too) is the variant "no refinement; 2 corrections";
// the exit code is non-zero if the shipped function or that variant differs anywhere.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/experiments/sqrt_exhaustive.hip -o /tmp/sq && /tmp/sq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../mpilattice-boltzmann_amd/csrc/kernels/exact_math.h"

template <int CORRECTIONS>
__device__ __forceinline__ double sqrt_variant(float xf)
{
  const double x = static_cast<double>(xf);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
#pragma unroll
  for (int i = 0; i < CORRECTIONS; ++i) {
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
  }
  return __builtin_amdgcn_class(x, 0x260) ? x : g;     // +-0 and +inf map to themselves
}

// MODE 0: as above (g and h refined once, then CORRECTIONS corrections)   1: h left unrefined   2: no refinement at all
template <int MODE, int CORRECTIONS>
__device__ __forceinline__ double sqrt_short(float xf)
{
  const double x = static_cast<double>(xf);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  if (MODE < 2) {
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    if (MODE == 0) h = __builtin_fma(h, r, h);
  }
#pragma unroll
  for (int i = 0; i < CORRECTIONS; ++i) {
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
  }
  return __builtin_amdgcn_class(x, 0x260) ? x : g;
}

constexpr int kVariants = 10;
__global__ void check(unsigned long long* bad)
{
  unsigned long long n[kVariants] = {0};
  for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= 0x7f800000ull; b += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (uint32_t)b);
    const uint64_t ref = __builtin_bit_cast(uint64_t, __builtin_sqrt(static_cast<double>(x)));
    n[0] += __builtin_bit_cast(uint64_t, sqrt_short<0, 0>(x)) != ref;
    n[1] += __builtin_bit_cast(uint64_t, sqrt_short<0, 1>(x)) != ref;
    n[2] += __builtin_bit_cast(uint64_t, sqrt_short<0, 2>(x)) != ref;
    n[3] += __builtin_bit_cast(uint64_t, sqrt_short<1, 1>(x)) != ref;
    n[4] += __builtin_bit_cast(uint64_t, sqrt_short<1, 2>(x)) != ref;
    n[5] += __builtin_bit_cast(uint64_t, sqrt_short<2, 1>(x)) != ref;
    n[6] += __builtin_bit_cast(uint64_t, sqrt_short<2, 2>(x)) != ref;
    n[7] += __builtin_bit_cast(uint64_t, sqrt_short<2, 3>(x)) != ref;
    n[8] += __builtin_bit_cast(uint64_t, sqrt_variant<2>(x)) != ref;
    n[9] += __builtin_bit_cast(uint64_t, sqrt_of_float(x)) != ref;
  }
  for (int i = 0; i < kVariants; ++i) atomicAdd(&bad[i], n[i]);
}

int main()
{
  unsigned long long* bad;
  if (hipMalloc(&bad, kVariants * sizeof *bad) != hipSuccess) return 2;
  hipMemset(bad, 0, kVariants * sizeof *bad);
  check<<<4096, 256>>>(bad);
  unsigned long long h[kVariants];
  if (hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  const char* name[kVariants] = {"g, h refined; 0 corrections (7 instructions after the conversion)", "g, h refined; 1 correction  (9)",
                                 "g, h refined; 2 corrections (11, what hipcc emits)", "g refined, h not; 1 correction (8)", "g refined, h not; 2 corrections (10)",
                                 "no refinement; 1 correction (5)", "no refinement; 2 corrections (7)  <- the shipped form", "no refinement; 3 corrections (9)", "sqrt_variant<2> (same as line 3)",
                                 "SHIPPED kernels/exact_math.h sqrt_of_float"};
  std::printf("all %llu non-negative floats (0 .. +inf): values whose result differs from the correctly rounded sqrt((double)x)\n", 0x7f800001ull);
  for (int i = 0; i < kVariants; ++i) std::printf("  %-70s %llu\n", name[i], h[i]);
  return (h[6] != 0 || h[9] != 0) ? 1 : 0;
}
