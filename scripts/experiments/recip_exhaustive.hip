// Exhaustive check (every positive normal float): which short reciprocal sequences equal the IEEE division 1.0f / d that
// relax_core uses (d2q9-bgk.c:574 `1.0f / local_density`; hipcc emits div_scale / rcp / 6 fma / div_fmas / div_fixup)?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/experiments/recip_exhaustive.hip -o /tmp/recip && /tmp/recip
// The functions under test are the SHIPPED ones: kernels/exact_math.h is the header common.h includes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../mpilattice-boltzmann_amd/csrc/kernels/exact_math.h"

__device__ __forceinline__ float recip_a(float d) { return recip_newton(d); }        // rcp + one Newton step (exact_math.h)
__device__ __forceinline__ float recip_b(float d)        // rcp + Newton + one residual correction
{
  const float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  const float q = __builtin_fmaf(e, r, r);
  const float rem = __builtin_fmaf(-d, q, 1.0f);
  return __builtin_fmaf(rem, r, q);
}

// The form the kernels use: the short sequence where the magnitude of its RESULT is at least 2^-126, the division elsewhere.
__global__ void check_guarded(unsigned long long* out /* [0] mismatches on the short path, [1] values on the short path */)
{
  unsigned long long bad = 0, fast = 0;
  for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
    const float d = __builtin_bit_cast(float, (uint32_t)b);
    const float a = recip_a(d);
    if (__builtin_fabsf(a) >= 0x1p-126f) {                  // the kernels' test: a normal result (NaN fails)
      ++fast;
      const float ref = 1.0f / d;
      bad += __builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, ref);
    }
  }
  atomicAdd(&out[0], bad);
  atomicAdd(&out[1], fast);
}

// recip_exact itself, scalar and packed forms, as relax_core calls them (wave-uniform choice between the short path and
// the division): all 2^32 bit patterns, consecutive patterns in consecutive lanes; the packed form takes pattern b in one
// half and b ^ 0xa5a5a5a5 in the other.  Bits must equal 1.0f / d, NaNs included.
__global__ void check_shipped_scalar(unsigned long long* out)
{
  unsigned long long bad = 0;
  for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
    const float d = __builtin_bit_cast(float, (uint32_t)b);
    bad += __builtin_bit_cast(uint32_t, recip_exact(d)) != __builtin_bit_cast(uint32_t, 1.0f / d);
  }
  atomicAdd(&out[0], bad);
}

__global__ void check_shipped_packed(unsigned long long* out, const uint32_t* flip /* = 0xa5a5a5a5, from memory */)
{
  unsigned long long bad_x = 0, bad_y = 0;
  const uint32_t m = *flip;
  for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
    f2 v;
    v.x = __builtin_bit_cast(float, (uint32_t)b);
    v.y = __builtin_bit_cast(float, (uint32_t)b ^ m);
    const f2 q = recip_exact(v);
    // (element copies first: __builtin_bit_cast applied to `q.y` itself reads the vector's FIRST element with this clang)
    const float qx = q.x, qy = q.y, rx = 1.0f / v.x, ry = 1.0f / v.y;
    bad_x += __builtin_bit_cast(uint32_t, qx) != __builtin_bit_cast(uint32_t, rx);
    bad_y += __builtin_bit_cast(uint32_t, qy) != __builtin_bit_cast(uint32_t, ry);
  }
  atomicAdd(&out[1], bad_x);
  atomicAdd(&out[2], bad_y);
}

__global__ void check(unsigned long long* bad /* [2] counts + [2][8] examples */, uint32_t lo, uint32_t hi)
{
  for (uint64_t b = lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
    const float d = __builtin_bit_cast(float, (uint32_t)b);
    const float ref = 1.0f / d;
    const float a = recip_a(d), c = recip_b(d);
    if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, ref)) {
      const unsigned long long n = atomicAdd(&bad[0], 1ull);
      if (n < 8) bad[2 + n] = b;
    }
    if (__builtin_bit_cast(uint32_t, c) != __builtin_bit_cast(uint32_t, ref)) {
      const unsigned long long n = atomicAdd(&bad[1], 1ull);
      if (n < 8) bad[10 + n] = b;
    }
  }
}

int main()
{
  unsigned long long* bad;
  hipMalloc(&bad, 18 * sizeof *bad);
  struct { const char* name; uint32_t lo, hi; } ranges[] = {
    {"[2^-6, 2^4)", 0x3c800000u, 0x41800000u},
    {"all positive normals", 0x00800000u, 0x7f800000u},
  };
  for (auto& r : ranges) {
    hipMemset(bad, 0, 18 * sizeof *bad);
    check<<<4096, 256>>>(bad, r.lo, r.hi);
    unsigned long long h[18];
    hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
    std::printf("%s: %llu values; rcp+newton differs on %llu, rcp+newton+residual differs on %llu\n", r.name,
                (unsigned long long)(r.hi - r.lo), h[0], h[1]);
    for (int i = 0; i < 8 && i < (int)h[0]; ++i) std::printf("  a: 0x%08llx\n", h[2 + i]);
    for (int i = 0; i < 8 && i < (int)h[1]; ++i) std::printf("  b: 0x%08llx\n", h[10 + i]);
  }
  hipMemset(bad, 0, 18 * sizeof *bad);
  check_guarded<<<4096, 256>>>(bad);
  unsigned long long g[2];
  hipMemcpy(g, bad, sizeof g, hipMemcpyDeviceToHost);
  std::printf("guarded by |result| >= 2^-126, all 2^32 bit patterns: %llu take the short path, %llu of them differ from 1.0f / d\n", g[1], g[0]);
  hipMemset(bad, 0, 18 * sizeof *bad);
  const uint32_t flip_host = 0xa5a5a5a5u;
  uint32_t* flip;
  hipMalloc(&flip, sizeof *flip);
  hipMemcpy(flip, &flip_host, sizeof flip_host, hipMemcpyHostToDevice);
  check_shipped_scalar<<<4096, 256>>>(bad);
  check_shipped_packed<<<4096, 256>>>(bad, flip);
  unsigned long long sh[3];
  hipMemcpy(sh, bad, sizeof sh, hipMemcpyDeviceToHost);
  std::printf("shipped recip_exact (kernels/exact_math.h), all 2^32 bit patterns: scalar form %llu differ, packed form %llu differ from 1.0f / d\n", sh[0], sh[1] + sh[2]);
  return (g[0] != 0 || sh[0] != 0 || sh[1] != 0 || sh[2] != 0) ? 1 : 0;
}
