// What would a persistent small-deck kernel pay per hand-off?  ("a persistent LDS-resident kernel with
// a grid barrier" against what lbm_tile_kernel does today: one launch per 8 steps.)
// Measures on one GPU, with 256 co-resident blocks (one per CU, cooperative launch):
//   a) a grid barrier: one agent-scope atomic per block on one counter (flat) or on its XCD's counter whose last arriver
//      reports to a global one (two-level); lane 0 of every block spins on the global counter with relaxed loads;
//   b) the same plus the data hand-off a tile needs: each block stores 3 KB (its outgoing ghost ring) before the
//      barrier and loads its neighbour's 3 KB after it (agent scope: across XCDs this goes through memory);
//   c) back-to-back dependent launches of an empty 256-block kernel (what the tile kernel pays today).
// Every spin is bounded (a block that waits more than ~50 ms raises a flag and all blocks leave).
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { unsigned* counter; int* failed; float* ring; int iters; int exchange; int tree; };   // counter: [0] global, [32 * (1 + x)] per XCD

__global__ void __launch_bounds__(256) persistent(Args a)
{
  const unsigned nb = gridDim.x;
  __shared__ int stop;
  float keep = 0.0f;
  for (int it = 0; it < a.iters; ++it) {
    if (a.exchange) {                       // outgoing ghost ring: 768 floats per block, double-buffered by iteration parity
      float* mine = a.ring + ((it & 1) * nb + blockIdx.x) * 768;
      for (int i = threadIdx.x; i < 768; i += 256) __hip_atomic_store(mine + i, keep + i + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      stop = 0;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
      unsigned* flag = a.counter;                                          // what this block polls
      unsigned want = (it + 1u) * nb;
      if (!a.tree) {
        __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // two levels: blocks b, b+8, ... run on one XCD (round-robin dispatch) and meet on that XCD's counter (its own
        // 128-byte line); the last of them to arrive reports to the global counter; everybody polls the global one
        const unsigned x = blockIdx.x & 7u, per = nb >> 3;
        unsigned* mine = a.counter + 32 * (1 + x);
        const unsigned seen = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (seen + 1u == (it + 1u) * per) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        want = (it + 1u) * 8u;
      }
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (__hip_atomic_load(a.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
            __builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) {         // 50 ms
          __hip_atomic_store(a.failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          stop = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);                             // one acquire after the relaxed polls
    }
    __syncthreads();
    if (stop) return;                         // every block sees the flag at its next poll: the grid drains
    if (a.exchange) {
      const float* theirs = a.ring + ((it & 1) * nb + (blockIdx.x + 1) % nb) * 768;
      for (int i = threadIdx.x; i < 768; i += 256) keep += __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (keep == 123.456f) a.ring[0] = keep;
}

__global__ void __launch_bounds__(256) empty_kernel(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.0f; }

int main()
{
  unsigned* counter; int* failed; float* ring;
  CHECK(hipMalloc(&counter, 4 * 32 * 9)); CHECK(hipMalloc(&failed, 4)); CHECK(hipMalloc(&ring, 2 * 256 * 768 * sizeof(float)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int mode = 0; mode < 4; ++mode) {
    const int exchange = mode & 1, tree = mode >> 1;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(counter, 0, 4 * 32 * 9)); CHECK(hipMemset(failed, 0, 4));
      Args a{counter, failed, ring, iters, exchange, tree};
      void* params[] = {&a};
      CHECK(hipEventRecord(e0, 0));
      CHECK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(persistent), dim3(256), dim3(256), params, 0, 0));
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      int f; CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
      std::printf("%s grid barrier of 256 blocks%s: %.3f us per iteration over %d iterations%s\n", tree ? "two-level (per-XCD counters)" : "flat",
                  exchange ? " + 3 KB ring hand-off per block" : "", ms * 1e3 / iters, iters, f ? "  [TIMED OUT]" : "");
    }
  }
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) empty_kernel<<<256, 256>>>(nullptr);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("dependent launches of an empty 256-block kernel: %.3f us per launch\n", ms * 1e3 / iters);
  }
  return 0;
}
