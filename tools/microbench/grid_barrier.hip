// What a grid-wide barrier costs on gfx950 (one persistent launch, G resident workgroups, K barriers):
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/grid_barrier.hip -o tools/microbench/grid_barrier && ./grid_barrier
// variant 0: arrive (atomic add, agent scope) + spin, no fences; 1: + __threadfence() before and after (what a barrier that
// publishes plain stores across the XCDs' L2s needs); 2: fences + 64 KB of plain stores per workgroup between barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Sync { unsigned arrived, error; };
template <int V>
__global__ __launch_bounds__(256) void k(Sync* g, float* buf, int K) {
  __shared__ int go;
  unsigned epoch = 0;
  const unsigned G = gridDim.x;
  for (int b = 0; b < K; ++b) {
    if (V == 2) for (int i = threadIdx.x; i < 16384; i += 256) buf[(size_t)blockIdx.x * 16384 + i] = (float)(b + i);
    if (V >= 1) __threadfence();
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&g->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(&g->arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * G) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000000ll) break;
      }
      go = 1;
    }
    __syncthreads();
    if (V >= 1) __threadfence();
  }
  if (go == 123456) buf[0] = 1.f;
}
template <int V> void run(int G, int K, Sync* s, float* buf) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(s, 0, sizeof(Sync));
    hipEventRecord(e0);
    k<V><<<G, 256>>>(s, buf, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("variant %d  G %4d  K %3d: %8.1f us total, %6.2f us per barrier\n", V, G, K, ms * 1e3, ms * 1e3 / K);
}
int main() {
  Sync* s; float* buf; hipMalloc(&s, sizeof(Sync)); hipMalloc(&buf, (size_t)512 * 16384 * 4);
  for (int G : {5, 64, 256, 512}) { run<0>(G, 100, s, buf); run<1>(G, 100, s, buf); run<2>(G, 100, s, buf); }
  return 0;
}
