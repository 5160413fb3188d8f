// Microbenchmark (round 5): what does a grid barrier cost against a kernel boundary on gfx950?
// Three dependent phases over n words (phase p reads a hashed index of phase p-1's output, so every workgroup depends on all others):
//   A: three launches     B: one launch, two grid barriers (counter + spin, agent-scope release / acquire), grid <= resident set
// build: hipcc --offload-arch=gfx950 -O3 bar.hip -o bar      run: ./bar [n] [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void phase(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n, uint32_t bid, uint32_t nwg) {
  for (uint32_t i = bid * 256u + threadIdx.x; i < n; i += nwg * 256u) {
    const uint32_t j = (uint32_t)(((unsigned long long)i * 2654435761ull) % n);
    out[i] = in[j] + 1u;
  }
}
__global__ __launch_bounds__(256) void k_phase(const uint32_t *in, uint32_t *out, uint32_t n) { phase(in, out, n, blockIdx.x, gridDim.x); }

__device__ __forceinline__ void grid_barrier(uint32_t *ctr, uint32_t target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (every wave: its own vector cache lines)
}
__global__ __launch_bounds__(256) void k_fused(const uint32_t *a, uint32_t *b, uint32_t *c, uint32_t *d, uint32_t n, uint32_t *ctr,
                                               uint32_t base) {
  phase(a, b, n, blockIdx.x, gridDim.x);
  grid_barrier(ctr, base + gridDim.x);
  phase(b, c, n, blockIdx.x, gridDim.x);
  grid_barrier(ctr, base + 2u * gridDim.x);
  phase(c, d, n, blockIdx.x, gridDim.x);
}
__global__ void k_empty() {}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 20);
  const int wgs = argc > 2 ? atoi(argv[2]) : 512;
  uint32_t *a, *b, *c, *d, *ctr;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&ctr, 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(ctr, 0, 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 300;
  float ms;
  for (int pass = 0; pass < 2; pass++) {
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++) {
      hipLaunchKernelGGL(k_phase, dim3(wgs), dim3(256), 0, s, a, b, n);
      hipLaunchKernelGGL(k_phase, dim3(wgs), dim3(256), 0, s, b, c, n);
      hipLaunchKernelGGL(k_phase, dim3(wgs), dim3(256), 0, s, c, d, n);
    }
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf("n %u wgs %d  three launches      %.2f us per triple\n", n, wgs, 1e3 * ms / reps);
  }
  uint32_t base = 0;
  for (int pass = 0; pass < 2; pass++) {
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++) {
      hipLaunchKernelGGL(k_fused, dim3(wgs), dim3(256), 0, s, a, b, c, d, n, ctr, base);
      base += 2u * (uint32_t)wgs;
    }
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf("n %u wgs %d  one launch, 2 barriers %.2f us per triple\n", n, wgs, 1e3 * ms / reps);
  }
  for (int pass = 0; pass < 2; pass++) {
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_phase, dim3(wgs), dim3(256), 0, s, a, b, n);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf("n %u wgs %d  one phase alone         %.2f us\n", n, wgs, 1e3 * ms / reps);
  }
  for (int pass = 0; pass < 2; pass++) {
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf("empty kernel                  %.2f us\n", 1e3 * ms / reps);
  }
  std::vector<uint32_t> h(4);
  CK(hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost));
  printf("check d[0..3] = %u %u %u %u (3 expected)\n", h[0], h[1], h[2], h[3]);
  return 0;
}
