// Random-access rates on MI355X: how many independent narrow gathers / scatters
// per second the chip sustains, by element width and table footprint.  These are
// the ceilings that matter for the slicer's irregular passes (k_sample gathers,
// k_scatter pair stores, k_bucket flag stores).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

template <typename T, int U>
__global__ void gather(const T* __restrict__ tab, uint32_t mask, uint32_t* out, uint32_t per_thread) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t k = 0; k < per_thread; k += U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = tab[mix(tid * 131u + (k + u) * 2654435761u) & mask];
#pragma unroll
    for (int u = 0; u < U; u++) acc += (uint32_t)v[u];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename T, int U>
__global__ void scatter(T* tab, uint32_t mask, uint32_t per_thread) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t k = 0; k < per_thread; k += U) {
#pragma unroll
    for (int u = 0; u < U; u++) tab[mix(tid * 131u + (k + u) * 2654435761u) & mask] = (T)(k + u);
  }
}

// runs of `run` consecutive elements at random bases (what an LDS-sorted scatter produces)
template <typename T>
__global__ void scatter_runs(T* tab, uint32_t mask, uint32_t per_thread, uint32_t run) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t lane = threadIdx.x & 63, wave = tid >> 6;
  for (uint32_t k = 0; k < per_thread; k++) {
    uint32_t grp = lane / run;
    uint32_t base = mix((wave * 977u + k) * 64u + grp) & mask;
    tab[(base + lane % run) & mask] = (T)k;
  }
}

template <typename F>
double time_ms(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
  const uint32_t blocks = 256 * 16, threads = 256, per_thread = 64;
  const double n = (double)blocks * threads * per_thread;
  uint32_t* out; hipMalloc(&out, 4);
  for (int lg = 19; lg <= 28; lg += 3) {  // elements: 512K .. 256M
    uint32_t elems = 1u << lg, mask = elems - 1;
    void* tab; hipMalloc(&tab, (size_t)elems * 8); hipMemset(tab, 1, (size_t)elems * 8);
    double g4 = time_ms([&] { gather<uint32_t, 8><<<blocks, threads>>>((uint32_t*)tab, mask, out, per_thread); });
    double g8 = time_ms([&] { gather<unsigned long long, 8><<<blocks, threads>>>((unsigned long long*)tab, mask, out, per_thread); });
    double g1 = time_ms([&] { gather<uint8_t, 8><<<blocks, threads>>>((uint8_t*)tab, mask, out, per_thread); });
    double s1 = time_ms([&] { scatter<uint8_t, 8><<<blocks, threads>>>((uint8_t*)tab, mask, per_thread); });
    double s4 = time_ms([&] { scatter<uint32_t, 8><<<blocks, threads>>>((uint32_t*)tab, mask, per_thread); });
    double s8 = time_ms([&] { scatter<unsigned long long, 8><<<blocks, threads>>>((unsigned long long*)tab, mask, per_thread); });
    double r8_8 = time_ms([&] { scatter_runs<unsigned long long><<<blocks, threads>>>((unsigned long long*)tab, mask, per_thread, 8); });
    double r8_16 = time_ms([&] { scatter_runs<unsigned long long><<<blocks, threads>>>((unsigned long long*)tab, mask, per_thread, 16); });
    printf("elems 2^%d (4B table %.0f MB): Gacc/s  gather1 %.0f gather4 %.0f gather8 %.0f | scatter1 %.0f scatter4 %.0f scatter8 %.0f | scatter8 runs8 %.0f runs16 %.0f\n",
           lg, elems * 4.0 / 1e6, n / g1 / 1e6, n / g4 / 1e6, n / g8 / 1e6, n / s1 / 1e6, n / s4 / 1e6, n / s8 / 1e6,
           n / r8_8 / 1e6, n / r8_16 / 1e6);
    hipFree(tab);
  }
  return 0;
}
