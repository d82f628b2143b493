// LDS operation throughput on MI355X: wave-instructions per second per CU for plain reads, stores and the
// atomics the dedup tables use (compare-and-swap, min, exchange) at pseudo-random addresses of a 16 KiB
// table; 3 blocks x 512 threads per CU like k_bucket.  Build: hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

template <int OP>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters) {
  __shared__ uint32_t t[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) t[i] = 0xFFFFFFFFu;
  __syncthreads();
  uint32_t acc = 0, x = mix(blockIdx.x * 512 + threadIdx.x);
  for (int i = 0; i < iters; i++) {
    x = x * 1664525u + 1013904223u;
    const uint32_t a = (x >> 10) & 4095u;
    if (OP == 0) acc += t[a];
    if (OP == 1) t[a] = x;
    if (OP == 2) acc += atomicCAS(&t[a], 0xFFFFFFFFu, x);
    if (OP == 3) atomicMin(&t[a], x);
    if (OP == 4) acc += atomicMin(&t[a], x);
    if (OP == 5) acc += atomicExch(&t[a], x);
    if (OP == 6) atomicAdd(&t[a], 1u);
  }
  if (acc == 0x12345u) out[0] = acc;
}

template <int OP>
void run(const char* name, uint32_t* out) {
  const int blocks = 256 * 3 * 8, iters = 2048;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<OP><<<blocks, 512>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<blocks, 512>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double wave_ops = (double)blocks * 8 * iters;  // 8 waves per block
  printf("%-28s %7.2f ms  %6.1f G lane-ops/s  %5.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, ms,
         wave_ops * 64 / ms / 1e6, 2.4e9 * (ms * 1e-3) * 256 / wave_ops);
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 4);
  run<0>("ds_read", out);
  run<1>("ds_write", out);
  run<2>("atomicCAS (returning)", out);
  run<3>("atomicMin (no return)", out);
  run<4>("atomicMin (returning)", out);
  run<5>("atomicExch (returning)", out);
  run<6>("atomicAdd (no return)", out);
  return 0;
}
