// Write bandwidth on MI355X as a function of the length of the contiguous runs a kernel emits:
// every wave stores `run` consecutive 4-byte elements at a pseudo-random (4-byte aligned) base of a
// 2 GiB buffer, again and again.  Short runs are what a ballot-ranked stable compaction produces
// (one run per wave, step and list); long runs are what an LDS-staged tile produces.
// Build: hipcc --offload-arch=gfx950 -O3 write_runs.hip -o write_runs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

// runs shorter than a wave: 64/run runs per store instruction; longer: run/64 consecutive instructions
__global__ void write_runs(uint32_t* buf, uint32_t mask, uint32_t steps, uint32_t run) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (run < 64) {
    for (uint32_t k = 0; k < steps; k++) {
      const uint32_t grp = lane / run;
      if (grp * run + run > 64) continue;
      const uint32_t base = mix((wave * 7919u + k) * 64u + grp) & mask;
      buf[(base + lane % run) & mask] = k;
    }
  } else {
    const uint32_t per = run / 64;
    for (uint32_t k = 0; k < steps; k += per) {
      const uint32_t base = mix(wave * 7919u + k) & mask;
      for (uint32_t j = 0; j < per; j++) buf[(base + j * 64 + lane) & mask] = k;
    }
  }
}

// same for loads
__global__ void read_runs(const uint32_t* __restrict__ buf, uint32_t mask, uint32_t steps, uint32_t run, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  if (run < 64) {
    for (uint32_t k = 0; k < steps; k++) {
      const uint32_t grp = lane / run;
      const uint32_t base = mix((wave * 7919u + k) * 64u + grp) & mask;
      acc += buf[(base + lane % run) & mask];
    }
  } else {
    const uint32_t per = run / 64;
    for (uint32_t k = 0; k < steps; k += per) {
      const uint32_t base = mix(wave * 7919u + k) & mask;
      for (uint32_t j = 0; j < per; j++) acc += buf[(base + j * 64 + lane) & mask];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// the compaction pattern: a block appends to `lists` private lists; per step each of its 4 waves adds
// `r` elements to every list (ranked order: wave 0's elements first).  staged=1: the block keeps a step
// count of `depth` in LDS-like registers first and emits runs of 4*r*depth elements instead.
__global__ void append_lists(uint32_t* buf, uint32_t list_cap, uint32_t steps, uint32_t lists, uint32_t r, uint32_t depth) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t* mine = buf + (size_t)blockIdx.x * lists * list_cap;
  if (depth <= 1) {
    for (uint32_t k = 0; k < steps; k++)
      for (uint32_t l = 0; l < lists; l++)
        if (lane < r) mine[(size_t)l * list_cap + (k * 4 + w) * r + lane] = k;
  } else {
    // each wave writes contiguous pieces of the block's staged run of 4*r*depth elements
    const uint32_t runlen = 4 * r * depth;
    for (uint32_t k = 0; k < steps; k += depth)
      for (uint32_t l = 0; l < lists; l++)
        for (uint32_t o = threadIdx.x; o < runlen; o += blockDim.x) mine[(size_t)l * list_cap + k * 4 * r + o] = k;
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
  const uint32_t elems = 1u << 29, mask = elems - 1;  // 2 GiB of u32
  uint32_t* buf; hipMalloc(&buf, (size_t)elems * 4); hipMemset(buf, 0, (size_t)elems * 4);
  uint32_t* out; hipMalloc(&out, 4);
  const uint32_t blocks = 256 * 32, threads = 256, steps = 256;
  for (uint32_t run : {8u, 10u, 16u, 32u, 64u, 128u, 256u, 1024u, 4096u, 16384u}) {
    double used = run < 64 ? (double)(64 / run) * run / 64.0 : 1.0;
    double bytes = (double)blocks * threads * steps * 4.0 * used;
    double w = time_ms([&] { write_runs<<<blocks, threads>>>(buf, mask, steps, run); });
    double r = time_ms([&] { read_runs<<<blocks, threads>>>(buf, mask, steps, run, out); });
    printf("run %6u elems (%6u B): write %7.0f GB/s   read %7.0f GB/s\n", run, run * 4, bytes / w / 1e6, bytes / r / 1e6);
  }
  // compaction pattern: 4096 blocks x lists x 96 steps x 4 waves x r elements
  for (uint32_t lists : {1u, 4u, 16u}) {
    for (uint32_t r : {10u, 40u}) {
      const uint32_t nblk = 4096, st = 96;
      const uint32_t list_cap = st * 4 * r;
      if ((size_t)nblk * lists * list_cap > elems) continue;
      for (uint32_t depth : {1u, 6u, 24u, 96u}) {
        double bytes = (double)nblk * lists * st * 4 * r * 4.0;
        double t = time_ms([&] { append_lists<<<nblk, 256>>>(buf, list_cap, st, lists, r, depth); });
        printf("append: %2u lists, %2u elems/wave/step, staged depth %2u (runs of %5u B): %7.0f GB/s\n", lists, r, depth,
               depth <= 1 ? r * 4 : 4 * r * depth * 4, bytes / t / 1e6);
      }
    }
  }
  return 0;
}
