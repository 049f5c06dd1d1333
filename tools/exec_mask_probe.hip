// Does a gfx950 SIMD issue a wave64 VALU instruction faster when only part of the wave's lanes are active?
// One block of 1024, 512 or 256 threads per CU (4, 2 or 1 waves per SIMD); every wave runs 16 independent v_fma_f32 per turn with
// the lanes of one pattern switched on.  Prints cycles per wave instruction per SIMD for each pattern.
//
//   hipcc -O3 --offload-arch=gfx950 -o exec_mask_probe tools/exec_mask_probe.hip && ./exec_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(call)                                                                            \
  do {                                                                                         \
    const hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                          \
      std::exit(1);                                                                            \
    }                                                                                          \
  } while (0)

constexpr int ITERS = 2048, UNROLL = 16;

__global__ void probe(float* out, unsigned long long* cycles, float seed, unsigned long long mask) {
  float a[UNROLL];
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) a[k] = seed + (float)(threadIdx.x + k);
  const float m = seed * 0.5f + 1.0f, c = seed + 0.25f;
  const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (on) {
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) s += a[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, (size_t)n_cu * 1024 * sizeof(float)));
  CHECK(hipMalloc(&cyc, (size_t)n_cu * 16 * sizeof(unsigned long long)));
  struct { const char* name; unsigned long long mask; } patterns[] = {
      {"all 64 lanes", ~0ull},
      {"lanes 0-31", 0x00000000FFFFFFFFull},
      {"lanes 32-63", 0xFFFFFFFF00000000ull},
      {"even lanes (32)", 0x5555555555555555ull},
      {"lanes 0-15", 0x000000000000FFFFull},
      {"lanes 0-15 and 32-47", 0x0000FFFF0000FFFFull},
      {"lane 0", 1ull},
      {"40 scattered lanes", 0xB6DB6DB6DB6DB6DBull & 0x0FFFFFFFFFFFFFFFull},
      {"lanes 0-23", 0x0000000000FFFFFFull},
      {"lanes 0-7", 0x00000000000000FFull},
      {"lanes 0-3", 0x000000000000000Full},
      {"every 4th lane (16)", 0x1111111111111111ull},
      {"every 8th lane (8)", 0x0101010101010101ull},
      {"lanes 0 and 32", 0x0000000100000001ull},
  };
  for (int per_simd = 4; per_simd >= 1; per_simd /= 2)
  for (const auto& p : patterns) {
    const int threads = 256 * per_simd, waves = n_cu * threads / 64;
    if (per_simd != 4 && __builtin_popcountll(p.mask) != 64 && __builtin_popcountll(p.mask) != 16 &&
        __builtin_popcountll(p.mask) != 8 && __builtin_popcountll(p.mask) != 1) continue;
    hipLaunchKernelGGL(probe, dim3(n_cu), dim3(threads), 0, 0, out, cyc, 1.0f, p.mask);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(probe, dim3(n_cu), dim3(threads), 0, 0, out, cyc, 1.0f, p.mask);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(waves);
    CHECK(hipMemcpy(c.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (unsigned long long v : c) mx = v > mx ? v : mx;
    std::printf("{\"active\": \"%s\", \"lanes\": %d, \"waves_per_simd\": %d, \"cycles_per_instruction_per_simd\": %.3f}\n", p.name,
                __builtin_popcountll(p.mask), per_simd, (double)mx / ((double)ITERS * UNROLL * per_simd));
  }
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
  return 0;
}
