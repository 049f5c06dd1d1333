// How many shader cycles does one SIMD of a gfx950 CU need per wave64 VALU instruction, with 1, 2, 4 or 8
// waves resident on the SIMD?  One block per CU; a block of 256 * w threads puts w waves on each of the CU's four
// SIMDs.  Every wave runs the same stream of independent (or, in the "chain" rows, dependent) instructions
// between two s_memtime reads; the slowest wave of the block sets the SIMD's time.
//
//   hipcc -O3 --offload-arch=gfx950 -o valu_issue_probe tools/valu_issue_probe.hip && ./valu_issue_probe
//
// Prints one JSON object per line: instruction, waves per SIMD, cycles per instruction per SIMD.
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

constexpr int ITERS = 2048;   // loop turns
constexpr int UNROLL = 16;    // instructions per turn

enum Kind { FMA_INDEP = 0, ADD_INDEP = 1, FMA_CHAIN = 2, PK_FMA_INDEP = 3, MIX_FMA_MAX = 4, RCP_INDEP = 5, MAD_U64_INDEP = 6, MUL_HI_INDEP = 7, MUL_LO_INDEP = 8, N_KINDS = 9 };
const char* kind_name[N_KINDS] = {"v_fma_f32 x16 independent", "v_add_f32 x16 independent", "v_fma_f32 one dependent chain",
                                  "v_pk_fma_f32 x8 independent (2 f32 per lane each)", "v_fma_f32 / v_max_f32 / v_cndmask mix",
                                  "v_rcp_f32 x16 independent", "v_mad_u64_u32 x8 independent", "v_mul_hi_u32 x16 independent",
                                  "v_mul_lo_u32 x16 independent"};

template <int KIND>
__global__ void probe(float* out, unsigned long long* cycles, float seed) {
  float a[UNROLL];
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) a[k] = seed + (float)(threadIdx.x + k);
  const float m = seed * 0.5f + 1.0f, c = seed + 0.25f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
    if (KIND == FMA_INDEP) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
    } else if (KIND == ADD_INDEP) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
    } else if (KIND == FMA_CHAIN) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
    } else if (KIND == PK_FMA_INDEP) {
#pragma unroll
      for (int k = 0; k < UNROLL; k += 2) {
        typedef float float2v __attribute__((ext_vector_type(2)));
        float2v v = {a[k], a[k + 1]}, mm = {m, m}, cc = {c, c};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
        a[k] = v.x; a[k + 1] = v.y;
      }
    } else if (KIND == MIX_FMA_MAX) {
#pragma unroll
      for (int k = 0; k < UNROLL; k += 4) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k + 1]) : "v"(c));
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k + 2]) : "v"(m));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k + 3]) : "v"(c));
      }
    } else if (KIND == RCP_INDEP) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
    } else if (KIND == MAD_U64_INDEP) {
#pragma unroll
      for (int k = 0; k < UNROLL; k += 2) {
        unsigned long long v = ((unsigned long long)__float_as_uint(a[k + 1]) << 32) | __float_as_uint(a[k]);
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(__float_as_uint(m)), "v"(__float_as_uint(c)) : "vcc");
        a[k] = __uint_as_float((unsigned)v); a[k + 1] = __uint_as_float((unsigned)(v >> 32));
      }
    } else if (KIND == MUL_HI_INDEP) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
    } else {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) s += a[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(int n_cu, int waves_per_simd, float* out, unsigned long long* cyc_dev) {
  const int threads = 256 * waves_per_simd;
  if (threads > 1024) return;
  const int waves = n_cu * threads / 64;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<KIND>, dim3(n_cu), dim3(threads), 0, 0, out, cyc_dev, 1.0f);  // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(probe<KIND>, dim3(n_cu), dim3(threads), 0, 0, out, cyc_dev, 1.0f);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.0f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(waves);
  CHECK(hipMemcpy(cyc.data(), cyc_dev, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long mx = 0, mn = ~0ull;
  double sum = 0.0;
  for (int i = 0; i < waves; ++i) { mx = cyc[i] > mx ? cyc[i] : mx; mn = cyc[i] < mn ? cyc[i] : mn; sum += (double)cyc[i]; }
  const int per_turn = (KIND == PK_FMA_INDEP || KIND == MAD_U64_INDEP) ? UNROLL / 2 : UNROLL;
  const double instr = (double)ITERS * per_turn;  // wave instructions per wave
  // a SIMD issued waves_per_simd * instr instructions while its slowest wave ran
  std::printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instruction_one_wave\": %.3f, "
              "\"cycles_per_instruction_per_simd\": %.3f, \"wave_cycles_min\": %llu, \"wave_cycles_max\": %llu, "
              "\"kernel_ms\": %.4f, \"shader_clock_ghz_from_event\": %.3f}\n",
              kind_name[KIND], waves_per_simd, sum / waves / instr, (double)mx / (instr * waves_per_simd), mn, mx, ms,
              (double)mx / (ms * 1e6));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  std::printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"iters\": %d, \"unroll\": %d}\n", prop.gcnArchName,
              n_cu, prop.clockRate / 1000, ITERS, UNROLL);
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, (size_t)n_cu * 1024 * sizeof(float)));
  CHECK(hipMalloc(&cyc, (size_t)n_cu * 16 * sizeof(unsigned long long)));
  for (int w = 1; w <= 4; ++w) {
    run<FMA_INDEP>(n_cu, w, out, cyc);
    run<ADD_INDEP>(n_cu, w, out, cyc);
    run<FMA_CHAIN>(n_cu, w, out, cyc);
    run<PK_FMA_INDEP>(n_cu, w, out, cyc);
    run<MIX_FMA_MAX>(n_cu, w, out, cyc);
    run<RCP_INDEP>(n_cu, w, out, cyc);
    run<MAD_U64_INDEP>(n_cu, w, out, cyc);
    run<MUL_HI_INDEP>(n_cu, w, out, cyc);
    run<MUL_LO_INDEP>(n_cu, w, out, cyc);
  }
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
  return 0;
}
