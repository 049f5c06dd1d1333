// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the render kernel's walk over a scene in
// global memory: every lane reads one whole 64-byte record (four 16-byte loads) at a random place of a table far
// larger than the 256 MiB Infinity Cache, or streams 64-byte path records in and out.  The byte counts are
// known, so the counter's reading of these shapes can be corrected:
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calibrate ; rocprofv3 --pmc WRITE_SIZE -- ./fetch_calibrate
// Kernels: k_gather64 (random 64-B records), k_stream64 (consecutive 64-B records, one per lane, read + written),
// k_stream16 (records stored quad by quad: a wave reads and writes 64 consecutive 16-byte quads of each of four
// arrays -- the layout of the path pool).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(call)                                                      \
  do {                                                                   \
    const hipError_t e_ = (call);                                        \
    if (e_ != hipSuccess) {                                              \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));    \
      std::exit(1);                                                      \
    }                                                                    \
  } while (0)

__global__ void k_gather64(const float4* __restrict__ table, unsigned long long n_records, float* __restrict__ out,
                           unsigned rounds) {
  unsigned long long x = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
  float acc = 0.0f;
  for (unsigned r = 0; r < rounds; ++r) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const float4* rec = table + 4 * (x % n_records);
    const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
    acc += a.x + b.y + c.z + d.w;
  }
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}

__global__ void k_stream64(const float4* __restrict__ in, float4* __restrict__ out, unsigned long long n_records) {
  const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  if (i >= n_records) return;
  const float4 a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
  out[4 * i] = d; out[4 * i + 1] = c; out[4 * i + 2] = b; out[4 * i + 3] = a;
}

__global__ void k_stream16(const float4* __restrict__ in, float4* __restrict__ out, unsigned long long n_records) {
  const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  if (i >= n_records) return;
  const float4 a = in[i], b = in[n_records + i], c = in[2 * n_records + i], d = in[3 * n_records + i];
  out[i] = d; out[n_records + i] = c; out[2 * n_records + i] = b; out[3 * n_records + i] = a;
}

int main() {
  const unsigned long long n_records = 48ull << 20;  // 3 GiB of 64-byte records
  float4* table;
  float4* copy;
  float* out;
  CHECK(hipMalloc(&table, n_records * 64));
  CHECK(hipMalloc(&copy, n_records * 64));
  CHECK(hipMemset(table, 0, n_records * 64));
  const unsigned threads = 1u << 22, rounds = 16;
  CHECK(hipMalloc(&out, threads * sizeof(float)));
  hipLaunchKernelGGL(k_gather64, dim3(threads / 256), dim3(256), 0, 0, table, n_records, out, rounds);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_stream64, dim3((unsigned)(n_records / 256)), dim3(256), 0, 0, table, copy, n_records);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_stream16, dim3((unsigned)(n_records / 256)), dim3(256), 0, 0, table, copy, n_records);
  CHECK(hipDeviceSynchronize());
  std::printf("{\"k_gather64_bytes_read\": %llu, \"k_stream64_bytes_read\": %llu, \"k_stream64_bytes_written\": %llu, "
              "\"k_stream16_bytes_read\": %llu, \"k_stream16_bytes_written\": %llu}\n",
              (unsigned long long)threads * rounds * 64ull, n_records * 64ull, n_records * 64ull, n_records * 64ull,
              n_records * 64ull);
  return 0;
}
