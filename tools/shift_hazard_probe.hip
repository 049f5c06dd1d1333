// gfx950: a 64-bit shift (v_lshlrev_b64, v_lshrrev_b64, v_ashrrev_i64) whose 32-bit shift amount sits in the LAST
// vector register of the wave's allocation (v119 of 120, v127 of 128) shifts by 0 -- on every wave of a 1024-thread
// block but the first on its SIMD, about a quarter of the time; never with 256-thread blocks, never with the amount
// one register lower or copied through a v_mov_b32 first, whatever wrote the register and however many wait states
// lie in between; 32-bit shifts, v_mad_u64_u32 and v_lshl_add_u64 on the same registers are right.
//
// Found while diagnosing a build of the render kernel scheduled with -mllvm -amdgpu-sched-strategy=max-ilp, which
// lost and duplicated paths (and, on long frames, took a memory access fault).  In the prologue of its shading function
//     v_mbcnt_lo_u32_b32 v4, -1, 0 ; 4 x SALU ; v_mbcnt_hi_u32_b32 v119, -1, v4 ; 2 x SALU ; v_lshlrev_b64 v[0:1], v119, -1
// (lane masks for the survivors' prefix sums; the kernel allocates 120 registers) gave -1 in every lane, only on waves
// 4..15 of the block.  This probe runs that sequence and variations of it in a loop, 16 waves per CU and 4, and counts
// wrong results per wave of the block (cases 0-12 in a kernel of 120 registers, 13-14 of 128).
//   hipcc -O2 --offload-arch=gfx950 -o shift_hazard_probe shift_hazard_probe.hip && ./shift_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define SALU4 "s_add_i32 s46, s43, s17\n s_mul_i32 s17, s6, 48\n s_mul_i32 s6, s15, s12\n s_lshl_b32 s12, s14, 6\n"
#define SALU2 "s_lshl_b32 s44, s8, 4\n s_add_i32 s8, s43, s12\n"
#define CLOBBERS "v0", "v1", "v2", "v3", "v4", "v5", "v46", "v66", "v118", "v119", "s6", "s8", "s12", "s17", "s19", "s28", "s44", "s46"
// every case leaves: %0 = the lane id as written, (%1, %2) = the shifted value
#define CASE(ID, BODY)                                                                                       \
  if (WHICH == ID) asm volatile("v_mov_b32 v119, 0\n v_mov_b32 v118, 0\n s_nop 7\n" BODY                     \
                                : "=&v"(amount), "=&v"(lo), "=&v"(hi) : : CLOBBERS);

template <int WHICH>
__global__ __launch_bounds__(1024) void probe(unsigned long long* wrong, float* sink, int iters) {
  __shared__ float lds[4096];
  const uint32_t true_lane = __lane_id();
  float acc = (float)threadIdx.x + 1.0f;
  unsigned long long mix = threadIdx.x * 0x9E3779B97F4A7C15ull + 1;
  unsigned long long bad = 0;
  lds[threadIdx.x] = acc;
  for (int i = 0; i < iters; ++i) {
    // what the other waves of a render are busy with: divisions, square roots, 64-bit multiplies, LDS traffic
    const int spin = 4 + ((threadIdx.x >> 6) * 7 + i) % 13;
    for (int k = 0; k < spin; ++k) {
      acc = sqrtf(acc * 1.0001f + 2.0f) / (acc + 1.5f) + lds[(threadIdx.x * 17 + k * 64) & 4095];
      mix = mix * 0xD2511F53ull + (mix >> 29);
      lds[(threadIdx.x + k) & 4095] = acc;
    }
    acc += (float)(mix & 1);
    uint32_t amount = 0, lo = 0, hi = 0;
    // 0: the kernel's sequence
    CASE(0, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2
            "v_lshlrev_b64 v[0:1], v119, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 1: another destination
    CASE(1, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2
            "v_lshlrev_b64 v[2:3], v119, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v2\n v_mov_b32 %2, v3")
    // 2: the shift amount in v118 instead of the allocation's last register
    CASE(2, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v118, -1, v4\n" SALU2
            "v_lshlrev_b64 v[0:1], v118, -1\n v_mov_b32 %0, v118\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 3: the shift amount in v5
    CASE(3, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v5, -1, v4\n" SALU2
            "v_lshlrev_b64 v[0:1], v5, -1\n v_mov_b32 %0, v5\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 4: no scalar instructions in between
    CASE(4, "v_mbcnt_lo_u32_b32 v4, -1, 0\n v_mbcnt_hi_u32_b32 v119, -1, v4\n"
            "v_lshlrev_b64 v[0:1], v119, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 5: sixteen wait states in front of the shift
    CASE(5, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2 "s_nop 7\n s_nop 7\n"
            "v_lshlrev_b64 v[0:1], v119, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 6: the shift amount copied to another register first (what cured the kernel)
    CASE(6, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2 "v_mov_b32 v5, v119\n"
            "v_lshlrev_b64 v[0:1], v5, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 7: a plain move as the producer
    CASE(7, "v_mbcnt_lo_u32_b32 v4, -1, 0\n v_mbcnt_hi_u32_b32 v4, -1, v4\n" SALU4 "v_mov_b32 v119, v4\n" SALU2
            "v_lshlrev_b64 v[0:1], v119, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 8: a 32-bit shift as the consumer (the high word is filled in so that the check below still applies)
    CASE(8, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2
            "v_lshlrev_b32 v0, v119, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, -1")
    // 9: a 64-bit left shift written the other way round (v_lshlrev_b64 is the only encoding; the amount from an SGPR-free add)
    CASE(9, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2
            "v_ashrrev_i64 v[0:1], v119, -1\n v_lshlrev_b64 v[0:1], v119, v[0:1]\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 10: v_lshrrev_b64 with the amount in v119
    CASE(10, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2
             "v_lshrrev_b64 v[0:1], v119, -1\n v_not_b32 v0, v0\n v_not_b32 v1, v1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 11: v_mad_u64_u32 with a 32-bit factor in v119 (1 << lane by multiplication is not possible: lane * 0x100000001 + 0)
    CASE(11, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2
             "v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mad_u64_u32 v[0:1], vcc, v119, 1, v[2:3]\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 12: v_lshl_add_u64 with the pair v[118:119] as its 64-bit operand
    CASE(12, "v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n v_mov_b32 v118, v119\n" SALU2
             "v_lshl_add_u64 v[0:1], v[118:119], 0, 0\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1")
    // 13: the kernel's sequence in a wave whose allocation goes on to v127 (v119 is not its last register)
    if (WHICH == 13) asm volatile("v_mov_b32 v119, 0\n v_mov_b32 v127, 0\n s_nop 7\n v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v119, -1, v4\n" SALU2
             "v_lshlrev_b64 v[0:1], v119, -1\n v_mov_b32 %0, v119\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1"
             : "=&v"(amount), "=&v"(lo), "=&v"(hi) : : CLOBBERS, "v127");
    // 14: ... and with the amount in that wave's last register, v127
    if (WHICH == 14) asm volatile("v_mov_b32 v127, 0\n s_nop 7\n v_mbcnt_lo_u32_b32 v4, -1, 0\n" SALU4 "v_mbcnt_hi_u32_b32 v127, -1, v4\n" SALU2
             "v_lshlrev_b64 v[0:1], v127, -1\n v_mov_b32 %0, v127\n v_mov_b32 %1, v0\n v_mov_b32 %2, v1"
             : "=&v"(amount), "=&v"(lo), "=&v"(hi) : : CLOBBERS, "v127");
    const unsigned long long shifted = ((unsigned long long)hi << 32) | lo;
    unsigned long long want = WHICH == 8 ? (0xFFFFFFFF00000000ull | (uint32_t)(~0u << (true_lane & 31u))) : (~0ull << true_lane);
    if (WHICH == 10) want = ~(~0ull >> true_lane);
    if (WHICH == 11) want = true_lane;
    if (WHICH == 12) want = ((unsigned long long)true_lane << 32) | true_lane;
    if (shifted != want || amount != true_lane) ++bad;
  }
  if (bad) atomicAdd(&wrong[threadIdx.x >> 6], bad);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WHICH>
static int run(const char* what, int threads) {
  unsigned long long* wrong;
  float* sink;
  CHECK(hipMalloc(&wrong, 16 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 256 * 1024 * sizeof(float)));
  CHECK(hipMemset(wrong, 0, 16 * sizeof(unsigned long long)));
  const int iters = 4000;
  hipLaunchKernelGGL(probe<WHICH>, dim3(256), dim3(threads), 0, 0, wrong, sink, iters);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[16];
  CHECK(hipMemcpy(h, wrong, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long total = 0;
  for (int k = 0; k < 16; ++k) total += h[k];
  std::printf("{\"case\": \"%s\", \"threads_per_block\": %d, \"lane_iterations\": %llu, \"wrong\": %llu, \"wrong_by_wave_of_block\": [", what,
              threads, (unsigned long long)256 * threads * iters, total);
  for (int k = 0; k < threads / 64; ++k) std::printf("%s%llu", k ? ", " : "", h[k]);
  std::printf("]}\n");
  (void)hipFree(wrong);
  (void)hipFree(sink);
  return 0;
}

int main() {
  for (int threads : {1024, 256}) {
    if (run<0>("the kernel's sequence: v_mbcnt_hi v119 ; 2 SALU ; v_lshlrev_b64 v[0:1], v119, -1", threads)) return 1;
    if (run<1>("destination v[2:3]", threads)) return 1;
    if (run<2>("shift amount in v118", threads)) return 1;
    if (run<3>("shift amount in v5", threads)) return 1;
    if (run<4>("no scalar instructions in between", threads)) return 1;
    if (run<5>("16 wait states in front of the shift", threads)) return 1;
    if (run<6>("shift amount copied to v5 first", threads)) return 1;
    if (run<7>("producer v_mov_b32 v119, v4", threads)) return 1;
    if (run<8>("consumer v_lshlrev_b32", threads)) return 1;
    if (run<9>("v_ashrrev_i64 in front of the shift, same amount register", threads)) return 1;
    if (run<10>("v_lshrrev_b64 with the amount in v119", threads)) return 1;
    if (run<11>("v_mad_u64_u32 with a factor in v119", threads)) return 1;
    if (run<12>("v_lshl_add_u64 with v[118:119]", threads)) return 1;
    if (run<13>("the kernel's sequence, allocation of 128 registers (v119 not the last)", threads)) return 1;
    if (run<14>("amount in v127, the last of 128 registers", threads)) return 1;
  }
  return 0;
}
