// mfma_issue.hip — how fast waves can issue independent MFMAs of the two fp16 shapes of gfx950, v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16
// (round 4: the conv kernels with one compute wave per SIMD run their K loops at ~55 % of the MFMA rate, with two at ~100 %; no operand stream explained
// it.  Round 5, VERDICT r04 item 2 (ii): the 16x16x32 shape, which /opt/skills/guides/MI355X_MICROARCH.md (DVFS item 7) saw at 1.12-1.15 x the FLOP/s of
// 32x32x16 on random data at equal cycles — the chip holds a higher clock under it).
// One workgroup per CU; W waves per SIMD each run `iters` rounds of NACC independent accumulators (NACC MFMAs per round).  Operands: RANDOM binary16 data
// (zero or regular operands draw less power and hide the clock effect), either held in registers (LDS = 0) or re-read from LDS by ds_read_b128 in front of
// every MFMA (LDS = 1: conflict-free lane-linear fragments, what the pilot's K loops do).  Every configuration runs for >= 25 ms (the clocks of an idle
// MI355X need that long), then is timed over the same length.  s_memtime ticks (100 MHz x the shader clock ratio) per MFMA per SIMD and the TFLOP/s by HIP
// events are printed.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ h16x8 random_fragment(unsigned seed)
{   // eight binary16 values in (-1, 1) with random mantissas
    h16x8 v;
    for (int i = 0; i < 8; ++i) { const unsigned r = hash32(seed * 8u + (unsigned)i); v[i] = (_Float16)(((float)(r & 0xffffu) - 32768.0f) * (1.0f / 32768.0f)); }
    return v;
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// SHAPE 0: 32x32x16 (acc 16 floats, 32768 flops... 2*32*32*16), SHAPE 1: 16x16x32 (acc 4 floats, 2*16*16*32 = half the flops per instruction)
template <int SHAPE, int NACC, int VALU, int LDS>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, long long* ticks, int iters)
{
    h16x8 a = random_fragment(threadIdx.x * 2u + 1u + blockIdx.x * 4096u), b = random_fragment(threadIdx.x * 2u + 2u + blockIdx.x * 4096u);
    u4v* lds = reinterpret_cast<u4v*>(smem);
    if (LDS) {                                                  // 8 fragments per wave and operand: a ring the K loop walks (lane-linear: conflict-free ds_read_b128)
        for (int k = 0; k < 8; ++k) {
            lds[(k * 2 + 0) * blockDim.x + threadIdx.x] = __builtin_bit_cast(u4v, random_fragment(threadIdx.x * 64u + k * 2u + 3u));
            lds[(k * 2 + 1) * blockDim.x + threadIdx.x] = __builtin_bit_cast(u4v, random_fragment(threadIdx.x * 64u + k * 2u + 4u));
        }
    }
    using Acc = typename std::conditional<SHAPE == 0, f32x16, f32x4>::type;
    Acc acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < (SHAPE == 0 ? 16 : 4); ++i) acc[j][i] = 0.0f;
    unsigned v = threadIdx.x;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (LDS && (SHAPE == 0 || (j & 1) == 0)) {           // the same LDS bytes per flop for both shapes: a fragment pair per 32x32x16, or per TWO 16x16x32
                const int k = (it * NACC + j) & 7;
                a = __builtin_bit_cast(h16x8, lds[(k * 2 + 0) * blockDim.x + threadIdx.x]);
                b = __builtin_bit_cast(h16x8, lds[(k * 2 + 1) * blockDim.x + threadIdx.x]);
            }
            if constexpr (SHAPE == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < VALU; ++k) { v = v * 3u + 1u; asm volatile("" : "+v"(v)); }
        }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    const int done = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s));
    asm volatile("" :: "s"(done));
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)v;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int NACC, int VALU, int LDS>
void run(int waves_per_simd, float* d_out, long long* d_ticks, int cus)
{
    const int threads = 256 * waves_per_simd;
    const size_t lds = LDS ? (size_t)16 * threads * 16 : 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(issue_kernel<SHAPE, NACC, VALU, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // one launch ~ 5 ms at the full rate of the 32x32x16 shape; five launches to warm the clocks, five timed
    const int iters = (int)(5e-3 * 1.65e9 / ((SHAPE == 0 ? 32.0 : 16.0) * NACC * waves_per_simd));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((issue_kernel<SHAPE, NACC, VALU, LDS>), dim3(cus), dim3(threads), lds, 0, d_out, d_ticks, iters);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((issue_kernel<SHAPE, NACC, VALU, LDS>), dim3(cus), dim3(threads), lds, 0, d_out, d_ticks, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5.0f;
    long long t; CK(hipMemcpy(&t, d_ticks + 7, sizeof t, hipMemcpyDeviceToHost));
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    const double flops = SHAPE == 0 ? 32768.0 : 16384.0;
    printf("  %s %s  %d wave(s)/SIMD, %d acc, %2d VALU: %6.1f ticks per MFMA per SIMD, %6.2f ns per MFMA (launch %.2f ms) -> %6.0f TFLOP/s, clock %.0f MHz\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", LDS ? "LDS-fed " : "register", waves_per_simd, NACC, VALU, (double)t / mfma_per_simd, ms * 1e6 / mfma_per_simd, ms,
           mfma_per_simd * 4 * cus * flops / (ms * 1e-3) / 1e12, (double)t / (ms * 1e-3) / 1e6);
}

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz  (s_memtime counts shader clocks on gfx950: `clock` below = ticks per second of kernel time)\n", pr.name, cus, pr.clockRate / 1000);
    float* d_out; long long* d_ticks;
    CK(hipMalloc(&d_out, (size_t)cus * 1024 * sizeof(float))); CK(hipMalloc(&d_ticks, cus * sizeof(long long)));
    for (int round = 0; round < 2; ++round) {
        printf("round %d\n", round);
        for (int w = 1; w <= 2; ++w) {
            run<0, 4, 0, 0>(w, d_out, d_ticks, cus);
            run<1, 4, 0, 0>(w, d_out, d_ticks, cus);
            run<1, 8, 0, 0>(w, d_out, d_ticks, cus);
            run<0, 4, 0, 1>(w, d_out, d_ticks, cus);
            run<1, 4, 0, 1>(w, d_out, d_ticks, cus);
            run<1, 8, 0, 1>(w, d_out, d_ticks, cus);
            run<0, 4, 2, 1>(w, d_out, d_ticks, cus);
            run<1, 8, 1, 1>(w, d_out, d_ticks, cus);
        }
    }
    run<0, 1, 0, 0>(1, d_out, d_ticks, cus);
    run<0, 2, 0, 0>(1, d_out, d_ticks, cus);
    run<0, 4, 6, 0>(1, d_out, d_ticks, cus);
    run<0, 4, 6, 0>(2, d_out, d_ticks, cus);
    return 0;
}
