// mfma_issue.hip — how fast ONE wave can issue independent v_mfma_f32_32x32x16_f16, alone on its SIMD and beside a second wave (round 4: the conv kernels with
// one compute wave per SIMD run their K loops at ~55 % of the MFMA rate, with two at ~100 %; no operand stream explained it).  One workgroup per CU;
// W waves per SIMD each run `iters` rounds of NACC independent accumulators (NACC MFMAs per round, operands in registers, no memory);
// s_memtime ticks per MFMA per SIMD are printed with the shader clock from hipDeviceProp.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC, int VALU>   // VALU: dependent integer ops between the MFMAs of a round (stands for the address arithmetic of a k-step)
__global__ __launch_bounds__(1024) void issue_kernel(float* out, long long* ticks, int iters)
{
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    unsigned v = threadIdx.x;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < VALU; ++k) { v = v * 3u + 1u; asm volatile("" : "+v"(v)); }
        }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][7];
    const int done = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s));
    asm volatile("" :: "s"(done));
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)v;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC, int VALU>
void run(int waves_per_simd, int iters, float* d_out, long long* d_ticks, int cus)
{
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((issue_kernel<NACC, VALU>), dim3(cus), dim3(threads), 0, 0, d_out, d_ticks, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((issue_kernel<NACC, VALU>), dim3(cus), dim3(threads), 0, 0, d_out, d_ticks, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long t; CK(hipMemcpy(&t, d_ticks + 7, sizeof t, hipMemcpyDeviceToHost));
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    printf("  %d wave(s) per SIMD, %d accumulators, %2d VALU ops between MFMAs: %6.1f ticks per MFMA per SIMD, %6.1f ns per MFMA (kernel %.1f us) -> %.0f TFLOP/s on %d CUs\n",
           waves_per_simd, NACC, VALU, (double)t / mfma_per_simd, ms * 1e6 / mfma_per_simd, ms * 1e3, mfma_per_simd * 4 * cus * 32768.0 / (ms * 1e-3) / 1e12, cus);
}

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", pr.name, cus, pr.clockRate / 1000);
    float* d_out; long long* d_ticks;
    CK(hipMalloc(&d_out, (size_t)cus * 1024 * sizeof(float))); CK(hipMalloc(&d_ticks, cus * sizeof(long long)));
    const int iters = 2000;
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>(w, iters, d_out, d_ticks, cus);
        run<2, 0>(w, iters, d_out, d_ticks, cus);
        run<4, 0>(w, iters, d_out, d_ticks, cus);
        run<4, 2>(w, iters, d_out, d_ticks, cus);
        run<4, 6>(w, iters, d_out, d_ticks, cus);
        run<4, 10>(w, iters, d_out, d_ticks, cus);
    }
    return 0;
}
