// lds_unaligned.hip — does ds_read_b128 take a 4-byte-aligned address on this chip, and what does it cost against two ds_read2_b32?  (conv1 of the fused head reads
// 16-byte fragments at a 12-byte lane stride: hipcc emits two ds_read2_b32 per fragment.)  One workgroup per CU, 16 waves; every lane reads `iters` fragments at
// 12 * lane + 4 * (it & 3) (+ 960 per row) and XORs them; results are checked against a host replica.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kLds = 64 * 1024;

template <int MODE>   // 0: two ds_read2_b32 (what hipcc emits for an align-4 16-byte read), 1: one ds_read_b128 by inline asm
__global__ __launch_bounds__(1024) void k(unsigned* out, long long* ticks, int iters)
{
    extern __shared__ unsigned sm[];
    for (int i = threadIdx.x; i < kLds / 4; i += 1024) sm[i] = 0x9E3779B9u * (unsigned)i + 12345u;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u4v acc = {0u, 0u, 0u, 0u};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned addr = (12u * lane + 4u * (it & 3) + 960u * ((wave + it) & 31)) & (kLds - 64);   // 4-byte aligned, inside LDS
        u4v v;
        if constexpr (MODE == 0) {
            const unsigned* p = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(sm) + addr);
            v = u4v{p[0], p[1], p[2], p[3]};
        } else {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        }
        acc ^= v;
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[(blockIdx.x * 1024 + threadIdx.x) * 4 + 0] = acc.x; out[(blockIdx.x * 1024 + threadIdx.x) * 4 + 1] = acc.y;
    out[(blockIdx.x * 1024 + threadIdx.x) * 4 + 2] = acc.z; out[(blockIdx.x * 1024 + threadIdx.x) * 4 + 3] = acc.w;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount, iters = 4000;
    unsigned* d_out; long long* d_t;
    CK(hipMalloc(&d_out, (size_t)cus * 1024 * 16)); CK(hipMalloc(&d_t, cus * sizeof(long long)));
    std::vector<unsigned> ref(1024 * 4), got(1024 * 4);
    std::vector<unsigned> sm(kLds / 4);
    for (int i = 0; i < kLds / 4; ++i) sm[i] = 0x9E3779B9u * (unsigned)i + 12345u;
    for (int t = 0; t < 1024; ++t) {
        unsigned a[4] = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            const unsigned addr = (12u * (t & 63) + 4u * (it & 3) + 960u * (((t >> 6) + it) & 31)) & (kLds - 64);
            for (int j = 0; j < 4; ++j) a[j] ^= sm[addr / 4 + j];
        }
        for (int j = 0; j < 4; ++j) ref[t * 4 + j] = a[j];
    }
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mode ? k<1> : k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        if (mode) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(1024), kLds, 0, d_out, d_t, iters); else hipLaunchKernelGGL(k<0>, dim3(cus), dim3(1024), kLds, 0, d_out, d_t, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), d_out + (size_t)7 * 1024 * 4, 1024 * 16, hipMemcpyDeviceToHost));
        long long t; CK(hipMemcpy(&t, d_t + 7, sizeof t, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024 * 4; ++i) bad += got[i] != ref[i];
        printf("%s: %d wrong words of 4096, %.1f ticks per fragment and wave (16 waves per CU reading at once)\n", mode ? "one ds_read_b128 at a 4-byte-aligned address" : "two ds_read2_b32", bad, (double)t / iters);
    }
    return 0;
}
