// store_ceiling.hip — the same-box ceiling of streaming stores: what a bare kernel that does nothing but
// store reaches on this MI355X, by store width (dwordx3 = the raster's 12 B per lane / 768 B per wave instruction, dwordx4),
// cache policy (plain, nt, sc1, sc0 sc1 = write-through) and footprint (118 MB = two 1024-env frame sets, inside the 256 MiB
// Infinity Cache; 472 MB = 4096 envs; 1.9 GB = 16384 envs).  Build: hipcc --offload-arch=gfx950 -O3 -o store_ceiling store_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u3v __attribute__((ext_vector_type(3)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every wave instruction writes 64 x W contiguous bytes; a workgroup of 512 threads walks pieces of 512 x W bytes, grid-strided
template <int W, int AUX>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* dst, size_t bytes, unsigned seed)
{
    const size_t piece = (size_t)512 * W;
    const size_t npieces = bytes / piece;
    const unsigned v = seed + threadIdx.x;
    for (size_t pc = blockIdx.x; pc < npieces; pc += gridDim.x) {
        unsigned char* base = dst + pc * piece;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)piece, 0x00020000);
        if constexpr (W == 16) {
            const u4v x = {v, v + 1, v + 2, v + 3};
            __builtin_amdgcn_raw_buffer_store_b128(x, rs, threadIdx.x * 16, 0, AUX);
        } else {
            const u3v x = {v, v + 1, v + 2};
            __builtin_amdgcn_raw_buffer_store_b96(x, rs, threadIdx.x * 12, 0, AUX);
        }
    }
}

// the raster's shape: a workgroup owns 4 "frames" of 57,600 B at a time (one per 2 waves... here: 8 waves x 768 B per pass over
// ONE frame, 4 frames one after the other), as trs_step_kernel / trs_worker_kernel write them
template <int AUX>
__global__ __launch_bounds__(512) void frame_kernel(unsigned char* dst, int nframes, unsigned seed)
{
    const int frame_bytes = 57600;
    const unsigned v = seed + threadIdx.x;
    const u3v x = {v, v + 1, v + 2};
    for (int f = blockIdx.x * 4; f < nframes; f += gridDim.x * 4) {
        for (int j = 0; j < 4 && f + j < nframes; ++j) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)(f + j) * frame_bytes, 0, frame_bytes, 0x00020000);
            if (threadIdx.x < 480)
                for (int pass = 0; pass < 10; ++pass)
                    __builtin_amdgcn_raw_buffer_store_b96(x, rs, pass * 5760 + threadIdx.x * 12, 0, AUX);
        }
    }
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t sizes[] = {(size_t)2 * 1024 * 57600, (size_t)2 * 4096 * 57600, (size_t)2 * 16384 * 57600};
    const char* names[] = {"118 MB (2 x 1024 frames)", "472 MB (2 x 4096 frames)", "1.9 GB (2 x 16384 frames)"};
    unsigned char* buf = nullptr;
    CK(hipMalloc(&buf, sizes[2] + (1 << 20)));
    CK(hipMemset(buf, 0, sizes[2]));
    printf("# bare streaming stores, one MI355X; GB/s of bytes stored; each timed launch writes HALF the footprint, alternating halves (like the env's two frame buffers)\n");
    for (int grid : {256, 1024}) {
        for (int s = 0; s < 3; ++s) {
            const size_t half = sizes[s] / 2;
            const int reps = s == 0 ? 200 : s == 1 ? 60 : 20;
            printf("== grid %d x 512 threads, footprint %s\n", grid, names[s]);
#define RUN(W, AUX, label) { \
                double ms = time_ms([&](int i) { hipLaunchKernelGGL((store_kernel<W, AUX>), dim3(grid), dim3(512), 0, 0, buf + (size_t)(i & 1) * half, half, (unsigned)i); }, reps); \
                printf("  %-28s %8.1f us  %8.1f GB/s\n", label, ms * 1e3, half / (ms * 1e-3) / 1e9); }
            RUN(16, 0, "dwordx4 plain")
            RUN(16, 2, "dwordx4 nt")
            RUN(16, 16, "dwordx4 sc1")
            RUN(16, 17, "dwordx4 sc0 sc1")
            RUN(16, 19, "dwordx4 sc0 sc1 nt")
            RUN(12, 0, "dwordx3 plain")
            RUN(12, 2, "dwordx3 nt")
            RUN(12, 16, "dwordx3 sc1")
            RUN(12, 17, "dwordx3 sc0 sc1")
            RUN(12, 19, "dwordx3 sc0 sc1 nt")
#undef RUN
            if (grid == 256) {
                const int nframes = (int)(half / 57600);
#define RUNF(AUX, label) { \
                double ms = time_ms([&](int i) { hipLaunchKernelGGL((frame_kernel<AUX>), dim3(256), dim3(512), 0, 0, buf + (size_t)(i & 1) * half, nframes, (unsigned)i); }, reps); \
                printf("  %-28s %8.1f us  %8.1f GB/s\n", label, ms * 1e3, half / (ms * 1e-3) / 1e9); }
                RUNF(0, "raster shape dwordx3 plain")
                RUNF(2, "raster shape dwordx3 nt")
                RUNF(17, "raster shape sc0 sc1")
                RUNF(19, "raster shape sc0 sc1 nt")
#undef RUNF
            }
        }
    }
    CK(hipFree(buf));
    return 0;
}
