// Round trip host -> resident kernel -> host through a mailbox word the kernel polls, for the two places the word can live:
//   host:   pinned host memory (hipHostMalloc), polled by the GPU over PCIe (what the resident worker does)
//   device: fine-grained device memory written by the HOST through the PCIe BAR, polled by the GPU in its own memory
// The kernel echoes every new value into a pinned host word; the host times post -> echo.  One wave; build:
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ab_bin/mailbox_latency scripts/mailbox_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void echo_kernel(volatile unsigned* box, volatile unsigned* echo, unsigned last)
{
    if (threadIdx.x != 0) return;
    unsigned seen = 0;
    while (seen != last) {
        const unsigned v = __hip_atomic_load(const_cast<unsigned*>(box), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v != seen) { seen = v; __hip_atomic_store(const_cast<unsigned*>(echo), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

static void run(const char* name, volatile unsigned* box_host_view, unsigned* box_dev_view, unsigned* echo_h, unsigned* echo_d)
{
    const unsigned n = 20000;
    *box_host_view = 0; *echo_h = 0;
    __sync_synchronize();
    hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, 0, box_dev_view, echo_d, n);
    CK(hipGetLastError());
    std::vector<double> us(n);
    for (unsigned i = 1; i <= n; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        *box_host_view = i;
        __sync_synchronize();
        while (*(volatile unsigned*)echo_h != i) { }
        us[i - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    CK(hipDeviceSynchronize());
    std::sort(us.begin(), us.end());
    std::printf("%-28s round trip: median %.2f us, p10 %.2f, p90 %.2f (n = %u)\n", name, us[n / 2], us[n / 10], us[9 * n / 10], n);
}

int main()
{
    unsigned *echo_h = nullptr, *box_h = nullptr, *box_d = nullptr;
    CK(hipHostMalloc((void**)&echo_h, 64, hipHostMallocMapped));
    CK(hipHostMalloc((void**)&box_h, 64, hipHostMallocMapped));
    unsigned *echo_d = nullptr, *box_hd = nullptr;
    CK(hipHostGetDevicePointer((void**)&echo_d, echo_h, 0));
    CK(hipHostGetDevicePointer((void**)&box_hd, box_h, 0));
    run("mailbox in pinned host memory", box_h, box_hd, echo_h, echo_d);
    hipError_t e = hipExtMallocWithFlags((void**)&box_d, 4096, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e)); return 0; }
    hipPointerAttribute_t at{};
    CK(hipPointerGetAttributes(&at, box_d));
    std::printf("fine-grained device allocation: device ptr %p, host ptr %p\n", at.devicePointer, at.hostPointer);
    std::fflush(stdout);
    if (std::getenv("TRY_HOST_WRITE")) {
        run("mailbox in device memory (BAR)", box_d, box_d, echo_h, echo_d);   // the host dereferences the device pointer: works only with the whole VRAM mapped (large BAR)
    }
    return 0;
}
