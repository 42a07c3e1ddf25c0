/*
 * spin.hip — TEST-ONLY: a kernel that holds every wave slot of the device for a given time, on its own stream.
 * tests/test_gpu_parity.py runs map updates while it is resident: the wavefront kernel's grid barrier has to
 * wait the intruder out, not time out (VERDICT r1 "k_waves relies on co-residency").
 *   hipcc --offload-arch=gfx950 -O2 -shared -fPIC spin.hip -o libspin.so
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void k_spin(const long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); n++; }
    if (n < 0) *sink = n;
}

static hipStream_t g_stream;
static int *g_sink;

/* occupies `wgs_per_cu` x (number of compute units) workgroups of 1024 threads for about `ms` milliseconds; returns at once */
extern "C" int spin_start(int wgs_per_cu, double ms)
{
    if (!g_stream) {
        if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
        if (hipMalloc(&g_sink, 4) != hipSuccess) return 1;
    }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return 1;
    int rate = 0;                                              /* wall_clock64 ticks per millisecond */
    if (hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0) != hipSuccess || rate <= 0) rate = 100000;
    hipLaunchKernelGGL(k_spin, dim3(pr.multiProcessorCount * wgs_per_cu), dim3(1024), 0, g_stream, (long long)(ms * rate), g_sink);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int spin_wait(void) { return hipStreamSynchronize(g_stream) == hipSuccess ? 0 : 1; }
