// tools/pmc_calib.hip — measurement only (not part of the library): what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for
// streams of KNOWN size, by access width.  VERDICT r4 'next' #6: the x2 FETCH correction of MI355X_MICROARCH.md is calibrated for
// 16-byte-per-lane streams; the map update's kernels read 1, 2, 4 and 8 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/bin/pmc_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o f -- tools/bin/pmc_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o w -- tools/bin/pmc_calib
//   python tools/pmc_calib_summary.py out/f/*.db out/w/*.db          -> reported KiB vs true bytes per kernel, the factor per width
// Every kernel streams the same 512 MiB buffer once, coalesced (lane i of a wave touches element base + i), far above the 256 MB
// of last-level cache; reads are folded into one word written per workgroup so that nothing is optimised away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static const size_t BYTES = (size_t)512 << 20;

template <class T> __device__ __forceinline__ uint32_t fold(const T &v);
template <> __device__ __forceinline__ uint32_t fold<uint8_t>(const uint8_t &v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<uint16_t>(const uint16_t &v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<uint32_t>(const uint32_t &v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<uint2>(const uint2 &v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ uint32_t fold<uint4>(const uint4 &v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <class T> __device__ __forceinline__ void k_read_body(const T *in, uint32_t *sink, size_t n, uint32_t magic)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += fold<T>(in[i]);
    if (acc == magic) sink[blockIdx.x] = acc;            // (a run-time value the sums never reach: keeps the loads alive)
}
__global__ __launch_bounds__(256) void k_read_b1(const uint8_t *in, uint32_t *s, size_t n, uint32_t magic) { k_read_body(in, s, n, magic); }
__global__ __launch_bounds__(256) void k_read_b2(const uint16_t *in, uint32_t *s, size_t n, uint32_t magic) { k_read_body(in, s, n, magic); }
__global__ __launch_bounds__(256) void k_read_b4(const uint32_t *in, uint32_t *s, size_t n, uint32_t magic) { k_read_body(in, s, n, magic); }
__global__ __launch_bounds__(256) void k_read_b8(const uint2 *in, uint32_t *s, size_t n, uint32_t magic) { k_read_body(in, s, n, magic); }
__global__ __launch_bounds__(256) void k_read_b16(const uint4 *in, uint32_t *s, size_t n, uint32_t magic) { k_read_body(in, s, n, magic); }
// the block-plane pattern of the global map: 8-byte records, a wave touches eight 64-byte runs 4 KB apart
__global__ __launch_bounds__(256) void k_read_b8_runs(const uint2 *in, uint32_t *s, size_t n)
{
    uint32_t acc = 0;
    const size_t nb = n >> 9;                             // blocks of 512 records
    for (size_t v = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); v < nb * 8; v += (size_t)gridDim.x * 4) {
        const size_t blk8 = (v >> 6) << 3, z = v & 63;   // eight consecutive blocks, one z row of 8 records in each
        const int lane = threadIdx.x & 63;
        acc ^= fold<uint2>(in[((blk8 + (lane >> 3)) << 9) | (z << 3) | (lane & 7)]);
    }
    if (acc == 0x12345678u) s[blockIdx.x] = acc;
}
template <class T> __device__ __forceinline__ void k_write_body(T *out, size_t n, T v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void k_write_b1(uint8_t *o, size_t n) { k_write_body<uint8_t>(o, n, 7); }
__global__ __launch_bounds__(256) void k_write_b4(uint32_t *o, size_t n) { k_write_body<uint32_t>(o, n, 7u); }
__global__ __launch_bounds__(256) void k_write_b8(uint2 *o, size_t n) { k_write_body<uint2>(o, n, make_uint2(7u, 9u)); }
__global__ __launch_bounds__(256) void k_write_b16(uint4 *o, size_t n) { k_write_body<uint4>(o, n, make_uint4(7u, 9u, 1u, 2u)); }

int main()
{
    void *buf = nullptr; uint32_t *sink = nullptr;
    CK(hipMalloc(&buf, BYTES)); CK(hipMalloc((void **)&sink, 1 << 20));
    CK(hipMemset(buf, 1, BYTES));
    const dim3 g(256 * 16), b(256);
    const uint32_t magic = (uint32_t)(getenv("PMC_CALIB_MAGIC") ? atoi(getenv("PMC_CALIB_MAGIC")) : 3);      // (sums of bytes of value 1 over >= 128 K elements per thread never equal 3)
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read_b1, g, b, 0, 0, (const uint8_t *)buf, sink, BYTES, magic);
        hipLaunchKernelGGL(k_read_b2, g, b, 0, 0, (const uint16_t *)buf, sink, BYTES / 2, magic);
        hipLaunchKernelGGL(k_read_b4, g, b, 0, 0, (const uint32_t *)buf, sink, BYTES / 4, magic);
        hipLaunchKernelGGL(k_read_b8, g, b, 0, 0, (const uint2 *)buf, sink, BYTES / 8, magic);
        hipLaunchKernelGGL(k_read_b16, g, b, 0, 0, (const uint4 *)buf, sink, BYTES / 16, magic);
        hipLaunchKernelGGL(k_read_b8_runs, g, b, 0, 0, (const uint2 *)buf, sink, BYTES / 8);
        hipLaunchKernelGGL(k_write_b1, g, b, 0, 0, (uint8_t *)buf, BYTES);
        hipLaunchKernelGGL(k_write_b4, g, b, 0, 0, (uint32_t *)buf, BYTES / 4);
        hipLaunchKernelGGL(k_write_b8, g, b, 0, 0, (uint2 *)buf, BYTES / 8);
        hipLaunchKernelGGL(k_write_b16, g, b, 0, 0, (uint4 *)buf, BYTES / 16);
        CK(hipDeviceSynchronize());
    }
    printf("pmc_calib: every kernel moved %zu bytes (%.1f MiB), 3 launches each\n", BYTES, BYTES / 1048576.0);
    return 0;
}
