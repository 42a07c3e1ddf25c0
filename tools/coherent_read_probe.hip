// tools/coherent_read_probe.hip — measurement only (not part of the library): the rate at which one wavefront per 8x8x8 block can
// fetch block records the way the waves kernel does (512 voxels x 8 bytes per plane, a few planes per block, blocks scattered
// over a large pool), with agent-scope (write-through-coherent, `sc1`) loads against plain loads, by workgroups and block count.
//   hipcc --offload-arch=gfx950 -O3 tools/coherent_read_probe.hip -o tools/bin/coherent_read_probe && tools/bin/coherent_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool COHERENT> __device__ __forceinline__ uint64_t ld(const uint64_t *p)
{
    if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// one wave per listed block: PLANES x 8 loads of 8 bytes per lane (the block's records), then — dependent on the first batch, like
// the halo behind the neighbour lookups — 6 more loads from six other blocks
template <bool COHERENT, int PLANES, bool HALO> __global__ __launch_bounds__(512) void k_blocks(const uint64_t *const *planes, const int *list, int n, size_t pool_blocks, uint64_t *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t acc = 0;
    for (int i = (int)blockIdx.x + (int)gridDim.x * wave; i < n; i += (int)gridDim.x * 8) {
        const size_t base = (size_t)list[i] * 512;
        uint64_t v[PLANES][8];
#pragma unroll
        for (int p = 0; p < PLANES; p++)
#pragma unroll
            for (int j = 0; j < 8; j++) v[p][j] = ld<COHERENT>(&planes[p][base + lane + 64 * j]);
        uint64_t s = 0;
#pragma unroll
        for (int p = 0; p < PLANES; p++)
#pragma unroll
            for (int j = 0; j < 8; j++) s += v[p][j];
        if (HALO) {
            uint64_t h[6];
#pragma unroll
            for (int f = 0; f < 6; f++) { const size_t nb = ((size_t)list[i] * 2654435761u + f * 40503u + (s & 1)) % pool_blocks; h[f] = ld<COHERENT>(&planes[0][nb * 512 + lane]); }
#pragma unroll
            for (int f = 0; f < 6; f++) s += h[f];
        }
        acc += s;
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

template <bool COHERENT, int PLANES, bool HALO> static float run(const uint64_t *const *d_planes, const int *d_list, int n, size_t pool, uint64_t *d_out, int wgs)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_blocks<COHERENT, PLANES, HALO>), dim3(wgs), dim3(512), 0, 0, d_planes, d_list, n, pool, d_out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    return best;
}

int main()
{
    const size_t pool = 360000;                       // blocks in the pool (the C5 run holds 360 k)
    const int PL = 3;
    uint64_t *planes[PL]; const uint64_t **d_planes; uint64_t *d_out;
    for (int p = 0; p < PL; p++) { CK(hipMalloc(&planes[p], pool * 512 * 8)); CK(hipMemset(planes[p], 1, pool * 512 * 8)); }
    CK(hipMalloc(&d_planes, sizeof(planes))); CK(hipMemcpy(d_planes, planes, sizeof(planes), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 8));
    std::mt19937 rng(3);
    printf("%-8s %-9s %-6s %-5s %9s %9s %10s\n", "blocks", "loads", "halo", "WGs", "us", "GB/s", "us/1000blk");
    for (int n : { 700, 2400, 8000, 32000 }) {
        std::vector<int> list(n);
        for (int i = 0; i < n; i++) list[i] = (int)(rng() % pool);
        int *d_list; CK(hipMalloc(&d_list, n * 4)); CK(hipMemcpy(d_list, list.data(), n * 4, hipMemcpyHostToDevice));
        for (int wgs : { 128, 256 }) {
            const double bytes = (double)n * (PL * 4096);
            struct { const char *name; float ms; const char *halo; } r[4] = {
                { "plain", run<false, PL, false>(d_planes, d_list, n, pool, d_out, wgs), "no" },
                { "coherent", run<true, PL, false>(d_planes, d_list, n, pool, d_out, wgs), "no" },
                { "plain", run<false, PL, true>(d_planes, d_list, n, pool, d_out, wgs), "yes" },
                { "coherent", run<true, PL, true>(d_planes, d_list, n, pool, d_out, wgs), "yes" } };
            for (auto &e : r) printf("%-8d %-9s %-6s %-5d %9.1f %9.0f %10.1f\n", n, e.name, e.halo, wgs, e.ms * 1000.0, bytes / (e.ms * 1e-3) / 1e9, e.ms * 1000.0 / n * 1000.0);
        }
        CK(hipFree(d_list));
    }
    return 0;
}
