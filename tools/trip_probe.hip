// tools/trip_probe.hip — measurement only (not part of the library): what ONE dependent round trip of a block-run of the waves
// kernel costs (a wave fetches P 512-byte rows whose addresses depend on the rows before), by where the rows lie:
//   planes : row p of a block in plane p of the pool (the library's layout: one array per field, 4 KB per block and plane)
//   record : the P rows of a block next to each other (one record per block)
// with agent-scope (`sc1`) and plain loads, by the number of wavefronts that do the same at the same time.
//   hipcc --offload-arch=gfx950 -O3 tools/trip_probe.hip -o tools/bin/trip_probe && tools/bin/trip_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool COHERENT> __device__ __forceinline__ uint64_t ld(const uint64_t *p)
{
    if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// rows of 512 B = 64 lanes x 8 B; a block has P rows per "trip" here (one row of each of P planes)
template <bool COHERENT, int P, bool RECORD> __global__ __launch_bounds__(512) void k_trips(const uint64_t *buf, size_t pool, int trips, int waves_per_wg, uint64_t *out, uint32_t seed)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_per_wg) return;
    uint64_t slot = ((uint64_t)(blockIdx.x * 8 + wave) * 2654435761u + seed) % pool;
    uint64_t acc = 0;
    for (int t = 0; t < trips; t++) {
        uint64_t v[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const size_t row = RECORD ? (slot * P + p) : ((size_t)p * pool + slot);          /* in units of 4 KB blocks-of-a-plane */
            v[p] = ld<COHERENT>(&buf[row * 512 + lane + 64 * (t & 7)]);
        }
        uint64_t s = 0;
#pragma unroll
        for (int p = 0; p < P; p++) s += v[p];
        acc += s;
        slot = (slot * 6364136223846793005ull + 1442695040888963407ull + (s & 1)) % pool;   /* the next block depends on what came back */
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

template <bool COHERENT, int P, bool RECORD> static float run(const uint64_t *buf, size_t pool, int trips, int wgs, int wpw, uint64_t *d_out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_trips<COHERENT, P, RECORD>), dim3(wgs), dim3(512), 0, 0, buf, pool, trips, wpw, d_out, (uint32_t)(rep * 977 + 13));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    return best * 1e6f / trips;      /* ns per trip */
}

int main()
{
    constexpr int P = 5;
    const size_t pool = 1100000;                       /* blocks: the 512^3 mapper's pool */
    const size_t bytes = pool * P * 4096;
    uint64_t *buf, *d_out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&d_out, 8));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    printf("pool %zu blocks, %d rows of 512 B per trip, %.1f GB\n", pool, P, bytes / 1e9);
    printf("%-28s %10s %10s %10s %10s\n", "ns per dependent trip", "planes", "record", "planes.sc1", "record.sc1");
    const int trips = 200;
    const int cfg[][2] = { { 1, 1 }, { 160, 1 }, { 160, 8 }, { 256, 8 } };
    for (auto &c : cfg) {
        const float a = run<false, P, false>(buf, pool, trips, c[0], c[1], d_out);
        const float b = run<false, P, true>(buf, pool, trips, c[0], c[1], d_out);
        const float cc = run<true, P, false>(buf, pool, trips, c[0], c[1], d_out);
        const float d = run<true, P, true>(buf, pool, trips, c[0], c[1], d_out);
        printf("%4d workgroups x %d waves     %10.0f %10.0f %10.0f %10.0f\n", c[0], c[1], a, b, cc, d);
    }
    /* the same with a small pool (everything within a few MB: TLB and L2 hits) */
    {
        const size_t small = 2000;
        const float a = run<false, P, false>(buf, small, trips, 160, 8, d_out);
        const float cc = run<true, P, false>(buf, small, trips, 160, 8, d_out);
        printf("160 x 8, pool of %zu blocks   %10.0f %10s %10.0f\n", small, a, "", cc);
    }
    return 0;
}
