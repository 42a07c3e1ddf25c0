// tools/sweep_probe.hip — measurement only (not part of the library): what a streaming sweep with the byte mix of the
// Mark + commit kernel can reach on this device, by access geometry.
//   hipcc --offload-arch=gfx950 -O3 tools/sweep_probe.hip -o /tmp/sweep_probe && /tmp/sweep_probe
// N = 512^3 voxels.  Local planes are x-fastest (type 1 B, bcoc 4 B, pair 8 B, edt 4 B); "global" planes are laid out in 8x8x8
// blocks (dist 4 B, coc 8 B; in-block index x | y<<3 | z<<6), blocks in x-fastest order (identity block table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static const int X = 512, Y = 512, Z = 512;
static const size_t N = (size_t)X * Y * Z;

struct P { const int8_t *type; const uint32_t *bcoc; uint64_t *pair; float *edt; int32_t *gdist; uint64_t *gcoc; uint64_t *gpair; };

__device__ __forceinline__ size_t gaddr(int x, int y, int z)
{ return ((size_t)(((z >> 3) * (Y >> 3) + (y >> 3)) * (X >> 3) + (x >> 3)) << 9) | (size_t)(((z & 7) << 6) | ((y & 7) << 3) | (x & 7)); }

// the work of one voxel: mode bits — 1: read old record (dist + coc), 2: write gpair, 4: write edt, 8: write gdist
template <int MODE> __device__ __forceinline__ void one(const P &p, size_t id, size_t a, uint32_t bc, int8_t ty, int dold, uint64_t ococ)
{
    if (ty == 0) return;
    const uint32_t d = (bc & 1023u) * (bc & 1023u) + ((bc >> 10) & 1023u) + (uint32_t)(dold & 1) + (uint32_t)(ococ & 1);
    const uint64_t pr = ((uint64_t)d << 42) | bc;
    p.pair[id] = pr;
    if (MODE & 4) p.edt[id] = sqrtf((float)d);
    p.gcoc[a] = pr ^ 0x5555u;
    if (MODE & 8) p.gdist[a] = (int)d;
    if (MODE & 2) p.gpair[a] = pr;
}

// geometry A: thread = z-column of 8 voxels, lanes 32 along x * 2 along y (the library's sweep)
template <int MODE, int LX> __global__ __launch_bounds__(256) void k_zcol(const P p)
{
    constexpr int LY = 64 / LX, WY = 4 * LY;
    const int lane = threadIdx.x & 63;
    const int gx = X / LX, gy = Y / WY, gz = Z / 8, nv = gx * gy * gz;
    const int per = (nv + gridDim.x - 1) / gridDim.x;
    const int lx = lane % LX, ly = (threadIdx.x >> 6) * LY + lane / LX;
    for (int v = blockIdx.x * per; v < nv && v < (int)(blockIdx.x + 1) * per; v++) {
        const int x = (v % gx) * LX + lx, y = ((v / gx) % gy) * WY + ly, z0 = (v / (gx * gy)) * 8;
        uint32_t bc[8]; int8_t ty[8]; int dold[8]; uint64_t oc[8]; size_t id[8], a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { id[k] = ((size_t)(z0 + k) * Y + y) * X + x; a[k] = gaddr(x, y, z0 + k); ty[k] = p.type[id[k]]; bc[k] = p.bcoc[id[k]]; }
#pragma unroll
        for (int k = 0; k < 8; k++) { dold[k] = 0; oc[k] = 0; if (MODE & 1) { dold[k] = p.gdist[a[k]]; oc[k] = p.gcoc[a[k]]; } }
#pragma unroll
        for (int k = 0; k < 8; k++) one<MODE>(p, id[k], a[k], bc[k], ty[k], dold[k], oc[k]);
    }
}

// geometry B: linear over the local planes, thread = 4 voxels along x (16-byte vectors for the 4-byte planes), grid-stride
template <int MODE> __global__ __launch_bounds__(256) void k_lin4(const P p)
{
    const size_t nq = N / 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const size_t id0 = q * 4;
        const int x = (int)(id0 % X), y = (int)((id0 / X) % Y), z = (int)(id0 / ((size_t)X * Y));
        const uint4 b4 = reinterpret_cast<const uint4 *>(p.bcoc)[q];
        const uint32_t t4 = reinterpret_cast<const uint32_t *>(p.type)[q];
        const size_t a0 = gaddr(x, y, z);          // 4 consecutive x inside one block row
        int4 d4 = make_int4(0, 0, 0, 0); uint64_t oc[4] = { 0, 0, 0, 0 };
        if (MODE & 1) { d4 = *reinterpret_cast<const int4 *>(p.gdist + a0); const ulonglong2 c01 = *reinterpret_cast<const ulonglong2 *>(p.gcoc + a0), c23 = *reinterpret_cast<const ulonglong2 *>(p.gcoc + a0 + 2); oc[0] = c01.x; oc[1] = c01.y; oc[2] = c23.x; oc[3] = c23.y; }
        const uint32_t bc[4] = { b4.x, b4.y, b4.z, b4.w }; const int dd[4] = { d4.x, d4.y, d4.z, d4.w };
        uint64_t pr[4]; float e[4]; int gd[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t d = (bc[i] & 1023u) * (bc[i] & 1023u) + ((bc[i] >> 10) & 1023u) + (uint32_t)(dd[i] & 1) + (uint32_t)(oc[i] & 1) + ((t4 >> (8 * i)) & 1u);
            pr[i] = ((uint64_t)d << 42) | bc[i]; e[i] = sqrtf((float)d); gd[i] = (int)d;
        }
        ulonglong2 w01, w23; w01.x = pr[0]; w01.y = pr[1]; w23.x = pr[2]; w23.y = pr[3];
        reinterpret_cast<ulonglong2 *>(p.pair + id0)[0] = w01; reinterpret_cast<ulonglong2 *>(p.pair + id0)[1] = w23;
        if (MODE & 4) *reinterpret_cast<float4 *>(p.edt + id0) = make_float4(e[0], e[1], e[2], e[3]);
        w01.x ^= 0x5555u;
        reinterpret_cast<ulonglong2 *>(p.gcoc + a0)[0] = w01; reinterpret_cast<ulonglong2 *>(p.gcoc + a0)[1] = w23;
        if (MODE & 8) *reinterpret_cast<int4 *>(p.gdist + a0) = make_int4(gd[0], gd[1], gd[2], gd[3]);
        if (MODE & 2) { reinterpret_cast<ulonglong2 *>(p.gpair + a0)[0] = w01; reinterpret_cast<ulonglong2 *>(p.gpair + a0)[1] = w23; }
    }
}

// geometry C: wave = one 8x8x8 block/tile; lane = (x, y) row pair ... thread = 8 voxels along x of one (y, z) row, 64 rows per wave:
// every global access is a 32/64-byte piece of a 2/4 KB contiguous block plane, every local access a 32/64-byte row piece
template <int MODE> __global__ __launch_bounds__(256) void k_blockrows(const P p)
{
    const int nb = (X / 8) * (Y / 8) * (Z / 8);
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int per = (nb + nw - 1) / nw;
    for (int b = wid * per; b < nb && b < (wid + 1) * per; b++) {
        const int bx = b % (X / 8), by = (b / (X / 8)) % (Y / 8), bz = b / ((X / 8) * (Y / 8));
        const int y = by * 8 + (lane & 7), z = bz * 8 + (lane >> 3), x0 = bx * 8;
        const size_t id0 = ((size_t)z * Y + y) * X + x0, a0 = ((size_t)b << 9) | (size_t)(lane << 3);
        const uint4 b0 = *reinterpret_cast<const uint4 *>(p.bcoc + id0), b1 = *reinterpret_cast<const uint4 *>(p.bcoc + id0 + 4);
        const uint64_t t8 = *reinterpret_cast<const uint64_t *>(p.type + id0);
        int dd[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; uint64_t oc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (MODE & 1) {
            const int4 d0 = *reinterpret_cast<const int4 *>(p.gdist + a0), d1 = *reinterpret_cast<const int4 *>(p.gdist + a0 + 4);
            dd[0] = d0.x; dd[1] = d0.y; dd[2] = d0.z; dd[3] = d0.w; dd[4] = d1.x; dd[5] = d1.y; dd[6] = d1.z; dd[7] = d1.w;
#pragma unroll
            for (int i = 0; i < 4; i++) { const ulonglong2 c2 = *reinterpret_cast<const ulonglong2 *>(p.gcoc + a0 + 2 * i); oc[2 * i] = c2.x; oc[2 * i + 1] = c2.y; }
        }
        const uint32_t bc[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
        uint64_t pr[8]; float e[8]; int gd[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t d = (bc[i] & 1023u) * (bc[i] & 1023u) + ((bc[i] >> 10) & 1023u) + (uint32_t)(dd[i] & 1) + (uint32_t)(oc[i] & 1) + (uint32_t)((t8 >> (8 * i)) & 1u);
            pr[i] = ((uint64_t)d << 42) | bc[i]; e[i] = sqrtf((float)d); gd[i] = (int)d;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { ulonglong2 w; w.x = pr[2 * i]; w.y = pr[2 * i + 1]; reinterpret_cast<ulonglong2 *>(p.pair + id0)[i] = w; w.x ^= 0x5555u; reinterpret_cast<ulonglong2 *>(p.gcoc + a0)[i] = w;
                                      if (MODE & 2) reinterpret_cast<ulonglong2 *>(p.gpair + a0)[i] = w; }
        if (MODE & 4) { *reinterpret_cast<float4 *>(p.edt + id0) = make_float4(e[0], e[1], e[2], e[3]); *reinterpret_cast<float4 *>(p.edt + id0 + 4) = make_float4(e[4], e[5], e[6], e[7]); }
        if (MODE & 8) { *reinterpret_cast<int4 *>(p.gdist + a0) = make_int4(gd[0], gd[1], gd[2], gd[3]); *reinterpret_cast<int4 *>(p.gdist + a0 + 4) = make_int4(gd[4], gd[5], gd[6], gd[7]); }
    }
}

// ceilings: pure linear copy (read R bytes, write W bytes per "voxel" through 16-byte vectors)
__global__ __launch_bounds__(256) void k_copy(const uint4 *src, uint4 *dst, size_t nr, size_t nw)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t; i < nr; i += stride) { const uint4 v = src[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    for (size_t i = t; i < nw; i += stride) dst[i] = make_uint4(acc.x + (uint32_t)i, acc.y, acc.z, acc.w);
}

template <class F> static float timeit(F f, int reps = 5)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best;
}

int main()
{
    P p; void *q;
    CK(hipMalloc(&q, N)); CK(hipMemset(q, 1, N)); p.type = (int8_t *)q;
    CK(hipMalloc(&q, N * 4)); CK(hipMemset(q, 3, N * 4)); p.bcoc = (uint32_t *)q;
    CK(hipMalloc(&q, N * 8)); p.pair = (uint64_t *)q;
    CK(hipMalloc(&q, N * 4)); p.edt = (float *)q;
    CK(hipMalloc(&q, N * 4)); CK(hipMemset(q, 0, N * 4)); p.gdist = (int32_t *)q;
    CK(hipMalloc(&q, N * 8)); CK(hipMemset(q, 0, N * 8)); p.gcoc = (uint64_t *)q;
    CK(hipMalloc(&q, N * 8)); p.gpair = (uint64_t *)q;
    int cus = 256; { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount; }
    auto rep = [&](const char *name, float ms, double rb, double wb) {
        printf("%-44s %7.3f ms   R %5.2f GB  W %5.2f GB   %6.2f TB/s\n", name, ms, rb * N / 1e9, wb * N / 1e9, (rb + wb) * N / ms / 1e9);
    };
    // ceilings
    for (int wfrac = 0; wfrac <= 4; wfrac++) {
        const double tot = 29.0, wb = tot * wfrac / 4.0, rb = tot - wb;
        const size_t nr = (size_t)(rb * N / 16), nw = (size_t)(wb * N / 16);
        const uint4 *src = (const uint4 *)p.gpair; uint4 *dst = (uint4 *)p.pair;     // 1 GB each: loop over them
        const size_t cap = N * 8 / 16;
        const float ms = timeit([&] { size_t r = nr, w = nw; while (r || w) { const size_t cr = r < cap ? r : cap, cw = w < cap ? w : cap; hipLaunchKernelGGL(k_copy, dim3(cus * 16), dim3(256), 0, 0, src, dst, cr, cw); r -= cr; w -= cw; } });
        char nm[64]; snprintf(nm, sizeof nm, "linear copy, %d/4 of 29 B/voxel written", wfrac); rep(nm, ms, rb, wb);
    }
    const int grid = cus * 32;
#define RUN(K, MODE, name) do { const float ms = timeit([&] { hipLaunchKernelGGL((K), dim3(grid), dim3(256), 0, 0, p); }); \
        rep(name, ms, 5.0 + ((MODE) & 1 ? 12.0 : 0.0), 16.0 + ((MODE) & 2 ? 8.0 : 0.0) + ((MODE) & 4 ? 4.0 : 0.0) + ((MODE) & 8 ? 4.0 : 0.0)); } while (0)
    RUN((k_zcol<15, 32>), 15, "z-columns 32x2, round-2 bytes (49 B)");
    RUN((k_zcol<13, 32>), 13, "z-columns 32x2, no gpair (41 B)");
    RUN((k_zcol<12, 32>), 12, "z-columns 32x2, no gpair, no old read (29 B)");
    RUN((k_zcol<12, 64>), 12, "z-columns 64x1, no gpair, no old read");
    RUN((k_zcol<12, 16>), 12, "z-columns 16x4, no gpair, no old read");
    RUN((k_zcol<8, 32>), 8, "z-columns 32x2, pair + gcoc + gdist only (25 B)");
    RUN((k_zcol<0, 32>), 0, "z-columns 32x2, pair + gcoc only (21 B)");
    RUN((k_lin4<15>), 15, "linear x4, round-2 bytes (49 B)");
    RUN((k_lin4<13>), 13, "linear x4, no gpair (41 B)");
    RUN((k_lin4<12>), 12, "linear x4, no gpair, no old read (29 B)");
    RUN((k_lin4<0>), 0, "linear x4, pair + gcoc only (21 B)");
    RUN((k_blockrows<15>), 15, "block rows x8, round-2 bytes (49 B)");
    RUN((k_blockrows<13>), 13, "block rows x8, no gpair (41 B)");
    RUN((k_blockrows<12>), 12, "block rows x8, no gpair, no old read (29 B)");
    RUN((k_blockrows<0>), 0, "block rows x8, pair + gcoc only (21 B)");
    return 0;
}
