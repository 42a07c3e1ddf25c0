/*
 * gie_kernels.hip.h — gfx950 kernels of the map update.  HIP only (never built for the host).
 *
 *  k_vox<F>           streaming per-voxel sweeps (classify, fuse, Mark, frontiers, commit):
 *                     one wave = 64 consecutive x → every plane access is a coalesced segment.
 *  k_edt_y            EDT pass Y (EDTphase1, local_edt_core.h:14-82): the column's occupancy
 *                     lives in registers as a bit mask; one read of _glb_type, one 2-byte write.
 *  k_edt_x / k_edt_z  EDT passes X/Z (EDTphase2/3, :84-193): the Meijster lower envelope is
 *                     computed wave-cooperatively in LDS (monotone divide & conquer argmin), no
 *                     stacks in HBM, no physical transposes (replaces the six cuTT calls,
 *                     local_edt.cu:12-25).
 *  k_wave_a/b/c       persistent level-synchronous BFS (parWave + BFS_in_block/BFS_one_layer,
 *                     wave_helper.h:8-93, wave_core.cuh:395-523) with no host round trip.
 */
#ifndef GIE_KERNELS_HIP_H
#define GIE_KERNELS_HIP_H

#include <hip/hip_runtime.h>
#include "gie_functors.h"

/* ------------------------------------------------------------------ per-voxel sweeps */
#define GIE_VOX_BX 64
#define GIE_VOX_BY 4

template <class F>
__global__ __launch_bounds__(GIE_VOX_BX *GIE_VOX_BY) void k_vox(const gie_ctx c, const F f)
{
    const int x = blockIdx.x * GIE_VOX_BX + threadIdx.x;
    const int y = blockIdx.y * GIE_VOX_BY + threadIdx.y;
    const int z = blockIdx.z;
    if (x < c.X && y < c.Y) f(c, x, y, z);
}

template <class F>
__global__ __launch_bounds__(256) void k_lin(const gie_ctx c, const F f, const int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) f(c, i);
}

/* one workgroup per table cell: initialise the 512 voxels of a block created this frame */
__global__ __launch_bounds__(256) void k_block_init(const gie_ctx c, const int32_t *flag, const int32_t *rank)
{
    const int cell = blockIdx.x;
    if (!flag[cell]) return;
    const int slot = *c.pool_count + rank[cell];
    if (slot >= c.max_blocks) return;
    gie_init_voxel(c, slot, threadIdx.x);
    gie_init_voxel(c, slot, threadIdx.x + 256);
}
__global__ void k_pool_advance(const gie_ctx c, const int32_t *flag, const int32_t *rank, int ncell)
{
    const int total = rank[ncell - 1] + flag[ncell - 1];
    int pc = *c.pool_count + total;
    if (pc > c.max_blocks) pc = c.max_blocks;
    *c.pool_count = pc;
    c.cnt[GIE_CNT_NEWBLK] = total;
}

/* ------------------------------------------------------------------ EDT pass Y */
/* Nearest occupied voxel along y for every (x,z) column; ties go to the larger y
 * (EDTphase1's backward sweep overwrites on '<', local_edt_core.h:65-81). */
template <int YW>
__global__ __launch_bounds__(256) void k_edt_y(const gie_ctx c)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int z = blockIdx.y;
    if (x >= c.X) return;
    const int X = c.X, Y = c.Y;
    const int8_t *t = c.glb_type + (size_t)z * X * Y + x;
    uint32_t bits[YW];
#pragma unroll
    for (int w = 0; w < YW; w++) {
        uint32_t m = 0;
        if (w * 32 < Y) {
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int y = w * 32 + k;
                if (y < Y) m |= (uint32_t)(t[(size_t)y * X] == GIE_VOX_OCCUPIED) << k;
            }
        }
        bits[w] = m;
    }
    int prev_last[YW], next_first[YW];
    int last = -1;
#pragma unroll
    for (int w = 0; w < YW; w++) { prev_last[w] = last; if (bits[w]) last = w * 32 + 31 - __clz(bits[w]); }
    int first = -1;
#pragma unroll
    for (int w = YW - 1; w >= 0; w--) { next_first[w] = first; if (bits[w]) first = w * 32 + __ffs(bits[w]) - 1; }
    uint16_t *out = c.cy1 + (size_t)z * X * Y + x;
#pragma unroll
    for (int w = 0; w < YW; w++) {
        if (w * 32 >= Y) break;
        const uint32_t bw = bits[w];
        const int pl = prev_last[w], nf = next_first[w];
        for (int k = 0; k < 32; k++) {
            const int y = w * 32 + k;
            if (y >= Y) break;
            const uint32_t lo = bw & (0xffffffffu >> (31 - k));
            const int below = lo ? w * 32 + 31 - __clz(lo) : pl;
            const uint32_t hi = bw >> k;
            const int above = hi ? y + __ffs(hi) - 1 : nf;
            int r;
            if (above >= 0 && (below < 0 || above - y <= y - below)) r = above; else r = below;
            out[(size_t)y * X] = r < 0 ? (uint16_t)0xffff : (uint16_t)r;
        }
    }
}

/* ------------------------------------------------------------------ lower-envelope argmin */
/* One wave computes, for every position u of a row of L sites (L <= 64*CP), the site
 *     argmin_i (u-i)² + a_i     with ties going to the smaller i
 * which is exactly what the reference's Meijster scan with truncating Sep() returns
 * (tests/test_oracle_edt.py::test_meijster_tie_rule pins the equivalence).
 * b[i] = (a_i << 10) | i in LDS; 32-bit keys, so (max a) + L² must stay below 2^22.
 * The argmin is monotone in u, so after a brute-force pass for every CP-th position the rest
 * is found by divide & conquer inside [site(left), site(right)]. */
template <int CP>
__device__ __forceinline__ void gie_row_argmin(const uint32_t *b, uint16_t *site, const int L, const int lane)
{
    const int u0 = lane * CP;
    {
        uint32_t best = 0xffffffffu;
        int dp = u0 << 5;
        const int L4 = L & ~3;
        const uint4 *b4 = reinterpret_cast<const uint4 *>(b);
        for (int i = 0; i < L4; i += 4) {
            const uint4 v = b4[i >> 2];
            best = min(best, (uint32_t)__mul24(dp, dp) + v.x); dp -= 32;
            best = min(best, (uint32_t)__mul24(dp, dp) + v.y); dp -= 32;
            best = min(best, (uint32_t)__mul24(dp, dp) + v.z); dp -= 32;
            best = min(best, (uint32_t)__mul24(dp, dp) + v.w); dp -= 32;
        }
        for (int i = L4; i < L; i++) { best = min(best, (uint32_t)__mul24(dp, dp) + b[i]); dp -= 32; }
        site[u0] = (u0 < L) ? (uint16_t)(best & 1023u) : (uint16_t)(L - 1);
        if (lane == 63) site[64 * CP] = (uint16_t)(L - 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int step = CP / 2; step >= 1; step >>= 1) {
#pragma unroll
        for (int m = step; m < CP; m += 2 * step) {
            const int u = u0 + m;
            const int lo = site[u - step], hi = site[u + step];
            uint32_t best = 0xffffffffu;
            if (u < L) {
                int dp = (u - lo) << 5;
                for (int i = lo; i <= hi; i++) { best = min(best, (uint32_t)__mul24(dp, dp) + b[i]); dp -= 32; }
            }
            site[u] = (u < L) ? (uint16_t)(best & 1023u) : (uint16_t)(L - 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

/* ------------------------------------------------------------------ EDT pass X */
/* one wave per (y,z) row; rows are contiguous in memory so loads/stores are coalesced */
#define GIE_EDTX_WAVES 4
template <int CP>
__global__ __launch_bounds__(64 * GIE_EDTX_WAVES) void k_edt_x(const gie_ctx c)
{
    constexpr int LP = 64 * CP;
    __shared__ __attribute__((aligned(16))) uint32_t s_b[GIE_EDTX_WAVES][LP];
    __shared__ uint16_t s_site[GIE_EDTX_WAVES][LP + 2];
    __shared__ uint16_t s_cy[GIE_EDTX_WAVES][LP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * GIE_EDTX_WAVES + wave;        /* row = z*Y + y */
    if (row >= c.Y * c.Z) return;
    const int X = c.X;
    const int y = row % c.Y;
    const uint32_t a_inf = (uint32_t)c.max_loc_dist_sq + 1u;
    const uint16_t *in = c.cy1 + (size_t)row * X;
    uint32_t *b = s_b[wave];
    uint16_t *site = s_site[wave];
    uint16_t *cyr = s_cy[wave];
    int any = 0;
    for (int i = lane; i < X; i += 64) {
        const uint16_t cy = in[i];
        cyr[i] = cy;
        uint32_t a = a_inf;
        if (cy != 0xffff) { const int d = y - (int)cy; a = (uint32_t)(d * d); any = 1; }
        b[i] = (a << 10) | (uint32_t)i;
    }
    uint32_t *out = c.cxy2 + (size_t)row * X;
    if (!__any(any)) {                                          /* slice without obstacle */
        for (int i = lane; i < X; i += 64) out[i] = 0xffffffffu;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    gie_row_argmin<CP>(b, site, X, lane);
    for (int i = lane; i < X; i += 64) {
        const int s = site[i];
        out[i] = (uint32_t)s | ((uint32_t)cyr[s] << 16);       /* a valid site always wins when one exists */
    }
}

/* ------------------------------------------------------------------ EDT pass Z */
/* one workgroup per (y, tile of TX columns): the tile [Z][TX] of pass-X results is staged in
 * LDS with coalesced loads, each wave runs the envelope along z for its columns, the results
 * are staged back and written with coalesced stores. */
#define GIE_EDTZ_TX 16
#define GIE_EDTZ_TS 17 /* padded LDS row stride: column walks hit distinct banks */
#define GIE_EDTZ_WAVES 4
template <int CP>
__global__ __launch_bounds__(64 * GIE_EDTZ_WAVES) void k_edt_z(const gie_ctx c)
{
    constexpr int LP = 64 * CP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Z = c.Z, X = c.X, Y = c.Y;
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem);                    /* [Z][TX] cx|cy<<16 → bcoc */
    uint32_t *dtile = tile + (size_t)Z * GIE_EDTZ_TS;                       /* [Z][TX] dist² out        */
    uint32_t *s_b = dtile + (((size_t)Z * GIE_EDTZ_TS + 3) & ~(size_t)3);                        /* [WAVES][LP]              */
    uint16_t *s_site = reinterpret_cast<uint16_t *>(s_b + GIE_EDTZ_WAVES * LP); /* [WAVES][LP+2]        */
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * GIE_EDTZ_TX, y = blockIdx.y;
    const int tx = threadIdx.x & (GIE_EDTZ_TX - 1), tz = threadIdx.x / GIE_EDTZ_TX;
    const size_t plane = (size_t)X * Y;
    for (int z = tz; z < Z; z += (64 * GIE_EDTZ_WAVES) / GIE_EDTZ_TX) {
        const int x = x0 + tx;
        tile[z * GIE_EDTZ_TS + tx] = (x < X) ? c.cxy2[(size_t)z * plane + (size_t)y * X + x] : 0xffffffffu;
    }
    __syncthreads();
    const uint32_t a_inf = (uint32_t)c.max_loc_dist_sq + 1u;
    const uint32_t mw2 = (uint32_t)(c.max_width * c.max_width);
    uint32_t *b = s_b + wave * LP;
    uint16_t *site = s_site + wave * (LP + 2);
    for (int col = wave; col < GIE_EDTZ_TX; col += GIE_EDTZ_WAVES) {
        const int x = x0 + col;
        if (x >= X) break;
        int any = 0;
        for (int i = lane; i < Z; i += 64) {
            const uint32_t v = tile[i * GIE_EDTZ_TS + col];
            uint32_t a = a_inf;
            if (v != 0xffffffffu) {
                const int dx = x - (int)(v & 0xffffu), dy = y - (int)(v >> 16);
                a = (uint32_t)(dx * dx + dy * dy); any = 1;
            }
            b[i] = (a << 10) | (uint32_t)i;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!__any(any)) {                                      /* the whole volume is empty */
            for (int i = lane; i < Z; i += 64) { dtile[i * GIE_EDTZ_TS + col] = mw2; tile[i * GIE_EDTZ_TS + col] = GIE_BCOC_NONE; }
        } else {
            gie_row_argmin<CP>(b, site, Z, lane);
            /* gather first (own column only), then overwrite the column in place */
            uint32_t oc[CP], od[CP];
#pragma unroll
            for (int j = 0; j < CP; j++) {
                const int i = lane + 64 * j;
                if (i < Z) {
                    const int s = site[i];
                    const uint32_t v = tile[s * GIE_EDTZ_TS + col];
                    const int dz = i - s;
                    od[j] = (b[s] >> 10) + (uint32_t)(dz * dz);
                    oc[j] = gie_pack_bcoc((int)(v & 0xffffu), (int)(v >> 16), s);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < CP; j++) {
                const int i = lane + 64 * j;
                if (i < Z) { dtile[i * GIE_EDTZ_TS + col] = od[j]; tile[i * GIE_EDTZ_TS + col] = oc[j]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    for (int z = tz; z < Z; z += (64 * GIE_EDTZ_WAVES) / GIE_EDTZ_TX) {
        const int x = x0 + tx;
        if (x < X) {
            const size_t o = (size_t)z * plane + (size_t)y * X + x;
            c.aux[o] = (int32_t)dtile[z * GIE_EDTZ_TS + tx];
            c.bcoc[o] = tile[z * GIE_EDTZ_TS + tx];
        }
    }
}

/* ------------------------------------------------------------------ persistent BFS waves */
#define GIE_WAVE_THREADS 1024

__global__ __launch_bounds__(GIE_WAVE_THREADS) void k_wave_a(const gie_ctx c)
{
    __shared__ int s_n;
    const int tid = threadIdx.x;
    if (tid == 0) { s_n = gie_ld(&c.cnt[GIE_CNT_A]); c.cnt[GIE_CNT_SEED_A] = s_n; c.cnt[GIE_CNT_SEED_B] = gie_ld(&c.cnt[GIE_CNT_B]); }
    __syncthreads();
    int n = s_n, cur = 0;
    while (n > 0) {
        if (tid == 0) { gie_st(&c.cnt[GIE_CNT_NEXT], 0); c.cnt[GIE_CNT_VIS_A] += n; c.cnt[GIE_CNT_LVL_A] += 1; }
        for (int e = tid; e < n; e += GIE_WAVE_THREADS) gie_wave_a_phase1(c, c.qa[cur], e);
        __syncthreads();
        for (int e = tid; e < n; e += GIE_WAVE_THREADS) gie_wave_a_phase2(c, c.qa[cur], c.qa[cur ^ 1], e);
        __syncthreads();
        if (tid == 0) { int m = gie_ld(&c.cnt[GIE_CNT_NEXT]); s_n = m < c.qcap_ab ? m : c.qcap_ab; }
        __syncthreads();
        n = s_n; cur ^= 1;
        __syncthreads();
    }
}

__global__ __launch_bounds__(GIE_WAVE_THREADS) void k_wave_b(const gie_ctx c)
{
    __shared__ int s_n;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int m = gie_ld(&c.cnt[GIE_CNT_B]); s_n = m < c.qcap_ab ? m : c.qcap_ab;
        c.cnt[GIE_CNT_FRONT_B] = s_n; c.cnt[GIE_CNT_SEED_C] = gie_ld(&c.cnt[GIE_CNT_C]);
    }
    __syncthreads();
    int n = s_n, cur = 0, level = 0;
    while (n > 0) {
        if (tid == 0) { gie_st(&c.cnt[GIE_CNT_NEXT], 0); c.cnt[GIE_CNT_VIS_B] += n; c.cnt[GIE_CNT_LVL_B] += 1; }
        for (int e = tid; e < n; e += GIE_WAVE_THREADS) gie_wave_b_phase1(c, c.qb[cur], e);
        __syncthreads();
        for (int e = tid; e < n; e += GIE_WAVE_THREADS) gie_wave_b_phase2(c, c.qb[cur], c.qb[cur ^ 1], level, e);
        __syncthreads();
        for (int e = tid; e < n; e += GIE_WAVE_THREADS) gie_wave_b_phase3(c, c.qb[cur], e);
        __syncthreads();
        if (tid == 0) { int m = gie_ld(&c.cnt[GIE_CNT_NEXT]); s_n = m < c.qcap_ab ? m : c.qcap_ab; }
        __syncthreads();
        n = s_n; cur ^= 1; level++;
        __syncthreads();
    }
}

__global__ __launch_bounds__(GIE_WAVE_THREADS) void k_wave_c(const gie_ctx c, const int record_seeds)
{
    __shared__ int s_n;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int m = gie_ld(&c.cnt[GIE_CNT_C]); s_n = m < c.qcap_c ? m : c.qcap_c;
        c.cnt[GIE_CNT_FRONT_C] = s_n;
        if (record_seeds) { c.cnt[GIE_CNT_SEED_C] = s_n; c.cnt[GIE_CNT_SEED_A] = gie_ld(&c.cnt[GIE_CNT_A]); c.cnt[GIE_CNT_SEED_B] = gie_ld(&c.cnt[GIE_CNT_B]); }
    }
    __syncthreads();
    int n = s_n, cur = 0, level = 0;
    while (n > 0) {
        if (tid == 0) { gie_st(&c.cnt[GIE_CNT_NEXT], 0); c.cnt[GIE_CNT_VIS_C] += n; c.cnt[GIE_CNT_LVL_C] += 1; }
        for (int e = tid; e < n; e += GIE_WAVE_THREADS) gie_wave_c_phase1(c, c.qc[cur], e);
        __syncthreads();
        for (int e = tid; e < n; e += GIE_WAVE_THREADS) gie_wave_c_phase2(c, c.qc[cur], c.qc[cur ^ 1], level, e);
        __syncthreads();
        if (tid == 0) { int m = gie_ld(&c.cnt[GIE_CNT_NEXT]); s_n = m < c.qcap_c ? m : c.qcap_c; }
        __syncthreads();
        n = s_n; cur ^= 1; level++;
        __syncthreads();
    }
}

#endif /* GIE_KERNELS_HIP_H */
