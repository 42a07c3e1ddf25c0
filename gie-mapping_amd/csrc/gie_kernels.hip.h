/*
 * gie_kernels.hip.h — gfx950 kernels of the map update.  HIP only (never built for the host).
 *
 *  k_voxz<F>          streaming per-voxel sweep of the whole volume (projective OGM classify):
 *                     one wave = 64 consecutive x → every plane access is a coalesced segment.
 *  k_voxa<F>          the same functors over a tile list or the volume, chosen on the device
 *                     (fuse, Mark, obtainFrontiers, commit).
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

/* Stamps and per-section clock sums of the measurement builds (tools/measure/gie_timing.h, tools/wave_timing.py): no-ops here */
#if defined(GIE_WAVE_TIMING) || defined(GIE_RAY_TIMING)
#include "../../tools/measure/gie_timing.h"
#endif
#if !defined(GIE_WAVE_TIMING)
#define GIE_TS2(tag, n) do { } while (0)
#define GIE_WPROF_DECL do { } while (0)
#define GIE_WPROF_MARK(base) do { } while (0)
#define GIE_WPROF_ADD(i, v) do { } while (0)
#define GIE_WPROF_END(base) do { } while (0)
#define GIE_WPROF_DRAIN() do { } while (0)
#define GIE_WPROF_SUB(i) do { } while (0)
#define GIE_WPROF_SUBSTART() do { } while (0)
#define GIE_WPROF_DUMP() do { } while (0)
#define GIE_WPROF_CLK() 0ull
#define GIE_WAVE_TIMING_RESET(cond) do { } while (0)
#endif
#if !defined(GIE_RAY_TIMING)
#define GIE_RTS(k) do { } while (0)
#endif


/* ------------------------------------------------------------------ per-voxel sweeps */
#define GIE_VOX_BX 64
#define GIE_VOX_BY 4

/* volume sweep with a short z-column per thread (the sweeps over a mostly-unknown volume are bound
 * by workgroup dispatch and exposed latency, not by HBM); the skip tests of the whole column are
 * issued back to back before any voxel is processed. */
#define GIE_VOX_ZPER 8
/* optional per-column hook of the staged sweep (only op_fuse has one: tile known/unknown summaries) */
template <class F> __device__ __forceinline__ auto gie_column_hook_impl(const F &f, const gie_ctx &c, int x, int y, int z0, unsigned known, unsigned valid, int)
    -> decltype(f.column(c, x, y, z0, known, valid), void()) { f.column(c, x, y, z0, known, valid); }
template <class F> __device__ __forceinline__ void gie_column_hook_impl(const F &, const gie_ctx &, int, int, int, unsigned, unsigned, long) {}
/* ... or column_max(c, x, y, z0, committed mask, valid mask, largest value finish() returned) (op_markc) */
template <class F> __device__ __forceinline__ auto gie_column_max_hook_impl(const F &f, const gie_ctx &c, int x, int y, int z0, unsigned known, unsigned valid, int vmax, int)
    -> decltype(f.column_max(c, x, y, z0, known, valid, vmax), void()) { f.column_max(c, x, y, z0, known, valid, vmax); }
template <class F> __device__ __forceinline__ void gie_column_max_hook_impl(const F &, const gie_ctx &, int, int, int, unsigned, unsigned, int, long) {}
template <class F>
__global__ __launch_bounds__(GIE_VOX_BX *GIE_VOX_BY) void k_voxz(const gie_ctx c, const F f)
{
    const int x = blockIdx.x * GIE_VOX_BX + threadIdx.x;
    const int y = blockIdx.y * GIE_VOX_BY + threadIdx.y;
    const int z0 = blockIdx.z * GIE_VOX_ZPER;
    const bool in = (x < c.X && y < c.Y);
    if (!in || f.tile_skip(c, x, y, z0)) return;            /* one flag for the thread's whole z-column (one 8x8x8 tile) */
    bool sk[GIE_VOX_ZPER];
#pragma unroll
    for (int k = 0; k < GIE_VOX_ZPER; k++) {
        const int z = z0 + k;
        sk[k] = z >= c.Z || f.skip(c, z < c.Z ? gie_lid(c, x, y, z) : 0, x, y, z);
    }
    if (F::rolled) {            /* big bodies: keep one copy of the code (instruction cache) */
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < GIE_VOX_ZPER; k++) m |= (sk[k] ? 0u : 1u) << k;
#pragma unroll 1
        for (int k = 0; k < GIE_VOX_ZPER; k++)
            if ((m >> k) & 1u) f(c, x, y, z0 + k);
    } else {
#pragma unroll
        for (int k = 0; k < GIE_VOX_ZPER; k++)
            if (!sk[k]) f(c, x, y, z0 + k);
    }
}

template <class F>
__global__ __launch_bounds__(256) void k_lin(const gie_ctx c, const F f, const int n)
{
    if (GIE_GATE_CLOSED(c)) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) f(c, i);
}

/* gie_ogm_labels for X % 16 == 0 without a robot sphere: a thread moves 16 voxels of a row (one
 * 16-byte load, one 16-byte store) and flags the (at most three) blocks its observed voxels lie in */
template <bool STORE>
__global__ __launch_bounds__(256) void k_labels16(const gie_ctx c, const int8_t *labels, const int nvec)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const int xv = c.X >> 4;
    const int x0 = (v % xv) << 4, row = v / xv, y = row % c.Y, z = row / c.Y;
    const uint4 q = reinterpret_cast<const uint4 *>(labels)[v];
    uint32_t w[4] = { q.x, q.y, q.z, q.w };
    unsigned known = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t l = (w[j] >> (8 * b)) & 0xffu;
            const bool ok = (l == (uint32_t)GIE_VOX_FREE) | (l == (uint32_t)GIE_VOX_OCCUPIED);
            if (!ok) w[j] &= ~(0xffu << (8 * b));
            known |= (ok ? 1u : 0u) << (4 * j + b);
        }
    }
    if (STORE) reinterpret_cast<uint4 *>(c.inst_type)[v] = make_uint4(w[0], w[1], w[2], w[3]);      /* (!STORE: gie_fuse reads the caller's plane itself, c.scan_labels) */
    if (!known) return;
    const int gx0 = x0 + c.pvt[0], gy = y + c.pvt[1], gz = z + c.pvt[2];
    const int cell0 = (((gz >> 3) - c.tb0[2]) * c.tdim[1] + ((gy >> 3) - c.tb0[1])) * c.tdim[0] - c.tb0[0];
    const int b0 = gx0 >> 3, b1 = (gx0 + 15) >> 3;
    for (int bx = b0; bx <= b1; bx++) {
        const int lo = max(bx * 8 - gx0, 0), hi = min(bx * 8 + 7 - gx0, 15);     /* voxels of this thread inside block bx */
        const unsigned m = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
        if (known & m) c.blk_need[cell0 + bx] = 1;                              /* all writers store 1 */
    }
}

/* ------------------------------------------------------------------ ray casting, segmented */
/* freeLocObs with the walk of every ray cut into GIE_RAY_SEGS segments: lane = ray (64 rays
 * adjacent in the cloud per workgroup, so that the wave aggregation of the _ray_count atomics
 * still sees rays crossing the same near-sensor cells), wave = segment.  A ray walk is a chain of
 * dependent memory round trips and one thread per ray leaves the GPU almost empty (a 16 x 1800
 * cloud is 450 waves).  The DDA state at a segment's start is handed from wave to wave through
 * LDS: wave s waits for wave s-1, advances its own steps in registers (no memory), publishes the
 * state for wave s+1 and only then starts on memory — the arithmetic chain is walked once per
 * ray, the memory chains of the segments run side by side.  Phase 1 finds where the ray
 * stops (first OCCUPIED cell or the walk's end) as a minimum over the segments in LDS, phase 2
 * applies the decrements of the cells before that point — the same set of cells as the
 * sequential walk. */
#ifndef GIE_RAY_SEGS
#define GIE_RAY_SEGS 8       /* 4 / 8 / 16 measured: 0.108 / 0.080 / 0.083 ms (16 x 1800 cloud), 8 vs 16 on a 64 x 1800 cloud: 0.100 vs 0.166 ms */
#endif
#ifndef GIE_RAY_SPIN_SLEEP
#define GIE_RAY_SPIN_SLEEP 4
#endif
/* The per-step address arithmetic of the kernel: the same values as gie_in_loc / gie_lid /
 * gie_tile_index / gie_tab_index, written without short-circuit branches and with 24-bit
 * multiplies (sides <= 1024, checked by gie_create) — a step of the walk is a few dozen
 * instructions, and the compute units that hold the longest rays are instruction-bound. */
__device__ __forceinline__ int gie_ray_cell(const gie_ctx &c, const gie_dda &d, int &lx, int &ly, int &lz)
{
    lx = d.cur[0] - c.pvt[0]; ly = d.cur[1] - c.pvt[1]; lz = d.cur[2] - c.pvt[2];
    const bool in = ((unsigned)lx < (unsigned)c.X) & ((unsigned)ly < (unsigned)c.Y) & ((unsigned)lz < (unsigned)c.Z);
    return in ? __mul24(__mul24(lz, c.Y) + ly, c.X) + lx : -1;
}
__device__ __forceinline__ void gie_ray_touch_k(const gie_ctx &c, const gie_dda &d, const int lx, const int ly, const int lz, gie_ray_marks *last)
{
    const int t = __mul24(__mul24(lz >> 3, c.tfd[1]) + (ly >> 3), c.tfd[0]) + (lx >> 3);
    const int b = __mul24(__mul24((d.cur[2] >> 3) - c.tb0[2], c.tdim[1]) + ((d.cur[1] >> 3) - c.tb0[1]), c.tdim[0]) + ((d.cur[0] >> 3) - c.tb0[0]);
    if ((t != last->tile) | (b != last->blk)) { c.tray[t] = 1; c.blk_need[b] = 1; last->tile = t; last->blk = b; }   /* both stores are idempotent */
}
__device__ __forceinline__ bool gie_dda_before(const gie_dda &d, const float limit)
{
    return fminf(fminf(d.tMax[0], d.tMax[1]), d.tMax[2]) < limit;      /* the crossing the next step takes is the smallest tMax */
}

/* advance the walk's state past every border crossed before `limit`, without looking at the cells:
 * per axis the very additions the walk performs (same roundings), three independent chains */
__device__ __forceinline__ void gie_dda_skip_to(gie_dda &d, const float limit, const int max_steps)
{
    float t0 = d.tMax[0], t1 = d.tMax[1], t2 = d.tMax[2];
    int n0 = 0, n1 = 0, n2 = 0;
    for (int k0 = 0; k0 < max_steps && __any((t0 < limit) | (t1 < limit) | (t2 < limit)); k0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool a0 = t0 < limit, a1 = t1 < limit, a2 = t2 < limit;
            t0 += a0 ? d.tDelta[0] : 0.0f; n0 += a0 ? 1 : 0;
            t1 += a1 ? d.tDelta[1] : 0.0f; n1 += a1 ? 1 : 0;
            t2 += a2 ? d.tDelta[2] : 0.0f; n2 += a2 ? 1 : 0;
        }
    }
    d.cur[0] += d.step[0] * n0; d.tMax[0] = t0;
    d.cur[1] += d.step[1] * n1; d.tMax[1] = t1;
    d.cur[2] += d.step[2] * n2; d.tMax[2] = t2;
}

__global__ __launch_bounds__(64 * GIE_RAY_SEGS) void k_free_rays(const gie_ctx c, const float *g, const int n, const int max_steps)
{
    __shared__ int s_stop[64];
    __shared__ int s_cur[GIE_RAY_SEGS][3][64];
    __shared__ float s_tmax[GIE_RAY_SEGS][3][64];
    __shared__ int s_ready[GIE_RAY_SEGS];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const bool ray = i < n && gie_point_ok(g[3 * (i < n ? i : 0)], g[3 * (i < n ? i : 0) + 1], g[3 * (i < n ? i : 0) + 2]);
    if (seg == 0) s_stop[lane] = 0x7fffffff;
    if (threadIdx.x < GIE_RAY_SEGS) s_ready[threadIdx.x] = 0;
    gie_dda d;
    int s0[3] = { 0, 0, 0 };
    gie_ray_marks last_tile = { -1, -1 };
    const bool walk = ray && gie_dda_init(c, g, i, d, s0);      /* constants of the ray (direction, deltas, end cell) in every wave */
    if (!walk) {
#pragma unroll
        for (int k = 0; k < 3; k++) { d.cur[k] = 0; d.step[k] = 0; d.tMax[k] = 3.402823466e+38f; d.tDelta[k] = 0.0f; }
        d.len = 0.0f;
    }
    __syncthreads();
    /* Segment `seg` owns the border crossings whose time along the ray lies in [T(seg), T(seg+1)),
     * T(k) = k/SEGS of the ray's length (capped by the maximum length); the last one is open-ended
     * and runs into the reference's own stop test.  The walk takes crossings in the order of
     * their times, so every crossing before T(k) precedes every crossing after it, and the state
     * at T(k) is, per axis, the first border time >= T(k) — three independent chains of float
     * additions (the same additions the walk performs, so the same roundings), which a wave
     * follows for its own interval and hands to the next one: the chain that has to run ahead
     * of the memory work costs one add per crossing instead of a whole DDA step. */
    const float L = walk ? (d.len < d.max_length ? d.len : d.max_length) : 0.0f;
    /* Only the part of the ray inside the local volume has cells to look at (a straight ray never
     * comes back into a box it has left; the rays of a scan that ends outside a thin volume or —
     * for a tile of a larger volume — never touch this tile are most of the walk): the segments
     * share [Ts, Te] = the ray's stay in the volume grown by one cell, clipped to [0, L].  A
     * volume the walk is still inside at L keeps the open end. */
    float Ts = 0.0f, Te = L;
    bool open_end = true;
    if (walk) {
        const float w = c.voxel_width;
        const int dim[3] = { c.X, c.Y, c.Z };
        float tin = 0.0f, tout = 3.402823466e+38f;
        bool hit = true;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float lo = ((float)c.pvt[a] - 1.5f) * w, hi = ((float)(c.pvt[a] + dim[a]) + 0.5f) * w;
            if (d.step[a] != 0) {
                const float inv = (float)d.step[a] * d.tDelta[a] / w;       /* 1 / direction component */
                const float ta = (lo - c.origin[a]) * inv, tb = (hi - c.origin[a]) * inv;
                tin = fmaxf(tin, fminf(ta, tb)); tout = fminf(tout, fmaxf(ta, tb));
            } else if (c.origin[a] < lo || c.origin[a] > hi) hit = false;
        }
        const float slack = 2.0f * w;
        float te_box = tout + slack;
        Ts = fmaxf(0.0f, tin - slack);
        if (!hit || !(Ts < te_box)) { Ts = 0.0f; te_box = 0.0f; }             /* never inside */
        if (te_box < L) { Te = te_box; open_end = false; }
        if (Ts > Te) Ts = Te;
    }
    const float Tn = (seg + 1 < GIE_RAY_SEGS) ? Ts + (Te - Ts) * ((float)(seg + 1) * (1.0f / (float)GIE_RAY_SEGS))
                                              : (open_end ? __builtin_inff() : Te);
    if (seg == 0) {   /* clearRayLoc on the sensor's own cell */
        const int id0 = (ray && gie_in_loc(c, s0[0], s0[1], s0[2])) ? gie_lid(c, s0[0], s0[1], s0[2]) : -1;
        if (id0 >= 0) gie_ray_touch(c, s0[0], s0[1], s0[2], &last_tile);
        gie_wave_add(c, (id0 >= 0 && c.inst_type[id0] != GIE_VOX_OCCUPIED) ? id0 : -1, -1);
        gie_dda_skip_to(d, Ts, max_steps);                /* sensor outside the volume: up to where the ray enters it */
    } else {
        /* state at the end of segment seg-1 = at the start of mine */
        volatile int *rdy = &s_ready[seg - 1];
        while (*rdy == 0) __builtin_amdgcn_s_sleep(GIE_RAY_SPIN_SLEEP);   /* waiting waves must not eat the issue slots of the one that works */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int k = 0; k < 3; k++) { d.cur[k] = s_cur[seg - 1][k][lane]; d.tMax[k] = s_tmax[seg - 1][k][lane]; }
    }
    GIE_RTS(0);
    const int base = seg << 20;                           /* step index = (segment, step inside it): orders the stops of all segments */
    const gie_dda at_start = d;
    if (seg + 1 < GIE_RAY_SEGS) {                         /* where my interval ends, for the next wave: registers only */
        gie_dda nx = d;
        gie_dda_skip_to(nx, Tn, max_steps);
#pragma unroll
        for (int k = 0; k < 3; k++) { s_cur[seg][k][lane] = nx.cur[k]; s_tmax[seg][k][lane] = nx.tMax[k]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) { volatile int *w = &s_ready[seg]; *w = 1; }
    }
    GIE_RTS(1);
    /* phase 1 (starts as soon as this wave has its state; no workgroup barrier before it): types of the segment's cells → where does the ray stop? (exclusive step index) */
    if (walk) {
        int stop = 0x7fffffff;
        bool more = true;
        for (int k0 = 0; k0 < max_steps && more && stop == 0x7fffffff; k0 += GIE_RAY_BATCH) {
            if (*(volatile int *)&s_stop[lane] < base) break;        /* an earlier segment already ends this walk */
            int ids[GIE_RAY_BATCH], end[GIE_RAY_BATCH];
            bool went[GIE_RAY_BATCH];
#pragma unroll
            for (int j = 0; j < GIE_RAY_BATCH; j++) {
                went[j] = gie_dda_before(d, Tn);
                end[j] = 0; ids[j] = -1;
                if (went[j]) { int lx, ly, lz; end[j] = gie_dda_step(d); ids[j] = gie_ray_cell(c, d, lx, ly, lz); }
            }
            int8_t ty[GIE_RAY_BATCH];
#pragma unroll
            for (int j = 0; j < GIE_RAY_BATCH; j++) ty[j] = ids[j] >= 0 ? c.inst_type[ids[j]] : (int8_t)GIE_VOX_UNKNOWN;
#pragma unroll
            for (int j = 0; j < GIE_RAY_BATCH; j++) {
                if (stop != 0x7fffffff || !went[j]) continue;
                if (ty[j] == GIE_VOX_OCCUPIED) stop = base + k0 + j;           /* this cell is not cleared */
                else if (end[j]) stop = base + k0 + j + 1;                      /* this cell is the last one cleared */
            }
            more = went[GIE_RAY_BATCH - 1];
        }
        if (stop != 0x7fffffff) atomicMin(&s_stop[lane], stop);
    }
    GIE_RTS(2);
    __syncthreads();
    GIE_RTS(3);
    /* phase 2: clear the segment's cells that lie before the stop (all lanes take part in the
     * wave aggregation; lanes without work pass -1) */
    const int stop = s_stop[lane];
    d = at_start;
    for (int k = 0; k < max_steps; k++) {
        int id = -1;
        const bool went = walk && base + k < stop && gie_dda_before(d, Tn);
        if (went) {
            int lx, ly, lz;
            gie_dda_step(d);
            id = gie_ray_cell(c, d, lx, ly, lz);
            if (id >= 0) gie_ray_touch_k(c, d, lx, ly, lz, &last_tile);
        }
        if (__ballot(went) == 0ull) break;                /* a lane that did not step now never steps again */
        if (__ballot(id >= 0) == 0ull) continue;
        gie_wave_add(c, id, -1);
    }
    GIE_RTS(4);
}

/* ------------------------------------------------------------------ frame clear */
/* every array that has to be zero when a map update starts, in one launch: workgroup w of the clear part zeroes its share of the
 * region it falls into (gie_clear_wgs workgroups per region, by size) */
__device__ __forceinline__ void gie_clear_part(const gie_clear_list &l, int w)
{
    int r = 0, nw = gie_clear_wgs(l.bytes[0]);
    while (w >= nw && r + 1 < l.n) { w -= nw; r++; nw = gie_clear_wgs(l.bytes[r]); }      /* uniform over the workgroup; at most GIE_CLEAR_MAX trips */
    if (w >= nw) return;
    unsigned char *p = (unsigned char *)l.p[r];
    const uint32_t bytes = l.bytes[r];
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u) < bytes ? (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u) : bytes;
    const uint32_t nvec = (bytes - head) / 16u, tail0 = head + nvec * 16u;
    const uint32_t tid = (uint32_t)w * 256u + threadIdx.x, nth = (uint32_t)nw * 256u;
    for (uint32_t i = tid; i < head; i += nth) p[i] = 0;
    uint4 *v = (uint4 *)(p + head);
    for (uint32_t i = tid; i < nvec; i += nth) v[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = tail0 + tid; i < bytes; i += nth) p[i] = 0;
}
__global__ __launch_bounds__(256) void k_clear(const gie_clear_list l, const int32_t *gate)
{
    if (gate != nullptr && *(const volatile int32_t *)gate == 0) return;
    gie_clear_part(l, (int)blockIdx.x);
}

/* the frame clear and the flush of the stored pairs a fused update left out (gie_pair_flush_voxel) in one launch: the first
 * `nclr` workgroups clear, the others flush — they touch different arrays */
__global__ __launch_bounds__(256) void k_flush_clear(const gie_ctx c, const op_pair_flush f, const int n, const gie_clear_list l, const int nclr)
{
    if ((int)blockIdx.x >= nclr) { const int i = ((int)blockIdx.x - nclr) * 256 + (int)threadIdx.x; if (i < n) f(c, i); return; }
    gie_clear_part(l, (int)blockIdx.x);
}

/* ------------------------------------------------------------------ block allocation */
/* allocHashTB (glb_hash_map.cu:58-113) + the frame's block table in one sweep over the table
 * cells: look the block up; a cell the scan observed without a block gets a slot (one atomic per
 * wave on the pool counter), is inserted and queued for initialisation; either way the table
 * entry is final when the thread ends (nobody else handles this key in this launch, and inserts
 * of other keys cannot break a probe sequence). */
__global__ __launch_bounds__(256) void k_cell_alloc(const gie_ctx c, const int ncell)
{
    if (GIE_GATE_CLOSED(c)) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int found = -1;
    bool isnew = false;
    if (i < ncell) {
        const int bx = i % c.tdim[0], by = (i / c.tdim[0]) % c.tdim[1], bz = i / (c.tdim[0] * c.tdim[1]);
        found = gie_cell_prev_slot(c, bx, by, bz);   /* one coalesced read instead of a chain of hash probes for every block the table before knew */
        if (found < 0) found = gie_hash_find(c, bx + c.tb0[0], by + c.tb0[1], bz + c.tb0[2]);
        isnew = found < 0 && c.blk_need[i];
        c.blk_need[i] = 0;
    }
    /* Slots for the new blocks: ONE set of counter updates per workgroup (ballot inside the wavefronts, prefix across them in LDS).
     * On a straight drive the new blocks are one layer of the table — a cell in every row, i.e. one or two per wavefront — and
     * with a set of updates per wavefront the launch was 4 096 x 3 same-address atomics at ~10 ns each (0.08 ms). */
    __shared__ int s_cnt[4], s_base[4];
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(isnew);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    const int c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
    const int cnt = c0 + c1 + c2 + c3;
    if (cnt > 0) {                                        /* workgroup-uniform */
        if (threadIdx.x == 0) {
            /* slots of erased blocks first (only pops happen in this launch; a count driven below zero is reset by
             * k_block_init_list), the rest from the bump allocator */
            int ftop = 0, nf = 0, base = 0;
            if (c.retain > 0) { ftop = atomicSub(&c.pool_count[1], cnt); nf = ftop < 0 ? 0 : (ftop < cnt ? ftop : cnt); }
            if (cnt > nf) base = atomicAdd(&c.pool_count[0], cnt - nf);
            const int lbase = atomicAdd(&c.cnt[GIE_CNT_NEWLIST], cnt);
            atomicAdd(&c.cnt[GIE_CNT_NEWBLK], cnt);
            s_base[0] = base; s_base[1] = lbase; s_base[2] = nf; s_base[3] = ftop;
        }
        __syncthreads();
        if (isnew) {
            const int base = s_base[0], lbase = s_base[1], nf = s_base[2], ftop = s_base[3];
            const int before = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
            const int r = before + __popcll(m & ((1ull << lane) - 1ull));
            const int slot = gie_alloc_slot(c, r, nf, ftop, base);
            if (slot >= c.max_blocks) { gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_POOL); c.blk_new[lbase + r] = -1; }
            else { gie_cell_insert(c, i, slot); c.blk_new[lbase + r] = slot; found = slot; }
        }
    }
    if (i < ncell) {
        c.blk_tab[i] = found;
        if (found >= 0) gie_cell_mark_tiles(c, i);
    }
}
/* initialise the 512 voxels of every block on the list (one workgroup per block, grid-stride).  The workgroups behind the
 * first `ninit` build the fuse tile list (op_fuse_list over the `ntile` tiles of the volume; 0: none): the two only read what
 * k_cell_alloc has left and touch different data, so they share a launch. */
__global__ __launch_bounds__(256) void k_block_init_list(const gie_ctx c, const int ninit, const int ntile)
{
    if (GIE_GATE_CLOSED(c)) return;
    if ((int)blockIdx.x >= ninit) {
        const op_fuse_list f;
        const int nfl = (int)gridDim.x - ninit;
        for (int i0 = ((int)blockIdx.x - ninit) * 256; i0 < ntile; i0 += nfl * 256) {      /* whole workgroups call (block barriers inside) */
            const int i = i0 + (int)threadIdx.x;
            f(c, i < ntile ? i : -1);
        }
        return;
    }
    const int n = c.cnt[GIE_CNT_NEWLIST];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (c.pool_count[0] > c.max_blocks) c.pool_count[0] = c.max_blocks;
        if (c.pool_count[1] < 0) c.pool_count[1] = 0;
    }
    for (int e = blockIdx.x; e < n; e += ninit) {
        const int slot = c.blk_new[e];
        if (slot < 0) continue;
        gie_init_voxel(c, slot, threadIdx.x);
        gie_init_voxel(c, slot, threadIdx.x + 256);
        if (threadIdx.x < 6) gie_nbr_link(c, slot, (int)threadIdx.x);
    }
}

/* ------------------------------------------------------------------ EDT pass Y */
/* Nearest occupied voxel along y for every (x,z) column; ties go to the larger y
 * (EDTphase1's backward sweep overwrites on '<', local_edt_core.h:65-81).
 * The column's occupancy is a bit mask (YW 32-bit words).  Four threads share a column: each
 * reads a quarter of the types and publishes its mask words through LDS, then every thread
 * holds the whole mask in registers and writes the answers of its own quarter — 4x the waves
 * in flight of a thread-per-column scan, same HBM traffic (1 B read + 2 B written per voxel). */
#ifndef GIE_EDTY_COLS
#define GIE_EDTY_COLS 64
#endif
template <int YW, int TPC>
__global__ __launch_bounds__(GIE_EDTY_COLS * TPC) void k_edt_y(const gie_ctx c)
{
    constexpr int QW = YW / TPC;                         /* mask words per thread (TPC threads share a column) */
    __shared__ uint32_t s_bits[YW][GIE_EDTY_COLS];
    const int col = threadIdx.x, q = threadIdx.y;
    const int x = blockIdx.x * GIE_EDTY_COLS + col;
    const int z = blockIdx.y;
    if (!c.zocc[z]) return;                              /* plane without obstacle: passes X/Z never read its cy1 */
    const int X = c.X, Y = c.Y;
    const bool in = x < X;
    const int8_t *t = c.glb_type + (size_t)z * X * Y + (in ? x : 0);
#pragma unroll
    for (int j = 0; j < QW; j++) {
        const int w = q * QW + j;
        uint32_t m = 0;
        if (in && w * 32 < Y) {
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int y = w * 32 + k;
                if (y < Y) m |= (uint32_t)(t[(size_t)y * X] == GIE_VOX_OCCUPIED) << k;
            }
        }
        s_bits[w][col] = m;
    }
    __syncthreads();
    if (!in) return;
    uint32_t bits[YW];
#pragma unroll
    for (int w = 0; w < YW; w++) bits[w] = s_bits[w][col];
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
        if (w / QW != q) continue;                       /* only this thread's quarter (uniform per wave) */
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

/* The same pass for X % 4 == 0 with FOUR adjacent columns per lane: a lane reads the four type
 * bytes of (x..x+3, y, z) as one dword and stores its four answers as one 8-byte vector, so a
 * wave moves 128-byte row segments per instruction instead of 64 single bytes (the byte form is
 * bound by the number of memory instructions, not by bytes).  Thread (l, q) owns the YB
 * positions y = q·YB .. of its four columns; what lies below / above its own mask word comes
 * from the other threads' words in LDS (one 16-byte read per word covers the four columns). */
template <int YB, int LANES>
__global__ __launch_bounds__(32 * LANES) void k_edt_y4(const gie_ctx c)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_m[32][LANES * 4];
    const int l = threadIdx.x, q = threadIdx.y, nq = blockDim.y;
    const int x = (blockIdx.x * LANES + l) * 4;
    const int z = blockIdx.y;
    if (!c.zocc[z]) return;                              /* plane without obstacle: passes X/Z never read its cy1 */
    const int X = c.X, Y = c.Y;
    const bool in = x < X;
    const size_t base = (size_t)z * X * Y + (in ? x : 0);
    uint32_t m[4] = {0u, 0u, 0u, 0u};
    if (in) {
        const int8_t *t = c.glb_type + base;
        uint32_t v[YB];
#pragma unroll
        for (int k = 0; k < YB; k++) {
            const int y = q * YB + k;                     /* rows past the end re-read the last one and are masked out */
            v[k] = *reinterpret_cast<const uint32_t *>(t + (size_t)min(y, Y - 1) * X);
        }
        const uint32_t ymask = (q * YB + YB <= Y) ? 0xffffffffu : ((q * YB < Y) ? (0xffffffffu >> (32 - (Y - q * YB))) : 0u);
#pragma unroll
        for (int k = 0; k < YB; k++) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) m[cc] |= (uint32_t)(((v[k] >> (8 * cc)) & 0xffu) == (uint32_t)GIE_VOX_OCCUPIED) << k;
        }
#pragma unroll
        for (int cc = 0; cc < 4; cc++) m[cc] &= ymask;
    }
    *reinterpret_cast<uint4 *>(&s_m[q][4 * l]) = make_uint4(m[0], m[1], m[2], m[3]);
    __syncthreads();
    if (!in) return;
    int pl[4] = {-1, -1, -1, -1}, nf[4] = {-1, -1, -1, -1};
    for (int w = 0; w < q; w++) {                         /* last obstacle below this thread's positions */
        const uint4 o = *reinterpret_cast<const uint4 *>(&s_m[w][4 * l]);
        const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int cc = 0; cc < 4; cc++) if (ow[cc]) pl[cc] = w * YB + 31 - __clz(ow[cc]);
    }
    for (int w = nq - 1; w > q; w--) {                    /* first obstacle above them */
        const uint4 o = *reinterpret_cast<const uint4 *>(&s_m[w][4 * l]);
        const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int cc = 0; cc < 4; cc++) if (ow[cc]) nf[cc] = w * YB + __ffs(ow[cc]) - 1;
    }
    uint16_t *out = c.cy1 + base;
#pragma unroll
    for (int k = 0; k < YB; k++) {
        const int y = q * YB + k;
        if (y >= Y) break;
        uint32_t r4[4];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const uint32_t bw = m[cc];
            const uint32_t lo = bw & (0xffffffffu >> (31 - k));
            const int below = lo ? q * YB + 31 - __clz(lo) : pl[cc];
            const uint32_t hi = bw >> k;
            const int above = hi ? y + __ffs(hi) - 1 : nf[cc];
            int r;
            if (above >= 0 && (below < 0 || above - y <= y - below)) r = above; else r = below;
            r4[cc] = r < 0 ? 0xffffu : (uint32_t)r;
        }
        *reinterpret_cast<uint2 *>(out + (size_t)y * X) = make_uint2(r4[0] | (r4[1] << 16), r4[2] | (r4[3] << 16));
    }
}

/* ------------------------------------------------------------------ lower-envelope argmin */
/* One wave computes, for every position u of a row of L sites (L <= 64*CP), the site
 *     argmin_i (u-i)² + a_i     with ties going to the smaller i
 * which is exactly what the reference's Meijster scan with truncating Sep() returns
 * (tests/test_oracle_edt.py::test_meijster_tie_rule pins the equivalence).
 *
 * Sites without a value (no obstacle behind them) can never win while a real one exists, so
 * the row is first COMPACTED (wave64 ballot + prefix popcount) to its K real sites:
 *     ce[j] = { (a << 10) | j ,  32 * i }            j = rank of site i among the real ones
 * 32-bit keys  ((u-i)² + a) << 10 | j  order by value, then by rank (= by i); (max a) + L² must
 * stay below 2^22 (checked in gie_create).  The argmin rank is monotone in u, so after a
 * brute-force pass over the K sites for every CP-th position the rest is found by divide &
 * conquer inside [rank(left), rank(right)].  jsite[u] receives the winning RANK. */
__device__ __forceinline__ void gie_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* min key of position `up` (= u << 5) over the sites lo..hi.  The loop is bound by the LDS round
 * trip, not by ALU work, so four independent reads are issued per trip; indices past `hi` are
 * clamped to it (a repeated candidate does not change a minimum). */
__device__ __forceinline__ uint32_t gie_scan_sites(const uint2 *ce, const int up, const int lo, const int hi)
{
    uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu, b3 = 0xffffffffu;
    for (int j = lo; j <= hi; j += 4) {
        const uint2 v0 = ce[j], v1 = ce[min(j + 1, hi)], v2 = ce[min(j + 2, hi)], v3 = ce[min(j + 3, hi)];
        const int d0 = up - (int)(v0.y & 0xffffu), d1 = up - (int)(v1.y & 0xffffu);
        const int d2 = up - (int)(v2.y & 0xffffu), d3 = up - (int)(v3.y & 0xffffu);
        b0 = min(b0, (uint32_t)__mul24(d0, d0) + v0.x);
        b1 = min(b1, (uint32_t)__mul24(d1, d1) + v1.x);
        b2 = min(b2, (uint32_t)__mul24(d2, d2) + v2.x);
        b3 = min(b3, (uint32_t)__mul24(d3, d3) + v3.x);
    }
    return min(min(b0, b1), min(b2, b3));
}

template <int CP>
__device__ __forceinline__ void gie_row_argmin(const uint2 *ce, const int K, const int L, const int lane, int (&sj)[CP])
{
    /* lane owns positions u0 .. u0+CP-1; sj[m] receives the winning RANK of position u0+m.
     * Everything lives in registers: the only LDS traffic is reading candidate sites. */
    const int u0 = lane * CP;
    int s0;
    {
        /* first position of every lane, in two steps (the argmin rank is monotone in u):
         * (1) the eight positions owned by lanes 0, 8, .. 56: the eight lanes of a group split
         *     the K sites between them (site j goes to lane j mod 8) and min-reduce;
         * (2) every other lane searches only between its group's result and the next group's. */
        const int g8 = lane & ~7, r = lane & 7;
        const int ug = g8 * CP;
        uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu, b3 = 0xffffffffu;
        {
            const int up = ug << 5, kl = K - 1;
            for (int j = r; j < K; j += 32) {                  /* clamped repeats are harmless for a minimum */
                const uint2 v0 = ce[j], v1 = ce[min(j + 8, kl)], v2 = ce[min(j + 16, kl)], v3 = ce[min(j + 24, kl)];
                const int d0 = up - (int)(v0.y & 0xffffu), d1 = up - (int)(v1.y & 0xffffu);
                const int d2 = up - (int)(v2.y & 0xffffu), d3 = up - (int)(v3.y & 0xffffu);
                b0 = min(b0, (uint32_t)__mul24(d0, d0) + v0.x);
                b1 = min(b1, (uint32_t)__mul24(d1, d1) + v1.x);
                b2 = min(b2, (uint32_t)__mul24(d2, d2) + v2.x);
                b3 = min(b3, (uint32_t)__mul24(d3, d3) + v3.x);
            }
        }
        uint32_t bg = min(min(b0, b1), min(b2, b3));
        bg = min(bg, (uint32_t)__shfl_xor((int)bg, 1));
        bg = min(bg, (uint32_t)__shfl_xor((int)bg, 2));
        bg = min(bg, (uint32_t)__shfl_xor((int)bg, 4));
        const int sg = (ug < L) ? (int)(bg & 1023u) : K - 1;
        int sn = __shfl_down(sg, 8);
        if (lane >= 56) sn = K - 1;
        s0 = sg;
        if (r != 0 && u0 < L && sn > sg) s0 = (int)(gie_scan_sites(ce, u0 << 5, sg, sn) & 1023u);
        if (u0 >= L) s0 = K - 1;
    }
    /* upper bound of the chunk = the next lane's first position (monotone argmin) */
    int sup = __shfl_down(s0, 1);
    if (lane == 63) sup = K - 1;
    int sx[CP + 1];
    sx[0] = s0; sx[CP] = sup;
#pragma unroll
    for (int step = CP / 2; step >= 1; step >>= 1) {
#pragma unroll
        for (int m = step; m < CP; m += 2 * step) {
            const int u = u0 + m;
            const int lo = sx[m - step], hi = sx[m + step];
            int r = lo;
            if (u < L && hi > lo) r = (int)(gie_scan_sites(ce, u << 5, lo, hi) & 1023u);
            sx[m] = (u < L) ? r : K - 1;
        }
    }
#pragma unroll
    for (int m = 0; m < CP; m++) sj[m] = sx[m];
}

/* The same argmin for rows with few real sites, organised in BANDS: lane owns the positions
 * lane, 64 + lane, 128 + lane, ...; sj[m] = winning rank of position 64 m + lane.  The argmin at
 * the band starts 0, 64, 128, ... is found first (the 64/CP lanes of a group split the K sites);
 * the 64 positions of band m can then only be won by the sites sa[m] .. sa[m+1], a range that is
 * the same for the whole wave: a uniform loop with broadcast reads and no divergence.  Every
 * lane evaluates K + CP candidates in total (the divide & conquer form above evaluates fewer
 * when K is large, but pays per-lane loop set-up for every position). */
#define GIE_BAND_MAXK 160
template <int CP>
__device__ __forceinline__ void gie_row_argmin_banded(const uint2 *ce, const int K, const int L, const int lane, int (&sj)[CP], const unsigned bandneed = ~0u)
{
    constexpr int G = 64 / CP;                            /* lanes per band start */
    const int b = lane / G, r = lane % G;
    uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu;
    {
        const int up = (64 * b) << 5, kl = K - 1;
        for (int j = r; j < K; j += 2 * G) {
            const uint2 v0 = ce[j], v1 = ce[min(j + G, kl)];
            const int d0 = up - (int)(v0.y & 0xffffu), d1 = up - (int)(v1.y & 0xffffu);
            b0 = min(b0, (uint32_t)__mul24(d0, d0) + v0.x);
            b1 = min(b1, (uint32_t)__mul24(d1, d1) + v1.x);
        }
    }
    uint32_t bg = min(b0, b1);
#pragma unroll
    for (int w = 1; w < G; w <<= 1) bg = min(bg, (uint32_t)__shfl_xor((int)bg, w));
    const int sg = (64 * b < L) ? (int)(bg & 1023u) : K - 1;
    int sa[CP + 1];
#pragma unroll
    for (int m = 0; m < CP; m++) sa[m] = __builtin_amdgcn_readlane(sg, m * G);
    sa[CP] = K - 1;
#pragma unroll
    for (int m = 0; m < CP; m++) {
        if (!((bandneed >> m) & 1u)) { sj[m] = 0; continue; }      /* nobody reads this band (wave-uniform) */
        const int lo = sa[m], hi = sa[m + 1];             /* wave-uniform */
        const int up = (64 * m + lane) << 5;
        uint32_t c0 = 0xffffffffu, c1 = 0xffffffffu, c2 = 0xffffffffu, c3 = 0xffffffffu;
        for (int j = lo; j <= hi; j += 4) {
            const uint2 v0 = ce[j], v1 = ce[min(j + 1, hi)], v2 = ce[min(j + 2, hi)], v3 = ce[min(j + 3, hi)];
            const int d0 = up - (int)(v0.y & 0xffffu), d1 = up - (int)(v1.y & 0xffffu);
            const int d2 = up - (int)(v2.y & 0xffffu), d3 = up - (int)(v3.y & 0xffffu);
            c0 = min(c0, (uint32_t)__mul24(d0, d0) + v0.x);
            c1 = min(c1, (uint32_t)__mul24(d1, d1) + v1.x);
            c2 = min(c2, (uint32_t)__mul24(d2, d2) + v2.x);
            c3 = min(c3, (uint32_t)__mul24(d3, d3) + v3.x);
        }
        sj[m] = (int)(min(min(c0, c1), min(c2, c3)) & 1023u);
    }
}

/* The banded argmin with LINEAR keys.  For one position u the order of the candidates does not
 * change when u² << 10 is taken off every key:
 *     ((u-i)² + a) << 10 | j   -   (u² << 10)   =   u · M_j + B_j ,     M_j = -(2 i) << 10 ,   B_j = (i² + a) << 10 | j
 * so a candidate costs one 24-bit multiply-add and one SIGNED minimum; the low ten bits are still
 * the rank.  Needs (u-i)² + a < 2^21 and i² + a < 2^21 (pass X: always, sides are <= 1024).
 * mb[j] = { M_j, B_j }, j < K, followed by neutral entries { 0, INT_MAX } (gie_row_mb_pad);
 * the array is 16-byte aligned and GIE_BAND_MAXK + GIE_BAND_PAD entries long. */
#ifndef GIE_BAND_TRIP
#define GIE_BAND_TRIP 8                                   /* sites per trip of a band loop (4, 8 or 16), = the alignment of its start: the loops are bound by the LDS round trip of a trip, not by the candidates */
#endif
#define GIE_BAND_PAD 16                                   /* neutral entries behind the K sites (>= GIE_BAND_TRIP - 1) */
__device__ __forceinline__ void gie_row_mb_pad(int2 *mb, const int K, const int lane)
{
    if (lane < GIE_BAND_PAD - 1) mb[K + lane] = make_int2(0, 0x7fffffff);
}
template <int CP>
__device__ __forceinline__ void gie_row_argmin_banded_lin(const int2 *mb, const int K, const int L, const int lane, int (&sj)[CP])
{
    constexpr int G = 64 / CP;                            /* lanes per band start */
    const int b = lane / G, r = lane % G;
    int b0 = 0x7fffffff, b1 = 0x7fffffff, b2 = 0x7fffffff, b3 = 0x7fffffff;
    {
        const int u = 64 * b, kl = K - 1;
        for (int j = r; j < K; j += 4 * G) {
            const int2 v0 = mb[j], v1 = mb[min(j + G, kl)], v2 = mb[min(j + 2 * G, kl)], v3 = mb[min(j + 3 * G, kl)];
            b0 = min(b0, __mul24(u, v0.x) + v0.y);
            b1 = min(b1, __mul24(u, v1.x) + v1.y);
            b2 = min(b2, __mul24(u, v2.x) + v2.y);
            b3 = min(b3, __mul24(u, v3.x) + v3.y);
        }
    }
    int bg = min(min(b0, b1), min(b2, b3));
#pragma unroll
    for (int w = 1; w < G; w <<= 1) bg = min(bg, __shfl_xor(bg, w));
    const int sg = (64 * b < L) ? (bg & 1023) : K - 1;
    int sa[CP + 1];
#pragma unroll
    for (int m = 0; m < CP; m++) sa[m] = __builtin_amdgcn_readlane(sg, m * G);
    sa[CP] = K - 1;
#pragma unroll
    for (int m = 0; m < CP; m++) {
        /* the winner over ALL sites lies in sa[m] .. sa[m+1] and keys are distinct (they carry the
         * rank), so a candidate outside that range can never win: the trips are aligned groups of four
         * sites read as two 16-byte vectors, with no clamping (mb[K .. K+2] hold neutral entries) */
        const int lo = sa[m] & ~(GIE_BAND_TRIP - 1), hi = sa[m + 1];   /* wave-uniform */
        const int u = 64 * m + lane;
        int c0 = 0x7fffffff, c1 = 0x7fffffff, c2 = 0x7fffffff, c3 = 0x7fffffff;
        for (int j = lo; j <= hi; j += GIE_BAND_TRIP) {
            const int4 p0 = *reinterpret_cast<const int4 *>(mb + j), p1 = *reinterpret_cast<const int4 *>(mb + j + 2);
#if GIE_BAND_TRIP >= 8
            const int4 p2 = *reinterpret_cast<const int4 *>(mb + j + 4), p3 = *reinterpret_cast<const int4 *>(mb + j + 6);
#endif
#if GIE_BAND_TRIP == 16
            const int4 p4 = *reinterpret_cast<const int4 *>(mb + j + 8), p5 = *reinterpret_cast<const int4 *>(mb + j + 10);
            const int4 p6 = *reinterpret_cast<const int4 *>(mb + j + 12), p7 = *reinterpret_cast<const int4 *>(mb + j + 14);
#endif
            c0 = min(c0, __mul24(u, p0.x) + p0.y);
            c1 = min(c1, __mul24(u, p0.z) + p0.w);
            c2 = min(c2, __mul24(u, p1.x) + p1.y);
            c3 = min(c3, __mul24(u, p1.z) + p1.w);
#if GIE_BAND_TRIP >= 8
            c0 = min(c0, __mul24(u, p2.x) + p2.y);
            c1 = min(c1, __mul24(u, p2.z) + p2.w);
            c2 = min(c2, __mul24(u, p3.x) + p3.y);
            c3 = min(c3, __mul24(u, p3.z) + p3.w);
#endif
#if GIE_BAND_TRIP == 16
            c0 = min(c0, __mul24(u, p4.x) + p4.y);
            c1 = min(c1, __mul24(u, p4.z) + p4.w);
            c2 = min(c2, __mul24(u, p5.x) + p5.y);
            c3 = min(c3, __mul24(u, p5.z) + p5.w);
            c0 = min(c0, __mul24(u, p6.x) + p6.y);
            c1 = min(c1, __mul24(u, p6.z) + p6.w);
            c2 = min(c2, __mul24(u, p7.x) + p7.y);
            c3 = min(c3, __mul24(u, p7.z) + p7.w);
#endif
        }
        sj[m] = min(min(c0, c1), min(c2, c3)) & 1023;
    }
}

/* DENSE rows (most sites real — e.g. pass X behind a pass Y in a volume with obstacles in every
 * column): the argmin of position u cannot lie farther from u than the square root of ANY value
 * already seen for u, because (u-i)² alone would exceed it.  So every position scans a window
 * around itself that grows until its radius passes sqrt(best value) for every position of the
 * wave: with obstacles a few voxels apart that is a dozen candidates on either side instead of a
 * divide & conquer over hundreds of sites (2200 → 600 VALU instructions per 512-long row).
 *   sk[i] = (a_i << 10) | i  for real sites, GIE_WIN_NONE otherwise, valid for i in [-GIE_WIN_MAX, 64 CP + GIE_WIN_MAX);
 *   lane owns positions lane, 64 + lane, …; best[m] = min key of position 64 m + lane: its low ten
 *   bits are the winning site, ties going to the smaller site like the envelope forms.
 * All reads are consecutive words across the wave (no bank conflicts), the loop is wave-uniform.
 * Returns false when some position still has no bound inside GIE_WIN_MAX (a sparse stretch of the
 * row): the caller falls back to the envelope forms. */
#define GIE_WIN_MAX 48
#define GIE_WIN_NONE 0x7fffffffu                          /* (2^21 - 1) << 10 | 1023: adding (48² << 10) cannot wrap */
template <int CP>
__device__ __forceinline__ bool gie_row_window(const uint32_t *sk, const int L, const int lane, uint32_t (&best)[CP], const unsigned bandneed = ~0u)
{
    /* (round 5: a form without the per-band masks, tests and branches — scalar instructions, about as many as the vector ones they
     * steer — with all bands running until the last one has finished was measured on the 512^3 C5 volume: pass X 0.274 -> 0.299 ms.
     * The trips the early bands are spared outweigh the scalar work.) */
    /* (a position beyond the row starts at 0 and stays there: the finish test below is one comparison per band, without a mask) */
#pragma unroll
    for (int m = 0; m < CP; m++) best[m] = (64 * m + lane < L) ? sk[64 * m + lane] : 0u;
    unsigned active = 0;                                  /* bands (64 positions each) that still have a position without its bound */
#pragma unroll
    for (int m = 0; m < CP; m++) if (64 * m < L && ((bandneed >> m) & 1u)) active |= 1u << m;
    int W = 0;
#pragma unroll 1
    for (;;) {                                            /* one trip = four more candidates on either side (kept rolled: registers) */
        /* a band is finished when every candidate that could still win or tie one of its positions has been seen:
         * (W + 1)^2 above the position's current value.  Bands finish on their own (the sparse stretch of a row
         * keeps only its own band going). */
        const uint32_t w1 = (uint32_t)((W + 1) * (W + 1));
#pragma unroll
        for (int m = 0; m < CP; m++) {
            if (!((active >> m) & 1u)) continue;          /* wave-uniform */
            if (!__any(best[m] >= (w1 << 10))) active &= ~(1u << m);      /* w1 <= value of the key: the low ten bits are the site */
        }
        if (!active) return true;
        if (W >= GIE_WIN_MAX) return false;
        const uint32_t *lo = sk + (lane - W - 4), *hi = sk + (lane + W + 1);   /* u - (W+4) .. u - (W+1)  and  u + (W+1) .. u + (W+4) */
        const uint32_t a1 = w1 << 10, a2 = (uint32_t)((W + 2) * (W + 2)) << 10;
        const uint32_t a3 = (uint32_t)((W + 3) * (W + 3)) << 10, a4 = (uint32_t)((W + 4) * (W + 4)) << 10;
#pragma unroll
        for (int m = 0; m < CP; m++) {
            if (!((active >> m) & 1u)) continue;          /* wave-uniform */
            const uint32_t l4 = lo[64 * m], l3 = lo[64 * m + 1], l2 = lo[64 * m + 2], l1 = lo[64 * m + 3];
            const uint32_t h1 = hi[64 * m], h2 = hi[64 * m + 1], h3 = hi[64 * m + 2], h4 = hi[64 * m + 3];
            uint32_t b = best[m];                         /* min(l + a, h + a) = min(l, h) + a: one add per PAIR of candidates */
            b = min(b, min(l1, h1) + a1);
            b = min(b, min(l2, h2) + a2);
            b = min(b, min(l3, h3) + a3);
            b = min(b, min(l4, h4) + a4);
            best[m] = b;
        }
        W += 4;
    }
}

/* wave64 stream compaction of the row's real sites; returns K.  `a` = value or ~0u (none),
 * `hi16` is carried in the upper half of ce[].y (pass X keeps the site's closest y there). */
__device__ __forceinline__ int gie_row_compact_push(uint2 *ce, int base, const bool valid, const uint32_t a, const int i, const uint32_t hi16, const int lane, int2 *mb = nullptr)
{
    const unsigned long long m = __ballot(valid);
    if (valid) {
        const int j = base + __popcll(m & ((1ull << lane) - 1ull));
        ce[j] = make_uint2((a << 10) | (uint32_t)j, ((uint32_t)i << 5) | (hi16 << 16));
        if (mb && j < GIE_BAND_MAXK) mb[j] = make_int2(-(i << 11), (int)((((uint32_t)(i * i) + a) << 10) | (uint32_t)j));   /* linear keys of the banded form */
    }
    return base + __popcll(m);
}

/* ------------------------------------------------------------------ EDT pass X */
/* one wave per (y,z) row; rows are contiguous in memory so loads/stores are coalesced; every
 * lane ends up with the CP consecutive results of its chunk in registers and stores them as
 * 16-byte vectors (the wave covers one contiguous run) */
#ifndef GIE_EDTX_WAVES
#define GIE_EDTX_WAVES 4
#endif
template <int CP>
__global__ __launch_bounds__(64 * GIE_EDTX_WAVES) void k_edt_x(const gie_ctx c)
{
    constexpr int LP = 64 * CP;
    __shared__ __attribute__((aligned(16))) uint2 s_ce[GIE_EDTX_WAVES][LP];
    /* the linear site records of the banded form (K <= GIE_BAND_MAXK) live in the upper half of the
     * wave's site list when that is long enough — a longer row overwrites them with its own sites
     * and does not use them — so that eight workgroups fit a compute unit's LDS */
    constexpr bool OVL = LP >= 2 * (GIE_BAND_MAXK + GIE_BAND_PAD) && LP / 2 >= GIE_BAND_MAXK;
    __shared__ __attribute__((aligned(16))) int2 s_mb[OVL ? 1 : GIE_EDTX_WAVES][OVL ? 1 : GIE_BAND_MAXK + GIE_BAND_PAD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    /* grid = (rows of a plane / waves, planes): no division by a run-time size — most waves
     * belong to a plane without obstacle and leave after one flag read, and an integer division
     * was a quarter of all the instructions of the launch */
    const int z = blockIdx.y, y = blockIdx.x * GIE_EDTX_WAVES + wave;
    if (!c.zocc[z]) return;                                     /* empty plane: pass Z does not read its cxy2 */
    if (y >= c.Y) return;
    const int X = c.X;
    const size_t row = (size_t)z * c.Y + y;                     /* row = z*Y + y */
    const uint16_t *in = c.cy1 + row * X;
    uint2 *ce = s_ce[wave];
    int2 *mb = OVL ? reinterpret_cast<int2 *>(ce + LP / 2) : s_mb[OVL ? 0 : wave];
    int K = 0;
    uint16_t cyv[CP];                                           /* the whole row in flight at once: one memory round trip per row, not CP */
#pragma unroll
    for (int m = 0; m < CP; m++) { const int i = 64 * m + lane; cyv[m] = __builtin_nontemporal_load(&in[min(i, X - 1)]); }
    uint32_t *out = c.cxy2 + row * X;
    if (CP >= 4) {
        /* most sites real (a volume with obstacles in almost every column): windowed scan over the
         * direct-indexed row, kept in this wave's site-list memory */
        int kall = 0;
#pragma unroll
        for (int m = 0; m < CP; m++) if (64 * m < X) kall += __popcll(__ballot(64 * m + lane < X && cyv[m] != 0xffff));
        /* (round 6: ... and only when at least 7 of 8 columns hold a site — a lidar scene's wall planes have sites in a third of their columns,
         * the window gave up after twelve trips per band and the row was done again by the envelope forms: pass X 0.144 -> 0.104 ms on
         * BASELINE config 3, 0.065 -> 0.053 on config 4; the headline's planes hold a site in 99 % of their columns) */
        if (kall > GIE_BAND_MAXK && kall * 8 >= X * 7) {
            static_assert(CP < 4 || (LP + 2 * GIE_WIN_MAX) * 4 + LP * 2 <= LP * 8, "the windowed row must fit the wave's site list");
            uint32_t *sk = reinterpret_cast<uint32_t *>(ce) + GIE_WIN_MAX;
            uint16_t *scy = reinterpret_cast<uint16_t *>(ce) + 2 * (LP + 2 * GIE_WIN_MAX);
#pragma unroll
            for (int m = 0; m < CP; m++) {
                const int i = 64 * m + lane;
                const uint16_t cy = (i < X) ? cyv[m] : (uint16_t)0xffff;
                const int d = y - (int)cy;
                sk[i] = (cy != 0xffff) ? (((uint32_t)(d * d) << 10) | (uint32_t)i) : GIE_WIN_NONE;
                scy[i] = cy;
            }
            if (lane < GIE_WIN_MAX) { sk[-1 - lane] = GIE_WIN_NONE; sk[LP + lane] = GIE_WIN_NONE; }
            gie_wave_sync();
            uint32_t best[CP];
            if (gie_row_window<CP>(sk, X, lane, best)) {
#pragma unroll
                for (int m = 0; m < CP; m++) {
                    const int sx = (int)(best[m] & 1023u);
                    if (64 * m + lane < X) __builtin_nontemporal_store((uint32_t)sx | ((uint32_t)scy[sx] << 16), &out[64 * m + lane]);
                }
                return;
            }
            gie_wave_sync();                                    /* a sparse stretch: the envelope forms take the row (they rebuild the site list) */
        }
    }
#pragma unroll
    for (int m = 0; m < CP; m++) {
        const int i = 64 * m + lane;
        const uint16_t cy = (i < X) ? cyv[m] : (uint16_t)0xffff;
        const int d = y - (int)cy;
        if (64 * m < X) K = gie_row_compact_push(ce, K, cy != 0xffff, (uint32_t)(d * d), i, (uint32_t)cy, lane, mb);
    }
    const int u0 = lane * CP;
    uint32_t o[CP];
    if (K == 0) {                                               /* slice without obstacle */
#pragma unroll
        for (int m = 0; m < CP; m++) o[m] = 0xffffffffu;
    } else if (K <= GIE_BAND_MAXK) {
        /* few sites (the usual case: a handful of obstacle columns per row): banded form, results
         * for positions lane, 64 + lane, ... — stored as 4-byte coalesced rows */
        gie_row_mb_pad(mb, K, lane);
        gie_wave_sync();
        int sj[CP];
        gie_row_argmin_banded_lin<CP>(mb, K, X, lane, sj);
#pragma unroll
        for (int m = 0; m < CP; m++) {
            const uint32_t e = ce[sj[m]].y;
            if (64 * m + lane < X) out[64 * m + lane] = ((e & 0xffffu) >> 5) | (e & 0xffff0000u);
        }
        return;
    } else {
        gie_wave_sync();
        int sj[CP];
        gie_row_argmin<CP>(ce, K, X, lane, sj);
#pragma unroll
        for (int m = 0; m < CP; m++) {
            const uint32_t e = ce[sj[m]].y;
            o[m] = ((e & 0xffffu) >> 5) | (e & 0xffff0000u);     /* cx | cy << 16 */
        }
    }
    if (CP >= 4 && (X & 3) == 0) {
#pragma unroll
        for (int m = 0; m < CP; m += 4)
            if (u0 + m < X) *reinterpret_cast<uint4 *>(out + u0 + m) = make_uint4(o[m], o[m + 1 < CP ? m + 1 : m], o[m + 2 < CP ? m + 2 : m], o[m + 3 < CP ? m + 3 : m]);
    } else {
#pragma unroll
        for (int m = 0; m < CP; m++) if (u0 + m < X) out[u0 + m] = o[m];
    }
}

/* ------------------------------------------------------------------ EDT pass Z */
/* one workgroup per (y, tile of TX columns): the tile [Z][TX] of pass-X results is staged in
 * LDS with coalesced loads, each wave runs the envelope along z for its columns and overwrites
 * the column in place with the packed closest obstacle, written back with coalesced stores
 * (dist² is a function of it and is never stored). */
/* TX columns per workgroup (TX*4-byte row segments in HBM), padded LDS row stride TX+1 so that
 * column walks hit distinct banks, WAVES waves per workgroup. */
/* `full` = 0: only the tiles somebody reads are produced (c.zneed: tiles with a known voxel or on
 * a face of the volume); whole workgroup tiles, columns, 64-position bands and output rows
 * without a reader are skipped.  `full` = 1 (gie_read_batch_edt): every voxel. */
template <int CP, int TX, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (CP <= 8 ? 4 : 2)) void k_edt_z(const gie_ctx c, const int ntiles_x, const int ntiles, const int full)
{
    static_assert(TX == 16 && WAVES == 8, "a workgroup tile spans two 8-voxel tile columns, one column per wave and half");
    if (full == 0 && gie_z_use_lists(c, c.cnt[GIE_CNT_TL_KNOWN])) return;   /* few known tiles: k_edt_z_direct (launched next to this one) does the pass */
    const int repair = c.cnt[GIE_CNT_ZSTREAM];            /* the streaming form (k_edt_z_stream, launched before this one) has done the volume: only the tiles it flagged */
    if (repair && c.cnt[GIE_CNT_ZFAIL] == 0) return;      /* ... and it finished every column (a walk over the flags of 16 K tiles was 40 us of nothing) */
    constexpr int LP = 64 * CP;
    constexpr int TS = TX + 1;
    constexpr int NT = 64 * WAVES;
    constexpr int ZSTEP = NT / TX;                       /* z rows covered by one pass of the workgroup */
    constexpr int NLD = (LP + ZSTEP - 1) / ZSTEP;        /* loads per thread and tile (Z <= LP) */
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Z = c.Z, X = c.X, Y = c.Y;
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem);                               /* [Z][TS] cx|cy<<16 → bcoc */
    uint2 *s_ce = reinterpret_cast<uint2 *>(tile + (((size_t)Z * TS + 3) & ~(size_t)3));   /* [WAVES][LP] */
    uint16_t *s_zl = reinterpret_cast<uint16_t *>(s_ce + WAVES * LP);                       /* [LP] planes with obstacles, ascending */
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = threadIdx.x & (TX - 1), tz = threadIdx.x / TX;
    const size_t plane = (size_t)X * Y;
    uint2 *ce = s_ce + wave * LP;
    /* persistent workgroup: tiles t, t+G, t+2G, …; the NEXT tile is fetched into registers while
     * the envelopes of the current one are computed out of LDS */
    uint32_t pre[NLD];
    unsigned zmask = 0, zin = 0;                          /* this thread's z rows that lie in a plane with obstacles / inside the volume */
#pragma unroll
    for (int j = 0; j < NLD; j++) { const int z = tz + j * ZSTEP; if (z < Z) { zin |= 1u << j; if (c.zocc[z]) zmask |= 1u << j; } }
    /* the real sites of EVERY column are the planes with obstacles (a plane that holds one gives
     * every voxel of the plane a closest obstacle): one list for the whole launch (k_edt_prep) */
    const int K = *c.zcount;
    for (int j = threadIdx.x; j < K; j += NT) s_zl[j] = c.zlist[j];
    unsigned zvalid = 0;                                  /* bit m: plane 64 m + lane holds obstacles (windowed form) */
#pragma unroll
    for (int m = 0; m < CP; m++) { const int z = 64 * m + lane; if (z < Z && c.zocc[z]) zvalid |= 1u << m; }
    __syncthreads();

    /* A 16-column row segment is half a 128-byte line.  x-adjacent tiles — the two halves of the same lines — go to two workgroups
     * of the SAME XCD (workgroup ids are dealt round-robin over the eight XCDs: b and b ^ 8 share one) in the same iteration, so
     * that the second half is still in that XCD's L2 when it is asked for (round 4; before, one workgroup took the two tiles one
     * after the other — a whole tile's column work apart — and the pass fetched 1.46x its bytes).  A grid that is not a multiple
     * of 16 workgroups keeps the old order. */
    int it = 0;
    const bool xpair = (gridDim.x & 15u) == 0u;
    const int pq = ((int)blockIdx.x & 7) | (((int)blockIdx.x >> 4) << 3), ph = ((int)blockIdx.x >> 3) & 1, phalf = (int)gridDim.x >> 1;
#define GIE_ZTILE(i) (xpair ? (((i) * phalf + pq) * 2 + ph) : ((((i) >> 1) * (int)gridDim.x + (int)blockIdx.x) * 2 + ((i) & 1)))
    int t = GIE_ZTILE(0);
    /* rows are addressed as buffer offsets: a 32-bit byte offset per thread and tile + a scalar row stride (N * 4 bytes < 2^32 for
     * every volume gie_create accepts).  With 64-bit pointers the compiler kept sixteen row addresses per thread alive across the
     * column work and spilled them — and a reload from scratch waits for ALL of the wave's outstanding memory operations
     * (vmcnt counts in order): the prefetch below was waited for at once, the write-out went store by store. */
    /* A row a thread has no business with gets an offset beyond the buffer: the hardware's range check drops the access (a load
     * returns 0 — such rows are never staged — a store goes nowhere), so the sixteen loads and the sixteen stores of a thread are
     * straight-line code: no branch per row.  (Skipping a row none of the wave's lanes wants with a wave-uniform branch was
     * measured too: +0.015 ms on the dense workload, nothing gained on the sparse ones.) */
#define GIE_BUF_OOB 0xfffffff0u
    const unsigned nbytes = (unsigned)((size_t)X * Y * Z * 4u);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(c.cxy2), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(c.bcoc, 0, nbytes, 0x00020000);
    const unsigned zstride_b = (unsigned)(plane * ZSTEP * 4u);   /* bytes between a thread's consecutive rows */
    const unsigned tzoff_b = (unsigned)((size_t)tz * plane * 4u);
    /* reader masks of the two 8-wide tile columns a workgroup tile spans (workgroup-uniform) */
    uint64_t nd0, nd1, ndn0 = 0, ndn1 = 0;               /* bit tz: tile (tx, ty, tz) has a reader */
    /* (read through the scalar cache: a vector load here would sit behind the previous tile's stores in the wave's in-order memory
     * counter, and waiting for it would wait for them) */
#define GIE_LOAD_NEED(tt, o0, o1) do { \
        const int txc_ = (((tt) % ntiles_x) * TX) >> 3, tyc_ = ((tt) / ntiles_x) >> 3; \
        const uint64_t *p_ = c.zneed + ((size_t)tyc_ * c.tfd[0] + txc_); \
        o0 = ~0ull; o1 = ~0ull; \
        if (repair) { uint32_t r_; asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) : "s"(c.zredo + (tt)) : "memory"); if (!r_) { o0 = 0ull; o1 = 0ull; } } \
        if (!full && (o0 | o1) != 0ull) { \
            uint64_t a_, b_ = 0ull; \
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_) : "s"(p_) : "memory"); \
            if (txc_ + 1 < c.tfd[0]) asm volatile("s_load_dwordx2 %0, %1, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=s"(b_) : "s"(p_) : "memory"); \
            o0 = a_; o1 = b_; } } while (0)
    if (t < ntiles) {
        GIE_LOAD_NEED(t, ndn0, ndn1);
        const int x = (t % ntiles_x) * TX + tx, y = t / ntiles_x;
        const unsigned voff = tzoff_b + (unsigned)(y * X + x) * 4u;
        if ((ndn0 | ndn1) != 0ull) {
#pragma unroll
            for (int j = 0; j < NLD; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, (((zmask >> j) & 1u) && x < X) ? voff : GIE_BUF_OOB, j * zstride_b, 0);
        }
    }
    for (; t < ntiles; t = GIE_ZTILE(it)) {
        it++;
        const int x0 = (t % ntiles_x) * TX, y = t / ntiles_x;
        nd0 = ndn0; nd1 = ndn1;
        const bool work = (nd0 | nd1) != 0ull;            /* workgroup-uniform */
        if (work) {
#pragma unroll
            for (int j = 0; j < NLD; j++) { if ((zmask >> j) & 1u) tile[(tz + j * ZSTEP) * TS + tx] = pre[j]; }   /* only site rows are ever read */
        }
        __syncthreads();
        const int tn = GIE_ZTILE(it);
        if (tn < ntiles) {                                /* prefetch: in flight during the column work */
            GIE_LOAD_NEED(tn, ndn0, ndn1);
            if ((ndn0 | ndn1) != 0ull) {
                const int xn = (tn % ntiles_x) * TX + tx, yn = tn / ntiles_x;
                const unsigned voff = tzoff_b + (unsigned)(yn * X + xn) * 4u;
#pragma unroll
                for (int j = 0; j < NLD; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, (((zmask >> j) & 1u) && xn < X) ? voff : GIE_BUF_OOB, j * zstride_b, 0);
            }
        }
        if (!work) continue;                              /* nobody reads this tile: nothing loaded, nothing stored */
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int col = wave + 8 * half;
            const int x = x0 + col;
            if (x >= X) break;
            const uint64_t nc = half ? nd1 : nd0;
            if (nc == 0ull) continue;                     /* this 8-wide tile column has no reader */
            if (CP >= 4 && K > GIE_BAND_MAXK) {
                /* obstacles in most planes: windowed scan along z over the direct-indexed column (the
                 * wave's site-list memory holds it); planes without obstacles carry no value */
                unsigned bn = 0;
#pragma unroll
                for (int m = 0; m < CP && m < 8; m++) if ((nc >> (8 * m)) & 0xffull) bn |= 1u << m;
                if (full) bn = ~0u;
                uint32_t *sk = reinterpret_cast<uint32_t *>(ce) + GIE_WIN_MAX;
#pragma unroll
                for (int m = 0; m < CP; m++) {
                    const int i = 64 * m + lane;
                    uint32_t key = GIE_WIN_NONE;
                    if ((zvalid >> m) & 1u) {                    /* rows of planes without obstacles are never staged */
                        const uint32_t v = tile[i * TS + col];
                        if (v != 0xffffffffu) { const int dx = x - (int)(v & 0xffffu), dy = y - (int)(v >> 16); key = ((uint32_t)(dx * dx + dy * dy) << 10) | (uint32_t)i; }
                    }
                    sk[i] = key;
                }
                if (lane < GIE_WIN_MAX) { sk[-1 - lane] = GIE_WIN_NONE; sk[LP + lane] = GIE_WIN_NONE; }
                gie_wave_sync();
                uint32_t best[CP];
                const bool okw = gie_row_window<CP>(sk, Z, lane, best, bn);
                if (okw) {
                    uint32_t oc[CP];
#pragma unroll
                    for (int m = 0; m < CP; m++) {
                        if (64 * m + lane < Z && ((bn >> m) & 1u)) {
                            const int s = (int)(best[m] & 1023u);
                            const uint32_t v = tile[s * TS + col];
                            oc[m] = gie_pack_bcoc((int)(v & 0xffffu), (int)(v >> 16), s);
                        }
                    }
                    gie_wave_sync();
#pragma unroll
                    for (int m = 0; m < CP; m++)
                        if (64 * m + lane < Z && ((bn >> m) & 1u)) tile[(64 * m + lane) * TS + col] = oc[m];
                    gie_wave_sync();
                    continue;
                }
                gie_wave_sync();                                /* a sparse stretch of the column: envelope forms below */
            }
            for (int j = lane; j < K; j += 64) {                /* site j = plane s_zl[j]: value = in-plane distance² */
                const int i = s_zl[j];
                const uint32_t v = tile[i * TS + col];
                const int dx = x - (int)(v & 0xffffu), dy = y - (int)(v >> 16);
                ce[j] = make_uint2(((uint32_t)(dx * dx + dy * dy) << 10) | (uint32_t)j, (uint32_t)i << 5);
            }
            gie_wave_sync();
            if (K == 0) {                                       /* the whole volume is empty */
                for (int i = lane; i < Z; i += 64) tile[i * TS + col] = GIE_BCOC_NONE;
            } else {
                /* 64-position bands (8 z tiles each) with a reader */
                unsigned bandneed = 0;
#pragma unroll
                for (int m = 0; m < CP && m < 8; m++) if ((nc >> (8 * m)) & 0xffull) bandneed |= 1u << m;
                if (full) bandneed = ~0u;                       /* also covers Z > 512 (more than 64 z tiles) */
                int sj[CP];
                const bool banded = K <= GIE_BAND_MAXK;         /* wave-uniform: few sites → banded form */
                if (banded) gie_row_argmin_banded<CP>(ce, K, Z, lane, sj, bandneed);
                else gie_row_argmin<CP>(ce, K, Z, lane, sj);
                /* gather first (own column only), then overwrite the column in place.
                 * position of result m: 64 m + lane (banded) or lane CP + m */
                const int ub = banded ? lane : lane * CP, us = banded ? 64 : 1;
                uint32_t oc[CP];
#pragma unroll
                for (int m = 0; m < CP; m++) {
                    if (ub + m * us < Z && (!banded || ((bandneed >> m) & 1u))) {
                        const int s = (int)((ce[sj[m]].y & 0xffffu) >> 5);
                        const uint32_t v = tile[s * TS + col];
                        oc[m] = gie_pack_bcoc((int)(v & 0xffffu), (int)(v >> 16), s);
                    }
                }
                gie_wave_sync();
#pragma unroll
                for (int m = 0; m < CP; m++)
                    if (ub + m * us < Z && (!banded || ((bandneed >> m) & 1u))) tile[(ub + m * us) * TS + col] = oc[m];
            }
            gie_wave_sync();
        }
        __syncthreads();
        {
            const int x = x0 + tx;
            {
                const unsigned voff = tzoff_b + (unsigned)(y * X + x) * 4u;
                /* which of this thread's rows have a reader: one mask up front, then the row's LDS reads and stores go out back to
                 * back (with the 64-bit reader masks looked at inside the loop the compiler spilled them, and every reload waits for
                 * ALL outstanding memory operations — the stores went out one at a time) */
                static_assert(ZSTEP % 8 == 0, "a thread's rows are whole z tiles apart");
                const uint64_t nds = ((tx >> 3) ? nd1 : nd0) >> (tz >> 3);      /* bit (ZSTEP / 8) j: row j's z tile has a reader */
                unsigned wm = 0;
#pragma unroll
                for (int j = 0; j < NLD; j++) if (full || (j * (ZSTEP / 8) < 64 && ((nds >> (j * (ZSTEP / 8))) & 1ull))) wm |= 1u << j;
                {
                    unsigned zl = zin;
                    asm volatile("" : "+v"(zl));          /* (keeps the compiler from hoisting sixteen single-bit masks out of the tile loop — and spilling them) */
                    wm &= zl;
                    if (x >= X) wm = 0;                   /* (a column beyond the volume: all of its stores are dropped) */
                }
                {
#pragma unroll
                    for (int j0 = 0; j0 < NLD; j0 += 4) {
                        uint32_t ov[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) ov[u] = (j0 + u < NLD) ? tile[(tz + (j0 + u) * ZSTEP) * TS + tx] : 0u;   /* (a row beyond Z reads the site lists behind the tile: masked below) */
#pragma unroll
                        for (int u = 0; u < 4; u++) if (j0 + u < NLD) __builtin_amdgcn_raw_buffer_store_b32(ov[u], rs_out, ((wm >> (j0 + u)) & 1u) ? voff : GIE_BUF_OOB, (j0 + u) * zstride_b, 0);
                    }
                }
            }
        }
        __syncthreads();                                  /* tile is overwritten by the next trip */
    }
}

/* ------------------------------------------------------------------ EDT pass Z, streaming form (dense fields)
 * With obstacles a few voxels apart in every direction (BASELINE config 5: 1 % of the voxels) the closest obstacle of a voxel
 * lies within a handful of planes, and the column kernel above — tile staged in LDS, per-column windows out of LDS, two workgroup
 * barriers per tile, 4 waves per SIMD — spends its time waiting, not computing (round 4: 0.45 ms for 1.08 GB; SALU = VALU,
 * a quarter of the LDS cycles conflicts).  This form has no LDS round trip at all: a wave owns 64 consecutive columns of one
 * y (lane = x) and walks along z; every row of pass X's result it reads is one 256-byte run.  A window of
 * 48 planes lives in REGISTERS as 32-bit keys
 *     in-plane distance² (8 bits, <= 80) | plane index inside the window (6 bits) | x offset + 8 (5 bits) | y offset + 8 (5 bits) | 0 (8 bits)
 * and position z takes  min over |d| <= R = 8 of  key[z + d] + (d² << 24):  one v_min3-able chain of static register operands —
 * the minimum orders by distance², then by plane (the reference's tie rule: the smaller index), and carries the in-plane closest
 * obstacle along.  EXACT whenever the result is below (R + 1)² = 81: a site outside the window, or one whose in-plane distance² is
 * above 80 (stored as "none"), costs at least 81.  A column that meets a larger value anywhere makes its wave give the slab up and
 * flag the four 16-column tiles it spans (c.zredo): the column kernel, launched behind this one, redoes exactly those (the
 * envelope forms have no such limit).  GIE_ZS_C planes per trip: that many row loads in flight per wave, 2R keys carried over.
 * Measured on the C5 volume (1.08 GB of rows): 0.26 ms = 4.1 TB/s against 0.45 ms of the column kernel; this device streams a
 * read and a write of that size in 0.21 ms (tools/sweep_probe.hip).  R = 6 instead of 8 was tried: 457 of 134 M voxels have no
 * obstacle within 7 planes, their slabs go to the column kernel, and the pass takes 0.50 ms — R = 8 leaves none on that field.
 * Taken when the planes with obstacles are many (K > GIE_BAND_MAXK, the column kernel's own switch to its windowed form) and the
 * volume is swept, not listed. */
#ifndef GIE_ZS_R
#define GIE_ZS_R 8
#endif
#ifndef GIE_ZS_C
#define GIE_ZS_C 16                                       /* planes per trip: 32 measured 0.305 ms on the C5 volume (119 VGPRs, 4 waves per SIMD), 16: 0.296 (7 waves) */
#endif
#define GIE_ZS_W (GIE_ZS_C + 2 * GIE_ZS_R)
/* cache policy of the streaming form's row loads and stores (raw buffer aux bit 1 = nt on gfx950): every row is touched once by this
 * kernel — non-temporal rows leave the L2 to the other streams (pass Z 0.263 -> 0.255 ms on the C5 volume) */
#define GIE_ZS_LDAUX 2
#define GIE_ZS_STAUX 2
#define GIE_ZS_NONE 0xb0000000u                           /* 176 << 24: + 64 stays below 2^8, and above every finished value */
#define GIE_ZS_LIMIT ((uint32_t)((GIE_ZS_R + 1) * (GIE_ZS_R + 1)) << 24)
__device__ __forceinline__ uint32_t gie_zs_key(const uint32_t v, const bool plane_ok, const int x8, const int y8, const int j)
{
    const uint32_t ux = (uint32_t)(x8 - (int)(v & 0xffffu)), uy = (uint32_t)(y8 - (int)(v >> 16));      /* offset + 8: 0 .. 16 when it can matter */
    const int dx = (int)ux - 8, dy = (int)uy - 8;
    const uint32_t a = (uint32_t)(__mul24(dx, dx) + __mul24(dy, dy));
    const bool ok = plane_ok && max(ux, uy) <= 16u && a < (uint32_t)((GIE_ZS_R + 1) * (GIE_ZS_R + 1));
    return ok ? ((a << 24) | ((uint32_t)j << 18) | (ux << 13) | (uy << 8)) : (GIE_ZS_NONE | ((uint32_t)j << 18));
}
/* One trip of a slab again, for the few positions whose closest obstacle lies beyond the register window (BASELINE config 5's hash
 * world: about thirty of 134 M voxels per update have none within 8 planes — and sending their slabs to the column kernel cost
 * 44 us of a 0.24 ms pass): a window of GIE_ZS_R2 planes on either side with full 32-bit keys (value << 10 | plane, no clamp on the
 * in-plane distance), exact below (R2 + 1)².  Returns false when a position of the wave is still unfinished: the slab is given up. */
__device__ __forceinline__ void gie_edt_z_direct_body(const gie_ctx &c);      /* (below, with its own kernel) */
#define GIE_ZS_R2 24
__device__ __noinline__ bool gie_zs_wide_trip(const uint32_t *cxy2, uint32_t *bcoc, const unsigned nbytes, const unsigned voff, const unsigned pstride,
                                              const int zc, const int z1, const int Z, const int x, const int y, const bool inx, const uint8_t *occ)
{
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cxy2), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(bcoc, 0, nbytes, 0x00020000);
    uint32_t best[GIE_ZS_C];
#pragma unroll
    for (int t = 0; t < GIE_ZS_C; t++) best[t] = 0xffffffffu;
    const int lo = max(zc - GIE_ZS_R2, 0), hi = min(zc + GIE_ZS_C + GIE_ZS_R2, Z);
#pragma unroll 1
    for (int i0 = lo; i0 < hi; i0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (unsigned)min(i0 + k, Z - 1) * pstride, 0);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + k;
            if (i >= hi || !occ[i]) continue;             /* wave-uniform */
            const int dx = x - (int)(v[k] & 0xffffu), dy = y - (int)(v[k] >> 16);
            const uint32_t a = (v[k] == 0xffffffffu) ? 0x3fffffu : (uint32_t)(dx * dx + dy * dy);
#pragma unroll
            for (int t = 0; t < GIE_ZS_C; t++) { const int d = zc + t - i; best[t] = min(best[t], (min(a + (uint32_t)(d * d), 0x3fffffu) << 10) | (uint32_t)i); }
        }
    }
    uint32_t worst = 0;
#pragma unroll
    for (int t = 0; t < GIE_ZS_C; t++) if (zc + t < z1) worst = max(worst, best[t]);
    if (__any(inx && worst >= ((uint32_t)((GIE_ZS_R2 + 1) * (GIE_ZS_R2 + 1)) << 10))) return false;
#pragma unroll
    for (int t = 0; t < GIE_ZS_C; t++) {
        const int sp = (int)(best[t] & 1023u);
        const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (unsigned)sp * pstride, 0);
        __builtin_amdgcn_raw_buffer_store_b32(gie_pack_bcoc((int)(v & 0xffffu), (int)(v >> 16), sp), rs_out, (zc + t < z1) ? voff : GIE_BUF_OOB, (unsigned)min(zc + t, Z - 1) * pstride, 0);
    }
    return true;
}
__global__ __launch_bounds__(256, 6) void k_edt_z_stream(const gie_ctx c, const int full, const int nseg, const int seg_len)
{
    __shared__ uint8_t s_occ[1024 + 2 * (32 + 2 * GIE_ZS_R)];        /* plane holds obstacles, for planes -W .. Z + W (0 outside the volume) */
    const int Z = c.Z, X = c.X, Y = c.Y;
    if (full == 0 && gie_z_use_lists(c, c.cnt[GIE_CNT_TL_KNOWN])) { gie_edt_z_direct_body(c); return; }   /* few known tiles: the list form, in this launch (one launch less
                                                                                                         * per update than with a kernel of its own: 4.5 us each on the sparse workloads) */
    if (*c.zcount <= GIE_BAND_MAXK) return;               /* planes with obstacles are few: the column kernel's envelope forms (same answer in every workgroup) */
    if (blockIdx.x == 0 && threadIdx.x == 0) c.cnt[GIE_CNT_ZSTREAM] = 1;
    for (int i = threadIdx.x; i < Z + 2 * GIE_ZS_W; i += 256) { const int z = i - GIE_ZS_W; s_occ[i] = (z >= 0 && z < Z) ? c.zocc[z] : (uint8_t)0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nxr = (X + 63) >> 6;
    const int total = nxr * nseg * Y;
    const size_t plane = (size_t)X * Y;
    const unsigned nbytes = (unsigned)((size_t)X * Y * Z * 4u);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(c.cxy2), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(c.bcoc, 0, nbytes, 0x00020000);
    const unsigned pstride = (unsigned)(plane * 4u);
    for (int w = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); w < total; w += (int)gridDim.x * 4) {
        const int xr = w % nxr, seg = (w / nxr) % nseg, y = w / (nxr * nseg);
        const int x = xr * 64 + lane;
        const int z0 = seg * seg_len, z1 = min(Z, z0 + seg_len);
        if (z0 >= Z) continue;
        const unsigned voff = x < X ? (unsigned)(y * X + x) * 4u : GIE_BUF_OOB;
        const int x8 = x + 8, y8 = y + 8;
        uint32_t wk[GIE_ZS_W];
        /* the planes z0 - R .. z0 + R - 1 of the first trip (window places 0 .. 2R - 1) */
        {
            const int zz = z0 - GIE_ZS_R + lane;
            const unsigned long long pm = __ballot(lane < 2 * GIE_ZS_R && s_occ[zz + GIE_ZS_W] != 0);
            uint32_t v[2 * GIE_ZS_R];
#pragma unroll
            for (int j = 0; j < 2 * GIE_ZS_R; j++) { const int zp = min(max(z0 - GIE_ZS_R + j, 0), Z - 1); v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (unsigned)zp * pstride, 0); }
#pragma unroll
            for (int j = 0; j < 2 * GIE_ZS_R; j++) wk[j] = gie_zs_key(v[j], (pm >> j) & 1ull, x8, y8, j);
        }
        bool failed = false;
        int wide = 0, wz0 = 0, wz1 = 0;                   /* trips of this slab to redo with the wide window (more than two: a sparse field — the column kernel's) */
#pragma unroll 1
        for (int zc = z0; zc < z1; zc += GIE_ZS_C) {
            /* planes zc + R .. zc + C + R - 1 -> window places 2R .. W - 1 */
            {
                const int zz = zc + GIE_ZS_R + lane;
                const unsigned long long pm = __ballot(lane < GIE_ZS_C && s_occ[zz + GIE_ZS_W] != 0);
                uint32_t v[GIE_ZS_C];
#pragma unroll
                for (int j = 0; j < GIE_ZS_C; j++) { const int zp = min(zc + GIE_ZS_R + j, Z - 1); v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (unsigned)zp * pstride, GIE_ZS_LDAUX); }
#pragma unroll
                for (int j = 0; j < GIE_ZS_C; j++) wk[2 * GIE_ZS_R + j] = gie_zs_key(v[j], (pm >> j) & 1ull, x8, y8, 2 * GIE_ZS_R + j);
            }
            uint32_t wlo = 0, whi = 0;                    /* the largest key of the trip's first / last eight planes (a trip is two tiles high) */
#pragma unroll
            for (int t = 0; t < GIE_ZS_C; t++) {
                const int p = GIE_ZS_R + t;
                uint32_t b = wk[p];
#pragma unroll
                for (int d = 1; d <= GIE_ZS_R; d++) b = min(b, min(wk[p - d], wk[p + d]) + ((uint32_t)(d * d) << 24));
                const bool in = zc + t < z1;              /* wave-uniform */
                if (in) { if (t < 8) wlo = max(wlo, b); else whi = max(whi, b); }
                const int s = zc - GIE_ZS_R + (int)((b >> 18) & 63u);
                const int cx = x8 - (int)((b >> 13) & 31u), cy = y8 - (int)((b >> 8) & 31u);
                /* stored at once (no second copy of the trip in registers); a slab given up below is redone as a whole by the column kernel */
                __builtin_amdgcn_raw_buffer_store_b32(gie_pack_bcoc(cx, cy, s), rs_out, in ? voff : GIE_BUF_OOB, (unsigned)min(zc + t, Z - 1) * pstride, GIE_ZS_STAUX);
            }
            const uint32_t worst = max(wlo, whi);
            {   /* the largest distance of each of the sixteen tiles the trip has crossed a row of (exact below 81; "81 or more" otherwise):
                 * what the fused sweep bounds its lazy tiles by (k_markc) — eight lanes per tile, one atomic per tile and row */
                static_assert(GIE_ZS_C == 16, "a trip is two tiles high");
                uint32_t m0 = x < X ? (wlo >> 24) : 0u, m1 = x < X ? (whi >> 24) : 0u;
                m0 = max(m0, (uint32_t)__shfl_xor((int)m0, 1)); m1 = max(m1, (uint32_t)__shfl_xor((int)m1, 1));
                m0 = max(m0, (uint32_t)__shfl_xor((int)m0, 2)); m1 = max(m1, (uint32_t)__shfl_xor((int)m1, 2));
                m0 = max(m0, (uint32_t)__shfl_xor((int)m0, 4)); m1 = max(m1, (uint32_t)__shfl_xor((int)m1, 4));
                if ((lane & 7) == 0 && x < X) {
                    const int t0 = gie_tile_index(c, x, y, zc);
                    __hip_atomic_fetch_max(&c.tbmax[t0], (int32_t)m0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (zc + 8 < z1) __hip_atomic_fetch_max(&c.tbmax[t0 + c.tfd[0] * c.tfd[1]], (int32_t)m1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (__any(x < X && worst >= GIE_ZS_LIMIT)) {  /* wave-uniform: a position of this trip has no obstacle inside the window */
                if (wide == 2) { failed = true; break; }
                if (wide == 0) wz0 = zc; else wz1 = zc;   /* redone with the wide window when the slab is through (no call inside this loop: registers) */
                wide++;
            }
#pragma unroll
            for (int j = 0; j < 2 * GIE_ZS_R; j++) wk[j] = wk[GIE_ZS_C + j] - ((uint32_t)GIE_ZS_C << 18);      /* the last 2R planes are the next trip's first */
        }
        if (!failed && wide > 0) failed = !gie_zs_wide_trip(c.cxy2, c.bcoc, nbytes, voff, pstride, wz0, z1, Z, x, y, x < X, s_occ + GIE_ZS_W);
        if (!failed && wide > 1) failed = !gie_zs_wide_trip(c.cxy2, c.bcoc, nbytes, voff, pstride, wz1, z1, Z, x, y, x < X, s_occ + GIE_ZS_W);
        if (failed && lane < 4) {                         /* the column kernel redoes the slab's tiles (whole columns: every segment's stores are overwritten) */
            const int tx16 = xr * 4 + lane;
            if (tx16 * 16 < X) c.zredo[(size_t)y * ((X + 15) >> 4) + tx16] = 1u;
            if (lane == 0) atomicAdd(&c.cnt[GIE_CNT_ZFAIL], 1);
        }
    }
}

/* Before the EDT passes: the list of planes that hold obstacles (workgroup 0, wave 0) and the
 * reader masks of pass Z (one thread per (x,y) tile column). */
#define GIE_PREP_WAVES 16
__global__ __launch_bounds__(64 * GIE_PREP_WAVES) void k_edt_prep(const gie_ctx c, const int ncol)
{
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        /* all flag loads first (sides are <= 1024: 16 per lane), then the ballots: one memory round trip, not Z/64 */
        uint8_t f[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { const int z = 64 * i + (int)threadIdx.x; f[i] = c.zocc[min(z, c.Z - 1)]; }
        int k = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int z = 64 * i + (int)threadIdx.x;
            const bool v = z < c.Z && f[i];
            const unsigned long long m = __ballot(v);
            if (v) c.zlist[k + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = (uint16_t)z;
            k += __popcll(m);
        }
        if (threadIdx.x == 0) *c.zcount = k;
    }
    /* one wave per (x,y) tile column, lane = z tile: the ballot IS the mask (of the first 64 z tiles: pass Z only goes by the masks
     * when there are no more, gie_batch_edt) */
    const int col = blockIdx.x * GIE_PREP_WAVES + (threadIdx.x >> 6);
    const bool colok = col < ncol;
    const int tx = colok ? col % c.tfd[0] : 0, ty = colok ? col / c.tfd[0] : 0;
    for (int tz0 = 0; tz0 < c.tfd[2]; tz0 += 64) {           /* (volumes taller than 512 voxels: ADVICE r4 — their upper tiles were never listed nor flagged; uniform over the workgroup) */
        const int tz = tz0 + (int)(threadIdx.x & 63);
        const int t = (tz * c.tfd[1] + ty) * c.tfd[0] + tx;
        const bool k = colok && tz < c.tfd[2] && c.tknown[t];
        const unsigned long long m = __ballot(k);
        if (colok && tz == 0) c.zneed[col] = m;
        /* the same tiles as a list (mark / commit / pass Z visit only these when they are few): one atomic per workgroup */
        const int slot = gie_wg_reserve(&c.cnt[GIE_CNT_TL_KNOWN], k);
        if (slot >= 0) c.tl_known[slot] = t;
        /* ... and the ones among them that are not flagged 2 (gie_fuse has flagged): what the fused sweep walks when the others are lazy */
        const int slot2 = gie_wg_reserve(&c.cnt[GIE_CNT_TL_SWEPT], k && c.tskip[t] != 2);
        if (slot2 >= 0) c.tl_swept[slot2] = t;
        /* (the tiles whose stored records Mark need not read — tskip — are flagged by gie_fuse since round 5: be_tile_oldskip) */
    }
}

/* ------------------------------------------------------------------ tskip (gie_fuse, after the occupancy fusion)
 * a thread per tile: gie_tile_oldskip — unless the volume holds no obstacle at all (no plane flagged by fuse): that update's Mark
 * commits nothing, so no tile may count on its pair plane ("deferred records", gie_ops.h) */
__global__ __launch_bounds__(256) void k_tile_oldskip(const gie_ctx c, const int ntile, int32_t *list, int32_t *count)
{
    int any = 0;
    for (int z = threadIdx.x; z < c.Z; z += 256) any |= c.zocc[z];
    any = __syncthreads_or(any);
    const int t = blockIdx.x * 256 + threadIdx.x;
    /* without an obstacle no tile is cleared — and every tile whose records the update before left to its pair plane is listed */
    bool need = false;
    if (t < ntile) need = gie_tile_oldskip(c, t, any) != 0;
    /* (one counter update per WORKGROUP: a thousand wavefronts adding to the same word, each waiting for its answer, was 44 us of
     * this launch's 49 — same-address atomics with a return serialise at their L2 slice) */
    const int slot = gie_wg_reserve(count, need);
    if (slot >= 0) list[slot] = t;
}
/* a wave per tile k_tile_oldskip has listed, a lane per z-column */
__global__ __launch_bounds__(256) void k_coc_catchup_new(const gie_ctx c, const int pu0, const int pu1, const int pu2, const int32_t *list, const int32_t *count)
{
    const int n = *count;
    const int lane = threadIdx.x & 63;
    const int pupvt[3] = { pu0, pu1, pu2 };
    for (int e = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); e < n; e += (int)gridDim.x * 4) gie_coc_catchup_newcolumn(c, pupvt, list[e], lane);
}

/* "lazy pairs" (gie_ops.h): the flagged tiles that this update's sweep will not leave flagged again (stay: it will leave tskip-2 tiles
 * flagged) — listed by a thread per tile, then a wave per listed tile puts the tile's 512 pairs into the plane and takes the flag away */
__global__ __launch_bounds__(256) void k_pair_lazy_list(const gie_ctx c, const int ntile, const int stay, int32_t *list, int32_t *count)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool need = t < ntile && c.tlazy[t] != 0 && !(stay && c.tskip[t] == 2);
    const int slot = gie_wg_reserve(count, need);
    if (slot >= 0) list[slot] = t;
}
__global__ __launch_bounds__(256) void k_pair_lazy_run(const gie_ctx c, const int32_t *list, const int32_t *count)
{
    const int n = *count;
    const int lane = threadIdx.x & 63;
    for (int e = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); e < n; e += (int)gridDim.x * 4) {
        const int t = list[e];
        gie_pair_materialise_column(c, t, lane);
        if (lane == 0) c.tlazy[t] = 0;
    }
}

/* the records a fused update left to its pair plane, for the tiles that are no tskip tiles any more (gie_ops.h "deferred records"):
 * k_coc_catchup_list — a thread per tile of the flags' plane decides (gie_coc_catchup_tile: nearly always "nothing to do") and the
 * tiles found go onto a list (one atomic per wave); k_coc_catchup_run — a wave per listed tile, a lane per z-column.  Two launches
 * because tiles that flip together are neighbours: a wave that stored the records of the tiles its own lanes found would, when
 * a whole layer of tiles flips (a lidar's flood wave moving the bounds), store 64 tiles one after the other while the rest of
 * the launch is through (round 5: up to 0.35 ms on the projective lidar workload); spreading the TEST over the waves instead
 * makes its byte loads uncoalesced (0.55 ms on the C5 volume). */
__global__ __launch_bounds__(256) void k_coc_catchup_list(const gie_ctx c, const gie_catchup p, const int ntile, int32_t *list, int32_t *count)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool need = t < ntile && gie_coc_catchup_tile(c, p, t);
    const int slot = gie_wg_reserve(count, need);
    if (slot >= 0) list[slot] = t;
}
__global__ __launch_bounds__(256) void k_coc_catchup_run(const gie_ctx c, const gie_catchup p, const int32_t *list, const int32_t *count)
{
    const int n = *count;
    const int lane = threadIdx.x & 63;
    for (int e = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); e < n; e += (int)gridDim.x * 4) gie_coc_catchup_column(c, p, list[e], lane);
}

/* ------------------------------------------------------------------ adaptive sweeps */
/* The per-voxel functors of fuse / Mark / obtainFrontiers / commit over either the tiles on a list
 * (one wave per 8x8x8 tile, lane = (x,y) column of the tile) or the whole volume (the geometry of
 * k_voxz: 64 x 4 columns per virtual workgroup), chosen by the kernel itself from the length of
 * the list (gie_use_lists).  A fixed grid strides through the work either way: for a sparsely
 * observed volume this replaces 65 536 workgroups that each find out they have nothing to do. */
template <class F, bool STAGED>
__device__ __forceinline__ void gie_vox_column(const gie_ctx &c, const F &f, const int x, const int y, const int z0)
{
    if (x >= c.X || y >= c.Y || f.tile_skip(c, x, y, z0)) return;
    bool sk[8];
    int id[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int z = z0 + k;
        id[k] = (z < c.Z) ? gie_lid(c, x, y, z) : 0;
        sk[k] = z >= c.Z || f.skip(c, id[k], x, y, z);
    }
    if (STAGED) {
        typename F::st s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) if (!sk[k]) f.load1(c, id[k], x, y, z0 + k, s[k]);
#pragma unroll
        for (int k = 0; k < 8; k++) if (!sk[k]) f.load2(c, id[k], x, y, z0 + k, s[k]);
        unsigned known = 0, valid = 0;
        int vmax = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (z0 + k < c.Z) valid |= 1u << k;
            if (!sk[k]) { const int r = f.finish(c, id[k], x, y, z0 + k, s[k]); known |= (unsigned)(r != 0) << k; vmax = r > vmax ? r : vmax; }
        }
        gie_column_hook_impl(f, c, x, y, z0, known, valid, 0);
        gie_column_max_hook_impl(f, c, x, y, z0, known, valid, vmax, 0);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) if (!sk[k]) f(c, x, y, z0 + k);
    }
}
/* LX = lanes of a wave along x in the sweep form (64 / LX rows of y per wave): 64 gives the local
 * x-fastest planes whole 64-voxel rows per instruction but the global block planes 8-voxel pieces of
 * eight blocks; smaller LX trades row length for longer runs inside a block (a block plane holds
 * x | y<<3 | z<<6): LX = 16 reads 4 rows of 16 voxels and touches 2 blocks with 32 consecutive
 * voxels each. */
template <class F, bool STAGED, int LX>
__global__ __launch_bounds__(256) void k_voxa(const gie_ctx c, const F f, const int32_t *list, const int count_idx, const int always_list)
{
    if (GIE_GATE_CLOSED(c)) return;
    const int n = c.cnt[count_idx];
    const int lane = threadIdx.x & 63;
    if (always_list == 2 && !gie_use_lists(c, n)) return;        /* the dense form is another kernel's (k_fuse_rows) */
    if (always_list || gie_use_lists(c, n)) {
        const int waves = gridDim.x * 4;
        for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += waves) {
            const int t = list[e];
            const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
            gie_vox_column<F, STAGED>(c, f, tx * 8 + (lane & 7), ty * 8 + (lane >> 3), tz * 8);
        }
    } else {
        constexpr int LY = 64 / LX, WY = 4 * LY;               /* rows per wave / per workgroup */
        const int gx = (c.X + LX - 1) / LX, gy = (c.Y + WY - 1) / WY, gz = (c.Z + 7) / 8;
        const int nv = gx * gy * gz;
        /* a workgroup takes a contiguous run of virtual workgroups (whole x rows of one (y, z) strip):
         * measured a little faster than striding through the volume (fuse 0.18 -> 0.155 ms, dense) */
        const int per = (nv + (int)gridDim.x - 1) / (int)gridDim.x;
        const int lx = lane % LX, ly = (int)(threadIdx.x >> 6) * LY + lane / LX;
        int v = (int)blockIdx.x * per;
        const int vend = min(nv, ((int)blockIdx.x + 1) * per);
        int vx = v % gx, vy = (v / gx) % gy, vz = v / (gx * gy);
        for (; v < vend; v++) {
            gie_vox_column<F, STAGED>(c, f, vx * LX + lx, vy * WY + ly, vz * 8);
            if (++vx == gx) { vx = 0; if (++vy == gy) { vy = 0; vz++; } }
        }
    }
}

/* ------------------------------------------------------------------ Mark + commit, the dense sweep's own kernel */
/* MarkLimitedObserve + UpdateHashBatch as one sweep (gie_ops.h "Mark + commit"): the same per-voxel functions as op_markc
 * under k_voxa (gie_mark_logic, gie_commit_pair, gie_markc_column) and the same geometry (thread = z-column of one tile, lanes LX
 * along x), but every load of a column's batch goes out BEFORE anything is tested: types, batch obstacles, the two block slots a
 * column can touch and the tile's skip flag in one batch, the stored records in a second one, then arithmetic and stores.  The
 * generic staged sweep tests the type first and loads behind the branch — four dependent round trips per column instead of
 * two; on the C5 volume the sweep is bound by how many bytes it keeps in flight, and by its stores (tools/sweep_probe.hip:
 * this device writes at ~4.1 TB/s and reads at ~6.5, one after the other). */
/* "LAZY PAIRS" (round 6, gie_ops.h).  A tskip tile with deferred records inside the wave range — 3/4 of the headline volume — has
 * nothing to decide: no stored record can win, every voxel is known (the update before committed them all), its batch obstacle lies
 * inside the volume, so MarkLimitedObserve's answer is (batch distance, batch obstacle in wave-range coordinates): a function of the
 * batch-obstacle plane, which is NOT copied into the pair plane any more — the tile is flagged, readers derive (gie_pair_of_bcoc).
 * What is left of the sweep there: the columns' `ucol` bytes (Mark "has written" the indices; gie_markc_column_fast), and per TILE
 * the flag and the BOUND for the next update's gie_tile_oldskip, here.  Any upper bound of the largest distance in the tile will do
 * (a larger bound clears fewer tiles, never a wrong one), and the distance is the exact EDT's, so by the triangle inequality
 * sqrt(d(v)) <= sqrt(d(s)) + |v - s|: eight samples, the voxel (1, 1, 1) of every octant, none of whose voxels is farther than
 * |(2, 2, 2)| < 4 from it.  (A sample without a batch obstacle: an update without obstacles — it clears no tile, so it cannot be
 * here — but if it is: the bound is "infinite".)  A thread per tile: the sweep's threads would each wait for the samples' round
 * trip behind the flags', a chain per virtual workgroup that — with the bytes gone — was what the sweep's time consisted of. */
__device__ __forceinline__ void gie_markc_lazy_tile(const gie_ctx &c, const int t)
{
    if (!c.tknown[t] || c.tskip[t] != 2) return;
    const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
    const int x = tx * 8, y = ty * 8, z0 = tz * 8;
    const size_t plane = (size_t)c.X * c.Y;
    const size_t id0 = ((size_t)z0 * c.Y + y) * c.X + x;
    /* the bound: pass Z's streaming form has recorded the tile's largest distance on its way (exact below 81) — otherwise eight samples */
    const int wb = c.cnt[GIE_CNT_ZSTREAM] ? c.tbmax[t] : 0x7fffffff;
    int vmax = wb <= 80 ? wb + 1 : 0;
    if (wb > 80) {
        uint32_t sb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) sb[k] = c.bcoc[id0 + (size_t)(1 + 4 * (k >> 2)) * plane + (size_t)(1 + 4 * ((k >> 1) & 1)) * c.X + (size_t)(1 + 4 * (k & 1))];
        vmax = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int sx = x + 1 + 4 * (k & 1), sy = y + 1 + 4 * ((k >> 1) & 1), sz = z0 + 1 + 4 * (k >> 2);
            int v = GIE_TMAX_INF;
            if (sb[k] != GIE_BCOC_NONE) {
                const int dx = sx - (int)(sb[k] & 1023u), dy = sy - (int)((sb[k] >> 10) & 1023u), dz = sz - (int)(sb[k] >> 20);
                const int r = (int)ceilf(sqrtf((float)(dx * dx + dy * dy + dz * dz))) + 4;
                v = r * r + 1;
            }
            vmax = v > vmax ? v : vmax;
        }
    }
    c.tmax[t] = vmax;
    c.tlazy[t] = 1;
    /* Mark "has written" the tile's indices: its 64 `ucol` bytes, eight rows of eight (a whole tile: tskip 2) */
    uint8_t *const u0 = c.ucol + gie_ucol_index(c, x, y, z0);
    if ((c.X & 7) == 0) {
        uint64_t ub[8];
#pragma unroll
        for (int r = 0; r < 8; r++) ub[r] = *reinterpret_cast<const uint64_t *>(u0 + (size_t)r * c.X);
#pragma unroll
        for (int r = 0; r < 8; r++) if (ub[r]) *reinterpret_cast<uint64_t *>(u0 + (size_t)r * c.X) = 0ull;
    } else {
        for (int r = 0; r < 8; r++) for (int q = 0; q < 8; q++) if (u0[(size_t)r * c.X + q]) u0[(size_t)r * c.X + q] = 0;
    }
}
/* What a column's thread looks at first: is the tile known, how are its old records treated, the column's `ucol` byte (packed
 * into one word), and the (at most two) block slots its eight voxels lie in.  k_markc loads these for a run of virtual workgroups
 * at once, and the column then asks for its types, batch obstacles AND stored records in one batch: two round trips per column
 * instead of four (flags; types + slots; stored records; stores) — away from the lazy tiles the sweep is a chain of round
 * trips, not a stream (round 6: without the read of the stored records 0.35 -> 0.24 ms, of which their bytes explain a fifth). */
struct gie_markc_pre { uint32_t flags; int32_t slot_lo, slot_hi; };
__device__ __forceinline__ gie_markc_pre gie_markc_flags(const gie_ctx &c, const int x, const int y, const int z0)
{
    gie_markc_pre p = { 0u, -1, -1 };
    if (x >= c.X || y >= c.Y || z0 >= c.Z) return p;
    const int t = gie_tile_index(c, x, y, z0);
    const int nz = min(8, c.Z - z0);
    const int gx = x + c.pvt[0], gy = y + c.pvt[1], gz0 = z0 + c.pvt[2];
    const uint32_t kn = c.tknown[t], sk = c.tskip[t], ub = c.ucol[gie_ucol_index(c, x, y, z0)];
    p.slot_lo = c.blk_tab[gie_tab_index(c, gx, gy, gz0)];
    p.slot_hi = c.blk_tab[gie_tab_index(c, gx, gy, gz0 + nz - 1)];
    p.flags = (kn ? 1u : 0u) | (sk << 8) | (ub << 16);
    return p;
}
__device__ __forceinline__ void gie_markc_column_fast(const gie_ctx &c, const int x, const int y, const int z0, const gie_markc_pre pre)
{
    const uint32_t flags = pre.flags;
    if (!(flags & 1u)) return;                          /* (outside the volume, or a tile without a known voxel) */
    const int t = gie_tile_index(c, x, y, z0);
    const size_t plane = (size_t)c.X * c.Y;
    const size_t id0 = ((size_t)z0 * c.Y + y) * c.X + x;
    const int nz = min(8, c.Z - z0);
    const int gx = x + c.pvt[0], gy = y + c.pvt[1], gz0 = z0 + c.pvt[2];
    const int skipold = (int)((flags >> 8) & 255u);
    const bool nostore = c.coc_defer && skipold == 2;     /* a tskip tile with deferred records: the sweep neither reads nor writes the global map here */
    const size_t ui = gie_ucol_index(c, x, y, z0);
    const unsigned ub = flags >> 16;                    /* indices that have just turned known: their old pair says nothing about `_edt_D` */
    if (nostore && c.lazy_ok) return;                   /* (a lazy tile: gie_markc_lazy_tile has done what there is to do; never listed) */
    int8_t ty[8]; uint32_t bc[8]; gie_vaddr a[8]; uint64_t oc[8];
    const int slot_lo = pre.slot_lo, slot_hi = pre.slot_hi;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const size_t id = id0 + (size_t)(k < nz ? k : 0) * plane;
        ty[k] = c.glb_type[id]; bc[k] = __builtin_nontemporal_load(&c.bcoc[id]);      /* (the batch obstacles are read once, here; the pairs below written once: non-temporal, -3 %) */
        const int gz = gz0 + k;
        const int slot = ((gz >> 3) == (gz0 >> 3)) ? slot_lo : slot_hi;
        a[k] = slot < 0 ? (gie_vaddr)-1 : (gie_vaddr)slot * GIE_VBSZ + gie_vox_in_blk(gx, gy, gz);
        oc[k] = 0;
        if (!skipold && k < nz && a[k] >= 0) oc[k] = c.g_coc[a[k]];      /* (asked for before the type is known: an unknown voxel's record is read for nothing) */
    }
    if (nostore && c.wr_inside) {
        /* A tskip tile with deferred records — 82 % of the C5 volume — and nothing to decide: no stored record can win (dold is
         * "infinite"), the batch obstacle lies inside the volume and the volume inside the wave range, so MarkLimitedObserve's
         * answer is (batch distance, batch obstacle in wave-range coordinates), no tile flag, no `_edt_D` to keep, and nothing goes
         * to the global map — no block slot, no voxel address.  A voxel without a batch obstacle cannot occur here (an update
         * without obstacles clears no tile); if one does, the general path below takes the column. */
        unsigned want = 0;
        bool plain = true;
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < nz && ty[k] != GIE_VOX_UNKNOWN) { want |= 1u << k; if (bc[k] == GIE_BCOC_NONE) plain = false; }
        if (plain) {
            const uint32_t ox = (uint32_t)(c.pvt[0] - c.upvt[0]), oy = (uint32_t)(c.pvt[1] - c.upvt[1]), oz = (uint32_t)(c.pvt[2] - c.upvt[2]);
            int vmax = 0;
            uint64_t *const pp = c.pair + id0;
            /* (one tile of several — lazy_ok 0 — gets here and stores: its faces lie inside the whole volume and the refinement rounds read them) */
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!((want >> k) & 1u)) continue;
                const uint32_t b = bc[k];
                const uint32_t cx = b & 1023u, cy = (b >> 10) & 1023u, cz = b >> 20;
                const int dx = x - (int)cx, dy = y - (int)cy, dz = z0 + k - (int)cz;
                const uint32_t d = (uint32_t)(__mul24(dx, dx) + __mul24(dy, dy) + __mul24(dz, dz));
                const uint32_t wz = cz + oz;
                const uint32_t lo = (cx + ox) | ((cy + oy) << 14) | (wz << 28), hi = (wz >> 4) | (d << (GIE_PAIR_DIST_SHIFT - 32));
                __builtin_nontemporal_store(((uint64_t)hi << 32) | lo, &pp[(size_t)k * plane]);
                vmax = (int)d + 1 > vmax ? (int)d + 1 : vmax;
            }
            if (ub & want) c.ucol[ui] = (uint8_t)(ub & ~want);
            gie_markc_column(c, x, y, z0, want, (1u << nz) - 1u, vmax);
            return;
        }
    }
    int dold[8];
    unsigned want = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < nz && ty[k] != GIE_VOX_UNKNOWN) want |= 1u << k;
        dold[k] = GIE_TMAX_INF;
    }
    if (!skipold) {
#pragma unroll
        for (int k = 0; k < 8; k++) if (((want >> k) & 1u) && a[k] >= 0) dold[k] = gie_gdist(c, oc[k], gx, gy, gz0 + k);
    }
    unsigned known = 0, valid = 0;
    int vmax = 0, flag = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < nz) valid |= 1u << k;
        if (!((want >> k) & 1u)) continue;
        const size_t id = id0 + (size_t)k * plane;
        int r;
        if (a[k] < 0) { gie_commit_pair<false>(c, (int)id, -1, c.pair[id]); r = GIE_TMAX_INF; }      /* (a known voxel always has its block) */
        else {
            int ft;
            const uint64_t pr = gie_mark_logic(c, x, y, z0 + k, bc[k], dold[k], oc[k], &c.pair[id], &ft);
            flag |= ft;
            gie_edt_before_keep(c, (int)id, pr, (ub >> k) & 1u);
            c.pair[id] = pr;
            if (!nostore) gie_commit_pair<false>(c, (int)id, a[k], pr);      /* (a tskip tile that did not take the short way above) */
            const int d = gie_pair_dist(pr);
            r = d == c.empty_value ? GIE_TMAX_INF : d + 1;
        }
        known |= 1u << k; vmax = r > vmax ? r : vmax;
    }
    if (flag) c.tflag[t] = 1;
    if (ub & known) c.ucol[ui] = (uint8_t)(ub & ~known);           /* (the byte is this thread's) */
    gie_markc_column(c, x, y, z0, known, valid, vmax);
}
#ifndef GIE_MARKC_OCC
#define GIE_MARKC_OCC 5
#endif
template <int LX>
__global__ __launch_bounds__(256, GIE_MARKC_OCC) void k_markc(const gie_ctx c, const int32_t *list)   /* (five waves per SIMD: 96 registers; one more costs the sweep a tenth of its time) */
{
    const int n = c.cnt[GIE_CNT_TL_KNOWN];
    const int lane = threadIdx.x & 63;
    const bool lazy = c.coc_defer && c.lazy_ok;
    if (lazy) {
        if (blockIdx.x == 0 && threadIdx.x == 0) c.cnt[GIE_CNT_LAZY_EXACT] = c.cnt[GIE_CNT_ZSTREAM] ? 1 : 0;      /* (gie_tile_oldskip's margin, next update) */
        const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
        for (int t = (int)(blockIdx.x * 256 + threadIdx.x); t < ntile; t += (int)gridDim.x * 256) gie_markc_lazy_tile(c, t);
    }
    /* With lazy tiles the sweep is a WALK: a wave per tile of the swept list (k_edt_prep: known, not flagged 2).  The volume-order
     * sweep below started 16 K workgroups of which 7 K found four lazy runs and left, and 2 K found four runs to sweep one after
     * the other, 40 us each — per-workgroup stamps (round 6) showed 740 workgroups resident where 1280 fit, the launch as long as
     * the workgroups could be started, not as long as the work took. */
    if (lazy || gie_use_lists(c, n)) {
        const int waves = gridDim.x * 4;
        if (lazy) { list = c.tl_swept; }
        const int nl = lazy ? c.cnt[GIE_CNT_TL_SWEPT] : n;
        for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < nl; e += waves) {
            const int t = list[e];
            const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
            const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
            gie_markc_column_fast(c, x, y, tz * 8, gie_markc_flags(c, x, y, tz * 8));
        }
    } else {
        constexpr int LY = 64 / LX, WY = 4 * LY;
        const int gx = (c.X + LX - 1) / LX, gy = (c.Y + WY - 1) / WY, gz = (c.Z + 7) / 8;
        const int nv = gx * gy * gz;
        const int per = (nv + (int)gridDim.x - 1) / (int)gridDim.x;
        const int lx = lane % LX, ly = (int)(threadIdx.x >> 6) * LY + lane / LX;
        /* (the virtual workgroup's coordinates by carry, not by three divisions per trip: the scalar unit is shared by the four
         * SIMDs of a compute unit, and round 5's counters had this sweep issue 108 M scalar against 190 M vector instructions) */
        /* Workgroup ids are dealt round-robin over the eight XCDs, and a run of virtual workgroups is a fixed stretch of an x row: with
         * the identity mapping the runs that hold the volume's -x / +x faces (every fourth run each) all land on XCDs 0, 4 / 3, 7 —
         * with most tiles lazy those are the runs with work in them, and four XCDs swept while four had finished (round 6,
         * per-workgroup stamps: 0.16 against 0.30 ms of the launch).  The runs of eight consecutive workgroups are rotated by
         * the group's number, so that every XCD gets every eighth run of each kind. */
        const int wg = ((int)gridDim.x & 7) ? (int)blockIdx.x : (((int)blockIdx.x & ~7) | (((int)blockIdx.x + ((int)blockIdx.x >> 3)) & 7));
        int v = wg * per;
        const int vend = min(nv, (wg + 1) * per);
        int vx = v % gx, vy = (v / gx) % gy, vz = v / (gx * gy);
        /* four virtual workgroups per trip (the default grid gives a workgroup exactly four): their flags first, together */
        for (; v < vend; v += 4) {
            int qx[4], qy[4], qz[4]; gie_markc_pre fl[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                qx[j] = vx; qy[j] = vy; qz[j] = v + j < vend ? vz : gz;       /* (past the run's end: a slab outside the volume) */
                if (++vx == gx) { vx = 0; if (++vy == gy) { vy = 0; vz++; } }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) fl[j] = gie_markc_flags(c, qx[j] * LX + lx, qy[j] * WY + ly, qz[j] * 8);
#pragma unroll
            for (int j = 0; j < 4; j++) gie_markc_column_fast(c, qx[j] * LX + lx, qy[j] * WY + ly, qz[j] * 8, fl[j]);
        }
    }
}

/* ------------------------------------------------------------------ placement probe */
/* The memory pattern of the dense Mark + commit sweep on a mapper's own planes, without its arithmetic: a z-column of eight voxels
 * per thread, 32 lanes along x by 2 along y; reads the type and the batch obstacle, writes the pair and — at the address the voxel
 * would have in a pool filled in tile order — the stored obstacle.  How long it takes depends on where the planes lie in physical
 * memory (DESIGN.md 4 "placement"), which is what gie_create uses it for; it leaves garbage in `pair` and `g_coc` (the caller clears
 * the former, the latter is initialised when a block is handed out). */
__global__ __launch_bounds__(256) void k_place_probe(const gie_ctx c, const int nslot, const int streams)
{
    constexpr int LX = 32, LY = 2, WY = 4 * LY;
    const int lane = threadIdx.x & 63;
    const int gx = (c.X + LX - 1) / LX, gy = (c.Y + WY - 1) / WY, gz = (c.Z + 7) / 8;
    const int nv = gx * gy * gz;
    const int per = (nv + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lx = lane % LX, ly = (int)(threadIdx.x >> 6) * LY + lane / LX;
    const size_t plane = (size_t)c.X * c.Y;
    for (int v = blockIdx.x * per; v < nv && v < (int)(blockIdx.x + 1) * per; v++) {
        const int xg = v % gx;
        const int x = xg * LX + lx, y = ((v / gx) % gy) * WY + ly, z0 = (v / (gx * gy)) * 8;
        if (x >= c.X || y >= c.Y) continue;
        const size_t id0 = ((size_t)z0 * c.Y + y) * c.X + x;
        const int nz = min(8, c.Z - z0);
        int8_t ty[8]; uint32_t bc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const size_t id = id0 + (size_t)(k < nz ? k : 0) * plane; ty[k] = (streams & 1) ? c.glb_type[id] : (int8_t)k; bc[k] = (streams & 2) ? c.bcoc[id] : (uint32_t)x; }
        const int t = gie_tile_index(c, x, y, z0);
        const gie_vaddr base = (gie_vaddr)(t % nslot) * GIE_VBSZ + ((x & 7) | ((y & 7) << 3));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k >= nz) break;
            const uint64_t pr = ((uint64_t)bc[k] << 20) | (uint64_t)(uint8_t)ty[k];
            if (streams & 4) c.pair[id0 + (size_t)k * plane] = pr;
            if (streams & 8) c.g_coc[base + (k << 6)] = pr ^ 0x5555ull;
            if (!(streams & 12) && pr == 0x123456789abcull) c.pair[id0] = pr;       /* (keeps the loads alive) */
        }
    }
}

/* ------------------------------------------------------------------ obtainFrontiers: tiles out of LDS, faces one voxel per lane */
/* obtainFrontiers (unify_helper.cuh:275-446).  The per-voxel decisions are gie_frontier_finish_nb's (gie_ops.h); what the two
 * kernels here change is where their inputs come from and who looks at which voxel (the thread-per-column form walked a
 * column's eight voxels one after the other, each a chain of two or three dependent memory round trips — five to seven for a
 * voxel on a face of the volume: 0.32 ms for 0.4 GB of traffic on the C5 workload).
 *   k_frontier_tiles  voxels OFF the faces of the volume, over the listed tiles that have something to look at (tsum == 1): a
 *                     WAVE takes a tile, ONE batch of loads brings the tile's types and Mark-time pairs and those of the
 *                     one-voxel halo around it into LDS, every voxel's six neighbour types / pairs are LDS reads;
 *   k_frontier_faces  voxels ON a face — the ones that look at hashed voxels outside the volume (gie_frontier_outside: a chain
 *                     of dependent round trips, latency-bound) — one per lane, a wave = an 8x8 patch of a face (one tile: the
 *                     tile summary is wave-uniform), no LDS staging, as many waves in flight as the registers allow; a
 *                     voxel on several faces goes with the first of them.
 * Seeds are collected per workgroup / wave in LDS and appended with one atomic each.  The order of the voxels is irrelevant:
 * a voxel's decision reads only Mark-time state (pairs, tile flags), types up to the FREE / FNT distinction it does not look
 * at, and the one outside voxel across its own face. */
#define GIE_FR_CQ 256           /* C seeds a wave collects before it appends them with one atomic */
struct gie_fr_tile { uint64_t pair[512], hpair[6][64]; int32_t cq[GIE_FR_CQ]; uint8_t typ[512], htyp[6][64]; };        /* 8.9 KB per wave */
struct gie_nbpair_lds {
    const gie_fr_tile *L; int ex, ey, ez;
    __device__ __forceinline__ uint64_t operator()(int k, int) const {
        const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
        const int ux = ex + dx[k], uy = ey + dy[k], uz = ez + dz[k];
        const bool inside = (unsigned)ux < 8u && (unsigned)uy < 8u && (unsigned)uz < 8u;
        const int hp = (k < 2) ? (ey + 8 * ez) : ((k < 4) ? (ex + 8 * ez) : (ex + 8 * ey));
        return inside ? L->pair[ux + 8 * uy + 64 * uz] : L->hpair[k][hp];
    }
};
struct gie_absink_none { static constexpr bool outside = false; __device__ __forceinline__ void ab(const gie_ctx &, int, uint64_t, int) const {} };   /* (a voxel off the faces has no outside neighbour) */
/* append the wave's collected C seeds: one atomic */
__device__ __forceinline__ void gie_frontier_flush_c(const gie_ctx &c, const int32_t *cq, int &ncq, const int lane)
{
    if (ncq == 0) return;                                     /* wave-uniform */
    gie_wave_sync();
    int base = 0;
    if (lane == 0) base = atomicAdd(&c.cnt[GIE_CNT_C], ncq);
    base = __shfl(base, 0);
    for (int e = lane; e < ncq; e += 64) {
        if (base + e < c.qcap_c) c.qc[0][base + e] = cq[e];
        else atomicOr(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
    }
    gie_wave_sync();
    ncq = 0;
}

/* 16 when the closest obstacle of pair `pr` lies outside the local volume and inside the wave range: only such a voxel can
 * make a neighbour a seed of wave C (gie_frontier_finish_nb tests the same again on the pair itself) */
#define GIE_FR_SRC 16u
__device__ __forceinline__ uint32_t gie_fr_srcbit(const gie_ctx &c, const uint64_t pr)
{
    int nw[3];
    gie_unpack_wr(gie_pair_par(pr), &nw[0], &nw[1], &nw[2]);
    const int n0 = nw[0] + c.upvt[0] - c.pvt[0], n1 = nw[1] + c.upvt[1] - c.pvt[1], n2 = nw[2] + c.upvt[2] - c.pvt[2];
    return (!gie_in_loc(c, n0, n1, n2) && gie_in_wr(c, nw[0], nw[1], nw[2])) ? GIE_FR_SRC : 0u;
}
__device__ __forceinline__ void gie_frontier_tile(const gie_ctx &c, gie_fr_tile &L, const int te, const int lane, int &ncq)
{
    const int t = te & 0x7fffffff;                             /* (bit 31 of a list entry: the tile is lazy, op_tile_summary) */
    const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
    const int x0 = tx * 8, y0 = ty * 8, z0 = tz * 8;
    const int lx = lane & 7, ly = lane >> 3;
    const int x = x0 + lx, y = y0 + ly;
    const bool colin = x < c.X && y < c.Y;
    const size_t plane = (size_t)c.X * c.Y;
    const size_t col = (size_t)y * c.X + x;
    /* ---- one batch of loads: types and Mark-time pairs of the tile and of the one-voxel halo around it */
    uint64_t pv[8], hv[6];
    int8_t tv[8], ht[6];
    const bool lz = te < 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const bool in = colin && z0 + j < c.Z;
        const size_t id = in ? (size_t)(z0 + j) * plane + col : 0;
        tv[j] = c.glb_type[id];
        pv[j] = lz ? gie_pair_of_bcoc(c, c.bcoc_lazy[id], x, y, z0 + j) : c.pair[id];      /* ("lazy pairs": wave-uniform) */
        if (!in) tv[j] = -1;
    }
    {   /* halo face f (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z), the lane's position (a, b) on it */
        const int a = lane & 7, b = lane >> 3;
        const int hx[6] = { x0 - 1, x0 + 8, x0 + a, x0 + a, x0 + a, x0 + a };
        const int hy[6] = { y0 + a, y0 + a, y0 - 1, y0 + 8, y0 + b, y0 + b };
        const int hz[6] = { z0 + b, z0 + b, z0 + b, z0 + b, z0 - 1, z0 + 8 };
#pragma unroll
        for (int f = 0; f < 6; f++) {
            const bool in = gie_in_loc(c, hx[f], hy[f], hz[f]);
            const int id = in ? gie_lid(c, hx[f], hy[f], hz[f]) : 0;
            ht[f] = c.glb_type[id];
            hv[f] = gie_pair_get(c, id, in ? hx[f] : 0, in ? hy[f] : 0, in ? hz[f] : 0);      /* (the neighbour tile's flag: the same byte for the whole face) */
            if (!in) ht[f] = -1;
        }
    }
    /* staged type byte: type | GIE_FR_SRC; a position outside the volume (only ever next to a voxel on a face, which is not
     * looked at here): UNKNOWN, no source */
#pragma unroll
    for (int j = 0; j < 8; j++) {
        L.pair[lane + 64 * j] = pv[j];
        L.typ[lane + 64 * j] = tv[j] < 0 ? (uint8_t)GIE_VOX_UNKNOWN : (uint8_t)((uint32_t)(tv[j] & 15) | gie_fr_srcbit(c, pv[j]));
    }
#pragma unroll
    for (int f = 0; f < 6; f++) {
        L.hpair[f][lane] = hv[f];
        L.htyp[f][lane] = ht[f] < 0 ? (uint8_t)GIE_VOX_UNKNOWN : (uint8_t)((uint32_t)(ht[f] & 15) | gie_fr_srcbit(c, hv[f]));
    }
    gie_wave_sync();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool xyface = x == 0 || x == c.X - 1 || y == 0 || y == c.Y - 1;
    /* ---- the lane's column, voxels off the faces.  A voxel acts only next to a source (C seed) or, FREE itself, next to an
     * unknown voxel (FNT): a trip none of whose voxels does is skipped, the others run the reference's decisions in full. */
#pragma unroll 1
    for (int ez = 0; ez < 8; ez++) {
        const int vz = z0 + ez;
        const bool go = colin && vz < c.Z && !xyface && vz != 0 && vz != c.Z - 1;
        const int v = lane + 64 * ez;
        const uint32_t own = L.typ[v];
        uint32_t nn[6], src = 0;
        bool unk = false;
        const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int ux = lx + dx[k], uy = ly + dy[k], uz = ez + dz[k];
            const bool inside = (unsigned)ux < 8u && (unsigned)uy < 8u && (unsigned)uz < 8u;
            const int hp = (k < 2) ? (ly + 8 * ez) : ((k < 4) ? (lx + 8 * ez) : (lx + 8 * ly));
            nn[k] = inside ? L.typ[ux + 8 * uy + 64 * uz] : L.htyp[k][hp];
            src |= ((nn[k] >> 4) & 1u) << k;
            unk |= (nn[k] & 15u) == (uint32_t)GIE_VOX_UNKNOWN;
        }
        const bool need = go && (own & 15u) != (uint32_t)GIE_VOX_UNKNOWN && (src != 0u || ((own & 15u) == (uint32_t)GIE_VOX_FREE && unk));
        if (__ballot(need) == 0ull) continue;                 /* wave-uniform */
        int push = 0, id = 0;
        if (need) {
            id = gie_lid(c, x, y, vz);
            gie_frontier_st s;
            s.p0 = L.pair[v];
            s.tys = own & 15u;
            s.tfm = src;                                       /* (the reference path gates the neighbour's pair by its TILE's flag: a superset) */
#pragma unroll
            for (int k = 0; k < 6; k++) s.tys |= (nn[k] & 15u) << (4 * (k + 1));
            const gie_nbpair_lds nb = { &L, lx, ly, ez };
            push = gie_frontier_finish_nb(c, id, x, y, vz, s, nb, gie_absink_none());
        }
        const unsigned long long m = __ballot(push);         /* C seeds of the trip join the wave's list */
        if (m) {
            if (push) L.cq[ncq + __popcll(m & lt)] = id;
            ncq += __popcll(m);
            if (ncq > GIE_FR_CQ - 64) gie_frontier_flush_c(c, L.cq, ncq, lane);
        }
    }
    gie_wave_sync();                                            /* the LDS block is reused for the wave's next tile */
}
#define GIE_FR_WAVES 4
__global__ __launch_bounds__(64 * GIE_FR_WAVES) void k_frontier_tiles(const gie_ctx c, const int32_t *list, const int count_idx)
{
    __shared__ gie_fr_tile s_tiles[GIE_FR_WAVES];
    const int n = c.cnt[count_idx];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = gridDim.x * GIE_FR_WAVES;
    int ncq = 0;
    int e = blockIdx.x * GIE_FR_WAVES + wave;
    int t = e < n ? list[e] : 0;
    while (e < n) {
        const int tn = e + stride < n ? list[e + stride] : 0;   /* the next entry travels with this tile's batch */
        gie_frontier_tile(c, s_tiles[wave], t, lane, ncq);
        t = tn; e += stride;
    }
    gie_frontier_flush_c(c, s_tiles[wave].cq, ncq, lane);
}

/* the seeds of waves A / B the face voxels of a workgroup produce, collected in LDS (bit 63 of the coordinate: frontier A) */
#define GIE_FR_ABIT ((uint64_t)1 << 63)
#ifndef GIE_FF_WAVES
#define GIE_FF_WAVES 8                                      /* (4 -> 8 in round 4: half as many same-address appends to the three seed queues, 0.157 -> 0.149 ms; 16: no gain) */
#endif
#define GIE_FF_AB (64 * GIE_FF_WAVES * 3)                     /* three outside neighbours per voxel: a corner of the volume */
struct gie_ff_wg { uint64_t ab_crd[GIE_FF_AB]; gie_vaddr ab_addr[GIE_FF_AB]; int32_t cq[64 * GIE_FF_WAVES]; int32_t nab, ncq, base[3], pad_; };   /* 10.3 KB */
struct gie_absink_lds {
    static constexpr bool outside = true;
    gie_ff_wg *L;
    __device__ __forceinline__ void ab(const gie_ctx &c, int push, uint64_t crd, gie_vaddr a) const {
        if (!push) return;
        const int i = __hip_atomic_fetch_add(&L->nab, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (i < GIE_FF_AB) { L->ab_crd[i] = crd | (push == 2 ? GIE_FR_ABIT : (uint64_t)0); L->ab_addr[i] = (gie_vaddr)a; }
        else if (push == 2) gie_push64a(c, c.qa, c.qa_a, &c.cnt[GIE_CNT_A], c.qcap_ab, crd, a);    /* (volumes one or two voxels thick: more than three outside neighbours per voxel) */
        else gie_push64a(c, c.qb, c.qb_a, &c.cnt[GIE_CNT_B], c.qcap_ab, crd, a);
    }
};
/* patches of 8x8 voxels per face: face f (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z) spans (Y, Z), (X, Z) or (X, Y) */
struct gie_face_patches { int off[7], na[6]; };
/* The first `nsum` workgroups of the launch build the tile summary and the list k_frontier_tiles walks (op_tile_summary over
 * the known tiles) — the face voxels do not wait for it: a patch is one tile, and its summary value is a dozen wave-uniform
 * flag reads (op_tile_summary::value) — so the summary costs no launch of its own and runs under the face voxels' chains. */
__global__ __launch_bounds__(64 * GIE_FF_WAVES) void k_frontier_faces(const gie_ctx c, const gie_face_patches fp, const int32_t *known, const int known_idx, const int nsum)
{
    if ((int)blockIdx.x < nsum) {
        const int n = c.cnt[known_idx];
        const op_tile_summary f;
        for (int e0 = blockIdx.x * blockDim.x; e0 < n; e0 += nsum * blockDim.x) {     /* whole workgroups call (block barriers inside) */
            const int e = e0 + (int)threadIdx.x;
            f(c, e < n ? known[e] : -1);
        }
        return;
    }
    __shared__ gie_ff_wg W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { W.nab = 0; W.ncq = 0; }
    __syncthreads();
    const int p = ((int)blockIdx.x - nsum) * GIE_FF_WAVES + wave;             /* the wave's patch */
    int push = 0, id = 0;
    if (p < fp.off[6]) {
        int f = 0;
#pragma unroll
        for (int k = 1; k < 6; k++) if (p >= fp.off[k]) f = k;
        const int q = p - fp.off[f], pa = q % fp.na[f], pb = q / fp.na[f];
        const int a = pa * 8 + (lane & 7), b = pb * 8 + (lane >> 3);
        const int fix = (f & 1) ? ((f == 1) ? c.X - 1 : (f == 3) ? c.Y - 1 : c.Z - 1) : 0;
        const int vx = (f < 2) ? fix : a, vy = (f < 2) ? a : ((f < 4) ? fix : b), vz = (f < 4) ? b : fix;
        bool go = vx < c.X && vy < c.Y && vz < c.Z;
        const bool fx0 = vx == 0, fx1 = vx == c.X - 1, fy0 = vy == 0, fy1 = vy == c.Y - 1, fz0 = vz == 0;
        /* on an earlier face: that face's patch has it */
        if ((f >= 1 && fx0) || (f >= 2 && fx1) || (f >= 3 && fy0) || (f >= 4 && fy1) || (f >= 5 && fz0)) go = false;
        if (go && op_tile_summary::value(c, gie_tile_index(c, vx, vy, vz)) == 0) go = false;         /* (one tile per patch: wave-uniform) */
        if (go) {
            id = gie_lid(c, vx, vy, vz);
            /* the voxel's own loads and the lookup of the neighbour across the patch's face go out together, that neighbour's record
             * right behind (round 5: the face voxels were six dependent round trips each — 92 us for 1.5 M of them) */
            const int ox = vx + ((f == 0) ? -1 : (f == 1) ? 1 : 0), oy = vy + ((f == 2) ? -1 : (f == 3) ? 1 : 0), oz = vz + ((f == 4) ? -1 : (f == 5) ? 1 : 0);
            gie_out_pre pre;
            pre.k = f;
            pre.a = gie_gvox_tab(c, ox + c.pvt[0], oy + c.pvt[1], oz + c.pvt[2]);
            gie_frontier_st s;
            /* (a lazy tile never touches a face of the WHOLE volume, and one tile of several has no lazy tiles: the face voxel's pair and its
             * neighbours' are in the plane — no flag to wait for) */
            gie_frontier_load1<true>(c, id, vx, vy, vz, s);
            pre.nty = pre.a >= 0 ? c.g_type[pre.a] : (int8_t)0;
            pre.ncoc = pre.a >= 0 ? c.g_coc[pre.a] : (uint64_t)0;
            const gie_nbpair_plane nb = { c.pair };
            const gie_absink_lds sink = { &W };
            push = gie_frontier_finish_nb(c, id, vx, vy, vz, s, nb, sink, &pre);      /* (an UNKNOWN voxel: nothing, as before) */
        }
    }
    if (push) W.cq[__hip_atomic_fetch_add(&W.ncq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = id;
    __syncthreads();
    /* ---- one atomic per list and workgroup */
    const int nab = min(W.nab, GIE_FF_AB), ncq = W.ncq;
    if (nab == 0 && ncq == 0) return;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (wave == 0) {
        /* entries of frontier A first, then B: the ranks come from two passes of ballots over the list */
        int na = 0;
        for (int e0 = 0; e0 < nab; e0 += 64) na += __popcll(__ballot(e0 + lane < nab && (W.ab_crd[e0 + lane] & GIE_FR_ABIT) != 0));
        if (lane < 3) {                                          /* the three appends in flight together */
            const int cnt = (lane == 0) ? na : (lane == 1) ? nab - na : ncq;
            int32_t *const ctr = &c.cnt[(lane == 0) ? GIE_CNT_A : (lane == 1) ? GIE_CNT_B : GIE_CNT_C];
            W.base[lane] = cnt > 0 ? gie_aadd32(ctr, cnt) : 0;
        }
        gie_wave_sync();
        int ra = W.base[0], rb = W.base[1];
        for (int e0 = 0; e0 < nab; e0 += 64) {
            const int e = e0 + lane;
            const bool have = e < nab;
            const uint64_t crd = have ? W.ab_crd[e] : (uint64_t)0;
            const bool isa = have && (crd & GIE_FR_ABIT) != 0, isb = have && !isa;
            const unsigned long long ma = __ballot(isa), mb = __ballot(isb);
            if (isa) {
                const int i = ra + __popcll(ma & lt);
                if (i < c.qcap_ab) { gie_st(&c.qa[i], (uint64_t)(crd & ~GIE_FR_ABIT)); gie_st(&c.qa_a[i], W.ab_addr[e]); } else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
            }
            if (isb) {
                const int i = rb + __popcll(mb & lt);
                if (i < c.qcap_ab) { gie_st(&c.qb[i], crd); gie_st(&c.qb_a[i], W.ab_addr[e]); } else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
            }
            ra += __popcll(ma); rb += __popcll(mb);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ncq; e += 64 * GIE_FF_WAVES) {
        const int i = W.base[2] + e;
        if (i < c.qcap_c) c.qc[0][i] = W.cq[e];
        else atomicOr(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
    }
}

/* ------------------------------------------------------------------ block-row sweeps */
/* The sweeps that touch the global map (fuse, Mark + commit) in their dense form.  A thread owns one
 * ROW OF A GLOBAL BLOCK — the 8 voxels (8 bx .. 8 bx + 7, gy, gz) — for the eight z layers of the block,
 * a wave owns eight blocks along x times the eight rows of one block row along y:
 *     lane = (bx & 7) + 8 (gy & 7)
 * so that per wave instruction
 *   - a global block plane (in-block index x | y<<3 | z<<6) is read as 8 blocks x 64 CONSECUTIVE voxels
 *     (one z layer of each block: 512 B of an 8-byte field per block, whole lines), and
 *   - a local x-fastest plane is read as 8 rows x 64 consecutive voxels,
 * each lane moving its 8 voxels of a field as 16-byte vectors (one 8-byte word for the byte planes).
 * The volume's pivot need not be block aligned: the local side of a row starts at any x (the hardware
 * takes unaligned vectors), rows cut by a face of the volume fall back to per-voxel accesses.  The
 * thread-per-z-column sweeps (k_voxa) this replaces moved 8-voxel pieces of eight blocks per
 * instruction with one 1/4/8-byte access per voxel. */
typedef uint64_t __attribute__((aligned(1))) gie_u64u;
typedef uint32_t gie_v4 __attribute__((ext_vector_type(4)));
typedef gie_v4 __attribute__((aligned(4))) gie_v4u;                 /* 16 bytes at any 4-byte boundary */
__device__ __forceinline__ uint4 gie_ld16u(const void *p) { const gie_v4 v = *reinterpret_cast<const gie_v4u *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void gie_st16u(void *p, const uint4 v) { gie_v4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *reinterpret_cast<gie_v4u *>(p) = t; }

struct gie_rows { int b0[3], nb[3], ngx, nvw; };
__device__ __forceinline__ gie_rows gie_rows_of(const gie_ctx &c)
{
    gie_rows r;
    const int sz[3] = { c.X, c.Y, c.Z };
#pragma unroll
    for (int a = 0; a < 3; a++) { r.b0[a] = c.pvt[a] >> 3; r.nb[a] = ((c.pvt[a] + sz[a] - 1) >> 3) - r.b0[a] + 1; }
    r.ngx = (r.nb[0] + 7) >> 3;
    r.nvw = r.ngx * r.nb[1] * r.nb[2];
    return r;
}
/* what one thread of a block-row sweep owns */
struct gie_row {
    int slot;            /* block slot or -1 */
    int x0, ly, lz0;     /* local coordinates of the row's first voxel / first layer (any of them may lie outside the volume) */
    int xlo, xhi;        /* valid voxels of the row: xlo <= i < xhi (empty when the row misses the volume) */
    int arow;            /* in-block offset of the row in layer 0: (gy & 7) << 3 */
    bool full;           /* all eight voxels inside the volume */
};
__device__ __forceinline__ gie_row gie_row_of(const gie_ctx &c, const gie_rows &r, const int v, const int lane)
{
    gie_row w;
    const int gxb = v % r.ngx, by = (v / r.ngx) % r.nb[1], bz = v / (r.ngx * r.nb[1]);
    const int bx = gxb * 8 + (lane & 7);
    const int gy = (r.b0[1] + by) * 8 + (lane >> 3);
    w.ly = gy - c.pvt[1];
    w.x0 = (r.b0[0] + bx) * 8 - c.pvt[0];
    w.lz0 = (r.b0[2] + bz) * 8 - c.pvt[2];
    w.xlo = max(0, -w.x0); w.xhi = min(8, c.X - w.x0);
    if (bx >= r.nb[0] || (unsigned)w.ly >= (unsigned)c.Y) w.xhi = w.xlo = 0;
    w.full = (w.xlo == 0 && w.xhi == 8);
    w.arow = (gy & 7) << 3;
    w.slot = -1;
    if (w.xhi > w.xlo) {
        const int ti = __mul24(__mul24(r.b0[2] + bz - c.tb0[2], c.tdim[1]) + (r.b0[1] + by - c.tb0[1]), c.tdim[0]) + (r.b0[0] + bx - c.tb0[0]);
        w.slot = c.blk_tab[ti];
    }
    return w;
}
/* the (at most four) local 8x8x8 tiles a thread's 8 x 8 (x, z) voxels lie in: voxel (i, k) belongs to
 * quadrant (i >= ix) + 2 (k >= kz); tile index -1 = that quadrant holds no voxel of the volume */
struct gie_row_tiles { int ix, kz; int t[4]; };
__device__ __forceinline__ gie_row_tiles gie_row_tiles_of(const gie_ctx &c, const gie_row &w)
{
    gie_row_tiles q;
    q.ix = 8 - (w.x0 & 7); q.kz = 8 - (w.lz0 & 7);        /* 8 when aligned: everything in quadrant 0 */
    const int tx0 = w.x0 >> 3, tz0 = w.lz0 >> 3, ty = w.ly >> 3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int tx = tx0 + (j & 1), tz = tz0 + (j >> 1);
        const bool used = ((j & 1) ? q.ix < 8 : true) && ((j >> 1) ? q.kz < 8 : true);
        const bool in = used && (unsigned)tx < (unsigned)c.tfd[0] && (unsigned)tz < (unsigned)c.tfd[2] && w.xhi > w.xlo;
        q.t[j] = in ? __mul24(__mul24(tz, c.tfd[1]) + ty, c.tfd[0]) + tx : -1;
    }
    return q;
}
/* 8 bytes of a local byte plane starting at element `id` (any alignment); a row cut by the volume reads its valid bytes one by one */
__device__ __forceinline__ uint64_t gie_row_ld8(const int8_t *p, const long long id, const gie_row &w)
{
    if (w.full) return *reinterpret_cast<const gie_u64u *>(p + id);
    uint64_t v = 0;
    for (int i = w.xlo; i < w.xhi; i++) v |= (uint64_t)(uint8_t)p[id + i] << (8 * i);
    return v;
}
__device__ __forceinline__ void gie_row_st8(int8_t *p, const long long id, const gie_row &w, const uint64_t v)
{
    if (w.full) { *reinterpret_cast<gie_u64u *>(p + id) = v; return; }
    for (int i = w.xlo; i < w.xhi; i++) p[id + i] = (int8_t)(v >> (8 * i));
}
/* 8 dwords of a local 4-byte plane */
__device__ __forceinline__ void gie_row_ld32(const uint32_t *p, const long long id, const gie_row &w, uint32_t (&v)[8])
{
    if (w.full) {
        const uint4 a = gie_ld16u(p + id), b = gie_ld16u(p + id + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (i >= w.xlo && i < w.xhi) ? p[id + i] : 0u;
}

/* updateHashOGMWithPntCld / updateHashOGMWithSensor (unify_helper.cuh:35-197), dense form: the per-voxel
 * decisions are gie_fuse_finish's (shared gie_fuse_logic), the per-tile known / unknown summaries and
 * the plane flags come out the same */
/* (few tiles to look at: the list form — a wave per listed tile, the staged per-voxel functor as in k_voxa — in the same launch) */
/* (four workgroups per compute unit: with 64-bit block addresses the kernel wanted 130 registers and lost a quarter of its waves) */
/* (PNT: the scan is a ray-cast one — hit / miss counters instead of labels; a template parameter so that the label form does not carry the eight counters of a row in registers) */
template <bool PNT>
__global__ __launch_bounds__(256, PNT ? 3 : 4) void k_fuse_rows(const gie_ctx c, const op_fuse f, const int32_t *list)
{
    {
        const int n = c.cnt[GIE_CNT_TL_FUSE];
        if (gie_use_lists(c, n)) {
            const int lane = threadIdx.x & 63, waves = gridDim.x * 4;
            for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += waves) {
                const int t = list[e];
                const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
                gie_vox_column<op_fuse, true>(c, f, tx * 8 + (lane & 7), ty * 8 + (lane >> 3), tz * 8);
            }
            return;
        }
    }
    const gie_rows r = gie_rows_of(c);
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per = (r.nvw + nw - 1) / nw;                         /* a contiguous run of virtual waves per wave */
    for (int v = wid * per; v < r.nvw && v < (wid + 1) * per; v++) {
        const gie_row w = gie_row_of(c, r, v, lane);
        const gie_row_tiles q = gie_row_tiles_of(c, w);
        const bool any = w.xhi > w.xlo;
        /* nothing stored for these voxels and nothing left from earlier frames: all unknown */
        bool stale = false;
#pragma unroll
        for (int j = 0; j < 4; j++) if (q.t[j] >= 0 && c.tknown_prev[q.t[j]]) stale = true;
        const bool idle = !any || (w.slot < 0 && !stale);
        if (any && idle) {
#pragma unroll
            for (int j = 0; j < 4; j++) if (q.t[j] >= 0) c.tunk[q.t[j]] = 1;
        }
        unsigned kn = 0u, un = 0u;                                /* bit = quadrant */
        const long long plane = (long long)c.X * c.Y;
        const long long id0 = ((long long)w.lz0 * c.Y + w.ly) * c.X + w.x0;
        /* the row's place in its block's planes: one 64-bit address per row, 32-bit offsets per layer */
        uint8_t *const p_occ = c.g_occ + (w.slot < 0 ? 0 : (gie_vaddr)w.slot * GIE_VBSZ + w.arow);
        int8_t *const p_typ = c.g_type + (w.slot < 0 ? 0 : (gie_vaddr)w.slot * GIE_VBSZ + w.arow);
#pragma unroll 2
        for (int k = 0; k < 8; k++) {
            const int lz = w.lz0 + k;
            const bool live = !idle && (unsigned)lz < (unsigned)c.Z;
            bool occ_here = false;
            if (live) {
                const long long id = id0 + (long long)k * plane;
                const uint64_t it8 = gie_row_ld8(c.scan_labels ? c.scan_labels : c.inst_type, id, w);
                const uint64_t gt8 = gie_row_ld8(c.glb_type, id, w);
                uint32_t rc[8];
                if (PNT) gie_row_ld32(reinterpret_cast<const uint32_t *>(c.ray_count), id, w, rc);
                const int a = k << 6;
                uint64_t go8 = 0, gy8 = 0;
                if (w.slot >= 0) { go8 = *reinterpret_cast<const uint64_t *>(p_occ + a); gy8 = *reinterpret_cast<const uint64_t *>(p_typ + a); }
                uint64_t no8 = go8, ny8 = gy8, ng8 = gt8;
                bool rc_any = false;
                const int qz = (k >= q.kz) ? 2 : 0;
                /* Labelled scans (the projective sensors, gie_ogm_labels): the eight voxels of the row as ONE 64-bit word, byte-parallel
                 * (gie_fuse_row8_labels, gie_ops.h).  The per-voxel loop below costs ~55 vector instructions per voxel and a wave runs
                 * it whenever any of its 512 voxels needs it. */
                const uint64_t ONES = 0x0101010101010101ull, L7 = 0x7f7f7f7f7f7f7f7full, H8 = 0x8080808080808080ull;
                if (!PNT && w.full && w.slot >= 0 && c.nbox <= 0 && c.occ_thresh >= 127) {
                    gie_fuse_row8_labels(c.occ_thresh, it8, go8, gy8, &no8, &ny8);
                    ng8 = ny8;
                    const uint32_t km = (uint32_t)((((ny8 | (ny8 >> 1)) & ONES) * 0x0102040810204080ull) >> 56);      /* bit i: voxel i is known */
                    const uint32_t lo = (1u << q.ix) - 1u;                                                           /* (ix = 8: the whole row) */
                    if (km & lo) kn |= 1u << qz;
                    if (~km & lo & 0xffu) un |= 1u << qz;
                    if (km & ~lo & 0xffu) kn |= 1u << (qz + 1);
                    if (~km & ~lo & 0xffu) un |= 1u << (qz + 1);
                    const uint64_t t2 = ny8 ^ (2ull * ONES);                                                          /* a byte of t2 is zero where the type is OCCUPIED */
                    occ_here = ((((t2 & L7) + L7) | t2) & H8) != H8;
                } else
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (i < w.xlo || i >= w.xhi) continue;
                    const int count = PNT ? (int)rc[i] : 0;
                    rc_any |= count != 0;
                    const int8_t nt = (int8_t)(it8 >> (8 * i));
                    int8_t ty = GIE_VOX_UNKNOWN;
                    if (w.slot >= 0) {
                        uint8_t occ = (uint8_t)(go8 >> (8 * i));
                        ty = (int8_t)(gy8 >> (8 * i));
                        const int occ_flag = c.nbox > 0 ? gie_fuse_occ_flag(c, w.x0 + i + c.pvt[0], w.ly + c.pvt[1], lz + c.pvt[2]) : 0;
                        gie_fuse_logic(c, count, nt, occ_flag, &occ, &ty);
                        no8 = (no8 & ~(0xffull << (8 * i))) | ((uint64_t)occ << (8 * i));
                        ny8 = (ny8 & ~(0xffull << (8 * i))) | ((uint64_t)(uint8_t)ty << (8 * i));
                    }
                    ng8 = (ng8 & ~(0xffull << (8 * i))) | ((uint64_t)(uint8_t)ty << (8 * i));
                    occ_here |= ty == GIE_VOX_OCCUPIED;
                    const int qd = qz + ((i >= q.ix) ? 1 : 0);
                    if (ty != GIE_VOX_UNKNOWN) kn |= 1u << qd; else un |= 1u << qd;
                }
                /* write only what changes */
                if (rc_any) {
                    if (w.full) { gie_st16u(c.ray_count + id, make_uint4(0, 0, 0, 0)); gie_st16u(c.ray_count + id + 4, make_uint4(0, 0, 0, 0)); }
                    else for (int i = w.xlo; i < w.xhi; i++) c.ray_count[id + i] = 0;
                }
                if (it8 != 0ull && !c.scan_labels) gie_row_st8(c.inst_type, id, w, 0ull);
                if (ng8 != gt8) gie_row_st8(c.glb_type, id, w, ng8);
                if (w.slot >= 0) {
                    if (no8 != go8) *reinterpret_cast<uint64_t *>(p_occ + a) = no8;        /* bytes of voxels outside the volume keep their value */
                    if (ny8 != gy8) { *reinterpret_cast<uint64_t *>(p_typ + a) = ny8; if (c.track) c.g_dirty[w.slot] = 1; }
                }
            }
            if (__ballot(occ_here) != 0ull && lane == __ffsll((long long)__ballot(occ_here)) - 1) c.zocc[lz] = 1;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (q.t[j] < 0) continue;
            if ((kn >> j) & 1u) c.tknown[q.t[j]] = 1;
            if ((un >> j) & 1u) c.tunk[q.t[j]] = 1;
        }
    }
}

/* pass Z for the tiles on tl_known, straight from the pass-X planes: lane = (x,y) column of the
 * tile, loop over the planes with obstacles, eight running minima (the tile's z range).  Ties go
 * to the smaller z like the envelope (strict '<' while z ascends).  Cheaper than the column
 * kernel while the known tiles are few: K reads per column instead of a whole column's work. */
__device__ __forceinline__ void gie_edt_z_direct_body(const gie_ctx &c)
{
    const int n = c.cnt[GIE_CNT_TL_KNOWN];
    if (!gie_z_use_lists(c, n)) return;                     /* many known tiles: the column kernel (launched next to this one) does the pass */
    /* workgroup = known tile; its four waves split the planes with obstacles between them (the
     * pass is a chain of dependent plane reads per tile: a quarter of the chain each), merge their
     * minima through LDS, and each wave finishes two of the tile's eight z */
    __shared__ uint32_t s_best[4][8][64];
    __shared__ uint16_t s_zl[1024];                        /* the planes with obstacles (sides are <= 1024): read from LDS, not through a dependent load per trip */
    const int K = *c.zcount;
    for (int j = threadIdx.x; j < K; j += 256) s_zl[j] = c.zlist[j];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t plane = (size_t)c.X * c.Y;
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        const int t = c.tl_known[e];
        const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
        const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3), z0 = tz * 8;
        const bool ok = x < c.X && y < c.Y;
        const size_t o = ok ? (size_t)y * c.X + x : 0;
        /* key = (dist² << 10) | site rank: one v_min per voxel and site, ties go to the smaller rank =
         * smaller z (dist² < 2^22 is gie_create's envelope limit); four plane reads in flight per trip */
        uint32_t best[8];
#pragma unroll
        for (int k = 0; k < 8; k++) best[k] = 0xffffffffu;
        /* The planes are taken outwards from the tile — first the ones at or above its lowest layer (rank js, js + 1, ...),
         * then the ones below (js - 1, js - 2, ...), every wave each fourth plane of a side — and a side ends where the
         * plane's distance to the tile alone (dz², nearest layer) exceeds every minimum this wave holds: such a plane can
         * neither win nor tie, and the planes behind it are farther still.  (The wave's own minima bound the merged ones
         * from above, so the test is conservative.)  A tile near obstacles reads a handful of planes instead of all K. */
        int js = 0;
        for (int lo = 0, hi = K; lo < hi;) { const int mid = (lo + hi) >> 1; if ((int)s_zl[mid] < z0) lo = mid + 1; else hi = mid; js = lo; }
#pragma unroll 1
        for (int side = 0; side < 2; side++) {
            for (int i0 = 0;; i0 += 4) {
                /* this wave's planes of the trip: ranks base +- (4 (i0 + u) + w) */
                int jj[4], zj[4]; uint32_t v[4];
                bool any_plane = false;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = 4 * (i0 + u) + w;
                    const int j = side ? js - 1 - r : js + r;
                    const bool have = side ? j >= 0 : j < K;
                    jj[u] = have ? j : -1;
                    any_plane |= have;
                }
                if (!any_plane) break;                              /* wave-uniform */
                {   /* the nearest plane of the trip is the first one: out of reach for every voxel of the tile? */
                    const int zn = s_zl[jj[0] >= 0 ? jj[0] : 0];
                    const int dzn = side ? z0 - zn : max(zn - (z0 + 7), 0);
                    uint32_t mx = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) mx = max(mx, best[k] >> 10);
                    if (jj[0] < 0 || !__any((uint32_t)(dzn * dzn) <= mx)) break;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) { const int j = jj[u] >= 0 ? jj[u] : jj[0]; zj[u] = s_zl[j]; v[u] = c.cxy2[(size_t)zj[u] * plane + o]; jj[u] = j; }
#pragma unroll
                for (int u = 0; u < 4; u++) {                       /* (a repeated site does not change a minimum) */
                    const int dx = x - (int)(v[u] & 0xffffu), dy = y - (int)(v[u] >> 16);
                    const uint32_t a = (uint32_t)(dx * dx + dy * dy);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int dz = z0 + k - zj[u];
                        best[k] = min(best[k], ((a + (uint32_t)(dz * dz)) << 10) | (uint32_t)jj[u]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) s_best[w][k][lane] = best[k];
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int k = 2 * w + kk;
            const uint32_t bk = min(min(s_best[0][k][lane], s_best[1][k][lane]), min(s_best[2][k][lane], s_best[3][k][lane]));
            uint32_t win = GIE_BCOC_NONE;
            if (K > 0) {
                const int zw = s_zl[bk & 1023u];
                const uint32_t vw = c.cxy2[(size_t)zw * plane + o];
                win = gie_pack_bcoc((int)(vw & 0xffffu), (int)(vw >> 16), zw);
            }
            if (ok && z0 + k < c.Z) c.bcoc[(size_t)(z0 + k) * plane + o] = win;
        }
        __syncthreads();                                   /* s_best is rewritten by the next tile */
    }
}
__global__ __launch_bounds__(256) void k_edt_z_direct(const gie_ctx c) { gie_edt_z_direct_body(c); }


/* ------------------------------------------------------------------ persistent BFS waves */
/* One launch for the three waves, all workgroups co-resident; the rounds of a wave are separated by a grid barrier
 * (monotonic counter, agent-scope release / relaxed poll / acquire — cdna_hip_programming.md G16), so a whole wavefront
 * costs one launch and no host round trip (the reference pays ≈3 PCIe round trips per level, wave_helper.h:20-90).
 * Every shared word touched inside a round goes through agent-scope accesses (gie_ld / gie_st / atomics in gie_ops.h):
 * the per-XCD L2s are not coherent for plain accesses. */
#ifndef GIE_WAVE_THREADS
#define GIE_WAVE_THREADS 512          /* 8 waves: each takes one block / tile at a time out of its own ~19 KB of LDS, with a register budget of 256 (1024 threads: 128,
                                       * and the block routines spilled 127 registers) */
#endif
#define GIE_WAVE_SLOTS(want) ((want) < GIE_WAVE_THREADS / 64 ? (want) : GIE_WAVE_THREADS / 64)
#define GIE_BAR_SPIN_LIMIT (1 << 22)

/* The visit count of a round is a word every block-run of the round would add to: a thousand wavefronts within the same few
 * microseconds, and same-address atomics serialise at their L2 slice (~6-10 ns each; round 4: 15 us of a 278 us launch, found by
 * leaving them out).  They are summed per WORKGROUP in LDS (GIE_VIS_ADD) and handed to the round's word by one lane at the grid
 * barrier.  (The lengths of the next round's lists are such words too.  Tried: per-wavefront chunks of 16 list entries with the
 * unused ones filled with -1 — the holes outnumbered the entries ten to one and the next round read them one dependent load at a
 * time, 0.28 -> 0.39 ms; a per-workgroup LDS queue copied out at the barrier — no holes, but the reservation's round trip sits on
 * every round's critical path and gives back what it saves, 0.270 against 0.263 ms with the visit counts alone.) */
#define GIE_VIS_ADD(p, v) __hip_atomic_fetch_add(gb.s_vis, (int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
struct gie_gridbar { int32_t *word; int epoch; int failed; int nwg; int *s_fail; int *s_vis; int spin_limit; };   /* nwg = workgroups that meet at this barrier */

/* Everything the wave kernels share between workgroups is read and written with agent-scope
 * (write-through, L1-bypassing) accesses — gie_ld / gie_st / atomics — so the barrier needs no
 * cache write-back or invalidate (each costs microseconds per level): every wave drains its own
 * stores, the workgroup meets, one lane arrives on the counter and polls it. */
/* vis_word: where the visits of the round that ends here go */
__device__ __forceinline__ void gie_grid_sync(gie_gridbar &gb, const gie_ctx &c, int32_t *const vis_word = nullptr)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && vis_word) {                   /* the workgroup's visits of the round: one update of the round's word, performed before the arrival below */
        const int v = *gb.s_vis;
        if (v) { gie_aadd32(vis_word, v); *gb.s_vis = 0; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    if (gb.nwg <= 1) { __syncthreads(); return; }             /* one workgroup: its waves share the CU's L2 path */
    gb.epoch += 1;
    if (threadIdx.x == 0 && !gb.failed) {
        const int target = gb.epoch * gb.nwg;
        __hip_atomic_fetch_add(gb.word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(gb.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > gb.spin_limit) { gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_BARRIER); gie_st(&c.cnt[GIE_CNT_BARFAIL], (int32_t)1); break; }
        }
        /* a timed-out barrier poisons the launch: this block leaves; the others time out at their
         * next barrier (bounded spin), so nobody hangs */
        if (spins > gb.spin_limit) *gb.s_fail = 1;
    }
    __syncthreads();
    if (*gb.s_fail) gb.failed = 1;
}

__device__ __forceinline__ int gie_clampi(int v, int hi) { return v < hi ? v : hi; }

/* The three waves run inside ONE launch (k_waves), separated by grid barriers of all workgroups: a wave
 * without seeds costs a counter read instead of a launch. */
/* append `value` to a list for every lane with `first`: one counter update per wave (hundreds of single appends to one word
 * serialise at its L2 slice, ≈ 6 ns each).  Only executing lanes are looked at. */
__device__ __forceinline__ void gie_list_append_wave(int32_t *list, int32_t *counter, const bool first, const int32_t value)
{
    const unsigned long long m = __ballot(first);
    if (!m) return;
    const int lane = __lane_id(), leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = gie_aadd32(counter, __popcll(m));
    base = __shfl(base, leader);
    if (first) gie_st(&list[base + __popcll(m & ((1ull << lane) - 1ull))], value);
}

/* ------------------------------------------------------------------ wave A: checkerboard block rounds */
/* raise_outside (wave_core.cuh:103-224) in the canonical CHECKERBOARD BLOCK-ROUND schedule (oracle/gie_oracle.c wave_a):
 * the hashed 8x8x8 blocks are coloured by the parity of bx + by + bz and the rounds alternate between the colours, so the blocks
 * of a round are never 6-adjacent: what a block reads of its neighbours (closest obstacle, raised-or-not) is stable while it
 * runs, although the wave both reads and rewrites those records.  In a round every ACTIVE block of the round's colour — one
 * that holds seeds or raises proposed by a neighbour in the round before — is taken by ONE WAVE: closest obstacles, stamps and
 * proposals of the block and the obstacles of the one-voxel halo around it go into LDS (six hash probes per block and round),
 * level after level runs inside the block to exhaustion out of LDS — per level: the pending voxels are raised (min-resolved
 * proposal) and become entries, every entry reads the level-start state of its six neighbours, proposes raises (LDS minimum
 * inside the block, agent-scope minimum into the other proposal plane across a block border) and computes its own lowering,
 * and the lowerings are applied behind a wave barrier — and the records that changed are written back once.
 *   planes: round h reads P[(h + 1) & 1] and proposes into P[h & 1] (P[1] = g_prop, P[0] = g_prop2); the seeds mark themselves
 *           with key 0 in the plane their colour reads first (colour 0: P[1], colour 1: P[0]);
 *   blocks: list wb_list[h & 1] with lvla_next[h] entries, membership flag wb_flag[h & 1][slot] (shared with wave B, which
 *           starts behind this wave with all flags cleared by the waves that took the blocks). */
struct gie_wa_tile { uint64_t coc[512], pair[512], prop[512], halo[6][64]; uint8_t flag[512], hflag[6][64];
                     uint16_t list[512], pend[2][512]; int32_t npend[2], nslot[6]; };                                   /* 18.9 KB */
#define GIE_WA_WAVES GIE_WAVE_SLOTS(8)                                    /* waves of a workgroup that take blocks */
#define GIE_WA_OK 1u                                      /* known, stored obstacle and distance valid: may be raised, may lower an entry */
#define GIE_WA_RAISED 2u                                  /* raised in this map update (stamp -map_ct) */
#define GIE_WA_DIRTY 4u                                   /* obstacle / stamp changed in this round */
#define GIE_WA_PAIR 8u                                    /* ... and the pair */
#define GIE_WA_PUSHB 16u                                  /* lowered to an obstacle inside the wave range: joins wave B's seeds */
#define GIE_WA_VANISHED 32u                               /* its stored obstacle lies inside the volume and is not OCCUPIED any more (looked up once, when the
                                                           * block is loaded: a voxel that is lowered takes over an obstacle that has NOT vanished, a raised
                                                           * one is not looked at again) */
/* `_aux[coc] != 0` (wave_core.cuh:177-178): the batch distance of a voxel is 0 iff it is OCCUPIED, and Mark never turns a non-zero value into 0 */
__device__ __forceinline__ int gie_wa_vanish_lid(const gie_ctx &c, const uint64_t coc)
{
    int x, y, z;
    gie_unpack_crd(coc, &x, &y, &z);
    x -= c.pvt[0]; y -= c.pvt[1]; z -= c.pvt[2];
    return gie_in_loc(c, x, y, z) ? gie_lid(c, x, y, z) : -1;
}

__device__ __forceinline__ void gie_wave_a_block(const gie_ctx &c, gie_gridbar &gb, gie_wa_tile &L, const int slot, const int round, const int lane)
{
    /* the chain of dependent round trips of a block-run: [block key + the block's records] -> [the six neighbour lookups + the
     * vanished-obstacle look-ups of the block's voxels] -> [halo records] -> [their vanished-obstacle look-ups] */
    uint64_t *const rd = ((round + 1) & 1) ? c.g_prop : c.g_prop2, *const wr = (round & 1) ? c.g_prop : c.g_prop2;
    const gie_vaddr base = (gie_vaddr)slot * GIE_VBSZ;
    GIE_WPROF_DECL;
    const int32_t nraw = gie_ld(&c.g_nbr[8 * (size_t)slot + (lane < 6 ? lane : 0)]);      /* the six neighbour slots: first in the queue, in flight with everything below */
    const uint64_t bkey = gie_ld(&c.g_key[slot]);
    uint64_t cv[8], cc8[8];
    int8_t ty8[8];
    int32_t wl8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {                                       /* voxel lane + 64 j = (x, y) = lane, z = j */
        const gie_vaddr a = base + lane + 64 * j;
        cv[j] = gie_ld(&rd[a]); cc8[j] = gie_ld(&c.g_coc[a]) & ~GIE_COC_STALEPAIR; ty8[j] = gie_ld(&c.g_type[a]); wl8[j] = gie_ld(&c.g_wl[a]);
    }
    if (lane == 0) gie_st(&c.wb_flag[round & 1][slot], (int32_t)0);
    int bk[3];
    gie_unpack_crd(bkey, &bk[0], &bk[1], &bk[2]);
    const int g0[3] = { bk[0] * 8, bk[1] * 8, bk[2] * 8 };
    int vl8[8]; int8_t vt8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) vl8[j] = gie_wa_vanish_lid(c, cc8[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) vt8[j] = c.glb_type[vl8[j] < 0 ? 0 : vl8[j]];              /* one batch, in flight with the neighbour lookups */
    if (lane < 6) L.nslot[lane] = nraw;
    gie_wave_sync();
    GIE_WPROF_MARK(0);                                                   /* 0: neighbour slots (own loads in flight) */
    const unsigned long long lt = (1ull << lane) - 1ull;
    int np = 0;
    {
        const int p = lane & 7, q = lane >> 3;
        const int hidx[6] = { 7 | (p << 3) | (q << 6), 0 | (p << 3) | (q << 6), p | (7 << 3) | (q << 6), p | (0 << 3) | (q << 6), p | (q << 3) | (7 << 6), p | (q << 3) | (0 << 6) };
        const int hx[6] = { -1, 8, p, p, p, p }, hy[6] = { p, p, -1, 8, q, q }, hz[6] = { q, q, q, q, -1, 8 };
        uint64_t hc[6], nk[6]; int8_t ht[6]; int32_t hw[6];
#pragma unroll
        for (int f = 0; f < 6; f++) {
            const int ns = L.nslot[f];
            const gie_vaddr an = (ns < 0 ? base : (gie_vaddr)ns * GIE_VBSZ) + hidx[f];
            nk[f] = gie_ld(&c.g_key[ns < 0 ? slot : ns]);              /* does the slot still hold that neighbour?  (fetched with its records) */
            hc[f] = gie_ld(&c.g_coc[an]) & ~GIE_COC_STALEPAIR; ht[f] = gie_ld(&c.g_type[an]); hw[f] = gie_ld(&c.g_wl[an]);
        }
        int hl[6]; int8_t hv[6];
#pragma unroll
        for (int f = 0; f < 6; f++) hl[f] = gie_wa_vanish_lid(c, hc[f]);
#pragma unroll
        for (int f = 0; f < 6; f++) hv[f] = c.glb_type[hl[f] < 0 ? 0 : hl[f]];          /* one batch */
        /* the block's own records go into LDS while that batch is in flight */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int v = lane + 64 * j;
            int ncx, ncy, ncz;
            gie_unpack_crd(cc8[j], &ncx, &ncy, &ncz);
            const bool ok = ty8[j] != GIE_VOX_UNKNOWN && !gie_invalid_coc(ncx, ncy, ncz)
                            && !gie_invalid_dist(c, gie_gdist(c, cc8[j], g0[0] + (lane & 7), g0[1] + (lane >> 3), g0[2] + j));
            L.coc[v] = cc8[j]; L.prop[v] = cv[j];
            L.flag[v] = (uint8_t)((ok ? GIE_WA_OK : 0u) | (wl8[j] == -c.map_ct ? GIE_WA_RAISED : 0u) | ((vl8[j] >= 0 && vt8[j] != GIE_VOX_OCCUPIED) ? GIE_WA_VANISHED : 0u));
            const bool have = cv[j] != GIE_NOPROP;
            const unsigned long long m = __ballot(have);
            if (have) L.pend[0][np + __popcll(m & lt)] = (uint16_t)v;
            np += __popcll(m);
        }
        unsigned live = 0;
#pragma unroll
        for (int f = 0; f < 6; f++) {
            int ncx, ncy, ncz;
            gie_unpack_crd(hc[f], &ncx, &ncy, &ncz);
            const bool there = L.nslot[f] >= 0 && nk[f] == gie_pack_crd(bk[0] + (f == 1) - (f == 0), bk[1] + (f == 3) - (f == 2), bk[2] + (f == 5) - (f == 4));
            if (there) live |= 1u << f;
            const bool ok = there && ht[f] != GIE_VOX_UNKNOWN && !gie_invalid_coc(ncx, ncy, ncz)
                            && !gie_invalid_dist(c, gie_gdist(c, hc[f], g0[0] + hx[f], g0[1] + hy[f], g0[2] + hz[f]));
            L.halo[f][lane] = hc[f];
            L.hflag[f][lane] = (uint8_t)((ok ? GIE_WA_OK : 0u) | (hw[f] == -c.map_ct ? GIE_WA_RAISED : 0u) | ((hl[f] >= 0 && hv[f] != GIE_VOX_OCCUPIED) ? GIE_WA_VANISHED : 0u));
        }
        if (lane < 6 && !((live >> lane) & 1u)) L.nslot[lane] = -1;      /* (an erased neighbour: its row of the table outlived it) */
    }
#pragma unroll
    for (int j = 0; j < 8; j++) if (cv[j] != GIE_NOPROP) gie_st(&rd[base + lane + 64 * j], (uint64_t)GIE_NOPROP);         /* consumed */
    if (lane == 0) { L.npend[0] = np; L.npend[1] = 0; }
    gie_wave_sync();
    GIE_WPROF_MARK(0);                                                   /* 1: halo + own records into LDS */
    unsigned xmask = 0;
    int nvis = 0;
    for (int sub = 0; np > 0; sub++) {
        GIE_WPROF_ADD(6, 1);
        const int pi = sub & 1;
        /* ---- the pending voxels: a seed (key 0) enters as it is, a proposal raises its voxel; both are entries of this level */
        for (int e = lane; e < np; e += 64) {
            const int v = L.pend[pi][e];
            const uint64_t key = L.prop[v];
            L.prop[v] = GIE_NOPROP;
            if (key != 0ull) {
                int lw[3];
                gie_unpack_wr(gie_pair_par(key), &lw[0], &lw[1], &lw[2]);
                L.coc[v] = gie_pack_crd(lw[0] + c.upvt[0], lw[1] + c.upvt[1], lw[2] + c.upvt[2]);
                L.pair[v] = key;
                L.flag[v] |= (uint8_t)(GIE_WA_RAISED | GIE_WA_DIRTY | GIE_WA_PAIR);
            }
            L.list[e] = (uint16_t)v;
            nvis++;
        }
        const int nent = np;
        if (lane == 0) L.npend[pi] = 0;
        gie_wave_sync();
        /* ---- every entry against the level-start state of its six neighbours: one LANE per (entry, direction) — a level holds a
         * dozen entries, and a lane that walks all six directions of one entry is six times the instructions on the critical path */
        for (int idx = lane; idx < nent * 6; idx += 64) {
            const int e = idx / 6, k = idx - 6 * e;
            const int v = L.list[e];
            const int ex = v & 7, ey = (v >> 3) & 7, ez = v >> 6;
            const int g[3] = { g0[0] + ex, g0[1] + ey, g0[2] + ez };
            const uint64_t lcoc = L.coc[v];
            const int cd = gie_gdist(c, lcoc, g[0], g[1], g[2]);
            if (cd > c.cutoff_sq) continue;
            const int ax = k >> 1, sg = (k & 1) ? 1 : -1;
            const int ux = ex + (ax == 0 ? sg : 0), uy = ey + (ax == 1 ? sg : 0), uz = ez + (ax == 2 ? sg : 0);
            const int nb[3] = { g0[0] + ux - c.pvt[0], g0[1] + uy - c.pvt[1], g0[2] + uz - c.pvt[2] };
            if (gie_in_loc(c, nb[0], nb[1], nb[2]) || gie_in_whole(c, nb[0], nb[1], nb[2])) continue;     /* (tiling: not into another tile's territory) */
            const bool inside = (unsigned)ux < 8u && (unsigned)uy < 8u && (unsigned)uz < 8u;
            const int hp = (ax == 0) ? (ey + 8 * ez) : ((ax == 1) ? (ex + 8 * ez) : (ex + 8 * ey));
            const int nv = (ux & 7) + 8 * (uy & 7) + 64 * (uz & 7);
            const uint64_t ncoc = inside ? L.coc[nv] : L.halo[k][hp];
            const unsigned nf = inside ? L.flag[nv] : L.hflag[k][hp];
            if (!(nf & GIE_WA_OK) || (nf & GIE_WA_RAISED) || ncoc == lcoc) continue;
            if (nf & GIE_WA_VANISHED) {
                int lc[3];
                gie_unpack_crd(lcoc, &lc[0], &lc[1], &lc[2]);
                const uint64_t key = gie_pair_make(gie_d2(lc[0], lc[1], lc[2], g0[0] + ux, g0[1] + uy, g0[2] + uz),
                                                   gie_pack_wr(lc[0] - c.upvt[0], lc[1] - c.upvt[1], lc[2] - c.upvt[2]));
                if (inside) {
                    if (__hip_atomic_fetch_min(&L.prop[nv], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == GIE_NOPROP)
                        L.pend[pi ^ 1][__hip_atomic_fetch_add(&L.npend[pi ^ 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = (uint16_t)nv;
                } else { gie_amin64(&wr[(gie_vaddr)L.nslot[k] * GIE_VBSZ + nv], key); xmask |= 1u << k; }
            } else {
                /* the entry's own lowering: the nearest of the obstacles its neighbours hold, the first direction among equals
                 * (the reference walks the directions in order and replaces on a strict improvement).  Nobody proposes to an
                 * entry — it is raised — so its proposal slot is free to collect the minimum over its six lanes. */
                int nc[3];
                gie_unpack_crd(ncoc, &nc[0], &nc[1], &nc[2]);
                const int d = gie_d2(nc[0], nc[1], nc[2], g[0], g[1], g[2]);
                if (d < cd) __hip_atomic_fetch_min(&L.prop[v], ((uint64_t)(uint32_t)d << 3) | (uint64_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        gie_wave_sync();
        /* ---- the lowerings (the neighbour an entry takes its obstacle from is not an entry: it is not raised) */
        bool deferred = false;
        for (int e = lane; e < nent; e += 64) {
            const int v = L.list[e];
            const uint64_t t = L.prop[v];
            if (t == GIE_NOPROP) continue;
            const int k = (int)(t & 7u), ax = k >> 1, sg = (k & 1) ? 1 : -1;
            const int ex = v & 7, ey = (v >> 3) & 7, ez = v >> 6;
            const int ux = ex + (ax == 0 ? sg : 0), uy = ey + (ax == 1 ? sg : 0), uz = ez + (ax == 2 ? sg : 0);
            const bool inside = (unsigned)ux < 8u && (unsigned)uy < 8u && (unsigned)uz < 8u;
            const int hp = (ax == 0) ? (ey + 8 * ez) : ((ax == 1) ? (ex + 8 * ez) : (ex + 8 * ey));
            const uint64_t ncoc = inside ? L.coc[(ux & 7) + 8 * (uy & 7) + 64 * (uz & 7)] : L.halo[k][hp];
            int ncx, ncy, ncz;
            gie_unpack_crd(ncoc, &ncx, &ncy, &ncz);
            const int nw[3] = { ncx - c.upvt[0], ncy - c.upvt[1], ncz - c.upvt[2] };
            if (!gie_in_wr(c, nw[0], nw[1], nw[2])) { deferred = true; continue; }      /* the rare corner below; L.prop[v] keeps the minimum */
            L.prop[v] = GIE_NOPROP;
            L.coc[v] = ncoc;
            L.pair[v] = gie_pair_make((int)(t >> 3), gie_pack_wr(nw[0], nw[1], nw[2]));
            L.flag[v] = (uint8_t)((L.flag[v] & ~(GIE_WA_RAISED | GIE_WA_VANISHED)) | GIE_WA_DIRTY | GIE_WA_PAIR | GIE_WA_PUSHB);
        }
        if (__ballot(deferred) != 0ull) {
            /* An entry whose nearest offered obstacle lies OUTSIDE the wave range (it cannot be packed into a pair).  The reference
             * walks the directions in order (wave_core.cuh:199-221): every strict improvement overwrites distance and obstacle, and
             * only then an un-encodable obstacle `continue`s -- so the voxel ends up with the nearest obstacle, but its PAIR (and its
             * place in frontier B) is that of the LAST improvement whose obstacle was inside the wave range, if there was one.
             * Replayed here by the entry's own lane from the level-start state: a neighbour that was an entry of this level was
             * raised when the level began (it may have been lowered a moment ago by another lane: then it is in L.list). */
            gie_wave_sync();
            for (int e = lane; e < nent; e += 64) {
                const int v = L.list[e];
                const uint64_t t = L.prop[v];
                if (t == GIE_NOPROP) continue;
                const int ex = v & 7, ey = (v >> 3) & 7, ez = v >> 6;
                const int g[3] = { g0[0] + ex, g0[1] + ey, g0[2] + ez };
                const uint64_t lcoc = L.coc[v];
                int cd = gie_gdist(c, lcoc, g[0], g[1], g[2]);
                uint64_t fin = lcoc, spair = GIE_NOPROP;
                for (int k = 0; k < 6; k++) {
                    const int ax = k >> 1, sg = (k & 1) ? 1 : -1;
                    const int ux = ex + (ax == 0 ? sg : 0), uy = ey + (ax == 1 ? sg : 0), uz = ez + (ax == 2 ? sg : 0);
                    const int nb[3] = { g0[0] + ux - c.pvt[0], g0[1] + uy - c.pvt[1], g0[2] + uz - c.pvt[2] };
                    if (gie_in_loc(c, nb[0], nb[1], nb[2]) || gie_in_whole(c, nb[0], nb[1], nb[2])) continue;
                    const bool inside = (unsigned)ux < 8u && (unsigned)uy < 8u && (unsigned)uz < 8u;
                    const int hp = (ax == 0) ? (ey + 8 * ez) : ((ax == 1) ? (ex + 8 * ez) : (ex + 8 * ey));
                    const int nv = (ux & 7) + 8 * (uy & 7) + 64 * (uz & 7);
                    const uint64_t ncoc = inside ? L.coc[nv] : L.halo[k][hp];
                    const unsigned nf = inside ? L.flag[nv] : L.hflag[k][hp];
                    if (!(nf & GIE_WA_OK) || (nf & (GIE_WA_RAISED | GIE_WA_VANISHED)) || ncoc == lcoc) continue;
                    if (inside && (nf & GIE_WA_DIRTY)) {
                        bool was_entry = false;
                        for (int i = 0; i < nent; i++) was_entry |= (L.list[i] == (uint16_t)nv);
                        if (was_entry) continue;
                    }
                    int nc[3];
                    gie_unpack_crd(ncoc, &nc[0], &nc[1], &nc[2]);
                    const int d = gie_d2(nc[0], nc[1], nc[2], g[0], g[1], g[2]);
                    if (d >= cd) continue;
                    cd = d; fin = ncoc;
                    const int nw[3] = { nc[0] - c.upvt[0], nc[1] - c.upvt[1], nc[2] - c.upvt[2] };
                    if (gie_in_wr(c, nw[0], nw[1], nw[2])) spair = gie_pair_make(d, gie_pack_wr(nw[0], nw[1], nw[2]));
                }
                L.prop[v] = GIE_NOPROP;
                L.coc[v] = fin;
                unsigned f = (L.flag[v] & ~(GIE_WA_RAISED | GIE_WA_VANISHED)) | GIE_WA_DIRTY;
                if (spair != GIE_NOPROP) { L.pair[v] = spair; f |= GIE_WA_PAIR | GIE_WA_PUSHB; }
                L.flag[v] = (uint8_t)f;
            }
        }
        gie_wave_sync();
        np = L.npend[pi ^ 1];
    }
    GIE_WPROF_MARK(0);                                                   /* 2: levels inside the block */
    /* ---- write back what changed; the voxels that were lowered with a pair join wave B's seeds (one counter update per block);
     * neighbour blocks that received a proposal take part in the next round.  The counter update and the six flag exchanges are in
     * flight together, the append to the block list follows (two dependent round trips, not three). */
    {
        int npush = 0;
        unsigned f8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            f8[j] = L.flag[lane + 64 * j];
            npush += __popcll(__ballot((f8[j] & GIE_WA_PUSHB) != 0u));
        }
        unsigned any6 = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) if (__ballot((xmask >> k) & 1u) != 0ull) any6 |= 1u << k;   /* wave-uniform */
        const bool act = lane < 6 && ((any6 >> lane) & 1u);
        const int ns = act ? L.nslot[lane] : 0;
        /* ONE atomic instruction for both (a returning atomic is paid per instruction, whatever its lanes address): the flag is
         * "was 0 before", so an add serves as the exchange */
        int32_t r1 = 1;
        {
            int32_t *ap = nullptr; int av = 0;
            if (act) { ap = &c.wb_flag[(round + 1) & 1][ns]; av = 1; }
            if (lane == 63 && npush > 0) { ap = &c.cnt[GIE_CNT_B]; av = npush; }
            if (ap != nullptr) r1 = gie_aadd32(ap, av);
        }
        const int qbase = __shfl(r1, 63);
        const bool first = act && r1 == 0;
        const unsigned long long fm = __ballot(first);
        int ab0 = 0;
        if (fm != 0ull) {
            if (lane == 0) ab0 = gie_aadd32(&c.lvla_next[round + 1], __popcll(fm));
        }
        int before = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int v = lane + 64 * j;
            const unsigned f = f8[j];
            if (f & GIE_WA_DIRTY) {
                gie_st(&c.g_coc[base + v], L.coc[v]);
                gie_st(&c.g_wl[base + v], (f & GIE_WA_RAISED) ? (int32_t)-c.map_ct : (int32_t)1);
                if (f & GIE_WA_PAIR) gie_st(&c.g_pair[base + v], L.pair[v]);
                gie_touch(c, base + v);
            }
            const unsigned long long m = __ballot((f & GIE_WA_PUSHB) != 0u);
            if (f & GIE_WA_PUSHB) {
                const int i = qbase + before + __popcll(m & lt);
                if (i < c.qcap_ab) { gie_st(&c.qb[i], gie_pack_crd(g0[0] + (lane & 7), g0[1] + (lane >> 3), g0[2] + j)); gie_st(&c.qb_a[i], (gie_vaddr)(base + v)); }
                else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
            }
            before += __popcll(m);
        }
        if (fm != 0ull) {
            ab0 = __shfl(ab0, 0);
            if (first) gie_st(&c.wb_list[(round + 1) & 1][ab0 + __popcll(fm & ((1ull << lane) - 1ull))], (int32_t)ns);
        }
    }
    {
        int sv = nvis;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sv += __shfl_xor(sv, o);
        if (lane == 0 && sv > 0) GIE_VIS_ADD(&c.lvla_vis[round], sv);
    }
    GIE_WPROF_DRAIN();
    GIE_WPROF_MARK(0);                                                   /* 3: write-back, activation (drained) */
    GIE_WPROF_ADD(7, 1);
    GIE_WPROF_END(0);
    gie_wave_sync();                                       /* the LDS block is reused for the wave's next block */
}

__device__ __forceinline__ void gie_wave_a_run(const gie_ctx &c, gie_gridbar &gb, gie_wa_tile *tiles)
{
    const bool boss = (blockIdx.x == 0 && threadIdx.x == 0);
    const int n = gie_clampi(gie_ld(&c.cnt[GIE_CNT_A]), c.qcap_ab);
    if (boss) { c.cnt[GIE_CNT_SEED_A] = n; c.cnt[GIE_CNT_SEED_B] = gie_ld(&c.cnt[GIE_CNT_B]); }
    if (n == 0 || gb.failed) return;               /* same n everywhere */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x * GIE_WAVE_THREADS + threadIdx.x; e < n; e += gridDim.x * GIE_WAVE_THREADS) {
        const gie_vaddr a = gie_ld(&c.qa_a[e]);
        int g[3];
        gie_unpack_crd(gie_ld(&c.qa[e]), &g[0], &g[1], &g[2]);
        const int col = ((g[0] >> 3) + (g[1] >> 3) + (g[2] >> 3)) & 1;
        bool first = false;
        if (a >= 0) {
            gie_st(col ? &c.g_prop2[a] : &c.g_prop[a], (uint64_t)0);
            first = gie_axchg32(&c.wb_flag[col][(int)(a >> 9)], (int32_t)1) == 0;
        }
        gie_list_append_wave(c.wb_list[0], &c.lvla_next[0], first && col == 0, (int32_t)(a >> 9));
        gie_list_append_wave(c.wb_list[1], &c.lvla_next[1], first && col == 1, (int32_t)(a >> 9));
    }
    gie_grid_sync(gb, c);
    GIE_TS2(1, n);
    int round = 0;
    while (!gb.failed && round < GIE_MAX_LEVELS - 2) {
        /* (the round's length and the wave's first list entry in one round trip: see gie_wave_c_run) */
        const int32_t *list = c.wb_list[round & 1];
        const int i0 = (int)blockIdx.x + (int)gridDim.x * wave;
        const int nt = gie_ld(&c.lvla_next[round]);
        int snext = gie_ld(&list[i0 < c.max_blocks ? i0 : 0]);
        if (nt <= 0) { if (round == 0) { round++; continue; } break; }      /* (colour 0 may have no seeds) same everywhere */
        if (wave < GIE_WA_WAVES) {
            for (int i = i0; i < nt; i += (int)gridDim.x * GIE_WA_WAVES) {      /* (spread over the workgroups first) */
                const int sl = snext;
                const int in = i + (int)gridDim.x * GIE_WA_WAVES;
                if (in < nt) snext = gie_ld(&list[in]);
                gie_wave_a_block(c, gb, tiles[wave], sl, round, lane);
            }
        }
        gie_grid_sync(gb, c, &c.lvla_vis[round]);
        GIE_TS2(3, nt);
        round++;
    }
    if (round >= GIE_MAX_LEVELS - 2 && boss) gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
    if (boss) {
        int lv = 0; long long vis = 0;
        for (int l = 0; l < round; l++) { const int v = gie_ld(&c.lvla_vis[l]); if (v > 0) { lv++; vis += v; } }
        c.cnt[GIE_CNT_VIS_A] = (int)vis; c.cnt[GIE_CNT_LVL_A] = lv;
        *reinterpret_cast<long long *>(&c.cnt[GIE_CNT_TOT_A]) += vis;
    }
}

/* ------------------------------------------------------------------ wave B: block rounds */
/* lower_outside (wave_core.cuh:229-350) in the canonical BLOCK-ROUND schedule (the twin of wave C's tile rounds, over
 * the hashed 8x8x8 blocks of the global map): in a round every ACTIVE block — one that holds proposals from the round before, or
 * seeds — is taken by ONE WAVE: its pairs, its proposals, the distances stored in it and which of its voxels may be lowered go into
 * LDS together with the pairs of the one-voxel halo around it (the six neighbour blocks are looked up ONCE per block and round: six
 * hash probes instead of six per entry and level), a level-synchronous BFS runs inside the block to exhaustion out of LDS, what
 * it proposes to a voxel of a neighbouring block is min-resolved in that voxel's slot of the other proposal plane and activates the
 * neighbour for the next round; what it proposes to voxels INSIDE the volume is min-resolved in the face table (lprop) over the
 * whole wave and stored when the wave is over.  A front needs one grid barrier per BLOCK it crosses, not two per voxel.
 *   planes: round r reads P[(r + 1) & 1] (the seeds come in P[1] = g_prop) and proposes into P[r & 1] (P[0] = g_prop2);
 *   blocks: list wb_list[r & 1] with lvlb_next[r] entries, membership flag wb_flag[r & 1][slot];
 *   rule:   a proposal replaces a pair on a strict distance improvement over the value at the start of the (sub-)level, among
 *           proposals the smaller (dist, parent) wins; the seeds of round 0 enter with the pair they hold; a voxel that is
 *           taken up is committed and expanded unless the distance stored in it BEFORE the commit exceeds the cut-off (:262-266). */
/* Round 6: the block WITH its one-voxel halo, 10 x 10 x 10 cells (cell (ex, ey, ez), each in -1 .. 8, at GIE_WB_P), owned by the
 * lanes as a Latin cube and run without lists — see gie_wave_c_tile, whose twin this is.  What is different here: a cell has a
 * CLASS next to its pair (may be lowered / lies inside the volume / neither), a voxel that is taken up is only expanded when the
 * distance stored in it before passes the cut-off, and a proposal to a position inside the volume goes into that cell's PAIR (the
 * running minimum of the block-run, marked GIE_PAIR_NEW), not into its proposal cell. */
struct gie_wb_tile { uint64_t pair[1000], prop[1000]; uint8_t flag[1000]; int32_t nslot[8]; };                           /* 17 KB */
#define GIE_WB_P(ex, ey, ez) ((ex) + 10 * (ey) + 100 * (ez) + 111)
#define GIE_WB_WAVES GIE_WAVE_SLOTS(9)                                    /* waves of a workgroup that take blocks */
#define GIE_WB_OK 1u                                      /* the voxel may be lowered: known, and its stored obstacle is valid */

#define GIE_WB_INVOL 4u                                   /* the position lies inside the volume: its slot holds the distance a proposal has to beat
                                                           * (`_aux[n]`, wave_core.cuh:334: the Mark-time pair's distance of an observed voxel, the batch
                                                           * distance of an unknown one), fetched with the block — the levels never wait for memory */
__device__ __forceinline__ void gie_wave_b_block(const gie_ctx &c, gie_gridbar &gb, gie_wb_tile &L, const int slot, const int round, const int lane)
{
    const int32_t nraw = gie_ld(&c.g_nbr[8 * (size_t)slot + (lane < 6 ? lane : 0)]);      /* the six neighbour slots (see gie_wave_a_block) */
    int bk[3];
    gie_unpack_crd(gie_ld(&c.g_key[slot]), &bk[0], &bk[1], &bk[2]);
    const int g0[3] = { bk[0] * 8, bk[1] * 8, bk[2] * 8 };              /* global coordinate of the block's first voxel */
    uint64_t *const rd = ((round + 1) & 1) ? c.g_prop : c.g_prop2, *const wr = (round & 1) ? c.g_prop : c.g_prop2;
    const gie_vaddr base = (gie_vaddr)slot * GIE_VBSZ;
    /* (the lane number through an opaque move: everything below that depends on the lane alone — forty cell addresses — would
     * otherwise be computed once per launch, ahead of the rounds, and kept in registers the block routine then lacks: 75 spilled) */
    int lane_here = lane;
    asm volatile("" : "+v"(lane_here));
    const int la = lane_here & 7, lb = lane_here >> 3;
    GIE_WPROF_DECL;
    /* the lane's eight voxels (layer j: x = la - j, y = lb - j mod 8; their pairs live in L.pair): */
    unsigned sdok = 0;                                    /* ... the distance STORED in the voxel does not exceed the cut-off (wave_core.cuh:262-266) */
    unsigned inv8 = 0;                                    /* ... the voxel lies inside the volume */
    unsigned seedm = 0;                                   /* ... which hold a proposal of the round before (round 0: the seeds) */
    unsigned bdm = 0;                                     /* my cells inside the volume whose voxel is UNKNOWN (bits 0-7: my voxels, 8-13: my halo cells): the distance to beat is the batch distance */
    {
        uint64_t pv[8], cv[8], cc8[8];
        int8_t ty8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ex = (la - j) & 7, ey = (lb - j) & 7;
            const gie_vaddr a = base + (ex | (ey << 3) | (j << 6));
            const int nb[3] = { g0[0] + ex - c.pvt[0], g0[1] + ey - c.pvt[1], g0[2] + j - c.pvt[2] };
            const bool inv = gie_in_loc(c, nb[0], nb[1], nb[2]);
            const int nid = inv ? gie_lid(c, nb[0], nb[1], nb[2]) : 0;
            inv8 |= (inv ? 1u : 0u) << j;
            pv[j] = gie_ld(inv ? &c.pair[nid] : &c.g_pair[a]) & ~GIE_PAIR_NEW; cv[j] = gie_ld(&rd[a]); cc8[j] = gie_ld(&c.g_coc[a]);
            ty8[j] = gie_ld(inv ? &c.glb_type[nid] : &c.g_type[a]);
        }
        if (lane == 0) gie_st(&c.wb_flag[round & 1][slot], (int32_t)0); /* may be activated again (for round + 2) from now on */
        /* ---- the six neighbour blocks, out of the block's row of the neighbour table, with the block's own records in flight */
        if (lane < 6) L.nslot[lane] = nraw;
        gie_wave_sync();                                                /* the neighbour slots */
        GIE_WPROF_MARK(8);
        {   /* halo face f (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z): the lane's position (la, lb) on it -> in-block index of the voxel across the face */
            const int hidx[6] = { 7 | (la << 3) | (lb << 6), 0 | (la << 3) | (lb << 6), la | (7 << 3) | (lb << 6), la | (0 << 3) | (lb << 6), la | (lb << 3) | (7 << 6), la | (lb << 3) | (0 << 6) };
            const int hx[6] = { -1, 8, la, la, la, la }, hy[6] = { la, la, -1, 8, lb, lb }, hz[6] = { lb, lb, lb, lb, -1, 8 };
            uint64_t hp[6], hc[6], nk[6]; int8_t ht[6];
            unsigned hinv = 0;
#pragma unroll
            for (int f = 0; f < 6; f++) {
                const int ns = L.nslot[f];
                const gie_vaddr an = (ns < 0 ? base : (gie_vaddr)ns * GIE_VBSZ) + hidx[f];
                nk[f] = gie_ld(&c.g_key[ns < 0 ? slot : ns]);
                const int nb[3] = { g0[0] + hx[f] - c.pvt[0], g0[1] + hy[f] - c.pvt[1], g0[2] + hz[f] - c.pvt[2] };
                const bool inv = gie_in_loc(c, nb[0], nb[1], nb[2]);
                const int nid = inv ? gie_lid(c, nb[0], nb[1], nb[2]) : 0;
                if (inv) hinv |= 1u << f;
                hp[f] = gie_ld(inv ? &c.pair[nid] : &c.g_pair[an]) & ~GIE_PAIR_NEW; hc[f] = gie_ld(&c.g_coc[an]);
                ht[f] = gie_ld(inv ? &c.glb_type[nid] : &c.g_type[an]);
            }
            unsigned live = 0;
#pragma unroll
            for (int f = 0; f < 6; f++) {
                const int cell = GIE_WB_P(hx[f], hy[f], hz[f]);
                const bool there = L.nslot[f] >= 0 && nk[f] == gie_pack_crd(bk[0] + (f == 1) - (f == 0), bk[1] + (f == 3) - (f == 2), bk[2] + (f == 5) - (f == 4));
                if (there) live |= 1u << f;
                L.prop[cell] = GIE_NOPROP;
                if ((hinv >> f) & 1u) {
                    if (ht[f] == GIE_VOX_UNKNOWN) bdm |= 1u << (8 + f);                   /* (its batch distance: below, one copy of the code) */
                    L.pair[cell] = hp[f]; L.flag[cell] = GIE_WB_INVOL;
                    continue;
                }
                int ncx, ncy, ncz;
                gie_unpack_crd(hc[f], &ncx, &ncy, &ncz);
                const bool ok = there && ht[f] != GIE_VOX_UNKNOWN && !gie_invalid_coc(ncx, ncy, ncz)
                                && !gie_in_whole(c, g0[0] + hx[f] - c.pvt[0], g0[1] + hy[f] - c.pvt[1], g0[2] + hz[f] - c.pvt[2]);   /* (tiling: not into another tile's territory) */
                L.pair[cell] = hp[f]; L.flag[cell] = ok ? GIE_WB_OK : 0u;
            }
            if (lane < 6 && !((live >> lane) & 1u)) L.nslot[lane] = -1;      /* (an erased neighbour) */
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ex = (la - j) & 7, ey = (lb - j) & 7;
            const int cell = GIE_WB_P(ex, ey, j);
            const int nb[3] = { g0[0] + ex - c.pvt[0], g0[1] + ey - c.pvt[1], g0[2] + j - c.pvt[2] };
            if ((inv8 >> j) & 1u) {
                if (ty8[j] == GIE_VOX_UNKNOWN) bdm |= 1u << j;
                L.pair[cell] = pv[j]; L.prop[cell] = GIE_NOPROP; L.flag[cell] = GIE_WB_INVOL;
                continue;
            }
            int ncx, ncy, ncz;
            gie_unpack_crd(cc8[j], &ncx, &ncy, &ncz);
            L.pair[cell] = pv[j]; L.prop[cell] = cv[j];
            L.flag[cell] = (ty8[j] != GIE_VOX_UNKNOWN && !gie_invalid_coc(ncx, ncy, ncz) && !gie_in_whole(c, nb[0], nb[1], nb[2])) ? GIE_WB_OK : 0u;
            if (gie_gdist(c, cc8[j], g0[0] + ex, g0[1] + ey, g0[2] + j) <= c.cutoff_sq) sdok |= 1u << j;
            if (cv[j] != GIE_NOPROP) { seedm |= 1u << j; gie_st(&rd[base + (ex | (ey << 3) | (j << 6))], (uint64_t)GIE_NOPROP); }      /* consumed */
        }
    }
    while (bdm != 0u) {                                   /* (rare: only blocks at the volume's faces; no wave-level operation inside) */
        const int t = __ffs((int)bdm) - 1;
        bdm &= bdm - 1u;
        const int f = t - 8;
        const int ux = t < 8 ? ((la - t) & 7) : (f == 0 ? -1 : f == 1 ? 8 : la), uy = t < 8 ? ((lb - t) & 7) : (f < 2 ? la : f == 2 ? -1 : f == 3 ? 8 : lb),
                  uz = t < 8 ? t : (f < 4 ? lb : f == 4 ? -1 : 8);
        L.pair[GIE_WB_P(ux, uy, uz)] = gie_pair_make(gie_batch_dist_direct(c, g0[0] + ux - c.pvt[0], g0[1] + uy - c.pvt[1], g0[2] + uz - c.pvt[2]), 0);
    }
    gie_wave_sync();
    GIE_WPROF_MARK(8);
    /* ---- BFS inside the block: (a) every lane looks at the proposal cells of its eight voxels: the ones that improve are taken up and
     * committed, and expanded unless the distance stored before exceeds the cut-off; (b) every lane expands its entries, one after the other */
    unsigned done = 0, chg = 0;                           /* my voxels committed in this round / whose pair has changed */
    int nvis = 0;
    const int emax = c.empty_value;
    bool first = round == 0;                              /* round 0, first level: the seeds enter with the pair they hold */
    for (;;) {
        unsigned tk = 0;                                  /* my voxels that become entries in this level */
        {
            uint64_t cd[8];
            uint32_t ch[8];                               /* the upper word of my voxels' pairs: their distances */
#pragma unroll
            for (int j = 0; j < 8; j++) {                 /* sixteen reads in flight */
                cd[j] = L.prop[GIE_WB_P((la - j) & 7, (lb - j) & 7, j)];
                ch[j] = reinterpret_cast<const uint32_t *>(&L.pair[GIE_WB_P((la - j) & 7, (lb - j) & 7, j)])[1];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) L.prop[GIE_WB_P((la - j) & 7, (lb - j) & 7, j)] = GIE_NOPROP;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                /* ("no proposal" carries the largest distance there is: never below a voxel's own; a voxel inside the volume has no proposal cell in use) */
                const int pd = gie_pair_dist(cd[j]), od = (int)(ch[j] >> 10);
                const bool take = first ? ((seedm >> j) & 1u) != 0u : (pd < od && !((inv8 >> j) & 1u));
                if (take) {
                    GIE_WPROF_ADD(14, j == 0 ? 1 : 0);
                    if (!first) { L.pair[GIE_WB_P((la - j) & 7, (lb - j) & 7, j)] = cd[j]; chg |= 1u << j; }
                    nvis++;
                    /* the distance stored before the commit: the one loaded with the block, or — the voxel has been committed in this
                     * block-run before — the one that commit stored: its pair's until now */
                    if (((done >> j) & 1u) ? od <= c.cutoff_sq : ((sdok >> j) & 1u) != 0u) { done |= 1u << j; tk |= 1u << j; }
                }
            }
        }
        first = false;
        if (__ballot(tk != 0u) == 0ull) break;            /* wave-uniform */
        gie_wave_sync();
        while (__ballot(tk != 0u) != 0ull) {
            if (tk != 0u) {
                const int ez = __ffs((int)tk) - 1;
                tk &= tk - 1u;
                const int ex = (la - ez) & 7, ey = (lb - ez) & 7;
                const int cell = GIE_WB_P(ex, ey, ez) - 100;                                  /* (non-negative offsets from here: -z 0, -y 90, -x 99, self 100, +x 101, +y 110, +z 200) */
                uint64_t *const pb = &L.pair[cell];
                const uint32_t *const ph = reinterpret_cast<const uint32_t *>(pb) + 1;         /* a pair's distance is in its upper word */
                const uint8_t *const fb = &L.flag[cell];
                const uint64_t own = pb[100];
                const uint32_t s[6] = { ph[2 * 99], ph[2 * 101], ph[2 * 90], ph[2 * 110], ph[0], ph[2 * 200] };
                const unsigned nf[6] = { fb[99], fb[101], fb[90], fb[110], fb[0], fb[200] };
                const int po[6] = { 99, 101, 90, 110, 0, 200 };
                const uint64_t par = gie_pair_par(own);
                int cw[3];
                gie_unpack_wr(par, &cw[0], &cw[1], &cw[2]);
                const int cg[3] = { cw[0] + c.upvt[0], cw[1] + c.upvt[1], cw[2] + c.upvt[2] };          /* the entry's closest obstacle, global */
                const int cx = cg[0] - (g0[0] + ex), cy = cg[1] - (g0[1] + ey), cz = cg[2] - (g0[2] + ez);
                /* |closest obstacle - neighbour|^2 = |closest obstacle - voxel|^2 +- 2 c + 1, in 32 bits while the obstacle is less than
                 * 2^14 voxels away on every axis; farther (gie_d2 saturates there) no candidate is below any stored distance */
                const bool near = (unsigned)(cx + 16384) < 32768u && (unsigned)(cy + 16384) < 32768u && (unsigned)(cz + 16384) < 32768u;
                const int d0 = cx * cx + cy * cy + cz * cz + 1;
                const int d[6] = { d0 + 2 * cx, d0 - 2 * cx, d0 + 2 * cy, d0 - 2 * cy, d0 + 2 * cz, d0 - 2 * cz };
                /* a neighbour inside the volume (only blocks at its faces have any): not from an obstacle in another tile's territory (tiling: gie_frontier_outside) */
                const int cl3[3] = { cg[0] - c.pvt[0], cg[1] - c.pvt[1], cg[2] - c.pvt[2] };
                const bool invol_ok = !(gie_in_whole(c, cl3[0], cl3[1], cl3[2]) && !gie_in_loc(c, cl3[0], cl3[1], cl3[2]));
                const uint32_t plo = (uint32_t)par, phi = (uint32_t)(par >> 32);
                if (near) {
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const uint32_t sdk = s[k] >> 10;
                        /* a voxel that may be lowered: a strict improvement goes into its proposal cell; a position inside the volume: its
                         * cell holds the distance to beat, a proposal (marked) that beats it replaces it — the running minimum of this
                         * block-run, handed to the face table once, with the write-back (at equal distance the stored reference — no mark —
                         * stays below every proposal; among proposals the parent decides) */
                        const bool an = (nf[k] & GIE_WB_OK) && d[k] < emax && (uint32_t)d[k] < sdk;
                        const bool ai = (nf[k] & GIE_WB_INVOL) && invol_ok && (uint32_t)d[k] <= sdk;
                        if (an | ai) {
                            const uint64_t key = ((uint64_t)(((uint32_t)d[k] << 10) | phi | (ai ? (uint32_t)(GIE_PAIR_NEW >> 32) : 0u)) << 32) | plo;
                            (void)__hip_atomic_fetch_min(&pb[(ai ? 0 : 1000) + po[k]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
        }
        gie_wave_sync();
    }
    GIE_WPROF_MARK(8);
    /* ---- what the block proposes to its neighbours' voxels: the halo's proposal cells into the other proposal plane */
    unsigned xmask = 0;
    {
        const int hidx[6] = { 7 | (la << 3) | (lb << 6), 0 | (la << 3) | (lb << 6), la | (7 << 3) | (lb << 6), la | (0 << 3) | (lb << 6), la | (lb << 3) | (7 << 6), la | (lb << 3) | (0 << 6) };
        const int hcell[6] = { GIE_WB_P(-1, la, lb), GIE_WB_P(8, la, lb), GIE_WB_P(la, -1, lb), GIE_WB_P(la, 8, lb), GIE_WB_P(la, lb, -1), GIE_WB_P(la, lb, 8) };
        uint64_t hq[6];
#pragma unroll
        for (int f = 0; f < 6; f++) hq[f] = L.prop[hcell[f]];
#pragma unroll
        for (int f = 0; f < 6; f++)
            if (hq[f] != GIE_NOPROP) { gie_amin64(&wr[(gie_vaddr)L.nslot[f] * GIE_VBSZ + hidx[f]], hq[f]); xmask |= 1u << f; }     /* (only voxels of live neighbours may be lowered) */
    }
    /* ---- write back: changed pairs, the closest obstacle of every voxel committed in this round */
    {
        unsigned wm = (chg | done) & ~inv8;
        while (wm != 0u) {                                /* (stores only; no wave-level operation inside) */
            const int j = __ffs((int)wm) - 1;
            wm &= wm - 1u;
            const int ex = (la - j) & 7, ey = (lb - j) & 7;
            const gie_vaddr a = base + (ex | (ey << 3) | (j << 6));
            const uint64_t pr = L.pair[GIE_WB_P(ex, ey, j)];
            if ((chg >> j) & 1u) gie_st(&c.g_pair[a], pr);
            if ((done >> j) & 1u) {
                int cw[3];
                gie_unpack_wr(gie_pair_par(pr), &cw[0], &cw[1], &cw[2]);
                gie_st(&c.g_coc[a], gie_pack_crd(cw[0] + c.upvt[0], cw[1] + c.upvt[1], cw[2] + c.upvt[2]));
                gie_touch(c, a);
            }
        }
    }
    /* ---- (i) neighbour blocks that received a proposal take part in the next round; (ii) what the block-run proposes to voxels
     * inside the volume goes to the face table, which takes the minimum of the whole wave — whoever finds a voxel's slot empty
     * lists the voxel.  Two stages of returning atomics, each stage of BOTH in flight together (a block-run is a chain of dependent
     * round trips: four here would be a fifth of it): flag exchanges + table minima, then the two list appends. */
    {
        unsigned any6 = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) if (__ballot((xmask >> k) & 1u) != 0ull) any6 |= 1u << k;   /* wave-uniform */
        const bool act = lane < 6 && ((any6 >> lane) & 1u);
        const int ns = act ? L.nslot[lane] : 0;
        const int hx[6] = { -1, 8, la, la, la, la }, hy[6] = { la, la, -1, 8, lb, lb }, hz[6] = { lb, lb, lb, lb, -1, 8 };
        uint64_t old[14];
        unsigned hitm = 0;
#pragma unroll
        for (int t = 0; t < 14; t++) {
            /* t < 8: my voxel of layer t if it lies inside the volume; t >= 8: my halo cell of face t - 8 if it does */
            const int ux = t < 8 ? ((la - t) & 7) : hx[t < 8 ? 0 : t - 8], uy = t < 8 ? ((lb - t) & 7) : hy[t < 8 ? 0 : t - 8], uz = t < 8 ? t : hz[t < 8 ? 0 : t - 8];
            const int cell = GIE_WB_P(ux, uy, uz);
            const bool isin = t < 8 ? ((inv8 >> t) & 1u) != 0u : (L.flag[cell] & GIE_WB_INVOL) != 0u;
            const uint64_t pr = isin ? L.pair[cell] : 0ull;
            if (pr & GIE_PAIR_NEW) hitm |= 1u << t;
        }
        const bool anyhit = __ballot(hitm != 0u) != 0ull;
        /* stage 1 */
        int32_t fx = 1;
        if (act) fx = gie_axchg32(&c.wb_flag[(round + 1) & 1][ns], (int32_t)1);
        if (anyhit) {
#pragma unroll
            for (int t = 0; t < 14; t++) {
                old[t] = 0;
                if ((hitm >> t) & 1u) {
                    const int ux = t < 8 ? ((la - t) & 7) : hx[t < 8 ? 0 : t - 8], uy = t < 8 ? ((lb - t) & 7) : hy[t < 8 ? 0 : t - 8], uz = t < 8 ? t : hz[t < 8 ? 0 : t - 8];
                    old[t] = gie_amin64(&c.lprop[gie_bdr_index(c, g0[0] + ux - c.pvt[0], g0[1] + uy - c.pvt[1], g0[2] + uz - c.pvt[2])], L.pair[GIE_WB_P(ux, uy, uz)] & ~GIE_PAIR_NEW);
                }
            }
        }
        const bool firstact = act && fx == 0;
        const unsigned long long fm = __ballot(firstact);
        int mine = 0;
#pragma unroll
        for (int t = 0; t < 14; t++) if (((hitm >> t) & 1u) && old[t] == GIE_NOPROP) mine++; else hitm &= ~(1u << t);
        int incl = mine;                                    /* inclusive prefix sum over the lanes */
        if (anyhit) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        }
        const int total = anyhit ? __shfl(incl, 63) : 0;
        /* stage 2: lane 0 appends to the list of voxels, lane 1 to the list of blocks */
        int abase = 0;
        {   /* (one atomic instruction: see gie_wave_a_block) */
            int32_t *ap = nullptr; int av = 0;
            if (lane == 0 && total > 0) { ap = &c.cnt[GIE_CNT_INL]; av = total; }
            if (lane == 1 && fm != 0ull) { ap = &c.lvlb_next[round + 1]; av = __popcll(fm); }
            if (ap != nullptr) abase = gie_aadd32(ap, av);
        }
        const int qb0 = __shfl(abase, 0), ab0 = __shfl(abase, 1);
        if (firstact) gie_st(&c.wb_list[(round + 1) & 1][ab0 + __popcll(fm & ((1ull << lane) - 1ull))], (int32_t)ns);
        if (total > 0) {
            int qb = qb0 + incl - mine;
#pragma unroll
            for (int t = 0; t < 14; t++) if ((hitm >> t) & 1u) {
                const int ux = t < 8 ? ((la - t) & 7) : hx[t < 8 ? 0 : t - 8], uy = t < 8 ? ((lb - t) & 7) : hy[t < 8 ? 0 : t - 8], uz = t < 8 ? t : hz[t < 8 ? 0 : t - 8];
                if (qb < c.qcap_c) gie_st(&c.qc[1][qb], (int32_t)gie_lid(c, g0[0] + ux - c.pvt[0], g0[1] + uy - c.pvt[1], g0[2] + uz - c.pvt[2])); else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
                qb++;
            }
        }
    }
    {
        int sv = nvis;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sv += __shfl_xor(sv, o);
        if (lane == 0 && sv > 0) GIE_VIS_ADD(&c.lvlb_vis[round], sv);
    }
    GIE_WPROF_DRAIN();
    GIE_WPROF_MARK(8);
    GIE_WPROF_ADD(15, 1);
    GIE_WPROF_END(8);
    gie_wave_sync();                                       /* the LDS block is reused for the wave's next block */
}

__device__ __forceinline__ void gie_wave_b_run(const gie_ctx &c, gie_gridbar &gb, gie_wb_tile *tiles)
{
    const bool boss = (blockIdx.x == 0 && threadIdx.x == 0);
    const int n = gie_clampi(gie_ld(&c.cnt[GIE_CNT_B]), c.qcap_ab);
    if (boss) { c.cnt[GIE_CNT_FRONT_B] = n; c.cnt[GIE_CNT_SEED_C] = gie_ld(&c.cnt[GIE_CNT_C]); }
    if (n == 0 || gb.failed) return;               /* same n everywhere */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    /* the seeds mark themselves in the plane round 0 reads, their blocks are its active blocks (a voxel that was appended twice
     * marks itself twice: the frontier is a set) */
    for (int e = blockIdx.x * GIE_WAVE_THREADS + threadIdx.x; e < n; e += gridDim.x * GIE_WAVE_THREADS) {
        const gie_vaddr a = gie_ld(&c.qb_a[e]);
        bool first = false;
        if (a >= 0) {
            gie_st(&c.g_prop[a], (uint64_t)0);
            first = gie_axchg32(&c.wb_flag[0][(int)(a >> 9)], (int32_t)1) == 0;
        }
        gie_list_append_wave(c.wb_list[0], &c.lvlb_next[0], first, (int32_t)(a >> 9));
    }
    gie_grid_sync(gb, c);
    GIE_TS2(4, n);
    int round = 0;
    while (!gb.failed && round < GIE_MAX_LEVELS - 2) {
        /* (the round's length and the wave's first list entry in one round trip: see gie_wave_c_run) */
        const int32_t *list = c.wb_list[round & 1];
        const int i0 = (int)blockIdx.x + (int)gridDim.x * wave;
        const int nt = gie_ld(&c.lvlb_next[round]);
        int snext = gie_ld(&list[i0 < c.max_blocks ? i0 : 0]);
        if (nt <= 0) break;                        /* same everywhere */
        if (wave < GIE_WB_WAVES) {
            for (int i = i0; i < nt; i += (int)gridDim.x * GIE_WB_WAVES) {      /* (spread over the workgroups first) */
                const int sl = snext;
                const int in = i + (int)gridDim.x * GIE_WB_WAVES;
                if (in < nt) snext = gie_ld(&list[in]);
                gie_wave_b_block(c, gb, tiles[wave], sl, round, lane);
            }
        }
        gie_grid_sync(gb, c, &c.lvlb_vis[round]);
        GIE_TS2(6, nt);
        round++;
    }
    if (round >= GIE_MAX_LEVELS - 2 && boss) gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
    /* the stores into the volume: the smallest proposal of the whole wave, stored unconditionally (wave_core.cuh:336-346), applied
     * when wave C starts; a voxel that is not a seed of wave C yet becomes one */
    {
        const int ni = gie_clampi(gie_ld(&c.cnt[GIE_CNT_INL]), c.qcap_c);
        for (int e = blockIdx.x * GIE_WAVE_THREADS + threadIdx.x; e < ni; e += gridDim.x * GIE_WAVE_THREADS) {
            const int nid = gie_ld(&c.qc[1][e]);
            const int x = nid % c.X, y = (nid / c.X) % c.Y, z = nid / (c.X * c.Y);
            uint64_t *lp = &c.lprop[gie_bdr_index(c, x, y, z)];
            gie_st(&c.cand[1][nid], gie_ld(lp));
            gie_st(lp, (uint64_t)GIE_NOPROP);
            const uint32_t w = gie_ld(&c.wl[nid]);
            const bool push = !(w == GIE_WL_SEED(c) || w == GIE_WL_PUSHED(c));
            if (push) gie_st(&c.wl[nid], GIE_WL_PUSHED(c));
            gie_push32_wave(c, c.qc[0], &c.cnt[GIE_CNT_C], c.qcap_c, push, nid);
        }
    }
    if (boss) {
        int lv = 0; long long vis = 0;
        for (int l = 0; l < round; l++) { const int v = gie_ld(&c.lvlb_vis[l]); if (v > 0) { lv++; vis += v; } }
        c.cnt[GIE_CNT_VIS_B] = (int)vis; c.cnt[GIE_CNT_LVL_B] = lv;
        *reinterpret_cast<long long *>(&c.cnt[GIE_CNT_TOT_B]) += vis;
    }
}

/* ------------------------------------------------------------------ wave C: tile rounds */
/* lower_inside (wave_core.cuh:353-393) in the canonical TILE-ROUND schedule: in a round every ACTIVE 8x8x8
 * tile — one that holds proposals from the round before (or seeds) — is taken by ONE WAVE: its pairs, its proposals and the
 * pairs of the one-voxel halo around it go into LDS (one batch of loads), a level-synchronous BFS runs inside the tile to
 * exhaustion out of LDS (64-bit LDS atomic minima, wave barriers, no memory traffic: the reference's BFS_in_block idea),
 * what it proposes to voxels of a neighbouring tile is min-resolved in that voxel's slot of the other candidate plane (agent
 * atomics) and activates the neighbour for the next round.  A flood front needs one grid barrier per TILE it crosses, not
 * per voxel, and an entry's seven pair reads are LDS reads of a block that was fetched as whole rows.
 *   planes: round r reads cand[(r+1) & 1] (the seeds come in cand[1]) and proposes into cand[r & 1];
 *   tiles:  list wc_list[r & 1] with lvl_next[r] entries, membership flag wc_flag[r & 1][tile] (cleared by the wave that
 *           takes the tile, set by whoever activates it for round r + 2);
 *   rule:   a proposal replaces a pair on a strict distance improvement over the value at the start of the (sub-)level,
 *           among proposals the smaller (dist, parent) wins; the seeds of round 0 are assignments. */
#define GIE_WC_WAVES GIE_WAVE_SLOTS(10)                                   /* waves of a workgroup that take tiles: 16 KB of LDS each */
/* the tile WITH its one-voxel halo, 10 x 10 x 10: cell (ex, ey, ez), each in -1 .. 8, at GIE_WC_P; the edges and corners are not used */
struct gie_wc_tile { uint64_t pair[1000], prop[1000]; };                  /* 16 KB */
#define GIE_WC_P(ex, ey, ez) ((ex) + 10 * (ey) + 100 * (ez) + 111)

/* Round 6: the BFS inside the tile by instruction count.  A wavefront alone on its SIMD issues an instruction every ~5 cycles: what
 * a level costs is its instructions, not its LDS round trips (rounds 2-5: pending list, entry list, returning LDS atomics, a branch
 * per direction for "inside the tile or halo, inside the volume or not": ~1 100 instructions = 2.5 us per level, and a flood crosses
 * a tile in up to 25 levels — the projective lidar workloads: 10 M visits in 50 rounds of 35 us).  Now
 *  - every lane OWNS eight voxels of the tile, one per z-layer, placed as a Latin cube — layer j: x = (lane & 7) - j,
 *    y = (lane >> 3) - j (mod 8) — so that the voxels of ANY axis-aligned plane of the tile (a flood front's usual shape) belong to
 *    64 different lanes;
 *  - the halo lives INSIDE the LDS arrays (10 x 10 x 10 cells): the six neighbours of a voxel are six fixed offsets, a proposal
 *    across the border is the same LDS minimum as one inside, and the halo's proposal cells go to the other candidate plane in one
 *    pass when the tile is through.  A cell outside the volume holds distance 0 (never improved, never proposed to); in round 0 a
 *    halo cell inside the volume holds the largest distance (every proposal is sent on: see below);
 *  - a level is (a) every lane looks at the proposal cells of its own eight voxels and takes the ones that improve (the same test as
 *    before: strictly nearer than the voxel's pair at the start of the level), (b) every lane expands the voxels it took, one after
 *    the other (one for a plane front): d(neighbour) = d(voxel) -+ 2 c + 1 per direction, seven LDS reads, minima without a return
 *    value.  About 250 instructions.
 * What a level does — which proposals exist, who wins, what is counted as a visit — is the canonical schedule's as before. */
__device__ __forceinline__ void gie_wave_c_tile(const gie_ctx &c, gie_gridbar &gb, gie_wc_tile &L, const int t, const int round, const int lane)
{
    const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
    const int x0 = tx * 8, y0 = ty * 8, z0 = tz * 8;
    const int la = lane & 7, lb = lane >> 3;
    uint64_t *const rd = c.cand[(round + 1) & 1], *const wr = c.cand[round & 1];
    const int plane = c.X * c.Y;
    GIE_WPROF_DECL;
    /* ---- one batch of loads: the lane's eight voxels (pair, proposal, type), the halo's pairs, the tile's `ucol` bytes and block slots */
    int curd[8];                                          /* the distances of the lane's eight voxels (the pairs live in L.pair) */
    uint64_t pv[8];                                       /* ... their pairs as loaded (for `_edt_D` of an UNKNOWN voxel that changes) */
    uint64_t tys = 0;                                     /* ... and their types, for the commit of what changes */
    /* halo face f (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z), the lane's position (la, lb) on it */
    const int hcell[6] = { GIE_WC_P(-1, la, lb), GIE_WC_P(8, la, lb), GIE_WC_P(la, -1, lb), GIE_WC_P(la, 8, lb), GIE_WC_P(la, lb, -1), GIE_WC_P(la, lb, 8) };
    const int hx[6] = { x0 - 1, x0 + 8, x0 + la, x0 + la, x0 + la, x0 + la };
    const int hy[6] = { y0 + la, y0 + la, y0 - 1, y0 + 8, y0 + lb, y0 + lb };
    const int hz[6] = { z0 + lb, z0 + lb, z0 + lb, z0 + lb, z0 - 1, z0 + 8 };
    unsigned hin = 0;                                     /* my halo cells inside the volume */
    const bool lz = gie_ld(&c.tlazy[t]) != 0;              /* "lazy pairs": the tile's pairs are not in the plane */
    {
        uint64_t cv[8], hv[6];
        unsigned vin = 0;                                 /* my voxels inside the volume (kept as bits in a register: sixteen lane masks held across the loads cost more scalar registers than the kernel has) */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int x = x0 + ((la - j) & 7), y = y0 + ((lb - j) & 7), z = z0 + j;
            const bool in = x < c.X && y < c.Y && z < c.Z;
            const int id = in ? z * plane + y * c.X + x : 0;
            vin |= (in ? 1u : 0u) << j;
            /* ("lazy pairs": a flagged tile's pairs are derived from its batch obstacles — the flag is the tile's, the branch wave-uniform) */
            pv[j] = lz ? gie_pair_of_bcoc(c, gie_ld(&c.bcoc_lazy[id]), x, y, z) : gie_ld(&c.pair[id]);
            cv[j] = gie_ld(&rd[id]);
            tys |= (uint64_t)(uint8_t)c.glb_type[id] << (8 * j);
        }
#pragma unroll
        for (int f = 0; f < 6; f++) {
            const bool in = gie_in_loc(c, hx[f], hy[f], hz[f]);
            const int hid = in ? gie_lid(c, hx[f], hy[f], hz[f]) : 0;
            /* (the neighbour tile's flag, the same for the face's 64 lanes; a neighbour that is putting its pairs into the plane right now —
             * it runs in this round too — takes its flag away when they are there: a reader sees the flag and derives the Mark-time pair,
             * which is what the halo is for, or sees none and finds the pairs) */
            const bool hlz = in && gie_ld(&c.tlazy[gie_tile_index(c, hx[f], hy[f], hz[f])]) != 0;
            hv[f] = hlz ? gie_pair_of_bcoc(c, gie_ld(&c.bcoc_lazy[hid]), hx[f], hy[f], hz[f]) : gie_ld(&c.pair[hid]);
            hin |= (in ? 1u : 0u) << f;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) if (!((vin >> j) & 1u)) { pv[j] = 0ull; cv[j] = GIE_NOPROP; }      /* a voxel outside the volume: distance 0, never improved */
        /* (the pre-read of a neighbour's pair only drops proposals that cannot improve: values only decrease; a halo pair is the
         * neighbour's at the start of the round.  NOT in round 0: the neighbour may be a seed its own tile is about to ASSIGN, and the
         * seed wave B leaves on an unknown face voxel — accepted against the batch distance, wave_core.cuh:334 — can lie ABOVE the
         * stale pair the plane still holds for it: a proposal between the two was dropped here and the voxel kept the seed's
         * distance (round-4 fuzz, seed 83 #63).  Sent on, it meets the assigned pair in round 1 — what the sequential schedule does.) */
#pragma unroll
        for (int f = 0; f < 6; f++) { if (!((hin >> f) & 1u)) hv[f] = 0ull; else if (round == 0) hv[f] = GIE_NOPROP; }
        if (lane == 0) gie_st(&c.wc_flag[round & 1][t], (int32_t)0);      /* may be activated again (for round + 2) from now on */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ex = (la - j) & 7, ey = (lb - j) & 7;
            const int v = GIE_WC_P(ex, ey, j);
            curd[j] = gie_pair_dist(pv[j]);
            L.pair[v] = pv[j]; L.prop[v] = cv[j];
            if (cv[j] != GIE_NOPROP) {
                gie_st(&rd[(z0 + j) * plane + (y0 + ey) * c.X + (x0 + ex)], (uint64_t)GIE_NOPROP);      /* consumed */
                if (round == 0) curd[j] = 0x400000;                      /* the seeds of round 0 are assignments: any proposal is "nearer" */
            }
        }
#pragma unroll
        for (int f = 0; f < 6; f++) { L.pair[hcell[f]] = hv[f]; L.prop[hcell[f]] = GIE_NOPROP; }
    }
    /* the z-column (x0 + la, y0 + lb)'s byte of `ucol` (its eight voxels belong to eight lanes: the column's lane writes the byte back) */
    const bool colin = x0 + la < c.X && y0 + lb < c.Y;
    uint8_t *const ucp = &c.ucol[gie_ucol_index(c, colin ? x0 + la : 0, colin ? y0 + lb : 0, z0)];
    const unsigned ub = gie_ld(ucp);
    int myslot = -1;                                      /* lanes 0-7: the slot of block (lane & 1, lane >> 1 & 1, lane >> 2) of the (at most) 2x2x2 global blocks the tile overlaps */
    if (c.fused && lane < 8) {
        const int bx = ((x0 + c.pvt[0]) >> 3) + (lane & 1) - c.tb0[0], by = ((y0 + c.pvt[1]) >> 3) + ((lane >> 1) & 1) - c.tb0[1], bz = ((z0 + c.pvt[2]) >> 3) + (lane >> 2) - c.tb0[2];
        if (bx < c.tdim[0] && by < c.tdim[1] && bz < c.tdim[2]) myslot = c.blk_tab[(bz * c.tdim[1] + by) * c.tdim[0] + bx];
    }
    gie_wave_sync();
    GIE_WPROF_MARK(16);                                                  /* 16: the tile and its halo into LDS */
    /* ---- BFS inside the tile */
    unsigned dirty = 0;                                   /* my voxels whose pair has changed */
    int nvis = 0;
    /* the closest obstacle of local voxel (x0, y0, z0) relative to it is (wave-range coordinate) + off */
    const int off0 = c.upvt[0] - c.pvt[0] - x0, off1 = c.upvt[1] - c.pvt[1] - y0, off2 = c.upvt[2] - c.pvt[2] - z0;
    const int emax = c.empty_value;
    unsigned long long wq0 = GIE_WPROF_CLK(), wq_merge = 0, wq_exp = 0; int wq_pass = 0;
    for (;;) {
        unsigned tk = 0;                                  /* (a) my voxels that take their proposal in this level */
        {
            uint64_t cd[8];
#pragma unroll
            for (int j = 0; j < 8; j++) cd[j] = L.prop[GIE_WC_P((la - j) & 7, (lb - j) & 7, j)];       /* eight reads in flight */
#pragma unroll
            for (int j = 0; j < 8; j++) L.prop[GIE_WC_P((la - j) & 7, (lb - j) & 7, j)] = GIE_NOPROP;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                /* "no proposal" carries the largest distance there is: never below a voxel's own — except a seed's of round 0 (0x400000
                 * above), and a seed has its proposal */
                const int pd = gie_pair_dist(cd[j]);
                const bool take = pd < curd[j];
                curd[j] = take ? pd : curd[j];
                tk |= (take ? 1u : 0u) << j;
                if (take) L.pair[GIE_WC_P((la - j) & 7, (lb - j) & 7, j)] = cd[j];
            }
        }
        nvis += __popc(tk);
        dirty |= tk;
        if (__ballot(tk != 0u) == 0ull) break;            /* wave-uniform */
        GIE_WPROF_ADD(22, 1);
        gie_wave_sync();
        { const unsigned long long n_ = GIE_WPROF_CLK(); wq_merge += n_ - wq0; wq0 = n_; }
        while (__ballot(tk != 0u) != 0ull) {              /* (b) expand them: as many trips as the busiest lane took voxels */
            wq_pass++;
            if (tk != 0u) {
                const int ez = __ffs((int)tk) - 1;
                tk &= tk - 1u;
                const int ex = (la - ez) & 7, ey = (lb - ez) & 7;
                uint64_t *const pb = &L.pair[GIE_WC_P(ex, ey, ez) - 100];           /* (non-negative offsets from here: -z 0, -y 90, -x 99, self 100, +x 101, +y 110, +z 200) */
                const uint32_t *const ph = reinterpret_cast<const uint32_t *>(pb) + 1;   /* a pair's distance is in its upper word */
                const uint64_t own = pb[100];
                const uint32_t s0 = ph[2 * 99], s1 = ph[2 * 101], s2 = ph[2 * 90], s3 = ph[2 * 110], s4 = ph[0], s5 = ph[2 * 200];
                const uint64_t par = gie_pair_par(own);
                int cw[3];
                gie_unpack_wr(par, &cw[0], &cw[1], &cw[2]);
                const int cx = cw[0] + off0 - ex, cy = cw[1] + off1 - ey, cz = cw[2] + off2 - ez;
                /* |closest obstacle - neighbour|^2 = |closest obstacle - voxel|^2 +- 2 c + 1; 32 bits: wave-range coordinates are below 2^14 */
                const int d0 = cx * cx + cy * cy + cz * cz + 1;
                const int d[6] = { d0 + 2 * cx, d0 - 2 * cx, d0 + 2 * cy, d0 - 2 * cy, d0 + 2 * cz, d0 - 2 * cz };
                const uint32_t s[6] = { s0, s1, s2, s3, s4, s5 };
                const int po[6] = { 99, 101, 90, 110, 0, 200 };
                const uint32_t plo = (uint32_t)par, phi = (uint32_t)(par >> 32);
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    /* a proposal that cannot improve is dropped (values only decrease); cells outside the volume hold distance 0 */
                    if (d[k] < emax && (uint32_t)d[k] < (s[k] >> 10)) {
                        const uint64_t key = ((uint64_t)(((uint32_t)d[k] << 10) | phi) << 32) | plo;
                        (void)__hip_atomic_fetch_min(&pb[1000 + po[k]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        gie_wave_sync();
        { const unsigned long long n_ = GIE_WPROF_CLK(); wq_exp += n_ - wq0; wq0 = n_; }
    }
    GIE_WPROF_ADD(24, wq_merge); GIE_WPROF_ADD(25, wq_exp); GIE_WPROF_ADD(26, wq_pass);
    /* ---- what the tile proposes to its neighbours' voxels: the halo's proposal cells into the other candidate plane */
    unsigned xmask = 0;                                   /* neighbour tiles that received a proposal */
    {
        uint64_t hp[6];
#pragma unroll
        for (int f = 0; f < 6; f++) hp[f] = L.prop[hcell[f]];
#pragma unroll
        for (int f = 0; f < 6; f++)
            if (hp[f] != GIE_NOPROP) { gie_amin64(&wr[hz[f] * plane + hy[f] * c.X + hx[f]], hp[f]); xmask |= 1u << f; }      /* (only cells inside the volume receive proposals) */
    }
    GIE_WPROF_SUBSTART();
    GIE_WPROF_DRAIN();
    GIE_WPROF_SUB(27);
    GIE_WPROF_MARK(16);                                                  /* 17: levels inside the tile (drained: the proposals across the border) */
    /* ---- neighbour tiles that received a proposal take part in the next round: the exchanges on their flags are issued now and
     * looked at behind the write-back */
    unsigned any6 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) if (__ballot((xmask >> k) & 1u) != 0ull) any6 |= 1u << k;   /* wave-uniform */
    const bool actl = lane < 6 && ((any6 >> lane) & 1u);                    /* lane k activates the neighbour across face k: the six in flight together */
    int nt = 0, act_old = 1;
    if (actl) {
        const int dt = (lane == 0) ? -1 : (lane == 1) ? 1 : (lane == 2) ? -c.tfd[0] : (lane == 3) ? c.tfd[0]
                     : (lane == 4) ? -c.tfd[0] * c.tfd[1] : c.tfd[0] * c.tfd[1];
        nt = t + dt;
        act_old = gie_axchg32(&c.wc_flag[(round + 1) & 1][nt], (int32_t)1);
    }
    /* ---- write back what changed (wave C is the only writer of these pairs; agent-scope: another XCD's wave takes the tile next time) */
    if (__ballot(dirty != 0u) != 0ull) {
        /* `_edt_D` is derived from the pairs (gie_ops.h gie_edt_unknown_touch): a wave that changes the pair of an UNKNOWN voxel stores
         * the value of the pair before the change and marks the index in `ucol` — once.  The eight voxels of a z-column lie with
         * eight lanes: the column's lane gathers their marks out of ballots and writes the byte, a voxel's lane pulls the byte. */
        unsigned du = 0;                                  /* my voxels that changed and are UNKNOWN */
#pragma unroll
        for (int j = 0; j < 8; j++) if (((dirty >> j) & 1u) && (int8_t)(tys >> (8 * j)) == GIE_VOX_UNKNOWN) du |= 1u << j;
        if (__ballot(du != 0u) != 0ull) {
            unsigned newbits = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned long long mj = __ballot((du >> j) & 1u);
                newbits |= (unsigned)((mj >> (((la + j) & 7) + 8 * ((lb + j) & 7))) & 1ull) << j;
                const unsigned ubc = (unsigned)__shfl((int)ub, ((la - j) & 7) + 8 * ((lb - j) & 7));
                if (((du >> j) & 1u) && !((ubc >> j) & 1u) && !gie_pair_keeps_edt(c, pv[j]))       /* marked now: the byte did not hold the bit */
                    gie_st(&c.edt[(z0 + j) * plane + (y0 + ((lb - j) & 7)) * c.X + (x0 + ((la - j) & 7))], gie_edt_of_pair(c, pv[j]));
            }
            if (colin && (newbits & ~ub) != 0u) gie_st(ucp, (uint8_t)(ub | newbits));
        }
        /* ("lazy pairs": a flagged tile that changes puts ALL its pairs into the plane, then takes its flag away) */
        unsigned dm = dirty;
        if (lz) {
#pragma unroll
            for (int j = 0; j < 8; j++) if (x0 + ((la - j) & 7) < c.X && y0 + ((lb - j) & 7) < c.Y && z0 + j < c.Z) dm |= 1u << j;
        }
        while (__ballot(dm != 0u) != 0ull) {              /* as many trips as the busiest lane has changed voxels; stores only */
            const int j = dm != 0u ? __ffs((int)dm) - 1 : 0;
            const int ex = (la - j) & 7, ey = (lb - j) & 7;
            const int x = x0 + ex, y = y0 + ey, z = z0 + j;
            int slot = -1;
            if (c.fused) {
                const int which = (((x + c.pvt[0]) >> 3) - ((x0 + c.pvt[0]) >> 3)) | ((((y + c.pvt[1]) >> 3) - ((y0 + c.pvt[1]) >> 3)) << 1) | ((((z + c.pvt[2]) >> 3) - ((z0 + c.pvt[2]) >> 3)) << 2);
                slot = __shfl(myslot, which);
            }
            if (dm != 0u) {
                const int id = z * plane + y * c.X + x;
                const uint64_t pr = L.pair[GIE_WC_P(ex, ey, j)];
                gie_st(&c.pair[id], pr);
                if (c.fused && ((dirty >> j) & 1u)) gie_commit_merged(c, id, (int8_t)(tys >> (8 * j)), slot, x, y, z, pr);
            }
            dm &= dm - 1u;
        }
        if (lz) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              /* the pairs first, the flag behind them */
            if (lane == 0) gie_st(&c.tlazy[t], (uint8_t)0);
        }
    }
    GIE_WPROF_DRAIN();
    GIE_WPROF_MARK(16);                                                  /* 18: write-back (+ the flags' round trip) */
    if (any6 != 0u) gie_list_append_wave(c.wc_list[(round + 1) & 1], &c.lvl_next[round + 1], actl && act_old == 0, nt);
    {   /* visits of the tile (one atomic per wave) */
        int s = nvis;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && s > 0) GIE_VIS_ADD(&c.lvl_vis[round], s);
    }
    GIE_WPROF_DRAIN();
    GIE_WPROF_MARK(16);                                                  /* 19: activation of the neighbours */
    GIE_WPROF_ADD(23, 1);
    GIE_WPROF_END(16);
    gie_wave_sync();                                       /* the LDS block is reused for the wave's next tile */
}

__device__ __forceinline__ void gie_wave_c_run(const gie_ctx &c, gie_gridbar &gb, const int record_seeds, gie_wc_tile *tiles)
{
    const bool boss = (blockIdx.x == 0 && threadIdx.x == 0);
    const int n = gie_clampi(gie_ld(&c.cnt[GIE_CNT_C]), c.qcap_c);
    if (boss) {
        c.cnt[GIE_CNT_FRONT_C] = n;
        if (record_seeds) { c.cnt[GIE_CNT_SEED_C] = n; c.cnt[GIE_CNT_SEED_A] = gie_ld(&c.cnt[GIE_CNT_A]); c.cnt[GIE_CNT_SEED_B] = gie_ld(&c.cnt[GIE_CNT_B]); }
    }
    if (n == 0 || gb.failed) return;           /* same n everywhere; a barrier of waves A / B that timed out: the update is incomplete (GIE_ERR_TIMEOUT) */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    /* the tiles of the seeds are the active tiles of round 0 (lvl_next[] / lvl_vis[] — one word per round — and the tile flags
     * are zero when the launch starts) */
    for (int e = blockIdx.x * GIE_WAVE_THREADS + threadIdx.x; e < n; e += gridDim.x * GIE_WAVE_THREADS) {
        const int id = gie_ld(&c.qc[0][e]);
        const int x = id % c.X, y = (id / c.X) % c.Y, z = id / (c.X * c.Y);
        const int t = gie_tile_index(c, x, y, z);
        gie_list_append_wave(c.wc_list[0], &c.lvl_next[0], gie_axchg32(&c.wc_flag[0][t], (int32_t)1) == 0, t);
    }
    gie_grid_sync(gb, c);
    GIE_TS2(12, n);
    int round = 0;
    while (!gb.failed && round < GIE_MAX_LEVELS - 2) {
        /* the round's length and the wave's first list entry in ONE round trip (the entry is only looked at when it exists; the list
         * has a word per tile and the grid is no larger than that... or the index is clamped) */
        const int32_t *list = c.wc_list[round & 1];
        const int i0 = (int)blockIdx.x + (int)gridDim.x * wave;
        const int i0c = i0 < c.tfd[0] * c.tfd[1] * c.tfd[2] ? i0 : 0;
        const int nt = gie_ld(&c.lvl_next[round]);
        int tnext = gie_ld(&list[i0c]);
        if (nt <= 0) break;                    /* same everywhere */
        if (wave < GIE_WC_WAVES) {
            for (int i = i0; i < nt; i += (int)gridDim.x * GIE_WC_WAVES) {      /* (spread over the workgroups first) */
                const int t = tnext;
                const int in = i + (int)gridDim.x * GIE_WC_WAVES;
                if (in < nt) tnext = gie_ld(&list[in]);            /* (in flight while the tile runs) */
                gie_wave_c_tile(c, gb, tiles[wave], t, round, lane);
            }
        }
        gie_grid_sync(gb, c, &c.lvl_vis[round]);
        GIE_TS2(11, nt);
        round++;
    }
    if (round >= GIE_MAX_LEVELS - 2 && boss) gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
    /* statistics once, after the wave */
    if (boss) {
        int lv = 0; long long vis = 0;
        for (int l = 0; l < round; l++) { const int v = gie_ld(&c.lvl_vis[l]); if (v > 0) { lv++; vis += v; } }
        c.cnt[GIE_CNT_VIS_C] = (int)vis; c.cnt[GIE_CNT_LVL_C] = lv;
        *reinterpret_cast<long long *>(&c.cnt[GIE_CNT_TOT_C]) += vis;
    }
}

/* Waves A and B in one launch, wave C in the next (round 6; rounds 2-5: all three in one).  The boundary between the two launches
 * stands where a grid barrier stood (wave C starts from the seeds waves A / B leave), and each kernel is compiled on its own: the
 * block routines of waves A / B need the whole register file, wave C's tile routine 196 registers — together the register
 * allocator spilled (and ROCm 7.2's stack-slot colouring gave up on the result). */
__device__ __forceinline__ gie_gridbar gie_waves_bar(const gie_ctx &c, int32_t *word, int *s_fail, int *s_vis)
{
    /* (c.bar_fault: the timeout path on purpose — a barrier that waits for one workgroup more than there are, with a short limit) */
    gie_gridbar gb = { word, 0, 0, (int)gridDim.x + (c.bar_fault ? 1 : 0), s_fail, s_vis, c.bar_fault ? (1 << 10) : GIE_BAR_SPIN_LIMIT };
    return gb;
}
__global__ __launch_bounds__(GIE_WAVE_THREADS) void k_waves_ab(const gie_ctx c)
{
    constexpr size_t lds_a = sizeof(gie_wa_tile) * GIE_WA_WAVES, lds_b = sizeof(gie_wb_tile) * GIE_WB_WAVES;
    __shared__ __attribute__((aligned(16))) unsigned char s_lds[lds_a > lds_b ? lds_a : lds_b];
    gie_wa_tile *const s_ablocks = reinterpret_cast<gie_wa_tile *>(s_lds);    /* wave A: one 8x8x8 block of the global map (+ halo) per wave */
    gie_wb_tile *const s_blocks = reinterpret_cast<gie_wb_tile *>(s_lds);     /* wave B: one 8x8x8 block of the global map (+ halo) per wave */
    __shared__ int s_fail, s_vis;
    if (GIE_GATE_CLOSED(c)) return;
    if (threadIdx.x == 0) { s_fail = 0; s_vis = 0; }
    GIE_WAVE_TIMING_RESET(true);
    gie_gridbar gb = gie_waves_bar(c, &c.cnt[GIE_CNT_BAR_C], &s_fail, &s_vis);
    {   /* nothing seeded outside the volume (the usual case of a sparse scan over a settled map): the launch ends here — same
         * counters for every workgroup, no barrier */
        const int na = gie_ld(&c.cnt[GIE_CNT_A]), nb = gie_ld(&c.cnt[GIE_CNT_B]);
        if ((na | nb) == 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                c.cnt[GIE_CNT_SEED_A] = 0; c.cnt[GIE_CNT_SEED_B] = 0; c.cnt[GIE_CNT_FRONT_B] = 0; c.cnt[GIE_CNT_SEED_C] = gie_ld(&c.cnt[GIE_CNT_C]);
            }
            return;
        }
    }
    __syncthreads();
    GIE_TS2(0, 0);
    gie_wave_a_run(c, gb, s_ablocks);   /* (ends behind the barrier of its last round: wave B starts from the queue and the counters it leaves) */
    GIE_TS2(8, 0);
    gie_wave_b_run(c, gb, s_blocks);
    GIE_TS2(9, 0);
}
__global__ __launch_bounds__(GIE_WAVE_THREADS) void k_waves_c(const gie_ctx c, const int with_ab, const int record_seeds)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_lds[sizeof(gie_wc_tile) * GIE_WC_WAVES];
    gie_wc_tile *const s_tiles = reinterpret_cast<gie_wc_tile *>(s_lds);      /* wave C: one 8x8x8 tile (+ halo) per wave */
    __shared__ int s_fail, s_vis;
    if (GIE_GATE_CLOSED(c)) return;              /* a refinement round nobody needs (gie_round_gate): same answer in every workgroup, before any barrier */
    if (threadIdx.x == 0) { s_fail = 0; s_vis = 0; }
    GIE_WAVE_TIMING_RESET(!with_ab);
    gie_gridbar gb = gie_waves_bar(c, &c.cnt[GIE_CNT_BAR_B], &s_fail, &s_vis);
    if (gie_ld(&c.cnt[GIE_CNT_BARFAIL]) != 0) gb.failed = 1;       /* a barrier of waves A / B timed out: the update is incomplete (GIE_ERR_TIMEOUT) */
    __syncthreads();
    if (!with_ab) GIE_TS2(0, 0);
    gie_wave_c_run(c, gb, record_seeds, s_tiles);                  /* (no seeds: the counters, and out — no barrier) */
    GIE_TS2(10, 0);
    GIE_WPROF_DUMP();
}

/* ------------------------------------------------------------------ exchange rounds without the host (gie_refine_dev / gie_round_end)
 * One thread.  end == 0, after a refinement round: *changed = the voxels the round seeded from its ghosts (0 when the gate kept
 * the round from running), stats[0] += 1 (rounds enqueued), stats[1] += 1 if it ran.  end == 1, after the last round of a map
 * update: stats[2] += 1 (updates), stats[3] += 1 if `go` (the all-reduced "some tile changed" word of the last round) is still set —
 * the bound on the rounds was too small for this update. */
__global__ void k_round_note(const gie_ctx c, int32_t *changed, long long *stats, const int32_t *go, const int end)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!end) {
        const bool open = !GIE_GATE_CLOSED(c);
        if (changed) *changed = open ? c.cnt[GIE_CNT_FRONT_C] : 0;
        stats[0] += 1;
        if (open) stats[1] += 1;
    } else {
        stats[2] += 1;
        if (go != nullptr && *(const volatile int32_t *)go != 0) stats[3] += 1;
    }
}

#endif /* GIE_KERNELS_HIP_H */
