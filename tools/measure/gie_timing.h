/*
 * gie_timing.h — MEASUREMENT builds only (tools/wave_timing.py, -DGIE_WAVE_TIMING / -DGIE_RAY_TIMING=<workgroup>): the stamps and
 * per-section clock sums the product kernels carry as no-op macros.  Not part of the product: libgie_hip.so is built without
 * either macro and this file is not included.
 */
#ifndef GIE_TIMING_H
#define GIE_TIMING_H

#if defined(GIE_WAVE_TIMING)
/* measurement only (tools/wave_timing.py): the boss thread stamps the wall clock (10 ns ticks, 24 bits) and the level size at
 * every phase boundary of waves A / B / C into the middle row of the edt plane (interior voxels: no wave writes there) */
#define GIE_TS2(tag, n) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_ts_i < 500) { \
        float *p_ = c.edt + ((size_t)(c.Z / 2) * c.Y + c.Y / 2) * c.X + 2 * g_ts_i; \
        p_[0] = (float)(wall_clock64() & 0xffffff); p_[1] = (float)((tag) * 1000000 + ((n) < 999999 ? (n) : 999999)); g_ts_i++; } } while (0)
static __device__ int g_ts_i;
/* ... and per-section clock sums of the block routines (wave A: 0-7, wave B: 8-15, wave C: 16-23): [base + i] = ticks between marks i and i + 1, [base + 6] = levels inside
 * blocks, [base + 7] = blocks */
static __device__ unsigned int g_wprof[256 * 16][32];          /* one row per (workgroup, wave): no atomics, nothing shared while the waves run */
#define GIE_WPROF_DECL unsigned long long wp_t_ = wall_clock64(), wp_s_ = 0; const unsigned long long wp_t0_ = wp_t_; int wp_i_ = 0; unsigned int *const wp_ = g_wprof[blockIdx.x * 16 + (threadIdx.x >> 6)]
#define GIE_WPROF_MARK(base) do { const unsigned long long n_ = wall_clock64(); if (lane == 0) wp_[(base) + wp_i_] += (unsigned int)(n_ - wp_t_); wp_i_++; wp_t_ = n_; } while (0)
#define GIE_WPROF_ADD(i, v) do { if (lane == 0) wp_[i] += (unsigned int)(v); } while (0)
#define GIE_WPROF_END(base) do { const unsigned int d_ = (unsigned int)(wall_clock64() - wp_t0_); if (lane == 0) { if (d_ > wp_[(base) + 4]) wp_[(base) + 4] = d_; wp_[(base) + 5] += d_; } } while (0)
#define GIE_WPROF_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GIE_WPROF_SUBSTART() wp_s_ = wall_clock64()
#define GIE_WPROF_SUB(i) do { const unsigned long long n_ = wall_clock64(); if (lane == 0) wp_[i] += (unsigned int)(n_ - wp_s_); wp_s_ = n_; } while (0)
#define GIE_WPROF_CLK() wall_clock64()
#define GIE_WPROF_DUMP() do { gie_grid_sync(gb, c); if (blockIdx.x == 0 && threadIdx.x < 32) { unsigned long long s_ = 0; \
        const bool mx_ = (threadIdx.x & 7) == 4 || (threadIdx.x & 7) == 5; \
        for (int r_ = 0; r_ < 256 * 16; r_++) { const unsigned int v_ = __hip_atomic_load(&g_wprof[r_][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (mx_) { if (64ull * v_ > s_) s_ = 64ull * v_; } else s_ += v_; __hip_atomic_store(&g_wprof[r_][threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } \
        float *p_ = c.edt + ((size_t)(c.Z / 2) * c.Y + c.Y / 2) * c.X + 2 * (g_ts_i + (int)threadIdx.x); \
        p_[0] = 0.0f; p_[1] = (float)((20 + (int)threadIdx.x) * 1000000 + (int)((s_ / 64) < 999999 ? (s_ / 64) : 999999)); } } while (0)
#define GIE_WAVE_TIMING_RESET(cond) do { if ((cond) && blockIdx.x == 0 && threadIdx.x == 0) g_ts_i = 0; } while (0)
#endif /* GIE_WAVE_TIMING */

#if defined(GIE_RAY_TIMING)
#if GIE_RAY_TIMING < 0      /* every workgroup: first and last stamp of wave 0 */
#define GIE_RTS(k) do { if (seg == 0 && lane == 0 && ((k) == 0 || (k) == 4)) c.edt[blockIdx.x * 2 + ((k) ? 1 : 0)] = (float)(wall_clock64() & 0xffffff); } while (0)
#else
#define GIE_RTS(k) do { if (blockIdx.x == GIE_RAY_TIMING && lane == 0) c.edt[seg * 8 + (k)] = (float)(wall_clock64() & 0xffffff); } while (0)
#endif
#endif /* GIE_RAY_TIMING */

#endif /* GIE_TIMING_H */
