/*
 * gie_platform.h — what the per-voxel logic of gie_ops.h / gie_functors.h stands on: the memory primitives (agent-scope relaxed
 * accesses and atomics), the function qualifiers and the loop pragmas — for gfx950.  The test-only sequential backend of
 * tests/emu brings its own (tests/emu/gie_platform_emu.h: plain memory, a wavefront of one lane) and includes it FIRST; the
 * product build never sees that file.
 */
#ifndef GIE_PLATFORM_H
#define GIE_PLATFORM_H
#define GIE_PLATFORM_DEFINED 1

/* agent-scope relaxed accesses: served by L2, never by a stale per-CU L1 line */
template <class T> __device__ __forceinline__ T gie_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void gie_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t gie_amin64(uint64_t *p, uint64_t v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t gie_acas64(uint64_t *p, uint64_t c, uint64_t v) { __hip_atomic_compare_exchange_strong(p, &c, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return c; }
__device__ __forceinline__ uint64_t gie_aand64(uint64_t *p, uint64_t v) { return __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t gie_axchg32(uint32_t *p, uint32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t gie_axchg32(int32_t *p, int32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t gie_aadd32(int32_t *p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t gie_aor32(int32_t *p, int32_t v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define GIE_DEV __device__ __forceinline__
#define GIE_DEVM __device__ __forceinline__
#define GIE_DEV_MEMBER __device__ __forceinline__
#define GIE_DEV_COLD __device__ __forceinline__   /* (a real call would push the kernarg context through scratch: measured 5x slower) */
#define GIE_UNROLL6 _Pragma("unroll 6")
#define GIE_UNROLL_BATCH _Pragma("unroll")
#define GIE_UNROLL _Pragma("unroll")
#define GIE_COUNT_TSKIP(c) do { } while (0)       /* (a statistic of the test backend) */

/* Append slots for the threads of a WORKGROUP that have `flag` set: ballot inside the waves, LDS prefix across them, ONE
 * atomic on the counter per workgroup (the list builders below append from thousands of waves to one counter; per-wave
 * atomics on one word serialise at ~10 ns each).  Every thread of the workgroup has to call it (block barriers inside). */
__device__ __forceinline__ int gie_wg_reserve(int32_t *counter, const bool flag)
{
    __shared__ int s_cnt[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < nw; w++) tot += s_cnt[w];
        s_base = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wave; w++) base += s_cnt[w];
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                                       /* the scratch is reused by the next call */
    return flag ? slot : -1;
}

#endif /* GIE_PLATFORM_H */
