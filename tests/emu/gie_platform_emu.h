/*
 * gie_platform_emu.h — TEST-ONLY: what gie_ops.h / gie_functors.h stand on when the sequential backend of tests/emu runs them on
 * the host: plain memory instead of agent-scope accesses, and a "wavefront" of ONE lane (a ballot is the lane's own bit, a shuffle
 * returns the lane's own value, a workgroup reservation is a counter increment), so that the wave-aggregated code of the product
 * headers runs unchanged.  Included by tests/emu/gie_emu.cpp BEFORE the product headers; never by the product.
 */
#ifndef GIE_PLATFORM_EMU_H
#define GIE_PLATFORM_EMU_H
#define GIE_PLATFORM_DEFINED 1
#define GIE_HOST_EMU 1

#include <stdint.h>

template <class T> static inline T gie_ld(const T *p) { return *p; }
template <class T> static inline void gie_st(T *p, T v) { *p = v; }
static inline uint64_t gie_amin64(uint64_t *p, uint64_t v) { uint64_t o = *p; if (v < o) *p = v; return o; }
static inline uint64_t gie_acas64(uint64_t *p, uint64_t c, uint64_t v) { uint64_t o = *p; if (o == c) *p = v; return o; }
static inline uint64_t gie_aand64(uint64_t *p, uint64_t v) { uint64_t o = *p; *p = o & v; return o; }
static inline uint32_t gie_axchg32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = v; return o; }
static inline int32_t gie_axchg32(int32_t *p, int32_t v) { int32_t o = *p; *p = v; return o; }
static inline int32_t gie_aadd32(int32_t *p, int32_t v) { int32_t o = *p; *p = o + v; return o; }
static inline int32_t gie_aor32(int32_t *p, int32_t v) { int32_t o = *p; *p = o | v; return o; }
#define GIE_DEV static inline
#define GIE_DEVM inline
#define GIE_DEV_MEMBER inline
#define GIE_DEV_COLD static
#define GIE_UNROLL6
#define GIE_UNROLL_BATCH
#define GIE_UNROLL
#define GIE_COUNT_TSKIP(c) do { (c).cnt[GIE_CNT_TSKIP] += 1; } while (0)

/* a wavefront of one lane */
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
static inline int __lane_id() { return 0; }
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_xor(T v, int) { return v; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __builtin_amdgcn_update_dpp(int, int src, int, int, int, bool) { return src; }
#define __ATOMIC_RELAXED_EMU 0
#define __HIP_MEMORY_SCOPE_AGENT 0
static inline int __hip_atomic_fetch_max(int32_t *p, int32_t v, int, int) { const int32_t o = *p; if (v > o) *p = v; return o; }
/* a workgroup of one thread: its reservation is the counter's next value */
static inline int gie_wg_reserve(int32_t *counter, const bool flag) { return flag ? (*counter)++ : -1; }

#endif /* GIE_PLATFORM_EMU_H */
