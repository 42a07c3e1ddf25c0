/*
 * gie_hip.hip — libgie_hip.so: the MI355X (gfx950) implementation of include/gie.h.
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC gie_hip.hip -o libgie_hip.so
 * One HIP stream per mapper; a frame is a fixed sequence of kernel launches with no host
 * synchronisation inside (the reference syncs ≥12 times per frame through thrust scalars,
 * SURVEY.md §2.3).
 */
#include <cstdio>
#include <unistd.h>
#include <cstring>
#include <cstdlib>
#include <string>
#include <algorithm>
#include <vector>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rocprim/rocprim.hpp>
#include "gie_kernels.hip.h"

#define GIE_NEV 8

struct be_state {
    int device;
    int num_cu;     /* workgroups of the persistent wave kernels */
    int cu_total;   /* CUs of the device */
    hipStream_t stream;
    hipEvent_t ev[GIE_NEV];
    int ev_set[GIE_NEV];
    void *scan_tmp; size_t scan_bytes;
    void *arena;                        /* GIE_ARENA_MB (placement experiments): be_arena */
    hipEvent_t copy_ev[2];              /* completion of the async D2H copies (changed-block streaming) */
    hipStream_t side;                   /* copy stream of gie_costmap_publish (created on first use) */
    hipEvent_t side_ready, side_done; int side_on, side_busy;
    /* per-kernel event profiling */
    int prof_on;
    std::vector<hipEvent_t> *pool;      /* event pool */
    std::vector<int> *pending;          /* triples (kernel id, start event idx, stop event idx) */
    int pool_used;
    int cur_id;                         /* bracket the launches currently belong to (-1: none) */
    double acc_ms[32]; int acc_n[32];
};
#include <vector>

struct be_arena { char *base; size_t size, top; int k; std::vector<long long> skew; };
static void gie_set_err(const std::string &s);
#include <string>

#define GIE_HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { gie_set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)

#include <mutex>
static std::mutex g_waves_mutex;
static hipEvent_t g_waves_event[64];
static bool g_waves_event_set[64];
static int g_live_mappers[64];          /* mappers alive per device: waves launches are chained only when there is more than one */

static int be_init(be_state *b, int device, int wave_workgroups)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { gie_set_err("no HIP device visible (libgie_hip.so needs an AMD GPU; there is no CPU fallback)"); return 1; }
    if (device < 0 || device >= n) { gie_set_err("bad device_id"); return 1; }
    b->device = device;
    if (hipSetDevice(device) != hipSuccess) { gie_set_err("hipSetDevice failed"); return 1; }
    { hipDeviceProp_t pr; b->num_cu = (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 64; }
    b->cu_total = b->num_cu;
    /* the persistent wavefront grid: HALF the compute units by default, so that two mappers (or two processes) with default settings
     * fit one device side by side (160 + 160 did not, for 3 % — C5 waves 128: 0.270, 160: 0.261, 192: 0.266, 256: 0.271 ms);
     * gie_config.wave_workgroups asks for another size (bench.py: 160 on a device of its own; ranks that share a device: 192 / ranks) */
    b->num_cu = b->num_cu >= 64 ? b->num_cu / 2 : b->num_cu;
    if (wave_workgroups > 0) b->num_cu = wave_workgroups < b->cu_total ? wave_workgroups : b->cu_total;
    {   /* the waves kernel meets at a hand-rolled grid barrier: its grid must fit the device at once.  Checked here, once,
         * against the runtime's own occupancy answer (what hipLaunchCooperativeKernel would check at every launch, at
         * +15-19 us of host time each); what it cannot guard against — another PROCESS holding compute units — ends in a
         * bounded spin and GIE_ERR_TIMEOUT (include/gie.h). */
        int per_cu = 0, per_cu_c = 0;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_waves_ab), GIE_WAVE_THREADS, 0);
        if (oe == hipSuccess) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_c, reinterpret_cast<const void *>(&k_waves_c), GIE_WAVE_THREADS, 0);
        if (per_cu_c < per_cu) per_cu = per_cu_c;
        if (oe != hipSuccess || per_cu < 1) {
            gie_set_err(std::string("the wavefront kernels cannot be resident on this device (occupancy query: ") + hipGetErrorString(oe) + ", " + std::to_string(per_cu) + " workgroups per compute unit)");
            (void)hipGetLastError();
            return 1;
        }
        if (b->num_cu > per_cu * b->cu_total) b->num_cu = per_cu * b->cu_total;
    }
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { gie_set_err("hipStreamCreate failed"); return 1; }
    for (int i = 0; i < GIE_NEV; i++) { GIE_HIP_OK(hipEventCreate(&b->ev[i])); b->ev_set[i] = 0; }
    b->scan_tmp = nullptr; b->scan_bytes = 0; b->arena = nullptr; b->side_on = 0; b->side_busy = 0;
    for (int i = 0; i < 2; i++) GIE_HIP_OK(hipEventCreateWithFlags(&b->copy_ev[i], hipEventDisableTiming));
    b->prof_on = 0; b->cur_id = -1; b->pool = new std::vector<hipEvent_t>(); b->pending = new std::vector<int>(); b->pool_used = 0;
    for (int i = 0; i < 32; i++) { b->acc_ms[i] = 0; b->acc_n[i] = 0; }
    {   /* a second mapper on this device: whatever the first one has in flight was launched unchained — let it finish once */
        std::lock_guard<std::mutex> lock(g_waves_mutex);
        if (++g_live_mappers[device & 63] == 2) (void)hipDeviceSynchronize();
    }
    return 0;
}
static void be_fini(be_state *b)
{
    { std::lock_guard<std::mutex> lock(g_waves_mutex); g_live_mappers[b->device & 63]--; }
    if (b->scan_tmp) (void)hipFree(b->scan_tmp);
    for (hipEvent_t e : *b->pool) (void)hipEventDestroy(e);
    delete b->pool; delete b->pending;
    for (int i = 0; i < GIE_NEV; i++) (void)hipEventDestroy(b->ev[i]);
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(b->copy_ev[i]);
    (void)hipStreamDestroy(b->stream);
    if (b->side_on) { (void)hipStreamSynchronize(b->side); (void)hipEventDestroy(b->side_ready); (void)hipEventDestroy(b->side_done); (void)hipStreamDestroy(b->side); }
    if (b->arena) { be_arena *a = (be_arena *)b->arena; if (a->base) (void)hipFree(a->base); delete a; b->arena = nullptr; }
}
/* Placement experiments ("placement bands"): GIE_ARENA_MB=<MiB> takes ONE allocation of that size per mapper and
 * carves every plane of 64 MiB and more out of it, the k-th such plane GIE_ARENA_SKEW[k] MiB (comma list, default 0) behind the
 * 2 MiB-rounded end of the one before; smaller buffers and whatever does not fit go through hipMalloc as usual. */
static be_arena *be_arena_of(be_state *b)
{
#if !defined(GIE_TEST_HOOKS)
    (void)b; return nullptr;              /* (a measurement aid: test builds only) */
#else
    static const char *e = getenv("GIE_ARENA_MB");
    if (!e || atoll(e) <= 0) return nullptr;
    if (!b->arena) {
        be_arena *a = new be_arena();
        a->size = (size_t)atoll(e) << 20; a->top = 0; a->k = 0; a->base = nullptr;
        if (const char *sk = getenv("GIE_ARENA_SKEW")) { std::string t(sk); size_t i = 0; while (i < t.size()) { size_t j = t.find(',', i); if (j == std::string::npos) j = t.size(); a->skew.push_back(atoll(t.substr(i, j - i).c_str())); i = j + 1; } }
        if (hipMalloc((void **)&a->base, a->size) != hipSuccess) { a->base = nullptr; a->size = 0; }
        b->arena = a;
    }
    return (be_arena *)b->arena;
#endif
}
static void *be_alloc(be_state *b, size_t bytes, bool zero)
{
    void *p = nullptr;
    (void)hipSetDevice(b->device);
    be_arena *a = bytes >= ((size_t)64 << 20) ? be_arena_of(b) : nullptr;
    if (a && a->base) {
        const size_t sk = (size_t)((a->k < (int)a->skew.size() ? a->skew[a->k] : 0) << 20);
        const size_t at = ((a->top + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) + sk;
        if (at + bytes <= a->size) { p = a->base + at; a->top = at + bytes; a->k++; }
    }
    if (!p && hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    if (zero) GIE_HIP_OK(hipMemsetAsync(p, 0, bytes, b->stream));
    return p;
}
static void be_free(be_state *b, void *p)
{
    (void)hipStreamSynchronize(b->stream);
    be_arena *a = (be_arena *)b->arena;
    if (a && a->base && (char *)p >= a->base && (char *)p < a->base + a->size) return;      /* goes with the arena (be_fini) */
    (void)hipFree(p);
}
static void be_memset(be_state *b, void *p, int v, size_t bytes) { GIE_HIP_OK(hipMemsetAsync(p, v, bytes, b->stream)); }
static void be_h2d(be_state *b, void *d, const void *h, size_t bytes)
{   /* blocking like the reference's GPU_MEMCPY_H2D (cuda_macro.h:33): the caller may reuse h */
    GIE_HIP_OK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, b->stream));
    GIE_HIP_OK(hipStreamSynchronize(b->stream));
}
static void be_d2h(be_state *b, void *h, const void *d, size_t bytes)
{
    GIE_HIP_OK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b->stream));
    GIE_HIP_OK(hipStreamSynchronize(b->stream));
}
/* pinned host memory + asynchronous D2H with a completion event per staging slot */
static void *be_host_alloc(be_state *b, size_t bytes) { void *p = nullptr; (void)hipSetDevice(b->device); return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
static void be_host_free(be_state *, void *p) { (void)hipHostFree(p); }
static void be_d2h_async(be_state *b, void *h, const void *d, size_t bytes, int slot)
{
    GIE_HIP_OK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b->stream));
    GIE_HIP_OK(hipEventRecord(b->copy_ev[slot], b->stream));
}
static void be_wait(be_state *b, int slot) { GIE_HIP_OK(hipEventSynchronize(b->copy_ev[slot])); }
/* a copy that leaves the mapper's stream free: the side stream waits for what the mapper's stream has enqueued so far, copies into
 * pinned host memory and records its completion; be_side_copy_begin makes the mapper's stream wait for the copy before (its source
 * is about to be rewritten) */
static void be_side_on(be_state *b)
{
    if (b->side_on) return;
    (void)hipSetDevice(b->device);
    GIE_HIP_OK(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
    GIE_HIP_OK(hipEventCreateWithFlags(&b->side_ready, hipEventDisableTiming));
    GIE_HIP_OK(hipEventCreateWithFlags(&b->side_done, hipEventDisableTiming));
    b->side_on = 1;
}
static void be_side_copy_begin(be_state *b) { be_side_on(b); if (b->side_busy) GIE_HIP_OK(hipStreamWaitEvent(b->stream, b->side_done, 0)); }
static void be_side_copy(be_state *b, void *h, const void *d, size_t bytes)
{
    be_side_on(b);
    GIE_HIP_OK(hipEventRecord(b->side_ready, b->stream));
    GIE_HIP_OK(hipStreamWaitEvent(b->side, b->side_ready, 0));
    GIE_HIP_OK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b->side));
    GIE_HIP_OK(hipEventRecord(b->side_done, b->side));
    b->side_busy = 1;
}
static int be_side_copy_wait(be_state *b)
{
    if (!b->side_busy) return 0;
    const hipError_t e = hipEventSynchronize(b->side_done);
    if (e != hipSuccess) { gie_set_err(std::string("hipEventSynchronize: ") + hipGetErrorString(e)); return 1; }
    return 0;
}
static void *be_stream_handle(be_state *b) { return (void *)b->stream; }
static int be_sync(be_state *b)
{
    hipError_t e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) { gie_set_err(std::string("hipStreamSynchronize: ") + hipGetErrorString(e)); return 1; }
    e = hipGetLastError();
    if (e != hipSuccess) { gie_set_err(std::string("HIP error: ") + hipGetErrorString(e)); return 1; }
    return 0;
}
/* stage boundaries (us_ogm .. us_merge of gie_frame_stats): recorded while profiling is on — every
 * event is a marker packet the next kernel has to wait for */
static void be_time(be_state *b, int i)
{
    if (!b->prof_on) { b->ev_set[i] = 0; return; }
    GIE_HIP_OK(hipEventRecord(b->ev[i], b->stream)); b->ev_set[i] = 1;
}
static void be_times(be_state *b, float *ogm, float *fuse, float *edt, float *merge)
{
    float *out[4] = { ogm, fuse, edt, merge };
    for (int k = 0; k < 4; k++) {
        float ms = 0.f;
        if (b->ev_set[2 * k] && b->ev_set[2 * k + 1] && hipEventElapsedTime(&ms, b->ev[2 * k], b->ev[2 * k + 1]) == hipSuccess) *out[k] = ms * 1000.f;
        else *out[k] = 0.f;
    }
}

/* Per-kernel timing: the start / stop events ride on the kernel's own dispatch packet
 * (hipExtLaunchKernelGGL), so a profiled map update has no marker packets between its kernels
 * (event records between the launches cost the map update +18 %) and a kernel's time is its
 * execution, as rocprofv3 reports it, without the dispatch gap. */
static int be_prof_event(be_state *b)
{
    if (b->pool_used == (int)b->pool->size()) { hipEvent_t e; GIE_HIP_OK(hipEventCreate(&e)); b->pool->push_back(e); }
    return b->pool_used++;
}
/* GIE_TRACE_LAUNCHES=1: every kernel launch is announced on stderr (process id, kernel) and waited for — the last line a
 * process prints before a device fault names the kernel */
static int be_trace_on() { static const int on = GIE_SWITCH("GIE_TRACE_LAUNCHES", 0); return on; }
#define GIE_LAUNCH(b, kern, grid, block, lds, ...) do { \
        if (be_trace_on()) { fprintf(stderr, "[%d] launch %s\n", (int)getpid(), #kern); fflush(stderr); } \
        GIE_LAUNCH_(b, kern, grid, block, lds, __VA_ARGS__); \
        if (be_trace_on()) { hipError_t e_ = hipStreamSynchronize((b)->stream); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", (int)getpid(), #kern, hipGetErrorString(e_)); fflush(stderr); } } \
    } while (0)
#define GIE_LAUNCH_(b, kern, grid, block, lds, ...) do { \
        if ((b)->prof_on && (b)->cur_id >= 0) { \
            const int e0_ = be_prof_event(b), e1_ = be_prof_event(b); \
            hipExtLaunchKernelGGL(kern, grid, block, lds, (b)->stream, (*(b)->pool)[e0_], (*(b)->pool)[e1_], 0, __VA_ARGS__); \
            (b)->pending->push_back((b)->cur_id); (b)->pending->push_back(e0_); (b)->pending->push_back(e1_); \
        } else hipLaunchKernelGGL(kern, grid, block, lds, (b)->stream, __VA_ARGS__); \
    } while (0)
static void be_prof_resolve(be_state *b)
{
    if (b->pending->empty()) { b->pool_used = 0; return; }
    GIE_HIP_OK(hipStreamSynchronize(b->stream));
    for (size_t i = 0; i + 2 < b->pending->size(); i += 3) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, (*b->pool)[(*b->pending)[i + 1]], (*b->pool)[(*b->pending)[i + 2]]) == hipSuccess) {
            b->acc_ms[(*b->pending)[i]] += ms;
        }
    }
    b->pending->clear(); b->pool_used = 0;
}
static void be_prof(be_state *b, int id, int end)
{
    if (!b->prof_on) return;
    if (!end) {
        if (b->pool_used > 4000) be_prof_resolve(b);
        b->cur_id = id;
    } else { b->cur_id = -1; b->acc_n[id] += 1; }       /* acc_n counts brackets, acc_ms sums the kernels launched inside them */
}
static void be_prof_enable(be_state *b, int on) { be_prof_resolve(b); b->prof_on = on; b->cur_id = -1; }
static void be_prof_collect(be_state *b, float *ms, int *n, int num)
{
    be_prof_resolve(b);
    for (int i = 0; i < num; i++) { ms[i] = (float)b->acc_ms[i]; n[i] = b->acc_n[i]; b->acc_ms[i] = 0; b->acc_n[i] = 0; }
}

template <class F> static void be_vox(be_state *b, const gie_ctx &c, const F &f)
{
    dim3 blk(GIE_VOX_BX, GIE_VOX_BY, 1);
    dim3 grd((c.X + GIE_VOX_BX - 1) / GIE_VOX_BX, (c.Y + GIE_VOX_BY - 1) / GIE_VOX_BY, (c.Z + GIE_VOX_ZPER - 1) / GIE_VOX_ZPER);
    GIE_LAUNCH(b, k_voxz<F>, grd, blk, 0, c, f);
}
template <class F> static void be_lin(be_state *b, const gie_ctx &c, const F &f, int n)
{
    if (n <= 0) return;
    GIE_LAUNCH(b, k_lin<F>, dim3((n + 255) / 256), dim3(256), 0, c, f, n);
}
static void be_flush_clear(be_state *b, const gie_ctx &c, const op_pair_flush &f, int n, const gie_clear_list &l)
{
    const int nclr = gie_clear_total_wgs(l);
    GIE_LAUNCH(b, k_flush_clear, dim3(nclr + (n + 255) / 256), dim3(256), 0, c, f, n, l, nclr);
}
/* can the scan stay where it is (be_labels with in_place: only the blocks are flagged, gie_fuse reads the plane itself)? */
static int be_labels_in_place_ok(const gie_ctx &c, const int8_t *labels)
{ return (c.X & 15) == 0 && !c.for_motion_planner && ((uintptr_t)labels & 15) == 0; }
static void be_labels(be_state *b, const gie_ctx &c, const int8_t *labels, int in_place = 0)
{
    if ((c.X & 15) == 0 && !c.for_motion_planner && ((uintptr_t)labels & 15) == 0) {
        const int nvec = c.N >> 4;
        if (in_place) GIE_LAUNCH(b, k_labels16<false>, dim3((nvec + 255) / 256), dim3(256), 0, c, labels, nvec);
        else GIE_LAUNCH(b, k_labels16<true>, dim3((nvec + 255) / 256), dim3(256), 0, c, labels, nvec);
    } else { op_classify_labels op; op.labels = labels; be_vox(b, c, op); }
}
static void be_clear(be_state *b, const gie_clear_list &l, const int32_t *gate = nullptr)
{
    if (l.n > 0) GIE_LAUNCH(b, k_clear, dim3(gie_clear_total_wgs(l)), dim3(256), 0, l, gate);
}
static void be_round_note(be_state *b, const gie_ctx &c, int32_t *changed, long long *stats, const int32_t *go, int end)
{
    GIE_LAUNCH(b, k_round_note, dim3(1), dim3(64), 0, c, changed, stats, go, end);
}
/* allocHashTB + block table (see k_cell_alloc) */
/* fuse_list_ntile > 0: the fuse tile list (op_fuse_list over that many tiles) is built in the block-initialisation launch */
static void be_block_alloc(be_state *b, const gie_ctx &c, int ncell, int32_t *, int clear_list, int fuse_list_ntile = 0)
{
    if (clear_list) GIE_HIP_OK(hipMemsetAsync(&c.cnt[GIE_CNT_NEWLIST], 0, sizeof(int32_t), b->stream));
    GIE_LAUNCH(b, k_cell_alloc, dim3((ncell + 255) / 256), dim3(256), 0, c, ncell);
    static const int imult = GIE_SWITCH("GIE_INIT_MULT", 4);
    const int ninit = b->cu_total * imult;
    int nfl = (fuse_list_ntile + 255) / 256; if (nfl > 2 * b->cu_total) nfl = 2 * b->cu_total;
    GIE_LAUNCH(b, k_block_init_list, dim3(ninit + nfl), dim3(256), 0, c, ninit, fuse_list_ntile);
}
static void be_free_rays(be_state *b, const gie_ctx &c, const float *g, int n)
{
    if (n <= 0) return;
    GIE_LAUNCH(b, k_free_rays, dim3((n + 63) / 64), dim3(64 * GIE_RAY_SEGS), 0, c, g, n, gie_ray_max_steps(c));
}
static void be_exclusive_scan(be_state *b, const int32_t *flag, int32_t *rank, int n)
{
    size_t bytes = 0;
    (void)hipSetDevice(b->device);
    GIE_HIP_OK(rocprim::exclusive_scan(nullptr, bytes, flag, rank, 0, (size_t)n, rocprim::plus<int32_t>(), b->stream));
    if (bytes > b->scan_bytes) {
        if (b->scan_tmp) { (void)hipStreamSynchronize(b->stream); (void)hipFree(b->scan_tmp); }
        GIE_HIP_OK(hipMalloc(&b->scan_tmp, bytes)); b->scan_bytes = bytes;
    }
    GIE_HIP_OK(rocprim::exclusive_scan(b->scan_tmp, bytes, flag, rank, 0, (size_t)n, rocprim::plus<int32_t>(), b->stream));
}

template <int CP, int TX, int WAVES> static void gie_launch_edt_z(be_state *b, const gie_ctx &c, int full)
{
    constexpr int LP = 64 * CP;
    const size_t tile = ((size_t)c.Z * (TX + 1) + 3) & ~(size_t)3;
    const size_t lds = tile * 4 + (size_t)WAVES * LP * 8 + (size_t)LP * 2 + 16;      /* column tile + per-wave sites + plane list */
    static bool attr_done = false;
    if (!attr_done) {
        GIE_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_edt_z<CP, TX, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const int ntx = (c.X + TX - 1) / TX, ntiles = ntx * c.Y;
    int wgs = (int)((160 * 1024) / (lds + 256));            /* workgroups that fit one CU's LDS */
    if (wgs < 1) wgs = 1; if (wgs > 4) wgs = 4;
    int grid = wgs * b->cu_total; if (grid > (ntiles + 1) / 2) grid = (ntiles + 1) / 2;     /* tiles are taken in pairs */
    GIE_LAUNCH(b, (k_edt_z<CP, TX, WAVES>), dim3(grid), dim3(64 * WAVES), lds, c, ntx, ntiles, full);
}
template <int CP> static void gie_launch_edt_xz(be_state *b, const gie_ctx &c, bool zpass, int full)
{
    if (!zpass) {
        GIE_LAUNCH(b, k_edt_x<CP>, dim3((c.Y + GIE_EDTX_WAVES - 1) / GIE_EDTX_WAVES, c.Z), dim3(64 * GIE_EDTX_WAVES), 0, c);
    } else {
        gie_launch_edt_z<CP, 16, 8>(b, c, full);      /* 2 workgroups per CU overlap load / envelope / store phases */
    }
}
static void gie_launch_edt_dim(be_state *b, const gie_ctx &c, int L, bool zpass, int full)
{
    if (L <= 64) gie_launch_edt_xz<1>(b, c, zpass, full);
    else if (L <= 128) gie_launch_edt_xz<2>(b, c, zpass, full);
    else if (L <= 256) gie_launch_edt_xz<4>(b, c, zpass, full);
    else if (L <= 512) gie_launch_edt_xz<8>(b, c, zpass, full);
    else gie_launch_edt_xz<16>(b, c, zpass, full);
}
/* pass Z again over the whole volume (the passes before it are complete either way) */
static int be_edt_z_stream(be_state *b, const gie_ctx &c, int full);
static void be_edt_z(be_state *b, const gie_ctx &c, int full) { (void)be_edt_z_stream(b, c, full); gie_launch_edt_dim(b, c, c.Z, true, full); }
/* EDT_OCC::batchEDTUpdate, local_edt.cu:7-28 */
/* adaptive sweep (k_voxa): the kernel walks the list or sweeps the volume, whichever the list's
 * length calls for; staged = the functor's load1/load2/finish form; always_list = never sweep */
/* mode: 0 = the kernel walks its list or sweeps the volume (decided on the device), 1 = always the list,
 * 2 = the list or nothing (the dense form is a block-row kernel launched next to this one) */
template <bool STAGED, class F> static void be_vox_list(be_state *b, const gie_ctx &c, const F &f, const int32_t *list, int count_idx, int always_list, int lx = 64)
{
    /* workgroups per compute unit: 8 / 16 / 32 / 64 measured 0.33 / 0.27 / 0.25 / 0.25 ms for Mark on a densely known
     * volume (sweep side); the list side does not care */
    static const int mult = GIE_SWITCH("GIE_VOXA_MULT", 32);
    const dim3 g(b->cu_total * mult), t(256);
    if (lx == 32) GIE_LAUNCH(b, (k_voxa<F, STAGED, 32>), g, t, 0, c, f, list, count_idx, always_list);
    else if (lx == 16) GIE_LAUNCH(b, (k_voxa<F, STAGED, 16>), g, t, 0, c, f, list, count_idx, always_list);
    else if (lx == 8) GIE_LAUNCH(b, (k_voxa<F, STAGED, 8>), g, t, 0, c, f, list, count_idx, always_list);
    else GIE_LAUNCH(b, (k_voxa<F, STAGED, 64>), g, t, 0, c, f, list, count_idx, always_list);
}
/* Mark + commit as one sweep: its own kernel (k_markc); GIE_MARKC_GENERIC=1 keeps the staged functor sweep (tests run both) */
static void be_markc(be_state *b, const gie_ctx &c, const int32_t *list)
{
    static const int generic = GIE_SWITCH("GIE_MARKC_GENERIC", 0);
    static const int lx = GIE_SWITCH("GIE_MARKC_LX", 32);
    static const int mult = GIE_SWITCH("GIE_VOXA_MULT", 64);     /* workgroups per compute unit of the dense sweep: 16 / 32 / 48 / 64 / 96 / 128 measured 1.00 / 0.80 / 0.78 / 0.76 / 0.78 / 0.79 ms at 512^3 (round 4; 64 = exactly four virtual workgroups each) */
    if (generic) { be_vox_list<true>(b, c, op_markc(), list, GIE_CNT_TL_KNOWN, 0, lx == 16 || lx == 8 || lx == 32 ? lx : 64); return; }
    /* with lazy tiles the kernel walks the swept tiles, a wave each (k_markc): a grid that is resident at once */
    static const int lmult = GIE_SWITCH("GIE_MARKC_LGRID", 40);      /* (5 / 10 / 20 / 40 / 64 / 96 per compute unit: 0.193 / 0.186 / 0.178 / 0.174 / 0.179 / 0.180 ms on the headline — a wave per swept tile, no second tile behind it) */
    const dim3 g(b->cu_total * ((c.coc_defer && c.lazy_ok) ? lmult : mult)), t(256);
    if (lx == 16) GIE_LAUNCH(b, k_markc<16>, g, t, 0, c, list);
    else if (lx == 64) GIE_LAUNCH(b, k_markc<64>, g, t, 0, c, list);
    else GIE_LAUNCH(b, k_markc<32>, g, t, 0, c, list);
}
/* placement probe (k_place_probe): median of `reps` timed launches, ms */
static float be_place_probe(be_state *b, const gie_ctx &c, int reps, int streams = 15)
{
    static const int mult = GIE_SWITCH("GIE_VOXA_MULT", 64);     /* (the sweep's own grid: be_markc) */
    const long long ntile = (long long)c.tfd[0] * c.tfd[1] * c.tfd[2];
    const int nslot = (int)(ntile < c.max_blocks ? ntile : c.max_blocks);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    std::vector<float> ms;
    hipLaunchKernelGGL(k_place_probe, dim3(b->cu_total * mult), dim3(256), 0, b->stream, c, nslot, streams);      /* warm-up: page tables, clocks */
    for (int r = 0; r < reps; r++) {
        (void)hipEventRecord(e0, b->stream);
        hipLaunchKernelGGL(k_place_probe, dim3(b->cu_total * mult), dim3(256), 0, b->stream, c, nslot, streams);
        (void)hipEventRecord(e1, b->stream);
        (void)hipEventSynchronize(e1);
        float t = 0.f; (void)hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    std::sort(ms.begin(), ms.end());
    GIE_HIP_OK(hipMemsetAsync(c.pair, 0, (size_t)c.N * sizeof(uint64_t), b->stream));
    return ms.empty() ? -1.f : ms[ms.size() / 2];
}
/* dense (block-row) form of fuse; GIE_ROWS=0 keeps the thread-per-z-column sweep */
static int be_rows_mode() { static const int v = GIE_SWITCH("GIE_ROWS", 1); return v ? 2 : 0; }
/* fuse: one launch — the kernel walks its tile list (a wave per tile) or sweeps the volume by block rows, whichever the list's
 * length calls for; GIE_ROWS=0: the thread-per-z-column sweep of k_voxa instead of the block rows */
static void be_fuse(be_state *b, const gie_ctx &c, const int32_t *list)
{
    static const int mult = GIE_SWITCH("GIE_ROWS_MULT", 32);     /* 8 / 16 / 32 / 48 workgroups per compute unit: 0.225 / 0.215 / 0.195 / 0.19 ms at 512^3 (round 4: with the byte-parallel filter the kernel is short enough for the tail of its last workgroups to show; 32 = one virtual wavefront per wavefront) */
    if (be_rows_mode()) {
        if (c.pntcld_mode) GIE_LAUNCH(b, k_fuse_rows<true>, dim3(b->cu_total * mult), dim3(256), 0, c, op_fuse(), list);
        else GIE_LAUNCH(b, k_fuse_rows<false>, dim3(b->cu_total * mult), dim3(256), 0, c, op_fuse(), list);
    }
    else be_vox_list<true>(b, c, op_fuse(), list, GIE_CNT_TL_FUSE, 0);
}
/* obtainFrontiers: the voxels on the six faces of the volume one per lane (an 8x8 patch per wave) with the tile summary + tile
 * list in the first workgroups of the same launch, then a wave per listed tile for the voxels off the faces (tile + halo
 * staged in LDS) */
static void be_frontier_tiles(be_state *b, const gie_ctx &c, const int32_t *known, int known_idx, const int32_t *list, int count_idx)
{
    gie_face_patches fp;
    const int da[6] = { c.Y, c.Y, c.X, c.X, c.X, c.X }, db[6] = { c.Z, c.Z, c.Z, c.Z, c.Y, c.Y };
    fp.off[0] = 0;
    for (int f = 0; f < 6; f++) { fp.na[f] = (da[f] + 7) / 8; fp.off[f + 1] = fp.off[f] + fp.na[f] * ((db[f] + 7) / 8); }
    const long long ntile = (long long)c.tfd[0] * c.tfd[1] * c.tfd[2];
    int nsum = (int)((ntile + 64 * GIE_FF_WAVES - 1) / (64 * GIE_FF_WAVES));
    if (nsum > 2 * b->cu_total) nsum = 2 * b->cu_total;
    const int nface = (fp.off[6] + GIE_FF_WAVES - 1) / GIE_FF_WAVES;
    /* (tried in round 4: the face voxels on a side stream next to the tiles, forked and joined with events — 2.352 against 2.352 ms per
     * C5 update: what the overlap saves the two stream hand-overs cost) */
    GIE_LAUNCH(b, k_frontier_faces, dim3(nsum + nface), dim3(64 * GIE_FF_WAVES), 0, c, fp, known, known_idx, nsum);
    static int mult = GIE_SWITCH("GIE_FRONT_MULT", 0);
    if (mult <= 0) {    /* as many workgroups as are resident at once: every wave walks the same share of the list (a second round of workgroups would start when the first is done) */
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_frontier_tiles), 64 * GIE_FR_WAVES, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        mult = per_cu;
    }
    GIE_LAUNCH(b, k_frontier_tiles, dim3(b->cu_total * mult), dim3(64 * GIE_FR_WAVES), 0, c, list, count_idx);
}
static void be_edt_z_direct(be_state *b, const gie_ctx &c)
{
    GIE_LAUNCH(b, k_edt_z_direct, dim3(b->cu_total * 16), dim3(256), 0, c);
}
/* the streaming form of pass Z (k_edt_z_stream): a wave per (64 columns, y, z segment) */
/* returns 1 when launched: the launch also carries the list form of the pass (k_edt_z_direct's body) when `full` is 0 */
static int be_edt_z_stream(be_state *b, const gie_ctx &c, int full)
{
    if (c.Z < 64 || c.Z > 1024) return 0;          /* (short columns: the column kernel) */
    static const int on = GIE_SWITCH("GIE_ZSTREAM", 1);
    if (!on) return 0;
    const int nxr = (c.X + 63) / 64;
    /* z segments so that the launch has about eight waves per SIMD to overlap its rows' round trips (a segment re-reads 16 planes) */
    static const int target = GIE_SWITCH("GIE_ZSTREAM_WAVES", 8);
    const long long want = (long long)b->cu_total * 4 * target;
    int nseg = (int)((want + (long long)nxr * c.Y - 1) / ((long long)nxr * c.Y));
    const int maxseg = c.Z / 64 > 0 ? c.Z / 64 : 1;
    if (nseg > maxseg) nseg = maxseg; if (nseg < 1) nseg = 1;
    int seg_len = ((c.Z + nseg - 1) / nseg + 31) & ~31;
    nseg = (c.Z + seg_len - 1) / seg_len;
    const long long waves = (long long)nxr * nseg * c.Y;
    long long grid = (waves + 3) / 4;
    const long long cap = (long long)b->cu_total * 8 * 4;
    if (grid > cap) grid = cap;
    if (!full && grid < (long long)b->cu_total * 16) grid = (long long)b->cu_total * 16;      /* (the list form's grid: be_edt_z_direct) */
    GIE_LAUNCH(b, k_edt_z_stream, dim3((unsigned)grid), dim3(256), 0, c, full, nseg, seg_len);
    return 1;
}
/* tskip for this update; with c.catchup_fast the tiles whose deferred records have to be stored are listed and stored (pupvt: the
 * wave-range pivot of the update that left them) */
static void be_tile_oldskip(be_state *b, const gie_ctx &c, const int pupvt[3])
{
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    /* the list lives in wave C's tile list (ntile words, not in use between two merges), its length in a spare counter */
    int32_t *const count = &c.cnt[GIE_CNT_STATE1];        /* (zero: the frame clear) */
    GIE_LAUNCH(b, k_tile_oldskip, dim3((ntile + 255) / 256), dim3(256), 0, c, ntile, c.wc_list[0], count);
    if (c.catchup_fast) GIE_LAUNCH(b, k_coc_catchup_new, dim3(b->cu_total * 8), dim3(256), 0, c, pupvt[0], pupvt[1], pupvt[2], c.wc_list[0], count);
}
/* "lazy pairs": the flagged tiles that will not stay flagged get their pairs into the plane (gie_fuse) */
static void be_pair_materialise(be_state *b, const gie_ctx &c, int stay)
{
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    /* the list lives in wave C's second tile list (ntile words, not in use between two merges), its length in a spare counter (zero: the frame clear) */
    int32_t *const count = &c.cnt[GIE_CNT_STATE2];
    GIE_LAUNCH(b, k_pair_lazy_list, dim3((ntile + 255) / 256), dim3(256), 0, c, ntile, stay, c.wc_list[1], count);
    GIE_LAUNCH(b, k_pair_lazy_run, dim3(b->cu_total * 8), dim3(256), 0, c, c.wc_list[1], count);
}
static void be_coc_catchup(be_state *b, const gie_ctx &c, const gie_catchup &p)
{
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    /* the list lives in wave C's tile list (ntile words, not in use between two merges), its length in a spare counter */
    int32_t *const count = &c.cnt[GIE_CNT_STATE1];
    GIE_HIP_OK(hipMemsetAsync(count, 0, sizeof(int32_t), b->stream));
    GIE_LAUNCH(b, k_coc_catchup_list, dim3((ntile + 255) / 256), dim3(256), 0, c, p, ntile, c.wc_list[0], count);
    GIE_LAUNCH(b, k_coc_catchup_run, dim3(b->cu_total * 4), dim3(256), 0, c, p, c.wc_list[0], count);
}
static void be_edt_prep(be_state *b, const gie_ctx &c)
{
    const int ncol = c.tfd[0] * c.tfd[1];
    GIE_LAUNCH(b, k_edt_prep, dim3((ncol + GIE_PREP_WAVES - 1) / GIE_PREP_WAVES), dim3(64 * GIE_PREP_WAVES), 0, c, ncol);
}
/* full = 1: whole volume (column kernel); 0: only where Mark reads — the direct kernel over tl_known
 * and the column kernel are both launched and the one the known-tile count does not call for
 * returns at once */
static void be_edt(be_state *b, const gie_ctx &c, int full)
{
    dim3 gy((c.X + GIE_EDTY_COLS - 1) / GIE_EDTY_COLS, c.Z);
    be_prof(b, 6, 0);   /* GIE_K_EDT_Y */
    /* one 32-voxel mask word per thread: with only the planes that hold obstacles at work, the
     * pass is bound by how many loads are in flight, not by bytes */
    /* X % 4 == 0: four columns per lane, dword loads / 8-byte stores (k_edt_y4) */
    static const int y4 = GIE_SWITCH("GIE_EDTY4", 1);
    if (y4 && (c.X & 3) == 0) {
        /* y4: 1 = 16 lanes x 4 columns per workgroup (more, smaller workgroups), 3 = 32 lanes */
        const int yb = c.Y > 512 ? 32 : 16, nq = (c.Y + yb - 1) / yb;
        const int lanes = (y4 == 3) ? 32 : 16;
        dim3 g4((c.X / 4 + lanes - 1) / lanes, c.Z), b4(lanes, nq);
        if (yb == 16 && lanes == 16) GIE_LAUNCH(b, (k_edt_y4<16, 16>), g4, b4, 0, c);
        else if (yb == 16) GIE_LAUNCH(b, (k_edt_y4<16, 32>), g4, b4, 0, c);
        else if (lanes == 16) GIE_LAUNCH(b, (k_edt_y4<32, 16>), g4, b4, 0, c);
        else GIE_LAUNCH(b, (k_edt_y4<32, 32>), g4, b4, 0, c);
    }
    else if (c.Y <= 256) GIE_LAUNCH(b, (k_edt_y<8, 8>), gy, dim3(GIE_EDTY_COLS, 8), 0, c);
    else if (c.Y <= 512) GIE_LAUNCH(b, (k_edt_y<16, 16>), gy, dim3(GIE_EDTY_COLS, 16), 0, c);
    else GIE_LAUNCH(b, (k_edt_y<32, 16>), gy, dim3(GIE_EDTY_COLS, 16), 0, c);
    be_prof(b, 6, 1);
    be_prof(b, 7, 0); gie_launch_edt_dim(b, c, c.X, false, 1); be_prof(b, 7, 1);   /* GIE_K_EDT_X */
    be_prof(b, 8, 0);                                                              /* GIE_K_EDT_Z */
    /* dense fields: the streaming form does the whole pass and flags what it could not finish for the column kernel; few known
     * tiles: the list form — in the same launch */
    if (!be_edt_z_stream(b, c, full) && !full) be_edt_z_direct(b, c);
    gie_launch_edt_dim(b, c, c.Z, true, full);
    be_prof(b, 8, 1);

}
/* waves A, B and C in one launch; workgroups are co-resident by construction (512 threads and ≈ 151 KB of LDS each:
 * at most one per compute unit).  The frame clear has zeroed the barrier word and the per-level
 * arrays; a second launch inside the same map update (gie_refine) clears them itself. */
/* The waves kernel synchronises its workgroups with a grid barrier, so ALL of them have to be
 * resident at once.  One launch is (128 workgroups on 256 compute units); two launches from
 * different mappers on the same device at the same time need not be.  Waves launches of one
 * device are therefore chained through an event: a launch waits for the previous one, whichever
 * mapper (stream) it came from. */
static void be_waves(be_state *b, const gie_ctx &c, int with_ab, int record_seeds, int clear_first)
{
    std::lock_guard<std::mutex> lock(g_waves_mutex);
    const int dv = b->device & 63;
    (void)hipSetDevice(b->device);
    const bool chain = g_live_mappers[dv] > 1;       /* a lone mapper's launches follow each other on its stream anyway (an event is a marker packet the next kernel waits for) */
    if (chain) {
        if (g_waves_event_set[dv]) GIE_HIP_OK(hipStreamWaitEvent(b->stream, g_waves_event[dv], 0));
        else { GIE_HIP_OK(hipEventCreateWithFlags(&g_waves_event[dv], hipEventDisableTiming)); g_waves_event_set[dv] = true; }
    }
    if (clear_first) {
        GIE_HIP_OK(hipMemsetAsync(&c.cnt[GIE_CNT_BAR_B], 0, 2 * sizeof(int32_t), b->stream));      /* (BAR_B, BAR_C: the barrier words of the two launches) */
        GIE_HIP_OK(hipMemsetAsync(c.lvl_next, 0, 2 * GIE_MAX_LEVELS * sizeof(int32_t), b->stream));
    }
    if (with_ab) GIE_LAUNCH(b, k_waves_ab, dim3(b->num_cu), dim3(GIE_WAVE_THREADS), 0, c);
    GIE_LAUNCH(b, k_waves_c, dim3(b->num_cu), dim3(GIE_WAVE_THREADS), 0, c, with_ab, record_seeds);
    if (chain) GIE_HIP_OK(hipEventRecord(g_waves_event[dv], b->stream));
}

#include "gie_api.inc.h"
