/*
 * gie_api.inc.h — the C-ABI of include/gie.h: allocation, per-frame orchestration
 * (VOLMAPNODE::publishMap's call sequence, src/volumetric_mapper.cpp:138-224) and readers.
 * Written against the small `be_*` backend interface; gie_hip.hip supplies the HIP backend
 * (the product), tests/emu supplies a sequential stand-in used only by CPU-side logic tests.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>

enum { GIE_K_CLASSIFY = 0, GIE_K_RAY_REGISTER, GIE_K_RAY_FREE, GIE_K_RAY_FINAL, GIE_K_ALLOC, GIE_K_FUSE, GIE_K_EDT_Y, GIE_K_EDT_X,
       GIE_K_EDT_Z, GIE_K_MARK, GIE_K_FRONTIER, GIE_K_WAVE_A, GIE_K_WAVE_B, GIE_K_WAVE_C, GIE_K_COMMIT, GIE_K_EDT_ZFACES, GIE_K_MARKC, GIE_K_NUM };
#include <cstdio>
#include <unistd.h>
static const char *const gie_kernel_names[GIE_K_NUM] = { "ogm_classify", "ray_register", "ray_free", "ray_finalize", "block_alloc", "fuse",
       "edt_pass_y", "edt_pass_x", "edt_pass_z", "mark", "frontiers", "wave_a", "wave_b", "waves", "commit", "edt_prep", "mark_commit" };

#define GIE_REHASH_PERIOD 64                 /* map updates between two rebuilds of the hash table while blocks are being erased */
static thread_local std::string g_gie_err;
static void gie_set_err(const std::string &s) { g_gie_err = s; }
extern "C" const char *gie_last_error(void) { return g_gie_err.c_str(); }

struct gie_mapper {
    gie_config cfg;
    gie_ctx c;
    be_state be;
    int ncell;
    int has_pose, has_ogm, merge_open;
    int bar_fault_left;     /* map updates whose wavefront launch gets the barrier fault (gie_debug_fault_barrier) */
    int fuse_fresh;                       /* gie_fuse has run and no merge has consumed it yet (a merge needs the frame clear of its own map update) */
    int pool_base;                        /* GIE_DEBUG_POOL_BASE (tests): the slots below it are never handed out */
    int evictions;                        /* map updates with block erasure since the hash table was last rebuilt */
    long long tomb_bound;                 /* upper bound of the tombstones those erasures left in the table */
    int retain_box_valid, retain_box_lo[3], retain_box_hi[3];   /* block box +- retain of the last update that erased */
    int deferred;                         /* the last merge ran fused: the stored pairs of its volume's voxels are still to be written when they leave (gie_commit_pair) */
    int32_t *blk_tab2; int tab_valid, tab_tb0[3];   /* the other block table (the tables of consecutive fuses alternate) and where the current one was built */
    int flush_tab_ok;                     /* ... and _glb_type / the block table are still that update's (no gie_fuse since) */
    int commit_pvt[3], commit_upvt[3], commit_tb0[3];   /* pivots / block-table origin of that merge */
    int edt_partial;                      /* the last batch EDT skipped tiles nobody reads (gie_read_batch_edt completes it) */
    int ogm_unlabelled;                   /* ray-cast scan whose _inst_type labels have not been written (gie_read_ogm does it) */
    float msg_origin[3];
    float *d_sensor; size_t sensor_cap;   /* device copy of the last sensor frame */
    float *d_pts_g; size_t pts_cap;       /* ray casting: points in the global frame */
    float *d_box_ll, *d_box_ur; uint8_t *d_box_act; int box_cap;
    int32_t *d_rank;
    /* changed-block streaming: slot list + double-buffered staging (device and pinned host) */
    int32_t *d_srank, *d_slist;
    void *d_stage[2], *h_stage[2];
    std::vector<void *> allocs;
    void *scratch[2]; size_t scratch_cap[2];   /* export buffers of the readers (grow-only, reused from call to call) */
    int32_t h_cnt[GIE_CNT_NUM];
    int32_t next_off[3], next_whole[3];
    float us[4];
    int track_want;                       /* gie_stream_enable: taken over by the next gie_fuse (an update runs in ONE order of kernels from its fuse to its merge) */
    const int8_t *labels_pending;         /* the last scan is a label plane left in place (c.scan_labels until gie_fuse): materialised into `_inst_type` if anything else wants it */
    int coc_pending;                      /* voxels of the current tskip tiles have their records in the pair plane only (gie_ops.h "deferred records") */
    int lazy_pending;                     /* tiles may be flagged in tlazy: their pairs are not in the pair plane (gie_ops.h "lazy pairs") */
    uint32_t *bcoc_alt;                   /* the other batch-obstacle plane (pass Z never writes the one the lazy pairs are derived from) */
    int tsp_pvt[3];                       /* the pivot tskip_prev's tiles refer to */
    int flushed_ct;                       /* map tick (c.map_ct) of the pose the owed pairs were last written for ahead of gie_fuse (gie_owed_pairs_before_import) */
    /* CostMap publishing without a stall (gie_costmap_publish / gie_costmap_acquire): device staging + two pinned host buffers */
    gie_seendist *d_cm; void *h_cm[2]; size_t cm_bytes; int cm_slot, cm_pending;
    long long *d_round_stats;             /* exchange rounds without the host: rounds enqueued / run, updates / updates left unconverged (k_round_note) */
};

template <class T> static T *gie_dalloc(gie_mapper *m, size_t n, bool zero = true)
{
    void *p = be_alloc(&m->be, n * sizeof(T), zero);
    if (p) m->allocs.push_back(p);
    {   /* GIE_DEBUG_ALLOC=1: where the planes went (stderr) */
        static const int dbg = GIE_SWITCH("GIE_DEBUG_ALLOC", 0);
        if (dbg && (dbg > 1 || n * sizeof(T) >= (64u << 20))) fprintf(stderr, "[%d] gie alloc %2d: %p  %8.1f MiB\n", (int)getpid(), (int)m->allocs.size(), p, (double)(n * sizeof(T)) / 1048576.0);
    }
    return (T *)p;
}

/* export buffer `i` of at least `bytes` (nullptr + error text when the device is out of memory) */
static void *gie_scratch(gie_mapper *m, int i, size_t bytes, const char *who)
{
    if (bytes > m->scratch_cap[i]) {
        if (m->scratch[i]) be_free(&m->be, m->scratch[i]);
        m->scratch[i] = be_alloc(&m->be, bytes, false);
        m->scratch_cap[i] = m->scratch[i] ? bytes : 0;
    }
    if (!m->scratch[i]) gie_set_err(std::string(who) + ": device allocation of the export buffer failed");
    return m->scratch[i];
}

/* the full-volume debug readers (gie_read_local / _batch_edt / _costmap) export through buffers of 4-12 bytes per voxel — 2 GB at
 * 512^3: those go back to the device when the call is over (eight mappers on one device is a supported test configuration);
 * the small buffers of the per-update readers (halo layers, point queries) are kept and reused */
#define GIE_SCRATCH_KEEP ((size_t)64 << 20)
static void gie_scratch_trim(gie_mapper *m)
{
    for (int i = 0; i < 2; i++)
        if (m->scratch[i] && m->scratch_cap[i] > GIE_SCRATCH_KEEP) { be_free(&m->be, m->scratch[i]); m->scratch[i] = nullptr; m->scratch_cap[i] = 0; }
}

/* PLACEMENT of the dense sweep's planes.  The Mark + commit sweep walks four large planes in step — it reads the
 * types and the batch obstacles, writes the pairs and the stored obstacles — and how long that takes depends on where those planes
 * lie in physical memory relative to each other: the same kernel on the same data runs in 0.80 or 0.89 ms from one mapper to the
 * next (two write streams that overlap or take turns; tools/place_probe.py), virtual addresses and offsets inside one allocation
 * decide nothing.  What can be done from here is to draw again: k_place_probe times the sweep's memory pattern on the planes
 * themselves (it agrees with the real sweep to 1 %), and each of the four planes is re-allocated up to GIE_PLACE_TRIES - 1 times
 * (default 4; 0 / 1 = off) while the candidate it replaces is still held — so the allocator hands out other memory —, keeping the
 * faster placement.  Costs a few tens of milliseconds of gie_create for volumes of 16 M voxels and more, nothing below. */
static void gie_place_calibrate(gie_mapper *m, size_t N, size_t GV)
{
    const int tries = m->cfg.place_tries;
    if (tries <= 1 || N < ((size_t)1 << 24) || !m->c.g_coc || !m->c.pair || !m->c.bcoc || !m->c.glb_type) return;
    gie_ctx &c = m->c;
    (void)be_place_probe(&m->be, c, 12);                      /* warm-up: a device that has been idle clocks up during the first milliseconds */
    float best = be_place_probe(&m->be, c, 5);
    if (best <= 0.f) return;
    const float first = best;
    struct plane { void **pp; size_t bytes; bool zero; } planes[4] = {
        { reinterpret_cast<void **>(&c.g_coc), GV * sizeof(uint64_t), false }, { reinterpret_cast<void **>(&c.pair), N * sizeof(uint64_t), true },
        { reinterpret_cast<void **>(&c.bcoc), N * sizeof(uint32_t), true }, { reinterpret_cast<void **>(&c.glb_type), N, true } };
    int swaps = 0;
    for (plane &pl : planes) {
        for (int t = 1; t < tries; t++) {
            void *alt = be_alloc(&m->be, pl.bytes, pl.zero);
            if (!alt) break;                                  /* memory is short: keep what there is */
            void *old = *pl.pp;
            *pl.pp = alt;
            const float ms = be_place_probe(&m->be, c, 5);
            bool take = ms > 0.f && ms < 0.985f * best;
            if (take) {                                       /* measured against the incumbent once more, back to back: clocks drift */
                *pl.pp = old;
                const float again = be_place_probe(&m->be, c, 5);
                *pl.pp = alt;
                take = again > 0.f && ms < 0.985f * again;
                if (!take && again > 0.f) best = again;
            }
            if (take) {
                best = ms; swaps++;
                for (void *&q : m->allocs) if (q == old) q = alt;
                be_free(&m->be, old);
            } else { *pl.pp = old; be_free(&m->be, alt); }
        }
    }
    if (GIE_SWITCH("GIE_DEBUG_ALLOC", 0)) fprintf(stderr, "[%d] gie placement: probe %.4f -> %.4f ms after %d re-draws\n", (int)getpid(), first, best, swaps);
}

static int gie_pow2_ge(long long v) { int p = 1; while ((long long)p < v) p <<= 1; return p; }

#define GIE_MAX_POOL_BLOCKS (1 << 26)          /* slots are 32-bit, voxel addresses (slot * 512 + index) 64-bit: what bounds a pool is the device's memory
                                                * (38 B per voxel: 15 M blocks in 288 GB); this is where the hash table's 32-bit index arithmetic ends */
extern "C" gie_mapper *gie_create(const gie_config *cfg)
{
    if (!cfg || cfg->voxel_width <= 0.f || cfg->local_size[0] < 1 || cfg->local_size[1] < 1 || cfg->local_size[2] < 1) {
        gie_set_err("gie_create: bad config"); return nullptr;
    }
    const int X = cfg->local_size[0], Y = cfg->local_size[1], Z = cfg->local_size[2];
    const long long M = (long long)X * X + (long long)Y * Y + (long long)Z * Z;
    const int L = X > Z ? X : Z;
    if (X > 1024 || Y > 1024 || Z > 1024 || M + 1 + (long long)L * L >= (1ll << 22) || (long long)X * Y * Z > 0x7fffffffll) {
        gie_set_err("gie_create: local volume too large (side <= 1024 and X²+Y²+Z²+max(X,Z)² < 2^22)"); return nullptr;
    }
    if (cfg->max_blocks > GIE_MAX_POOL_BLOCKS) {         /* a voxel address is slot * 512 + in-block index in 32 bits */
        gie_set_err("gie_create: max_blocks above 67 108 864 is not supported (and would not fit the device: 19.5 KB per block); "
                    "bound the map with retain_radius_blocks instead");
        return nullptr;
    }
    gie_mapper *m = new gie_mapper();
    m->cfg = *cfg;
    m->has_pose = m->has_ogm = 0; m->merge_open = 0; m->bar_fault_left = 0; m->c.bar_fault = 0; m->edt_partial = 0; m->ogm_unlabelled = 0; m->evictions = 0; m->tomb_bound = 0; m->retain_box_valid = 0; m->deferred = 0; m->flush_tab_ok = 0; m->fuse_fresh = 0; m->flushed_ct = -1; m->coc_pending = 0; m->labels_pending = nullptr; m->track_want = 0;
    m->d_cm = nullptr; m->h_cm[0] = m->h_cm[1] = nullptr; m->cm_bytes = 0; m->cm_slot = 0; m->cm_pending = 0;
    for (int i = 0; i < 3; i++) { m->next_off[i] = 0; m->next_whole[i] = cfg->local_size[i]; }
    m->d_sensor = nullptr; m->sensor_cap = 0; m->d_pts_g = nullptr; m->pts_cap = 0;
    m->d_box_ll = m->d_box_ur = nullptr; m->d_box_act = nullptr; m->box_cap = 0;
    m->d_srank = m->d_slist = nullptr; m->d_stage[0] = m->d_stage[1] = m->h_stage[0] = m->h_stage[1] = nullptr;
    memset(m->h_cnt, 0, sizeof(m->h_cnt)); memset(m->us, 0, sizeof(m->us));
    m->scratch[0] = m->scratch[1] = nullptr; m->scratch_cap[0] = m->scratch_cap[1] = 0;
    if (cfg->wave_workgroups < 0 || cfg->place_tries < 0) { gie_set_err("gie_create: bad config (wave_workgroups / place_tries are >= 0)"); delete m; return nullptr; }
    if (be_init(&m->be, cfg->device_id, cfg->wave_workgroups) != 0) { delete m; return nullptr; }
    gie_ctx &c = m->c;
    memset(&c, 0, sizeof(c));
    c.X = X; c.Y = Y; c.Z = Z; c.N = X * Y * Z;
    c.voxel_width = cfg->voxel_width; c.occ_thresh = cfg->occupancy_threshold;
    c.min_h = cfg->ogm_min_h; c.max_h = cfg->ogm_max_h; c.cutoff_sq = cfg->cutoff_grids_sq;
    c.fast_mode = cfg->fast_mode; c.for_motion_planner = cfg->for_motion_planner; c.robot_r2 = cfg->robot_r2_grids;
    c.max_width = X + Y + Z; c.max_loc_dist_sq = (int)M;
    if (c.max_width < 1022) {   /* the reference's envelope: local_batch.h:51-58, voxmap_utils.cuh:8,161-165 */
        c.wr[0] = 2046; c.wr[1] = 2046; c.wr[2] = 1022; c.empty_value = GIE_EMPTY_VALUE; c.invalid_dist_min = 900000;
    } else {                    /* documented extension for larger volumes */
        c.wr[0] = 16382; c.wr[1] = 16382; c.wr[2] = 8190; c.empty_value = 4194303; c.invalid_dist_min = 4000000;
    }
    c.L2G = gie_se3_from_quat(1, 0, 0, 0, 0, 0, 0); c.G2L = gie_se3_inv(c.L2G);
    const size_t N = (size_t)c.N;
    c.ray_count = gie_dalloc<int32_t>(m, N);
    c.inst_type = gie_dalloc<int8_t>(m, N);
    c.glb_type = gie_dalloc<int8_t>(m, N);
    c.edt = gie_dalloc<float>(m, N);
    c.cy1 = gie_dalloc<uint16_t>(m, N);
    c.cxy2 = gie_dalloc<uint32_t>(m, N);
    c.bcoc = gie_dalloc<uint32_t>(m, N);
    m->bcoc_alt = gie_dalloc<uint32_t>(m, N);
    c.pair = gie_dalloc<uint64_t>(m, N);      /* zero-initialised: SURVEY App. B #3 */
    c.wl = gie_dalloc<uint32_t>(m, N);
    for (int i = 0; i < 3; i++) c.tfd[i] = (cfg->local_size[i] + 7) / 8;
    const size_t ntile = (size_t)c.tfd[0] * c.tfd[1] * c.tfd[2];
    c.tflag = gie_dalloc<uint8_t>(m, 7 * ntile);     /* tflag | tunk | tsum | tray | tact | tknown (this frame / previous frame) */
    c.tunk = c.tflag + ntile; c.tsum = c.tflag + 2 * ntile; c.tray = c.tflag + 3 * ntile; c.tact = c.tflag + 4 * ntile;
    c.tknown = c.tflag + 5 * ntile; c.tknown_prev = c.tflag + 6 * ntile;
    c.tmax = gie_dalloc<int32_t>(m, 2 * ntile);
    c.tbmax = gie_dalloc<int32_t>(m, ntile);
    c.tmax_prev = c.tmax ? c.tmax + ntile : nullptr;
    c.tskip = gie_dalloc<uint8_t>(m, 2 * ntile);
    c.tskip_prev = c.tskip ? c.tskip + ntile : nullptr;
    c.tlazy = gie_dalloc<uint8_t>(m, ntile);
    c.bcoc_lazy = c.bcoc; c.lazy_ok = 0; m->lazy_pending = 0;
    c.coc_defer = 0; c.qdefer = 0; c.scan_labels = nullptr;
    for (int i = 0; i < 3; i++) { c.ts_pvt[i] = 0; m->tsp_pvt[i] = 0; c.pp_pvt[i] = c.pp_upvt[i] = 0; }
    c.ucol = gie_dalloc<uint8_t>(m, (((size_t)X * Y * ((Z + 7) / 8)) + 3) & ~(size_t)3);
    c.zocc = gie_dalloc<uint8_t>(m, (size_t)c.Z);
    c.zneed = gie_dalloc<uint64_t>(m, (size_t)c.tfd[0] * c.tfd[1]);
    c.zredo = gie_dalloc<uint32_t>(m, (size_t)((X + 15) / 16) * Y);
    c.zlist = gie_dalloc<uint16_t>(m, (size_t)c.Z + 8);
    c.zcount = gie_dalloc<int32_t>(m, 4);
    c.tl_known = gie_dalloc<int32_t>(m, ntile);
    c.tl_swept = gie_dalloc<int32_t>(m, ntile);
    c.tl_front = gie_dalloc<int32_t>(m, ntile);
    { const int force = GIE_SWITCH("GIE_TILE_LIST", -1); c.force_lists = force < 0 ? -1 : (force != 0); }   /* tests force lists / sweeps */
    const int bdr = 2 * (X * Y + Y * Z + X * Z);
    c.lprop = gie_dalloc<uint64_t>(m, (size_t)bdr, false);
    c.cand[0] = gie_dalloc<uint64_t>(m, N, false);
    c.cand[1] = gie_dalloc<uint64_t>(m, N, false);
    for (int i = 0; i < 3; i++) c.tdim[i] = cfg->local_size[i] / 8 + 3;
    m->ncell = c.tdim[0] * c.tdim[1] * c.tdim[2];
    c.blk_tab = gie_dalloc<int32_t>(m, (size_t)m->ncell, false);
    m->blk_tab2 = gie_dalloc<int32_t>(m, (size_t)m->ncell, false);
    c.tab_prev = nullptr; m->tab_valid = 0;
    c.blk_need = gie_dalloc<uint8_t>(m, (size_t)m->ncell);
    c.blk_new = gie_dalloc<int32_t>(m, (size_t)m->ncell);
    m->d_rank = gie_dalloc<int32_t>(m, (size_t)m->ncell);
    long long mb = cfg->max_blocks > 0 ? cfg->max_blocks : 3ll * m->ncell + 4096;
    if (mb > GIE_MAX_POOL_BLOCKS) mb = GIE_MAX_POOL_BLOCKS;   /* (the default of a very large volume; an explicit request above the limit was refused) */
    c.max_blocks = (int)mb;
    const int hcap = gie_pow2_ge(4 * mb);              /* load factor <= 1/4: probe chains stay short (16 bytes per slot) */
    c.hmask = (uint32_t)(hcap - 1);
    c.hkeys = gie_dalloc<uint64_t>(m, (size_t)hcap, false);
    c.hvals = gie_dalloc<int32_t>(m, (size_t)hcap, false);
    c.pool_count = gie_dalloc<int32_t>(m, 4);
    m->pool_base = 0;
    c.retain = cfg->retain_radius_blocks > 0 ? cfg->retain_radius_blocks : 0;
    c.free_list = gie_dalloc<int32_t>(m, c.retain > 0 ? (size_t)mb : 1, false);
    const size_t GV = (size_t)mb * GIE_VBSZ;
    c.g_key = gie_dalloc<uint64_t>(m, (size_t)mb, false);
    c.g_nbr = gie_dalloc<int32_t>(m, 8 * (size_t)mb, false);
    c.g_occ = gie_dalloc<uint8_t>(m, GV, false);
    c.g_type = gie_dalloc<int8_t>(m, GV, false);
    c.g_coc = gie_dalloc<uint64_t>(m, GV, false);
    c.g_pair = gie_dalloc<uint64_t>(m, GV, false);
    c.g_prop = gie_dalloc<uint64_t>(m, GV, false);
    c.g_prop2 = gie_dalloc<uint64_t>(m, GV, false);
    for (int i = 0; i < 2; i++) c.wb_list[i] = gie_dalloc<int32_t>(m, (size_t)mb, false);
    c.wb_flag[0] = gie_dalloc<int32_t>(m, 2 * (size_t)mb);      /* one allocation: cleared as one region with the frame */
    c.wb_flag[1] = c.wb_flag[0] ? c.wb_flag[0] + mb : nullptr;
    c.g_wl = gie_dalloc<int32_t>(m, GV, false);
    c.g_dirty = gie_dalloc<int32_t>(m, (size_t)mb);
    c.track = 0;
    long long qab = 16ll * bdr; if (qab < 65536) qab = 65536; if (qab > (16 << 20)) qab = 16 << 20;
    long long qc = (long long)c.N; if (qc < 4096) qc = 4096; if (qc > (16 << 20)) qc = 16 << 20;
    c.qcap_ab = (int)qab; c.qcap_c = (int)qc;
    c.qa = gie_dalloc<uint64_t>(m, (size_t)qab, false);
    c.qb = gie_dalloc<uint64_t>(m, (size_t)qab, false);
    c.qa_a = gie_dalloc<gie_vaddr>(m, (size_t)qab, false);
    c.qb_a = gie_dalloc<gie_vaddr>(m, (size_t)qab, false);
    for (int i = 0; i < 2; i++) c.qc[i] = gie_dalloc<int32_t>(m, (size_t)qc, false);
    c.cnt = gie_dalloc<int32_t>(m, GIE_CNT_NUM);
    for (int i = 0; i < 2; i++) c.wc_list[i] = gie_dalloc<int32_t>(m, ntile, false);
    c.wc_flag[0] = gie_dalloc<int32_t>(m, 2 * ntile);          /* one allocation: cleared as one region with the frame */
    c.wc_flag[1] = c.wc_flag[0] ? c.wc_flag[0] + ntile : nullptr;
    c.lvl_next = gie_dalloc<int32_t>(m, 6 * GIE_MAX_LEVELS);
    c.lvl_vis = c.lvl_next + GIE_MAX_LEVELS;
    c.lvlb_next = c.lvl_next + 2 * GIE_MAX_LEVELS; c.lvlb_vis = c.lvl_next + 3 * GIE_MAX_LEVELS;
    c.lvla_next = c.lvl_next + 4 * GIE_MAX_LEVELS; c.lvla_vis = c.lvl_next + 5 * GIE_MAX_LEVELS;
    m->d_round_stats = gie_dalloc<long long>(m, 4);
    c.gate = nullptr;
    bool ok = c.cnt != nullptr;
    for (void *p : m->allocs) ok = ok && p != nullptr;
    if (!ok) { gie_set_err("gie_create: device allocation failed"); gie_destroy(m); return nullptr; }
    if (GIE_SWITCH("GIE_DEBUG_POOL_BASE", 0) > 0) {   /* tests: hand out slots from here on (voxel addresses beyond 2^31 without filling a pool) */
        const long long b = GIE_SWITCH("GIE_DEBUG_POOL_BASE", 0);
        if (b > 0 && b < mb) {
            const int32_t v = (int32_t)b; be_h2d(&m->be, c.pool_count, &v, sizeof(v)); m->pool_base = (int)b;
            be_memset(&m->be, c.g_key, 0xff, (size_t)b * sizeof(uint64_t));    /* the slots below the base were never handed out: erasure and re-hash walk every slot below the pool top and must find them empty */
        }
    }
    gie_place_calibrate(m, N, GV);
    be_memset(&m->be, c.hkeys, 0xff, (size_t)hcap * sizeof(uint64_t));
    be_memset(&m->be, c.lprop, 0xff, (size_t)bdr * sizeof(uint64_t));
    be_memset(&m->be, c.cand[0], 0xff, N * sizeof(uint64_t));
    be_memset(&m->be, c.cand[1], 0xff, N * sizeof(uint64_t));
    be_sync(&m->be);
    return m;
}

extern "C" void gie_destroy(gie_mapper *m)
{
    if (!m) return;
    be_sync(&m->be);
    for (void *p : m->allocs) be_free(&m->be, p);
    if (m->d_sensor) be_free(&m->be, m->d_sensor);
    if (m->d_pts_g) be_free(&m->be, m->d_pts_g);
    if (m->d_box_ll) { be_free(&m->be, m->d_box_ll); be_free(&m->be, m->d_box_ur); be_free(&m->be, m->d_box_act); }
    if (m->d_srank) { be_free(&m->be, m->d_srank); be_free(&m->be, m->d_slist); }
    for (int i = 0; i < 2; i++) if (m->scratch[i]) be_free(&m->be, m->scratch[i]);
    if (m->d_cm) be_free(&m->be, m->d_cm);
    for (int i = 0; i < 2; i++) if (m->h_cm[i]) be_host_free(&m->be, m->h_cm[i]);
    for (int i = 0; i < 2; i++) { if (m->d_stage[i]) be_free(&m->be, m->d_stage[i]); if (m->h_stage[i]) be_host_free(&m->be, m->h_stage[i]); }
    be_fini(&m->be);
    delete m;
}

extern "C" int gie_set_pose(gie_mapper *m, const float pos[3], const float q[4])
{
    if (!m || !pos || !q) { gie_set_err("gie_set_pose: null"); return GIE_ERR_INVALID; }
    gie_ctx &c = m->c;
    /* validate before anything changes: a rejected pose leaves the mapper exactly as it was */
    int crd[3];
    for (int i = 0; i < 3; i++) {
        if (!(fabsf(pos[i]) <= 1.0e6f)) { gie_set_err("gie_set_pose: position is not finite"); return GIE_ERR_INVALID; }
        crd[i] = gie_pos2coord(pos[i], c.voxel_width);
        if (crd[i] > 900000 || crd[i] < -900000) { gie_set_err("gie_set_pose: position outside the representable map"); return GIE_ERR_INVALID; }
    }
    c.map_ct += 1;                                        /* _time++, volumetric_mapper.cpp:144 */
    c.gate = nullptr;                                     /* a round gate belongs to ONE update's exchange: a caller that left between gie_round_gate and
                                                           * gie_round_end (an exception in its transport) must not find the next update's kernels gated */
    c.L2G = gie_se3_from_quat(q[0], q[1], q[2], q[3], pos[0], pos[1], pos[2]);
    c.G2L = gie_se3_inv(c.L2G);
    const int sz[3] = { c.X, c.Y, c.Z };
    for (int i = 0; i < 3; i++) {
        c.origin[i] = pos[i];
        c.pvt[i] = crd[i] - sz[i] / 2 + m->next_off[i];   /* calculate_pivot_origin, local_batch.h:128-142 (+ tile offset) */
        c.tile_off[i] = m->next_off[i];
        c.whole_lo[i] = -(m->next_whole[i] / 2) + sz[i] / 2 - m->next_off[i];
        c.whole_hi[i] = c.whole_lo[i] + m->next_whole[i];
        m->msg_origin[i] = (float)c.pvt[i] * c.voxel_width;
        c.upvt[i] = crd[i] - c.wr[i] / 2;                 /* calculate_update_pivot, :159-166 */
        c.tb0[i] = (c.pvt[i] - 1) >> 3;
        c.vb_lo[i] = (c.pvt[i] - 1) >> 3; c.vb_hi[i] = (c.pvt[i] + sz[i]) >> 3;
    }
    c.wr_inside = 1;
    for (int i = 0; i < 3; i++) if (c.pvt[i] - c.upvt[i] < 0 || c.pvt[i] - c.upvt[i] + sz[i] > c.wr[i]) c.wr_inside = 0;
    c.lazy_ok = c.wr_inside;                              /* "lazy pairs": not for one tile of several (its faces lie inside the whole volume: the refinement rounds read them) */
    for (int i = 0; i < 3; i++) if (m->next_off[i] != 0 || m->next_whole[i] != sz[i]) c.lazy_ok = 0;
    c.tab_prev = nullptr;                                 /* the block table belongs to the pose before: the next gie_fuse builds this one's (from it) */
    const uint32_t f = (uint32_t)c.map_ct & 0x3ffffu;
    if (f == 0) {                                         /* stamp wrap: clear the stamp planes once */
        be_memset(&m->be, c.wl, 0, (size_t)c.N * sizeof(uint32_t));
        be_memset(&m->be, c.g_wl, 0xff, (size_t)c.max_blocks * GIE_VBSZ * sizeof(int32_t));
    }
    c.stamp_base = (f + 1u) << 12;
    m->has_pose = 1;
    return GIE_OK;
}

static int gie_need_pose(gie_mapper *m, const char *who)
{
    if (!m) { gie_set_err(std::string(who) + ": null handle"); return GIE_ERR_INVALID; }
    if (!m->has_pose) { gie_set_err(std::string(who) + ": gie_set_pose has not been called"); return GIE_ERR_INVALID; }
    return GIE_OK;
}

/* a label scan left in place (gie_ogm_labels_dev) becomes an ordinary one: its labels go into `_inst_type` now — before another scan
 * is laid over it, or a reader asks for the plane */
static void gie_labels_materialise(gie_mapper *m)
{
    if (!m->labels_pending) return;
    be_labels(&m->be, m->c, m->labels_pending, 0);
    m->labels_pending = nullptr; m->c.scan_labels = nullptr;
}

static int gie_stage_sensor(gie_mapper *m, const float *host, size_t n)
{
    gie_labels_materialise(m);            /* (the plane of a label scan before this one may live in the staging buffer) */
    if (n > m->sensor_cap) {
        if (m->d_sensor) { be_sync(&m->be); be_free(&m->be, m->d_sensor); }
        m->d_sensor = (float *)be_alloc(&m->be, n * sizeof(float), false);
        m->sensor_cap = m->d_sensor ? n : 0;
        if (!m->d_sensor) { gie_set_err("sensor buffer allocation failed"); return GIE_ERR_DEVICE; }
    }
    be_h2d(&m->be, m->d_sensor, host, n * sizeof(float));
    return GIE_OK;
}

/* ---- OGM */
extern "C" int gie_ogm_depth_dev(gie_mapper *m, const float *d_depth, const gie_cam_param *p)
{
    int rc = gie_need_pose(m, "gie_ogm_depth"); if (rc) return rc;
    if (!d_depth || !p || p->rows < 1 || p->cols < 1) { gie_set_err("gie_ogm_depth: bad arguments"); return GIE_ERR_INVALID; }
    gie_labels_materialise(m);
    m->c.pntcld_mode = 0;
    be_time(&m->be, 0);
    op_classify_depth op; op.img = d_depth; op.p = *p;
    be_prof(&m->be, GIE_K_CLASSIFY, 0); be_vox(&m->be, m->c, op); be_prof(&m->be, GIE_K_CLASSIFY, 1);
    be_time(&m->be, 1);
    m->has_ogm = 1;
    return GIE_OK;
}
extern "C" int gie_ogm_depth(gie_mapper *m, const float *depth, const gie_cam_param *p)
{
    int rc = gie_need_pose(m, "gie_ogm_depth"); if (rc) return rc;
    if (!depth || !p || p->rows < 1 || p->cols < 1) { gie_set_err("gie_ogm_depth: bad arguments"); return GIE_ERR_INVALID; }
    rc = gie_stage_sensor(m, depth, (size_t)p->rows * p->cols); if (rc) return rc;
    return gie_ogm_depth_dev(m, m->d_sensor, p);
}
extern "C" int gie_ogm_multiscan_dev(gie_mapper *m, const float *d_ranges, const gie_multiscan_param *p)
{
    int rc = gie_need_pose(m, "gie_ogm_multiscan"); if (rc) return rc;
    if (!d_ranges || !p || p->scan_num < 1 || p->ring_num < 1) { gie_set_err("gie_ogm_multiscan: bad arguments"); return GIE_ERR_INVALID; }
    gie_labels_materialise(m);
    m->c.pntcld_mode = 0;
    be_time(&m->be, 0);
    op_classify_multiscan op; op.img = d_ranges; op.p = *p;
    {   /* conservative elevation bounds for the early-out (host double precision, widened) */
        const double lo = (double)p->phi_min - 0.5 * (double)p->phi_inc, hi = (double)p->phi_min + ((double)p->ring_num - 0.5) * (double)p->phi_inc;
        const double a = (lo < hi ? lo : hi) - 1e-3, b = (lo < hi ? hi : lo) + 1e-3;
        op.fov_test = (a > -1.5 && b < 1.5) ? 1 : 0;
        op.tan_lo = op.fov_test ? (float)(tan(a) - 1e-4 * (1.0 + fabs(tan(a)))) : 0.f;
        op.tan_hi = op.fov_test ? (float)(tan(b) + 1e-4 * (1.0 + fabs(tan(b)))) : 0.f;
    }
    be_prof(&m->be, GIE_K_CLASSIFY, 0); be_vox(&m->be, m->c, op); be_prof(&m->be, GIE_K_CLASSIFY, 1);
    be_time(&m->be, 1);
    m->has_ogm = 1;
    return GIE_OK;
}
extern "C" int gie_ogm_multiscan(gie_mapper *m, const float *ranges, const gie_multiscan_param *p)
{
    int rc = gie_need_pose(m, "gie_ogm_multiscan"); if (rc) return rc;
    if (!ranges || !p || p->scan_num < 1 || p->ring_num < 1) { gie_set_err("gie_ogm_multiscan: bad arguments"); return GIE_ERR_INVALID; }
    rc = gie_stage_sensor(m, ranges, (size_t)p->scan_num * p->ring_num); if (rc) return rc;
    return gie_ogm_multiscan_dev(m, m->d_sensor, p);
}
extern "C" int gie_ogm_scan2d(gie_mapper *m, const float *ranges, const gie_scan_param *p)
{
    int rc = gie_need_pose(m, "gie_ogm_scan2d"); if (rc) return rc;
    if (!ranges || !p || p->scan_num < 1) { gie_set_err("gie_ogm_scan2d: bad arguments"); return GIE_ERR_INVALID; }
    rc = gie_stage_sensor(m, ranges, (size_t)p->scan_num); if (rc) return rc;
    gie_labels_materialise(m);
    m->c.pntcld_mode = 0;
    be_time(&m->be, 0);
    op_classify_scan2d op; op.img = m->d_sensor; op.p = *p;
    be_prof(&m->be, GIE_K_CLASSIFY, 0); be_vox(&m->be, m->c, op); be_prof(&m->be, GIE_K_CLASSIFY, 1);
    be_time(&m->be, 1);
    m->has_ogm = 1;
    return GIE_OK;
}
static int gie_ogm_labels_dev_impl(gie_mapper *m, const int8_t *d_labels, int may_borrow, int *borrowed)
{
    if (borrowed) *borrowed = 0;
    int rc = gie_need_pose(m, "gie_ogm_labels"); if (rc) return rc;
    if (!d_labels) { gie_set_err("gie_ogm_labels: bad arguments"); return GIE_ERR_INVALID; }
    gie_labels_materialise(m);
    m->c.pntcld_mode = 0;
    be_time(&m->be, 0);
    /* The plane stays where it is when the caller allows it and the device form applies: this launch only flags the blocks of the
     * observed voxels, gie_fuse reads the labels from d_labels itself (1 byte written and 1 byte re-read and reset per voxel less:
     * 0.4 GB of a 512^3 update).  d_labels must then not change before gie_fuse has run (include/gie.h: gie_ogm_labels_dev_borrow). */
    const int in_place = may_borrow && be_labels_in_place_ok(m->c, d_labels);
    be_prof(&m->be, GIE_K_CLASSIFY, 0); be_labels(&m->be, m->c, d_labels, in_place); be_prof(&m->be, GIE_K_CLASSIFY, 1);
    be_time(&m->be, 1);
    if (in_place) { m->labels_pending = d_labels; m->c.scan_labels = d_labels; }
    m->has_ogm = 1;
    if (borrowed) *borrowed = in_place;
    return GIE_OK;
}
extern "C" int gie_ogm_labels_dev(gie_mapper *m, const int8_t *d_labels)
{
    return gie_ogm_labels_dev_impl(m, d_labels, 0, nullptr);
}
extern "C" int gie_ogm_labels_dev_borrow(gie_mapper *m, const int8_t *d_labels, int *borrowed)
{
    return gie_ogm_labels_dev_impl(m, d_labels, 1, borrowed);
}
extern "C" int gie_ogm_labels(gie_mapper *m, const int8_t *labels)
{
    int rc = gie_need_pose(m, "gie_ogm_labels"); if (rc) return rc;
    if (!labels) { gie_set_err("gie_ogm_labels: bad arguments"); return GIE_ERR_INVALID; }
    gie_labels_materialise(m);                                      /* (the plane of a scan before this one may live in the staging buffer) */
    const size_t nfl = ((size_t)m->c.N + 3) / 4;                    /* the staging buffer is counted in floats */
    if (nfl > m->sensor_cap) {
        if (m->d_sensor) { be_sync(&m->be); be_free(&m->be, m->d_sensor); }
        m->d_sensor = (float *)be_alloc(&m->be, nfl * sizeof(float), false);
        m->sensor_cap = m->d_sensor ? nfl : 0;
        if (!m->d_sensor) { gie_set_err("sensor buffer allocation failed"); return GIE_ERR_DEVICE; }
    }
    be_h2d(&m->be, m->d_sensor, labels, (size_t)m->c.N);
    return gie_ogm_labels_dev_impl(m, (const int8_t *)m->d_sensor, 1, nullptr);      /* (the library's own staging buffer: borrowed until gie_fuse) */
}
extern "C" int gie_ogm_pointcloud_dev(gie_mapper *m, const float *d_xyz, int n)
{
    int rc = gie_need_pose(m, "gie_ogm_pointcloud"); if (rc) return rc;
    if (n < 0 || (n > 0 && !d_xyz)) { gie_set_err("gie_ogm_pointcloud: bad arguments"); return GIE_ERR_INVALID; }
    if ((size_t)n * 3 > m->pts_cap) {
        if (m->d_pts_g) { be_sync(&m->be); be_free(&m->be, m->d_pts_g); }
        m->d_pts_g = (float *)be_alloc(&m->be, (size_t)n * 3 * sizeof(float), false);
        m->pts_cap = m->d_pts_g ? (size_t)n * 3 : 0;
        if (!m->d_pts_g) { gie_set_err("point buffer allocation failed"); return GIE_ERR_DEVICE; }
    }
    gie_labels_materialise(m);
    m->c.pntcld_mode = 1;
    be_time(&m->be, 0);
    if (n > 0) {
        op_register_point r; r.xyz = d_xyz; r.g = m->d_pts_g;
        be_prof(&m->be, GIE_K_RAY_REGISTER, 0); be_lin(&m->be, m->c, r, n); be_prof(&m->be, GIE_K_RAY_REGISTER, 1);   /* registerLocObs */
        be_prof(&m->be, GIE_K_RAY_FREE, 0); be_free_rays(&m->be, m->c, m->d_pts_g, n); be_prof(&m->be, GIE_K_RAY_FREE, 1);   /* freeLocObs */
    }
    /* getAllocKeys: the ray kernels already flag the block of every cell they count in; what is left
     * is the scan label of those cells — which only gie_read_ogm looks at (fuse works from the
     * counts) — and the robot sphere of for_motion_planner */
    if (m->c.for_motion_planner) { be_prof(&m->be, GIE_K_RAY_FINAL, 0); be_vox(&m->be, m->c, op_raycast_finalize()); be_prof(&m->be, GIE_K_RAY_FINAL, 1); }
    else m->ogm_unlabelled = 1;
    be_time(&m->be, 1);
    m->has_ogm = 1;
    return GIE_OK;
}
extern "C" int gie_ogm_pointcloud(gie_mapper *m, const float *xyz, int n)
{
    int rc = gie_need_pose(m, "gie_ogm_pointcloud"); if (rc) return rc;
    if (n < 0 || (n > 0 && !xyz)) { gie_set_err("gie_ogm_pointcloud: bad arguments"); return GIE_ERR_INVALID; }
    if (n > 0) { rc = gie_stage_sensor(m, xyz, (size_t)n * 3); if (rc) return rc; }
    return gie_ogm_pointcloud_dev(m, m->d_sensor, n);
}

extern "C" int gie_set_ext_boxes(gie_mapper *m, const float *ll, const float *ur, const uint8_t *act, int n)
{
    if (!m || n < 0 || (n > 0 && (!ll || !ur || !act))) { gie_set_err("gie_set_ext_boxes: bad arguments"); return GIE_ERR_INVALID; }
    if (n > m->box_cap) {
        be_sync(&m->be);
        if (m->d_box_ll) { be_free(&m->be, m->d_box_ll); be_free(&m->be, m->d_box_ur); be_free(&m->be, m->d_box_act); }
        m->d_box_ll = (float *)be_alloc(&m->be, (size_t)n * 3 * sizeof(float), false);
        m->d_box_ur = (float *)be_alloc(&m->be, (size_t)n * 3 * sizeof(float), false);
        m->d_box_act = (uint8_t *)be_alloc(&m->be, (size_t)n, false);
        m->box_cap = n;
        if (!m->d_box_ll || !m->d_box_ur || !m->d_box_act) {
            if (m->d_box_ll) be_free(&m->be, m->d_box_ll);
            if (m->d_box_ur) be_free(&m->be, m->d_box_ur);
            if (m->d_box_act) be_free(&m->be, m->d_box_act);
            m->d_box_ll = m->d_box_ur = nullptr; m->d_box_act = nullptr; m->box_cap = 0; m->c.nbox = 0;
            gie_set_err("gie_set_ext_boxes: device allocation failed"); return GIE_ERR_DEVICE;
        }
    }
    if (n > 0) {
        be_h2d(&m->be, m->d_box_ll, ll, (size_t)n * 3 * sizeof(float));
        be_h2d(&m->be, m->d_box_ur, ur, (size_t)n * 3 * sizeof(float));
        be_h2d(&m->be, m->d_box_act, act, (size_t)n);
    }
    m->c.nbox = n; m->c.box_ll = m->d_box_ll; m->c.box_ur = m->d_box_ur; m->c.box_act = m->d_box_act;
    return GIE_OK;
}

/* The block table of the current pose is built from the one before: that one answers for the blocks it knew (k_cell_alloc), the
 * two tables alternate.  Called right before an allocation pass over all cells (be_block_alloc). */
static void gie_table_roll(gie_mapper *m)
{
    gie_ctx &c = m->c;
    static const int no_prev = GIE_SWITCH("GIE_NO_TAB_PREV", 0);      /* (debugging: every block through the hash) */
    if (m->tab_valid && !no_prev) {
        c.tab_prev = c.blk_tab;
        for (int i = 0; i < 3; i++) c.tab_prev_d[i] = c.tb0[i] - m->tab_tb0[i];
        int32_t *t = c.blk_tab; c.blk_tab = m->blk_tab2; m->blk_tab2 = t;
    } else c.tab_prev = nullptr;
    for (int i = 0; i < 3; i++) m->tab_tb0[i] = c.tb0[i];
    m->tab_valid = 1;
}

/* the stored pairs the last (fused) merge left out, for the voxels of ITS volume that the volume of the current pose no longer
 * holds (gie_pair_flush_voxel): the functor + how many voxels it covers.  rehash: types and block table on the device are no longer
 * that merge's — blocks are found through the hash and a record that has been flushed before (no mark) is left alone. */
static int gie_flush_op(gie_mapper *m, op_pair_flush *out, int rehash)
{
    const gie_ctx &c = m->c;
    op_pair_flush op;
    op.b.rehash = rehash;
    const int sz[3] = { c.X, c.Y, c.Z };
    int w[3];
    for (int i = 0; i < 3; i++) {
        op.b.opvt[i] = m->commit_pvt[i]; op.b.oupvt[i] = m->commit_upvt[i]; op.b.otb0[i] = m->commit_tb0[i];
        const long long s = (long long)m->commit_pvt[i] - c.pvt[i];          /* old local + s = new local */
        long long lo = -s > 0 ? -s : 0, hi = sz[i] - s < sz[i] ? sz[i] - s : sz[i];
        if (hi < lo) hi = lo;
        if (lo > sz[i]) { lo = sz[i]; hi = sz[i]; }
        op.b.lo[i] = (int)lo; op.b.hi[i] = (int)hi; w[i] = (int)(hi - lo);
    }
    if (w[0] == 0 || w[1] == 0 || w[2] == 0) { for (int i = 0; i < 3; i++) { op.b.lo[i] = op.b.hi[i] = 0; w[i] = 0; } }   /* nothing stays */
    op.b.n0 = (c.X - w[0]) * c.Y * c.Z; op.b.n1 = w[0] * (c.Y - w[1]) * c.Z; op.b.n2 = w[0] * w[1] * (c.Z - w[2]);
    *out = op;
    return op.b.n0 + op.b.n1 + op.b.n2;
}

static int gie_fused_mode(const gie_mapper *m);

static int gie_fused_mode(const gie_mapper *m);

/* ---- stages */
extern "C" int gie_fuse(gie_mapper *m)
{
    int rc = gie_need_pose(m, "gie_fuse"); if (rc) return rc;
    if (!m->has_ogm) { gie_set_err("gie_fuse: no scan has been fed since the last fuse (call gie_ogm_* first)"); return GIE_ERR_INVALID; }
    m->has_ogm = 0;
    m->ogm_unlabelled = 0;                /* fuse consumes the scan */
    m->c.track = m->track_want;           /* the order of kernels of this update — fused sweep or the reference's, with the changed-block flags — is settled here */
    const int unmerged = m->deferred && !m->flush_tab_ok;   /* types and block table are no longer those of the last fused merge: a gie_fuse
                                                              without a merge came in between (the staged ABI allows it) */
    m->fuse_fresh = 1;
    be_time(&m->be, 2);
    /* allocHashTB (glb_hash_map.cu:58-113): flag missing blocks, rank them with an exclusive
     * scan, insert + initialise, then resolve the frame's block table */
    be_prof(&m->be, GIE_K_ALLOC, 0);
    const int was_fused = m->deferred && !unmerged;    /* the map update before this one ran fused, and it was the one before */
    int nflush = 0;
    op_pair_flush flush_op;
    if (m->deferred) {
        /* the stored pairs the last (fused) merge left out, for the voxels that are not in this volume any more; before
         * anything of that update — types, pairs, block table — is overwritten, and before blocks are erased.
         * `deferred` stays set until the next merge: when fuses follow each other without one, every one of them
         * flushes what has left the volume of that LAST MERGE by now — the local pair plane is still that merge's — finding
         * the blocks through the hash (types and block table are the un-merged fuse's by then); a record flushed before has
         * lost its mark and is left alone. */
        const gie_ctx &c = m->c;
        op_pair_flush op;
        nflush = gie_flush_op(m, &op, unmerged);
        flush_op = op;
        if (m->c.retain > 0) { be_lin(&m->be, c, op, nflush); nflush = 0; }      /* before blocks are erased; otherwise it shares the frame clear's launch below */
    }
    if (m->c.retain > 0) {
        /* block-pool lifecycle (gie_config.retain_radius_blocks): erase what lies too far behind, before anything is allocated;
         * every GIE_REHASH_PERIOD-th map update the hash table is rebuilt from the live slots, which drops the tombstones */
        be_lin(&m->be, m->c, op_evict(), m->c.max_blocks);
        {   /* how many hash cells this erasure can have turned into tombstones: every live block lay inside the retention box of
             * the update before (what was farther was erased then, and blocks are only allocated inside the volume's box), so at
             * most the block coordinates of that box that are not in this update's.  The host never reads the device's count;
             * the bound decides when the table is rebuilt — a robot that jumps to new ground every update would otherwise use
             * up the table's EMPTY cells (lookups end at one) long before the period is over */
            long long vol_prev = m->retain_box_valid ? 1 : 0, vol_both = m->retain_box_valid ? 1 : 0;
            for (int i = 0; i < 3; i++) {
                const long long lo = (long long)m->c.vb_lo[i] - m->c.retain, hi = (long long)m->c.vb_hi[i] + m->c.retain;
                if (m->retain_box_valid) {
                    vol_prev *= (long long)m->retain_box_hi[i] - m->retain_box_lo[i] + 1;
                    const long long a = lo > m->retain_box_lo[i] ? lo : m->retain_box_lo[i], b = hi < m->retain_box_hi[i] ? hi : m->retain_box_hi[i];
                    vol_both *= b >= a ? b - a + 1 : 0;
                }
                m->retain_box_lo[i] = (int)lo; m->retain_box_hi[i] = (int)hi;
            }
            m->retain_box_valid = 1;
            long long gone = vol_prev - vol_both;
            if (gone > m->c.max_blocks) gone = m->c.max_blocks;
            m->tomb_bound += gone;
        }
        if (++m->evictions >= GIE_REHASH_PERIOD || m->tomb_bound > ((long long)m->c.hmask + 1) / 4) {
            m->evictions = 0; m->tomb_bound = 0;
            be_memset(&m->be, m->c.hkeys, 0xff, ((size_t)m->c.hmask + 1) * sizeof(uint64_t));
            be_lin(&m->be, m->c, op_rehash(), m->c.max_blocks);
        }
    }
    {   /* everything that has to be zero for this map update, in one launch.  Per-tile summaries:
         * last frame's "known" flags (tknown_prev) tell which tiles still hold stale _glb_type; the
         * ray-touch flags were consumed by the OGM stage and are cleared for the next scan. */
        gie_ctx &c = m->c;
        const size_t ntile = (size_t)c.tfd[0] * c.tfd[1] * c.tfd[2];
        uint8_t *t = c.tknown; c.tknown = c.tknown_prev; c.tknown_prev = t;
        {   /* the bounds of what the previous map update committed per tile (gie_tile_oldskip): valid when that update ran fused */
            int32_t *tm = c.tmax; c.tmax = c.tmax_prev; c.tmax_prev = tm;
            c.prev_valid = was_fused;
            for (int i = 0; i < 3; i++) c.prev_shift[i] = c.pvt[i] - m->commit_pvt[i];
            uint8_t *ts = c.tskip; c.tskip = c.tskip_prev; c.tskip_prev = ts;          /* (cleared below; flagged again behind the occupancy fusion) */
            c.skip2_ok = was_fused;       /* (... and tskip_prev is that update's: flagged at its pose, which is the merge's) */
            for (int i = 0; i < 3; i++) { if (c.ts_pvt[i] != m->commit_pvt[i]) c.skip2_ok = 0; m->tsp_pvt[i] = c.ts_pvt[i]; c.ts_pvt[i] = c.pvt[i]; }
            c.qdefer = 0;                 /* (until the flags are final: host-side state, no query can come in between) */
        }
        gie_clear_list l; l.n = 0;
        auto add = [&l](void *p, size_t bytes) { l.p[l.n] = p; l.bytes[l.n] = (uint32_t)bytes; l.n++; };
        add(c.tflag, 5 * ntile);                                   /* tflag | tunk | tsum | tray | tact */
        add(c.tknown, ntile);
        add(c.zocc, (size_t)c.Z);
        add(c.zredo, (size_t)((c.X + 15) / 16) * c.Y * sizeof(uint32_t));
        add(c.tmax, ntile * sizeof(int32_t));
        add(c.tbmax, ntile * sizeof(int32_t));
        add(c.tskip, ntile);
        add(c.cnt, GIE_CNT_ERR * sizeof(int32_t));                 /* per-frame counters (the sticky error flag survives) */
        add(c.cnt + GIE_CNT_ERR + 1, (GIE_CNT_FRAME_END - GIE_CNT_ERR - 1) * sizeof(int32_t));
        add(c.cnt + GIE_CNT_BAR_B, (GIE_CNT_AUX_END - GIE_CNT_BAR_B) * sizeof(int32_t));
        add(c.lvl_next, 6 * GIE_MAX_LEVELS * sizeof(int32_t));    /* waves C, B and A */
        add(c.wc_flag[0], 2 * ntile * sizeof(int32_t));          /* (zero again after every complete wave C; a wave cut short may leave flags) */
        if (!c.fast_mode) add(c.wb_flag[0], 2 * (size_t)c.max_blocks * sizeof(int32_t));      /* (likewise for wave B) */
        /* (the pair flush touches none of these arrays and none of the pointers swapped above: one launch for both) */
        if (nflush > 0) be_flush_clear(&m->be, c, flush_op, nflush, l); else be_clear(&m->be, l);
    }
    /* + the tiles fuse has to look at (an existing block overlaps them, or they still hold types from
     * earlier frames), listed in the block-initialisation launch; fuse, Mark, commit and pass Z walk their list or
     * sweep the volume — each kernel decides from the length of its list (gie_use_lists) */
    gie_table_roll(m);
    be_block_alloc(&m->be, m->c, m->ncell, m->d_rank, 0, (int)((size_t)m->c.tfd[0] * m->c.tfd[1] * m->c.tfd[2]));
    m->c.tab_prev = m->c.blk_tab; m->c.tab_prev_d[0] = m->c.tab_prev_d[1] = m->c.tab_prev_d[2] = 0;     /* (a second allocation pass of this update — ghost layers — looks into this table first) */
    be_prof(&m->be, GIE_K_ALLOC, 1);
    be_prof(&m->be, GIE_K_FUSE, 0);
    be_fuse(&m->be, m->c, m->c.tl_front);
    be_prof(&m->be, GIE_K_FUSE, 1);
    m->labels_pending = nullptr; m->c.scan_labels = nullptr;      /* (a label plane left in place has been read) */
    {   /* the tiles whose stored records this update's Mark need not read (gie_tile_oldskip: the previous update's bounds and this
         * pose — and an obstacle somewhere in the volume, known now that the types are fused), then the records the previous update
         * left to its pair plane for the tiles that are not among them any more (gie_ops.h "deferred records").  If the merge ends
         * up running in the reference's order after all (gie_stream_enable in between), gie_merge_begin catches up the rest. */
        gie_ctx &c = m->c;
        static const int use_bound = GIE_SWITCH("GIE_MARKC_BOUND", 1);     /* 0: always read the stored records (measurements) */
        /* (not for ray-cast scans: a scan observes well under 1 % of the volume, no tile is ever cleared — the launch would be 5 us of
         * a 0.2 - 0.4 ms update for nothing) */
        c.oldskip = (gie_fused_mode(m) && c.prev_valid && use_bound && !c.pntcld_mode) ? 1 : 0;
        be_prof(&m->be, GIE_K_ALLOC, 0);
        /* the usual case: tskip_prev is the update before's and lies at its pose (prev_shift) — the launch that flags this update's
         * tiles finds the ones to bring up to date on the way; otherwise (a fuse without a merge since; no bound to go by) the
         * tiles of the flags' own plane are enumerated (be_coc_catchup) */
        c.catchup_fast = (m->coc_pending && c.oldskip && c.skip2_ok) ? 1 : 0;
        if (c.oldskip) be_tile_oldskip(&m->be, c, m->commit_upvt);
        if (m->coc_pending && !c.catchup_fast) {
            gie_catchup p;
            /* (after a fuse WITHOUT a merge the flags are that fuse's, whose sweep never ran: also its tiles flagged 1 — "the sweep stores
             * them" — still hold voxels the merge before left to its pair plane: every flagged tile is caught up; round 6) */
            p.flags = c.tskip_prev; p.all = unmerged ? 1 : 0;
            for (int i = 0; i < 3; i++) { p.fpvt[i] = m->tsp_pvt[i]; p.ppvt[i] = m->commit_pvt[i]; p.pupvt[i] = m->commit_upvt[i]; }
            be_coc_catchup(&m->be, c, p);
        }
        if (m->coc_pending) m->coc_pending = c.oldskip;       /* what stays deferred lies in this update's tskip tiles (none without the bound) */
        /* "lazy pairs": the flagged tiles this update's sweep will not leave flagged again get their pairs into the plane now — the batch
         * obstacles they are derived from are still there, and the next Mark reads old pairs by local index */
        if (m->lazy_pending) {
            const int stay = (gie_fused_mode(m) && c.oldskip && c.lazy_ok) ? 1 : 0;      /* (what the sweep's short way will go by: tskip 2, records deferred) */
            be_pair_materialise(&m->be, c, stay);
            m->lazy_pending = stay;
        }
        be_prof(&m->be, GIE_K_ALLOC, 1);
        c.qdefer = m->coc_pending;
        for (int i = 0; i < 3; i++) { c.pp_pvt[i] = m->commit_pvt[i]; c.pp_upvt[i] = m->commit_upvt[i]; }
    }
    be_time(&m->be, 3);
    m->flush_tab_ok = 0;
    return GIE_OK;
}

/* Mark and commit as one sweep (gie_ops.h "Mark + commit") unless the changed-block flags are on;
 * GIE_FUSED=0 keeps the reference's order Mark -> obtainFrontiers -> waves -> commit (tests run both) */
static int gie_fused_mode(const gie_mapper *m)
{
    static const int env = GIE_SWITCH("GIE_FUSED", 1);
    return (env && !m->c.track) ? 1 : 0;
}

/* every record still left to the pair plane, now (gie_ops.h "deferred records"): before something reads or writes stored records of
 * voxels inside the volume out of turn — the reference's order of kernels, ghosts imported ahead of a pose's gie_fuse */
static void gie_catchup_everything(gie_mapper *m)
{
    if (!m->coc_pending) return;
    gie_ctx &c = m->c;
    gie_catchup p;
    p.flags = c.tskip; p.all = 1;
    for (int i = 0; i < 3; i++) { p.fpvt[i] = c.ts_pvt[i]; p.ppvt[i] = c.pp_pvt[i]; p.pupvt[i] = c.pp_upvt[i]; }
    be_coc_catchup(&m->be, c, p);
    m->coc_pending = 0; c.qdefer = 0;
}

extern "C" int gie_batch_edt(gie_mapper *m)
{
    int rc = gie_need_pose(m, "gie_batch_edt"); if (rc) return rc;
    be_time(&m->be, 4);
    if (m->c.bcoc == m->c.bcoc_lazy) { uint32_t *t = m->c.bcoc; m->c.bcoc = m->bcoc_alt; m->bcoc_alt = t; }      /* ("lazy pairs": pass Z writes the OTHER plane) */
    be_prof(&m->be, GIE_K_EDT_ZFACES, 0);
    be_edt_prep(&m->be, m->c);          /* plane list + reader masks (the tile skip flags are gie_fuse's since round 5) */
    be_prof(&m->be, GIE_K_EDT_ZFACES, 1);
    const int partial = m->c.tfd[2] <= 64;
    be_edt(&m->be, m->c, partial ? 0 : 1);   /* brackets its three passes itself (GIE_K_EDT_Y/X/Z); pass Z only where the result is read */
    m->edt_partial = partial;
    be_time(&m->be, 5);
    return GIE_OK;
}

/* first half of the merge: MarkLimitedObserve — and the commit of the Mark-time pairs, so that what a tiled run exports
 * between the two halves (gie_halo_export*) is this map update's state, not the previous one's */
extern "C" int gie_merge_begin(gie_mapper *m)
{
    int rc = gie_need_pose(m, "gie_merge"); if (rc) return rc;
    /* the seed counters, the barrier words of the waves and the per-round arrays are zeroed by gie_fuse's frame clear: a second
     * merge of the same map update would append to what the first one left */
    if (!m->fuse_fresh) { gie_set_err("gie_merge: no gie_fuse since the last merge (one merge per map update)"); return GIE_ERR_INVALID; }
    m->fuse_fresh = 0;
    be_time(&m->be, 6);
    m->c.fused = gie_fused_mode(m);
    const int kmark = m->c.fused ? GIE_K_MARKC : GIE_K_MARK;
    be_prof(&m->be, kmark, 0);
    /* (the tiles whose stored records need not be read — tskip — were flagged by gie_fuse) */
    {   /* deferred records (gie_ops.h): a tskip tile's voxels are not stored by the fused sweep.  The reference's order of kernels reads
         * every stored record: what is still owed to the tskip tiles is written first, while the pair plane is the previous update's. */
        gie_ctx &c = m->c;
        c.coc_defer = (c.fused && c.oldskip) ? 1 : 0;      /* (also for one tile of several: the face layers are exported through gie_deferred_coc) */
        if (!c.fused) gie_catchup_everything(m);
    }
    /* "lazy pairs": from this Mark on the flagged tiles' pairs are the ones of THIS update's batch obstacles at THIS update's pivots */
    m->c.bcoc_lazy = m->c.bcoc;
    for (int i = 0; i < 3; i++) { m->c.pp_pvt[i] = m->c.pvt[i]; m->c.pp_upvt[i] = m->c.upvt[i]; }
    if (m->c.fused && m->c.coc_defer && m->c.lazy_ok) m->lazy_pending = 1;
    if (m->c.fused) be_markc(&m->be, m->c, m->c.tl_known);
    else be_vox_list<false>(&m->be, m->c, op_mark(), m->c.tl_known, GIE_CNT_TL_KNOWN, 0);
    be_prof(&m->be, kmark, 1);
    m->merge_open = 1;
    if (m->c.fused) { m->deferred = 1; m->flush_tab_ok = 1; for (int i = 0; i < 3; i++) { m->commit_pvt[i] = m->c.pvt[i]; m->commit_upvt[i] = m->c.upvt[i]; m->commit_tb0[i] = m->c.tb0[i]; } }
    else m->deferred = 0;     /* the reference's order: its commit sweep stores every pair, also the ones an earlier fused update left out (gie_commit_finish) */
    /* from here on the pair plane is THIS update's: what the sweep left to it (coc_defer), or nothing (it stored every record it passed) */
    m->coc_pending = m->c.coc_defer;
    m->c.qdefer = m->coc_pending;
    for (int i = 0; i < 3; i++) { m->c.pp_pvt[i] = m->c.pvt[i]; m->c.pp_upvt[i] = m->c.upvt[i]; }
    return GIE_OK;
}
/* second half: obtainFrontiers, waves A / B / C, commit */
extern "C" int gie_merge_end(gie_mapper *m)
{
    int rc = gie_need_pose(m, "gie_merge"); if (rc) return rc;
    if (!m->merge_open) { gie_set_err("gie_merge_end: gie_merge_begin has not been called"); return GIE_ERR_INVALID; }
    m->merge_open = 0;
    be_prof(&m->be, GIE_K_FRONTIER, 0);
    /* obtainFrontiers looks at surfaces of the known space.  One launch: the tile summary (only tiles with a known voxel can have
     * anything to look at) + the voxels on the faces of the volume, one per lane; then the listed tiles (tsum == 1), a wave
     * per tile out of LDS (k_frontier_faces / k_frontier_tiles; 0.32 -> 0.15 ms on the C5 workload) */
    be_frontier_tiles(&m->be, m->c, m->c.tl_known, GIE_CNT_TL_KNOWN, m->c.tl_front, GIE_CNT_TL_FRONT);
    be_prof(&m->be, GIE_K_FRONTIER, 1);
    m->c.bar_fault = m->bar_fault_left > 0 ? 1 : 0;
    if (m->bar_fault_left > 0) m->bar_fault_left--;
    be_prof(&m->be, GIE_K_WAVE_C, 0); be_waves(&m->be, m->c, m->c.fast_mode ? 0 : 1, m->c.fast_mode ? 1 : 0, 0); be_prof(&m->be, GIE_K_WAVE_C, 1);
    m->c.bar_fault = 0;
    if (!m->c.fused) {
        be_prof(&m->be, GIE_K_COMMIT, 0);
        be_vox_list<true>(&m->be, m->c, op_commit(), m->c.tl_known, GIE_CNT_TL_KNOWN, 0);
        be_prof(&m->be, GIE_K_COMMIT, 1);
    }
    be_time(&m->be, 7);
    return GIE_OK;
}
/* the tiled sequence's first half in the reference's order of kernels: Mark, then (only when Mark and commit are not one
 * sweep) a commit of the Mark-time pairs for the export */
extern "C" int gie_merge_begin_tiled(gie_mapper *m)
{
    int rc = gie_merge_begin(m); if (rc) return rc;
    if (!m->c.fused) be_vox_list<true>(&m->be, m->c, op_commit(), m->c.tl_known, GIE_CNT_TL_KNOWN, 0);
    return GIE_OK;
}

extern "C" int gie_merge(gie_mapper *m)
{
    int rc = gie_merge_begin(m); if (rc) return rc;
    return gie_merge_end(m);
}

extern "C" int gie_step(gie_mapper *m)
{
    int rc = gie_fuse(m); if (rc) return rc;
    rc = gie_batch_edt(m); if (rc) return rc;
    return gie_merge(m);
}

static int gie_fetch_counters(gie_mapper *m)
{
    be_d2h(&m->be, m->h_cnt, m->c.cnt, sizeof(m->h_cnt));
    const int e = m->h_cnt[GIE_CNT_ERR];
    {   /* GIE_DEBUG_COUNTS=1: the lengths of the device-side lists of the last map update, on stderr */
        static const int dbg = GIE_SWITCH("GIE_DEBUG_COUNTS", 0);
        if (dbg) fprintf(stderr, "gie counts: tiles known %d, frontier tiles %d, fuse tiles %d, seeds A/B/C %d %d %d, tiles without a read of the stored records %d, tiles caught up %d, lazy tiles written out %d\n", m->h_cnt[GIE_CNT_TL_KNOWN],
                         m->h_cnt[GIE_CNT_TL_FRONT], m->h_cnt[GIE_CNT_TL_FUSE], m->h_cnt[GIE_CNT_SEED_A], m->h_cnt[GIE_CNT_SEED_B], m->h_cnt[GIE_CNT_SEED_C], m->h_cnt[GIE_CNT_TSKIP],
                         m->h_cnt[GIE_CNT_STATE1], m->h_cnt[GIE_CNT_STATE2]);
    }
    if (e & ~GIE_ERRF_BARRIER) {
        std::string s = "device capacity exceeded:";
        if (e & GIE_ERRF_POOL) s += " block pool (raise gie_config.max_blocks)";
        if (e & GIE_ERRF_QUEUE) s += " frontier queue";
        if (e & GIE_ERRF_HASH) s += " hash table";
        gie_set_err(s);
        return GIE_ERR_CAPACITY;
    }
    if (e & GIE_ERRF_BARRIER) {
        /* not sticky: reported once, then cleared (the per-frame clear leaves the error word alone) */
        be_memset(&m->be, &m->c.cnt[GIE_CNT_ERR], 0, sizeof(int32_t));
        be_sync(&m->be);
        gie_set_err("grid barrier of the wavefront kernel timed out (its workgroups were not all resident: another process is holding the "
                    "device); this map update is incomplete, the next one runs normally");
        return GIE_ERR_TIMEOUT;
    }
    return GIE_OK;
}

extern "C" int gie_sync(gie_mapper *m)
{
    if (!m) { gie_set_err("gie_sync: null handle"); return GIE_ERR_INVALID; }
    if (be_sync(&m->be) != 0) return GIE_ERR_DEVICE;
    return gie_fetch_counters(m);
}

/* ---- readers */
extern "C" int gie_read_local(gie_mapper *m, float *edt, int8_t *type, int32_t *dist_sq, int32_t *coc_xyz)
{
    if (!m) { gie_set_err("gie_read_local: null handle"); return GIE_ERR_INVALID; }
    const size_t N = (size_t)m->c.N;
    static const int edt_raw = GIE_SWITCH("GIE_EDT_RAW", 0);     /* measurement builds keep time stamps in the plane (tools/wave_timing.py) */
    if (edt && edt_raw) be_d2h(&m->be, edt, m->c.edt, N * sizeof(float));
    else if (edt) {      /* `_edt_D` is derived from the pairs where the reference would have written it (gie_ops.h gie_edt_value) */
        float *de = (float *)gie_scratch(m, 0, N * 4, "gie_read_local");
        if (!de) return GIE_ERR_DEVICE;
        op_export_edt op; op.out = de;
        be_lin(&m->be, m->c, op, m->c.N);
        be_d2h(&m->be, edt, de, N * sizeof(float));
    }
    if (type) be_d2h(&m->be, type, m->c.glb_type, N);
    if (dist_sq || coc_xyz) {
        int32_t *dd = dist_sq ? (int32_t *)gie_scratch(m, 0, N * 4, "gie_read_local") : nullptr;
        int32_t *dc = coc_xyz ? (int32_t *)gie_scratch(m, 1, N * 12, "gie_read_local") : nullptr;
        if ((dist_sq && !dd) || (coc_xyz && !dc)) return GIE_ERR_DEVICE;
        op_export_pair op; op.d = dd; op.coc = dc;
        be_lin(&m->be, m->c, op, m->c.N);
        if (dd) be_d2h(&m->be, dist_sq, dd, N * 4);
        if (dc) be_d2h(&m->be, coc_xyz, dc, N * 12);
        gie_scratch_trim(m);
    }
    return gie_sync(m);
}
extern "C" int gie_read_ogm(gie_mapper *m, int8_t *inst_type, int32_t *ray_count)
{
    if (!m) { gie_set_err("gie_read_ogm: null handle"); return GIE_ERR_INVALID; }
    if (m->ogm_unlabelled) { be_vox(&m->be, m->c, op_raycast_finalize()); m->ogm_unlabelled = 0; }   /* the scan labels of a ray-cast scan, on demand */
    gie_labels_materialise(m);                                                                        /* ... and of a label plane left in place */
    if (inst_type) be_d2h(&m->be, inst_type, m->c.inst_type, (size_t)m->c.N);
    if (ray_count) be_d2h(&m->be, ray_count, m->c.ray_count, (size_t)m->c.N * 4);
    return gie_sync(m);
}
extern "C" int gie_read_batch_edt(gie_mapper *m, int32_t *dist_sq, int32_t *coc)
{
    if (!m) { gie_set_err("gie_read_batch_edt: null handle"); return GIE_ERR_INVALID; }
    const size_t N = (size_t)m->c.N;
    if (m->edt_partial && !GIE_SWITCH("GIE_EDT_EXPORT_PARTIAL", 0)) {   /* complete the tiles the map update itself never reads */
        be_edt_z(&m->be, m->c, 1);
        m->edt_partial = 0;
    }
    if (dist_sq || coc) {
        int32_t *dd = dist_sq ? (int32_t *)gie_scratch(m, 0, N * 4, "gie_read_batch_edt") : nullptr;
        int32_t *dc = coc ? (int32_t *)gie_scratch(m, 1, N * 12, "gie_read_batch_edt") : nullptr;
        if ((dist_sq && !dd) || (coc && !dc)) return GIE_ERR_DEVICE;
        op_export_bcoc op; op.d = dd; op.coc = dc;
        be_lin(&m->be, m->c, op, m->c.N);
        if (dd) be_d2h(&m->be, dist_sq, dd, N * 4);
        if (dc) be_d2h(&m->be, coc, dc, N * 12);
        gie_scratch_trim(m);
    }
    return gie_sync(m);
}
static void gie_fill_costmap_hdr(const gie_mapper *m, gie_costmap_hdr *hdr);
extern "C" int gie_read_costmap(gie_mapper *m, gie_seendist *payload, gie_costmap_hdr *hdr)
{
    if (!m) { gie_set_err("gie_read_costmap: null handle"); return GIE_ERR_INVALID; }
    if (payload) {
        const size_t N = (size_t)m->c.N;
        gie_seendist *d = (gie_seendist *)gie_scratch(m, 1, N * sizeof(gie_seendist), "gie_read_costmap");
        if (!d) return GIE_ERR_DEVICE;
        op_costmap op; op.out = d;
        be_lin(&m->be, m->c, op, m->c.N);
        be_d2h(&m->be, payload, d, N * sizeof(gie_seendist));
        gie_scratch_trim(m);
    }
    if (hdr) gie_fill_costmap_hdr(m, hdr);
    return gie_sync(m);
}
static void gie_fill_costmap_hdr(const gie_mapper *m, gie_costmap_hdr *hdr)
{
    hdr->x_size = m->c.X; hdr->y_size = m->c.Y; hdr->z_size = m->c.Z;
    hdr->x_origin = m->msg_origin[0]; hdr->y_origin = m->msg_origin[1]; hdr->z_origin = m->msg_origin[2];
    hdr->width = m->c.voxel_width; hdr->type = 1; hdr->pad[0] = hdr->pad[1] = hdr->pad[2] = 0;
}
/* the payload into a DEVICE buffer of the caller's (GPU planners), asynchronous on the mapper's stream */
extern "C" int gie_read_costmap_dev(gie_mapper *m, gie_seendist *d_payload, gie_costmap_hdr *hdr)
{
    if (!m || !d_payload) { gie_set_err("gie_read_costmap_dev: bad arguments"); return GIE_ERR_INVALID; }
    op_costmap op; op.out = d_payload;
    be_lin(&m->be, m->c, op, m->c.N);
    if (hdr) gie_fill_costmap_hdr(m, hdr);
    return GIE_OK;
}
/* CostMap publishing that does not stall the mapper (include/gie.h): the SeenDist conversion on the mapper's stream into a device
 * staging buffer, then ONE asynchronous copy into pinned host memory on the library's copy stream — the next map update's kernels
 * do not queue up behind 8 bytes per voxel of PCIe traffic.  Two pinned buffers alternate: the one gie_costmap_acquire handed out
 * last stays untouched during the next publish. */
extern "C" int gie_costmap_publish(gie_mapper *m, gie_costmap_hdr *hdr)
{
    if (!m) { gie_set_err("gie_costmap_publish: null handle"); return GIE_ERR_INVALID; }
    const size_t bytes = (size_t)m->c.N * sizeof(gie_seendist);
    if (!m->d_cm) {
        m->d_cm = (gie_seendist *)be_alloc(&m->be, bytes, false);
        for (int i = 0; i < 2; i++) m->h_cm[i] = be_host_alloc(&m->be, bytes);
        m->cm_bytes = bytes;
        if (!m->d_cm || !m->h_cm[0] || !m->h_cm[1]) {
            if (m->d_cm) be_free(&m->be, m->d_cm);
            for (int i = 0; i < 2; i++) if (m->h_cm[i]) be_host_free(&m->be, m->h_cm[i]);
            m->d_cm = nullptr; m->h_cm[0] = m->h_cm[1] = nullptr;
            gie_set_err("gie_costmap_publish: staging allocation failed (8 bytes per voxel on the device, twice that in pinned host memory)"); return GIE_ERR_DEVICE;
        }
    }
    m->cm_slot ^= 1;
    op_costmap op; op.out = m->d_cm;
    be_side_copy_begin(&m->be);                       /* (the staging buffer is free again: the mapper's stream waits for the copy before) */
    be_lin(&m->be, m->c, op, m->c.N);
    be_side_copy(&m->be, m->h_cm[m->cm_slot], m->d_cm, bytes);
    m->cm_pending = 1;
    if (hdr) gie_fill_costmap_hdr(m, hdr);
    return GIE_OK;
}
extern "C" int gie_costmap_acquire(gie_mapper *m, const gie_seendist **payload)
{
    if (!m || !payload) { gie_set_err("gie_costmap_acquire: bad arguments"); return GIE_ERR_INVALID; }
    if (!m->cm_pending && !m->h_cm[0]) { gie_set_err("gie_costmap_acquire: nothing has been published"); return GIE_ERR_INVALID; }
    if (m->cm_pending) { if (be_side_copy_wait(&m->be) != 0) return GIE_ERR_DEVICE; m->cm_pending = 0; }
    *payload = (const gie_seendist *)m->h_cm[m->cm_slot];
    return GIE_OK;
}
/* GlbHashMap lookups for GPU planners: coordinates and results stay on the device, the kernel is enqueued on the mapper's stream */
extern "C" int gie_query_global_dev(gie_mapper *m, const int32_t *d_xyz, int n, gie_voxel *d_out)
{
    if (!m || n < 0 || (n > 0 && (!d_xyz || !d_out))) { gie_set_err("gie_query_global_dev: bad arguments"); return GIE_ERR_INVALID; }
    if (n == 0) return GIE_OK;
    op_query op; op.xyz = d_xyz; op.out = d_out;
    be_lin(&m->be, m->c, op, n);
    return GIE_OK;
}
extern "C" int gie_query_global(gie_mapper *m, const int32_t *xyz, int n, gie_voxel *out)
{
    if (!m || n < 0 || (n > 0 && (!xyz || !out))) { gie_set_err("gie_query_global: bad arguments"); return GIE_ERR_INVALID; }
    if (n == 0) return GIE_OK;
    int32_t *dx = (int32_t *)gie_scratch(m, 0, (size_t)n * 12, "gie_query_global");
    gie_voxel *dv = (gie_voxel *)gie_scratch(m, 1, (size_t)n * sizeof(gie_voxel), "gie_query_global");
    if (!dx || !dv) return GIE_ERR_DEVICE;
    be_h2d(&m->be, dx, xyz, (size_t)n * 12);
    op_query op; op.xyz = dx; op.out = dv;
    be_lin(&m->be, m->c, op, n);
    be_d2h(&m->be, out, dv, (size_t)n * sizeof(gie_voxel));
    return gie_sync(m);
}
extern "C" int gie_get_stats(gie_mapper *m, gie_frame_stats *s)
{
    if (!m || !s) { gie_set_err("gie_get_stats: bad arguments"); return GIE_ERR_INVALID; }
    int rc = gie_sync(m);
    int32_t pcs[2] = { 0, 0 };
    be_d2h(&m->be, pcs, m->c.pool_count, 8);
    const int32_t pc = pcs[0] - pcs[1] - m->pool_base;    /* live blocks: handed out minus the ones on the free list */
    const int32_t *h = m->h_cnt;
    memset(s, 0, sizeof(*s));
    s->frame = m->c.map_ct; s->blocks_total = pc; s->blocks_new = h[GIE_CNT_NEWBLK];
    s->seeds_a = h[GIE_CNT_SEED_A]; s->seeds_b = h[GIE_CNT_SEED_B]; s->seeds_c = h[GIE_CNT_SEED_C];
    s->front_b = h[GIE_CNT_FRONT_B]; s->front_c = h[GIE_CNT_FRONT_C];
    s->visits_a = h[GIE_CNT_VIS_A]; s->visits_b = h[GIE_CNT_VIS_B]; s->visits_c = h[GIE_CNT_VIS_C];
    s->levels_a = h[GIE_CNT_LVL_A]; s->levels_b = h[GIE_CNT_LVL_B]; s->levels_c = h[GIE_CNT_LVL_C];
    be_times(&m->be, &s->us_ogm, &s->us_fuse, &s->us_edt, &s->us_merge);
    s->known_tiles = h[GIE_CNT_TL_KNOWN]; s->frontier_tiles = h[GIE_CNT_TL_FRONT];
    memcpy(&s->total_visits_a, &h[GIE_CNT_TOT_A], 8); memcpy(&s->total_visits_b, &h[GIE_CNT_TOT_B], 8); memcpy(&s->total_visits_c, &h[GIE_CNT_TOT_C], 8);
    return rc;
}
/* ---- changed-block streaming */
#define GIE_STREAM_CHUNK 2048                /* blocks per staging buffer: 2048 x (10 KB + 12 B) = 20 MB */
static const size_t GIE_STREAM_BLK_BYTES = (size_t)GIE_VBSZ * sizeof(gie_voxel);
/* blocks per trip through the staging buffers; GIE_STREAM_CHUNK_BLOCKS (1..2048) shrinks it so that
 * small test volumes exercise the multi-chunk pipeline */
static int gie_stream_chunk_blocks()
{
    const int v = GIE_SWITCH("GIE_STREAM_CHUNK_BLOCKS", 0);
    return (v >= 1 && v <= GIE_STREAM_CHUNK) ? v : GIE_STREAM_CHUNK;
}

extern "C" int gie_stream_enable(gie_mapper *m, int on)
{
    if (!m) { gie_set_err("gie_stream_enable: null handle"); return GIE_ERR_INVALID; }
    m->track_want = on ? 1 : 0;           /* (from the next gie_fuse on: include/gie.h) */
    return GIE_OK;
}
extern "C" int gie_stream_changed(gie_mapper *m, int32_t *keys, gie_voxel *blocks, int max_blocks, int32_t *n_changed)
{
    if (!m || max_blocks < 0) { gie_set_err("gie_stream_changed: bad arguments"); return GIE_ERR_INVALID; }
    gie_ctx &c = m->c;
    const int cap = c.max_blocks;
    if (!m->d_srank) {
        m->d_srank = (int32_t *)be_alloc(&m->be, (size_t)cap * 4, false);
        m->d_slist = (int32_t *)be_alloc(&m->be, (size_t)cap * 4, false);
        if (!m->d_srank || !m->d_slist) { gie_set_err("gie_stream_changed: device allocation failed"); return GIE_ERR_DEVICE; }
    }
    be_exclusive_scan(&m->be, c.g_dirty, m->d_srank, cap);
    int32_t last[2] = { 0, 0 };
    be_d2h(&m->be, &last[0], m->d_srank + (cap - 1), 4);
    be_d2h(&m->be, &last[1], c.g_dirty + (cap - 1), 4);
    const int total = last[0] + last[1];
    if (n_changed) *n_changed = total;
    if (!keys || !blocks || total == 0 || max_blocks == 0) return gie_sync(m);
    const int deliver = total < max_blocks ? total : max_blocks;
    op_stream_list ol; ol.rank = m->d_srank; ol.list = m->d_slist;
    be_lin(&m->be, c, ol, cap);
    const size_t chunk_bytes = (size_t)GIE_STREAM_CHUNK * (GIE_STREAM_BLK_BYTES + 12);
    for (int i = 0; i < 2; i++) {
        if (!m->d_stage[i]) m->d_stage[i] = be_alloc(&m->be, chunk_bytes, false);
        if (!m->h_stage[i]) m->h_stage[i] = be_host_alloc(&m->be, chunk_bytes);
        if (!m->d_stage[i] || !m->h_stage[i]) { gie_set_err("gie_stream_changed: staging allocation failed"); return GIE_ERR_DEVICE; }
    }
    /* chunk k: gather on the device → async copy into pinned buffer k&1; meanwhile the host
     * unpacks chunk k-1 from the other pinned buffer into the caller's arrays */
    const int CH = gie_stream_chunk_blocks();
    const int nchunk = (deliver + CH - 1) / CH;
    for (int k = 0; k <= nchunk; k++) {
        if (k < nchunk) {
            const int first = k * CH, nb = deliver - first < CH ? deliver - first : CH;
            char *d = (char *)m->d_stage[k & 1];
            op_stream_gather og; og.list = m->d_slist; og.first = first;
            og.out = (gie_voxel *)d; og.keys = (int32_t *)(d + (size_t)GIE_STREAM_CHUNK * GIE_STREAM_BLK_BYTES);
            be_lin(&m->be, c, og, nb * GIE_VBSZ);
            op_stream_clear oc; oc.list = m->d_slist; oc.first = first;
            be_lin(&m->be, c, oc, nb);
            be_d2h_async(&m->be, m->h_stage[k & 1], d, (size_t)nb * GIE_STREAM_BLK_BYTES, k & 1);
            be_d2h_async(&m->be, (char *)m->h_stage[k & 1] + (size_t)GIE_STREAM_CHUNK * GIE_STREAM_BLK_BYTES,
                         d + (size_t)GIE_STREAM_CHUNK * GIE_STREAM_BLK_BYTES, (size_t)nb * 12, k & 1);
        }
        if (k > 0) {
            const int first = (k - 1) * CH, nb = deliver - first < CH ? deliver - first : CH;
            const char *h = (const char *)m->h_stage[(k - 1) & 1];
            be_wait(&m->be, (k - 1) & 1);
            memcpy(blocks + (size_t)first * GIE_VBSZ, h, (size_t)nb * GIE_STREAM_BLK_BYTES);
            memcpy(keys + 3 * (size_t)first, h + (size_t)GIE_STREAM_CHUNK * GIE_STREAM_BLK_BYTES, (size_t)nb * 12);
        }
    }
    return gie_sync(m);
}
extern "C" int gie_get_pivot(gie_mapper *m, int32_t pvt[3])
{
    if (!m || !pvt) { gie_set_err("gie_get_pivot: bad arguments"); return GIE_ERR_INVALID; }
    pvt[0] = m->c.pvt[0]; pvt[1] = m->c.pvt[1]; pvt[2] = m->c.pvt[2];
    return GIE_OK;
}

/* ---- tiling: halo exchange + refinement (include/gie.h) */
/* The halo entry points read and write the global map through the frame's block table (gie_gvox_tab) and the ABI only asks them
 * for a pose — so they may run between gie_set_pose and that pose's gie_fuse, when the table on the device is still the previous
 * pose's (the ghost-block pass then left a table the next gie_fuse misread; and an export read the cells of the new
 * origin out of the old table).  Before any of them touches the map:
 *  - the stored pairs the last fused merge still owes to the voxels that have left its volume are written (as gie_fuse would):
 *    ghosts land just outside the NEW volume, on exactly those voxels, and a flush after the import would overwrite them;
 *    gie_fuse's own flush then goes by the marks (rehash form);
 *  - the block table is rebuilt at this pose's origin (the allocation pass of gie_fuse, nothing flagged but what a scan of this
 *    pose has observed so far). */
static void gie_owed_pairs_before_import(gie_mapper *m)
{
    gie_ctx &c = m->c;
    gie_catchup_everything(m);            /* (the records of tskip tiles too: a ghost may land on one of their voxels) */
    if (!m->deferred || m->flushed_ct == c.map_ct) return;
    if (m->commit_pvt[0] == c.pvt[0] && m->commit_pvt[1] == c.pvt[1] && m->commit_pvt[2] == c.pvt[2]) return;
    op_pair_flush op;
    const int n = gie_flush_op(m, &op, m->flush_tab_ok ? 0 : 1);
    be_lin(&m->be, c, op, n);
    m->flushed_ct = c.map_ct;
    m->flush_tab_ok = 0;
}
static int gie_table_is_this_poses(const gie_mapper *m)
{ return m->tab_valid && m->tab_tb0[0] == m->c.tb0[0] && m->tab_tb0[1] == m->c.tb0[1] && m->tab_tb0[2] == m->c.tb0[2]; }
/* ghost voxels need their blocks: the allocation pass of gie_fuse again, over every cell of the block table */
static void gie_ghost_block_alloc(gie_mapper *m)
{
    gie_ctx &c = m->c;
    gie_owed_pairs_before_import(m);
    if (!gie_table_is_this_poses(m)) { gie_table_roll(m); m->flush_tab_ok = 0; }      /* (the table of the last fused merge is the OTHER one now) */
    be_block_alloc(&m->be, c, m->ncell, m->d_rank, 1);
    c.tab_prev = c.blk_tab; c.tab_prev_d[0] = c.tab_prev_d[1] = c.tab_prev_d[2] = 0;
}
static void gie_table_for_pose(gie_mapper *m)
{
    if (!gie_table_is_this_poses(m)) gie_ghost_block_alloc(m);
}
extern "C" int gie_set_tile(gie_mapper *m, const int32_t off[3], const int32_t whole[3])
{
    if (!m || !off || !whole) { gie_set_err("gie_set_tile: bad arguments"); return GIE_ERR_INVALID; }
    for (int i = 0; i < 3; i++) {
        if (whole[i] < m->cfg.local_size[i]) { gie_set_err("gie_set_tile: whole volume smaller than the tile"); return GIE_ERR_INVALID; }
        m->next_off[i] = off[i]; m->next_whole[i] = whole[i];
    }
    return GIE_OK;
}
extern "C" int gie_halo_count(gie_mapper *m, int face)
{
    if (!m || face < 0 || face > 5) { gie_set_err("gie_halo_count: bad arguments"); return -1; }
    return gie_face_count(m->c, face);
}
extern "C" int gie_halo_export_dev(gie_mapper *m, int face, gie_halo_voxel *d_out)
{
    int rc = gie_need_pose(m, "gie_halo_export"); if (rc) return rc;
    if (face < 0 || face > 5 || !d_out) { gie_set_err("gie_halo_export: bad arguments"); return GIE_ERR_INVALID; }
    gie_table_for_pose(m);
    op_halo_export op; op.face = face; op.out = d_out;
    be_lin(&m->be, m->c, op, gie_face_count(m->c, face));
    return GIE_OK;
}
extern "C" int gie_halo_export(gie_mapper *m, int face, gie_halo_voxel *out)
{
    int rc = gie_need_pose(m, "gie_halo_export"); if (rc) return rc;
    if (face < 0 || face > 5 || !out) { gie_set_err("gie_halo_export: bad arguments"); return GIE_ERR_INVALID; }
    const int n = gie_face_count(m->c, face);
    gie_halo_voxel *d = (gie_halo_voxel *)gie_scratch(m, 1, (size_t)n * sizeof(gie_halo_voxel), "gie_halo_export");
    if (!d) return GIE_ERR_DEVICE;
    rc = gie_halo_export_dev(m, face, d);
    be_d2h(&m->be, out, d, (size_t)n * sizeof(gie_halo_voxel));
    return rc;
}
extern "C" int gie_halo_import_dev(gie_mapper *m, int face, const gie_halo_voxel *d)
{
    int rc = gie_need_pose(m, "gie_halo_import"); if (rc) return rc;
    if (face < 0 || face > 5 || !d) { gie_set_err("gie_halo_import: bad arguments"); return GIE_ERR_INVALID; }
    const int n = gie_face_count(m->c, face);
    /* ghost voxels need their blocks: same allocation path as gie_fuse */
    op_halo_need nd; nd.face = face; nd.in = d;
    be_lin(&m->be, m->c, nd, n);
    gie_ghost_block_alloc(m);
    op_halo_import im; im.face = face; im.in = d;
    be_lin(&m->be, m->c, im, n);
    return GIE_OK;
}
extern "C" int gie_halo_import(gie_mapper *m, int face, const gie_halo_voxel *in)
{
    int rc = gie_need_pose(m, "gie_halo_import"); if (rc) return rc;
    if (face < 0 || face > 5 || !in) { gie_set_err("gie_halo_import: bad arguments"); return GIE_ERR_INVALID; }
    const int n = gie_face_count(m->c, face);
    gie_halo_voxel *d = (gie_halo_voxel *)gie_scratch(m, 1, (size_t)n * sizeof(gie_halo_voxel), "gie_halo_import");
    if (!d) return GIE_ERR_DEVICE;
    be_h2d(&m->be, d, in, (size_t)n * sizeof(gie_halo_voxel));
    rc = gie_halo_import_dev(m, face, d);
    be_sync(&m->be);
    return rc;
}
/* sparse face layers (include/gie.h) */
extern "C" int gie_halo_export_sparse_dev(gie_mapper *m, int face, gie_halo_entry *d_out, int32_t *d_count)
{
    int rc = gie_need_pose(m, "gie_halo_export_sparse"); if (rc) return rc;
    if (face < 0 || face > 5 || !d_out || !d_count) { gie_set_err("gie_halo_export_sparse: bad arguments"); return GIE_ERR_INVALID; }
    gie_table_for_pose(m);
    be_memset(&m->be, d_count, 0, sizeof(int32_t));
    op_halo_export_sparse op; op.face = face; op.out = d_out; op.count = d_count;
    be_lin(&m->be, m->c, op, gie_face_count(m->c, face));
    return GIE_OK;
}
extern "C" int gie_halo_import_sparse_dev(gie_mapper *m, int face, const gie_halo_entry *d_in, const int32_t *d_count)
{
    int rc = gie_need_pose(m, "gie_halo_import_sparse"); if (rc) return rc;
    if (face < 0 || face > 5 || !d_in || !d_count) { gie_set_err("gie_halo_import_sparse: bad arguments"); return GIE_ERR_INVALID; }
    const int n = gie_face_count(m->c, face);
    op_halo_need_sparse nd; nd.face = face; nd.nface = n; nd.in = d_in; nd.count = d_count;
    be_lin(&m->be, m->c, nd, n);
    gie_ghost_block_alloc(m);
    op_halo_import_sparse im; im.face = face; im.nface = n; im.in = d_in; im.count = d_count;
    be_lin(&m->be, m->c, im, n);
    return GIE_OK;
}
extern "C" int gie_halo_export_sparse(gie_mapper *m, int face, gie_halo_entry *out, int32_t *count)
{
    int rc = gie_need_pose(m, "gie_halo_export_sparse"); if (rc) return rc;
    if (face < 0 || face > 5 || !out || !count) { gie_set_err("gie_halo_export_sparse: bad arguments"); return GIE_ERR_INVALID; }
    const int n = gie_face_count(m->c, face);
    char *d = (char *)gie_scratch(m, 1, (size_t)n * sizeof(gie_halo_entry) + 16, "gie_halo_export_sparse");
    if (!d) return GIE_ERR_DEVICE;
    int32_t *d_count = (int32_t *)d;
    gie_halo_entry *d_out = (gie_halo_entry *)(d + 16);
    rc = gie_halo_export_sparse_dev(m, face, d_out, d_count);
    be_d2h(&m->be, count, d_count, sizeof(int32_t));
    if (*count > 0) be_d2h(&m->be, out, d_out, (size_t)*count * sizeof(gie_halo_entry));
    return rc;
}
extern "C" int gie_halo_import_sparse(gie_mapper *m, int face, const gie_halo_entry *in, int32_t count)
{
    int rc = gie_need_pose(m, "gie_halo_import_sparse"); if (rc) return rc;
    const int n = (face < 0 || face > 5) ? -1 : gie_face_count(m->c, face);
    if (n < 0 || count < 0 || count > n || (count > 0 && !in)) { gie_set_err("gie_halo_import_sparse: bad arguments"); return GIE_ERR_INVALID; }
    if (count == 0) return GIE_OK;
    char *d = (char *)gie_scratch(m, 1, (size_t)count * sizeof(gie_halo_entry) + 16, "gie_halo_import_sparse");
    if (!d) return GIE_ERR_DEVICE;
    be_h2d(&m->be, d, &count, sizeof(int32_t));
    be_h2d(&m->be, d + 16, in, (size_t)count * sizeof(gie_halo_entry));
    rc = gie_halo_import_sparse_dev(m, face, (const gie_halo_entry *)(d + 16), (const int32_t *)d);
    be_sync(&m->be);
    return rc;
}
/* several faces per call: one launch per step instead of one per face and step, and ONE block
 * allocation for all ghost layers (a 2x2x2 tile has three shared faces; the exchange is bound by
 * the number of small launches) */
static int gie_face_set_of(const gie_ctx &c, const void *const p[6], gie_face_set *fs)
{
    int n = 0;
    for (int f = 0; f < 6; f++) { fs->off[f] = n; if (p[f]) n += gie_face_count(c, f); }
    fs->off[6] = n;
    return n;
}
extern "C" int gie_halo_export_all_dev(gie_mapper *m, gie_halo_voxel *const d_out[6])
{
    int rc = gie_need_pose(m, "gie_halo_export_all"); if (rc) return rc;
    if (!d_out) { gie_set_err("gie_halo_export_all: bad arguments"); return GIE_ERR_INVALID; }
    gie_table_for_pose(m);
    op_halo_export_all op;
    const int n = gie_face_set_of(m->c, (const void *const *)d_out, &op.fs);
    for (int f = 0; f < 6; f++) op.out[f] = d_out[f];
    be_lin(&m->be, m->c, op, n);
    return GIE_OK;
}
extern "C" int gie_halo_import_all_dev(gie_mapper *m, const gie_halo_voxel *const d_in[6])
{
    int rc = gie_need_pose(m, "gie_halo_import_all"); if (rc) return rc;
    if (!d_in) { gie_set_err("gie_halo_import_all: bad arguments"); return GIE_ERR_INVALID; }
    op_halo_need_all nd; op_halo_import_all im;
    const int n = gie_face_set_of(m->c, (const void *const *)d_in, &nd.fs);
    im.fs = nd.fs;
    for (int f = 0; f < 6; f++) { nd.in[f] = d_in[f]; im.in[f] = d_in[f]; }
    if (n == 0) return GIE_OK;
    be_lin(&m->be, m->c, nd, n);
    gie_ghost_block_alloc(m);
    be_lin(&m->be, m->c, im, n);
    return GIE_OK;
}
extern "C" int gie_refine(gie_mapper *m, int32_t *seeded)
{
    int rc = gie_need_pose(m, "gie_refine"); if (rc) return rc;
    gie_table_for_pose(m);
    gie_ctx &c = m->c;
    {   /* what the second waves launch of this map update needs zeroed, in one launch */
        gie_clear_list l; l.n = 0;
        l.p[l.n] = c.cnt + GIE_CNT_C; l.bytes[l.n++] = sizeof(int32_t);
        l.p[l.n] = c.cnt + GIE_CNT_BAR_B; l.bytes[l.n++] = 2 * sizeof(int32_t);      /* (the barrier words of the two waves launches: wave C's, waves A / B's) */
        l.p[l.n] = c.lvl_next; l.bytes[l.n++] = 2 * GIE_MAX_LEVELS * sizeof(int32_t);
        be_clear(&m->be, l, c.gate);
    }
    const int nb = 2 * (c.X * c.Y + c.Y * c.Z + c.X * c.Z);
    be_lin(&m->be, c, op_refine(), nb);
    c.fused = gie_fused_mode(m);
    be_waves(&m->be, c, 0, 0, 0);                        /* (its own clear above) */
    if (!c.fused) be_vox_list<true>(&m->be, c, op_commit(), c.tl_known, GIE_CNT_TL_KNOWN, 0);   /* fused: wave C has committed what it merged */
    if (!seeded) return GIE_OK;          /* enqueue only: a fixed number of exchange rounds needs no answer */
    rc = gie_sync(m);
    *seeded = m->h_cnt[GIE_CNT_FRONT_C];
    return rc;
}
/* exchange rounds until no tile changed, without the host (include/gie.h) */
extern "C" int gie_round_gate(gie_mapper *m, const int32_t *d_go)
{
    if (!m) { gie_set_err("gie_round_gate: null handle"); return GIE_ERR_INVALID; }
    m->c.gate = d_go;
    return GIE_OK;
}
extern "C" int gie_refine_dev(gie_mapper *m, int32_t *d_changed)
{
    if (!d_changed) { gie_set_err("gie_refine_dev: bad arguments"); return GIE_ERR_INVALID; }
    int rc = gie_refine(m, nullptr); if (rc) return rc;
    be_round_note(&m->be, m->c, d_changed, m->d_round_stats, nullptr, 0);
    return GIE_OK;
}
extern "C" int gie_round_end(gie_mapper *m, const int32_t *d_go)
{
    if (!m) { gie_set_err("gie_round_end: null handle"); return GIE_ERR_INVALID; }
    m->c.gate = nullptr;
    be_round_note(&m->be, m->c, nullptr, m->d_round_stats, d_go, 1);
    return GIE_OK;
}
extern "C" int gie_round_stats(gie_mapper *m, int64_t out[4])
{
    if (!m || !out) { gie_set_err("gie_round_stats: bad arguments"); return GIE_ERR_INVALID; }
    long long v[4] = { 0, 0, 0, 0 };
    be_d2h(&m->be, v, m->d_round_stats, sizeof(v));
    for (int i = 0; i < 4; i++) out[i] = (int64_t)v[i];
    return gie_sync(m);
}
extern "C" int gie_get_stream(gie_mapper *m, void **stream)
{
    if (!m || !stream) { gie_set_err("gie_get_stream: bad arguments"); return GIE_ERR_INVALID; }
    *stream = be_stream_handle(&m->be);
    return GIE_OK;
}

#if defined(GIE_TEST_HOOKS)
/* measurement aid (not in gie.h): the placement probe on this mapper's planes, median of `reps` launches in ms; only before the
 * first map update (it scribbles over the pair plane and clears it again) */
extern "C" int gie_debug_place_probe(gie_mapper *m, int reps, float *ms)
{
    if (!m || !ms || m->has_pose) { gie_set_err("gie_debug_place_probe: only on a fresh mapper"); return GIE_ERR_INVALID; }
    *ms = be_place_probe(&m->be, m->c, reps > 0 ? (reps & 0xffff) : 3, (reps >> 16) ? (reps >> 16) : 15);      /* (streams to exercise in the upper half of `reps`: 1 type, 2 batch obstacle, 4 pair, 8 stored obstacle) */
    return gie_sync(m);
}
/* test hook (not in gie.h): the wavefront launches of the next `updates` map updates meet at a barrier that cannot complete — the
 * GIE_ERR_TIMEOUT path of include/gie.h without a second process holding the device (tests/test_gpu_parity.py) */
extern "C" int gie_debug_fault_barrier(gie_mapper *m, int updates)
{
    if (!m || updates < 0) { gie_set_err("gie_debug_fault_barrier: bad arguments"); return GIE_ERR_INVALID; }
    m->bar_fault_left = updates;
    return GIE_OK;
}
/* test hook (not in gie.h): rows of the neighbour table of waves A / B (c.g_nbr) that disagree with the hash, over every live block */
extern "C" int gie_debug_nbr_check(gie_mapper *m, int32_t *mismatches)
{
    if (!m || !mismatches) { gie_set_err("gie_debug_nbr_check: bad arguments"); return GIE_ERR_INVALID; }
    int32_t *d = (int32_t *)gie_scratch(m, 0, sizeof(int32_t), "gie_debug_nbr_check");
    if (!d) return GIE_ERR_DEVICE;
    be_memset(&m->be, d, 0, sizeof(int32_t));
    op_nbr_check op; op.bad = d;
    be_lin(&m->be, m->c, op, 6 * m->c.max_blocks);
    be_d2h(&m->be, mismatches, d, sizeof(int32_t));
    return gie_sync(m);
}
#endif /* GIE_TEST_HOOKS */
extern "C" int gie_profile_enable(gie_mapper *m, int on)
{
    if (!m) { gie_set_err("gie_profile_enable: null handle"); return GIE_ERR_INVALID; }
    be_prof_enable(&m->be, on);
    return GIE_OK;
}
extern "C" int gie_profile_read(gie_mapper *m, gie_kernel_time *out, int max_entries)
{
    if (!m || !out || max_entries < GIE_K_NUM) { gie_set_err("gie_profile_read: need room for all kernels"); return GIE_ERR_INVALID; }
    float ms[GIE_K_NUM]; int n[GIE_K_NUM];
    be_prof_collect(&m->be, ms, n, GIE_K_NUM);
    for (int i = 0; i < GIE_K_NUM; i++) {
        memset(out[i].name, 0, sizeof(out[i].name));
        strncpy(out[i].name, gie_kernel_names[i], sizeof(out[i].name) - 1);
        out[i].total_ms = ms[i]; out[i].launches = n[i];
    }
    return GIE_K_NUM;
}
