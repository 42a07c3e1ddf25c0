/*
 * gie_emu.cpp — TEST-ONLY sequential stand-in for the HIP backend of gie_api.inc.h.
 *
 * Built into tests/emu/libgie_emu.so by the CPU test-suite so that the per-voxel / per-entry
 * logic in gie_ops.h and the orchestration in gie_api.inc.h (shared verbatim with the product)
 * can be checked against the oracle in a container without a GPU.  It is never shipped, never
 * loaded by the gie package, and bench.py never touches it.  The wave-cooperative HIP kernels
 * (EDT passes, ballot compaction, persistent BFS kernels) are NOT exercised here — only
 * `pytest -m gpu` covers them.
 */
#include "gie_platform_emu.h"   /* plain memory + a wavefront of one lane under the product's gie_ops.h / gie_functors.h (defines GIE_HOST_EMU) */
#define GIE_TEST_HOOKS 1      /* the emulation is test infrastructure: the switches of gie_api.inc.h read the environment here */
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include "gie_emu_ops.h"

struct be_state { int dummy; };
static void gie_set_err(const std::string &s);
static int be_init(be_state *, int, int) { return 0; }
static void be_fini(be_state *) {}
static void *be_alloc(be_state *, size_t bytes, bool zero) { return zero ? calloc(bytes ? bytes : 1, 1) : malloc(bytes ? bytes : 1); }
static void be_free(be_state *, void *p) { free(p); }
static void be_memset(be_state *, void *p, int v, size_t n) { memset(p, v, n); }
static void be_h2d(be_state *, void *d, const void *h, size_t n) { memcpy(d, h, n); }
static void be_d2h(be_state *, void *h, const void *d, size_t n) { memcpy(h, d, n); }
static int be_sync(be_state *) { return 0; }
static float be_place_probe(be_state *, const gie_ctx &, int, int = 15) { return 0.f; }      /* (a device measurement: nothing to emulate) */
static void *be_stream_handle(be_state *) { return nullptr; }
static void *be_host_alloc(be_state *, size_t bytes) { return malloc(bytes ? bytes : 1); }
static void be_host_free(be_state *, void *p) { free(p); }
static void be_d2h_async(be_state *, void *h, const void *d, size_t n, int) { memcpy(h, d, n); }
static void be_wait(be_state *, int) {}
static void be_side_copy_begin(be_state *) {}
static void be_side_copy(be_state *, void *h, const void *d, size_t n) { memcpy(h, d, n); }
static int be_side_copy_wait(be_state *) { return 0; }
static void be_time(be_state *, int) {}
static void be_prof(be_state *, int, int) {}
static void be_prof_enable(be_state *, int) {}
static void be_prof_collect(be_state *, float *ms, int *n, int num) { for (int i = 0; i < num; i++) { ms[i] = 0; n[i] = 0; } }
static void be_times(be_state *, float *a, float *b, float *c, float *d) { *a = *b = *c = *d = 0.f; }
/* a device thread owns the z-column (x, y, z0..z0+7) of one 8x8x8 tile: same order here, so that the
 * per-tile skip flags are exercised exactly as on the device */
template <class F> static void be_vox(be_state *, const gie_ctx &c, const F &f)
{
    for (int z0 = 0; z0 < c.Z; z0 += 8) for (int y = 0; y < c.Y; y++) for (int x = 0; x < c.X; x++) {
        if (f.tile_skip(c, x, y, z0)) continue;
        for (int z = z0; z < z0 + 8 && z < c.Z; z++) if (!f.skip(c, gie_lid(c, x, y, z), x, y, z)) f(c, x, y, z);
    }
}
/* the fuse sweep also records the per-tile known/unknown summaries (column hook on the device) */
static void be_vox(be_state *, const gie_ctx &c, const op_fuse &f)
{
    for (int z0 = 0; z0 < c.Z; z0 += 8) for (int y = 0; y < c.Y; y++) for (int x = 0; x < c.X; x++) {
        if (f.tile_skip(c, x, y, z0)) continue;
        unsigned known = 0, valid = 0;
        for (int z = z0; z < z0 + 8 && z < c.Z; z++) { valid |= 1u << (z - z0); if (f(c, x, y, z)) known |= 1u << (z - z0); }
        f.column(c, x, y, z0, known, valid);
    }
}
/* ... and the Mark + commit sweep the bound of what it committed per tile */
static void be_vox(be_state *, const gie_ctx &c, const op_markc &f)
{
    for (int z0 = 0; z0 < c.Z; z0 += 8) for (int y = 0; y < c.Y; y++) for (int x = 0; x < c.X; x++) {
        if (f.tile_skip(c, x, y, z0)) continue;
        unsigned known = 0, valid = 0; int vmax = 0;
        for (int z = z0; z < z0 + 8 && z < c.Z; z++) { valid |= 1u << (z - z0); if (f.skip(c, gie_lid(c, x, y, z), x, y, z)) continue; const int r = f(c, x, y, z); if (r) known |= 1u << (z - z0); if (r > vmax) vmax = r; }
        f.column_max(c, x, y, z0, known, valid, vmax);
    }
}
template <class F> static void be_list(be_state *, const gie_ctx &c, const F &f, const int32_t *list, int count_idx)
{ const int n = c.cnt[count_idx]; for (int e = 0; e < n; e++) f(c, list[e]); }
/* the list form: only the listed tiles, one (x,y) column of a tile at a time like a device lane */
template <class F> static void be_vox_list_col(const gie_ctx &c, const F &f, int x, int y, int z0)
{ for (int z = z0; z < z0 + 8 && z < c.Z; z++) if (!f.skip(c, gie_lid(c, x, y, z), x, y, z)) f(c, x, y, z); }
static void be_vox_list_col(const gie_ctx &c, const op_fuse &f, int x, int y, int z0)
{
    unsigned known = 0, valid = 0;
    for (int z = z0; z < z0 + 8 && z < c.Z; z++) { valid |= 1u << (z - z0); if (f(c, x, y, z)) known |= 1u << (z - z0); }
    f.column(c, x, y, z0, known, valid);
}
static void be_vox_list_col(const gie_ctx &c, const op_markc &f, int x, int y, int z0)
{
    unsigned known = 0, valid = 0; int vmax = 0;
    for (int z = z0; z < z0 + 8 && z < c.Z; z++) { valid |= 1u << (z - z0); if (f.skip(c, gie_lid(c, x, y, z), x, y, z)) continue; const int r = f(c, x, y, z); if (r) known |= 1u << (z - z0); if (r > vmax) vmax = r; }
    f.column_max(c, x, y, z0, known, valid, vmax);
}
                       /* the block-row kernels are device-only forms of the same functors */

template <bool STAGED, class F> static void be_vox_list(be_state *b, const gie_ctx &c, const F &f, const int32_t *list, int count_idx, int always_list, int = 64)
{
    if (GIE_GATE_CLOSED(c)) return;
    const int n = c.cnt[count_idx];
    if (always_list != 1 && !gie_use_lists(c, n)) { be_vox(b, c, f); return; }        /* the same choice the device kernel makes */
    for (int e = 0; e < n; e++) {
        const int t = list[e];
        const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
        for (int ly = 0; ly < 8; ly++) for (int lx = 0; lx < 8; lx++) {
            const int x = tx * 8 + lx, y = ty * 8 + ly, z0 = tz * 8;
            if (x >= c.X || y >= c.Y || f.tile_skip(c, x, y, z0)) continue;
            be_vox_list_col(c, f, x, y, z0);
        }
    }
}
static void be_markc(be_state *b, const gie_ctx &c, const int32_t *list) { c.cnt[GIE_CNT_LAZY_EXACT] = 1; be_vox_list<true>(b, c, op_markc(), list, GIE_CNT_TL_KNOWN, 0); }      /* (every tile's bound is exact here) */
static void be_fuse(be_state *b, const gie_ctx &c, const int32_t *list) { be_vox_list<true>(b, c, op_fuse(), list, GIE_CNT_TL_FUSE, 0); }
/* the device takes the listed tiles (tsum == 1) a wave per tile and the face voxels by patches of the faces; here: every voxel
 * the tile summary does not rule out (op_frontier::tile_skip / skip), same decisions per voxel */
static void be_frontier_tiles(be_state *b, const gie_ctx &c, const int32_t *known, int known_idx, const int32_t *, int)
{ be_list(b, c, op_tile_summary(), known, known_idx); be_vox(b, c, op_frontier()); }
static int be_labels_in_place_ok(const gie_ctx &, const int8_t *) { return 0; }     /* (a device shortcut: the emulation always copies the plane into _inst_type) */
static void be_labels(be_state *b, const gie_ctx &c, const int8_t *labels, int = 0) { op_classify_labels op; op.labels = labels; be_vox(b, c, op); }
template <class F> static void be_lin(be_state *, const gie_ctx &c, const F &f, int n) { if (GIE_GATE_CLOSED(c)) return; for (int i = 0; i < n; i++) f(c, i); }
template <class F> static void be_range(be_state *, const gie_ctx &c, const F &f, int n) { for (int i = 0; i < n; i++) f(c, i); }
static void be_clear(be_state *, const gie_clear_list &l, const int32_t *gate = nullptr) { if (gate && *gate == 0) return; for (int i = 0; i < l.n; i++) memset(l.p[i], 0, l.bytes[i]); }
static void be_round_note(be_state *, const gie_ctx &c, int32_t *changed, long long *stats, const int32_t *go, int end)
{   /* k_round_note */
    if (!end) { const bool open = !GIE_GATE_CLOSED(c); if (changed) *changed = open ? c.cnt[GIE_CNT_FRONT_C] : 0; stats[0] += 1; if (open) stats[1] += 1; }
    else { stats[2] += 1; if (go && *go != 0) stats[3] += 1; }
}
static void be_flush_clear(be_state *b, const gie_ctx &c, const op_pair_flush &f, int n, const gie_clear_list &l) { be_lin(b, c, f, n); be_clear(b, l); }
static void be_exclusive_scan(be_state *, const int32_t *flag, int32_t *rank, int n);
static void be_block_init(be_state *, const gie_ctx &c, const int32_t *flag, const int32_t *rank, int ncell);
/* sequential allocHashTB: flag → rank → insert → initialise → table */
static void be_block_alloc(be_state *b, const gie_ctx &c, int ncell, int32_t *rank, int, int fuse_list_ntile = 0)
{
    if (GIE_GATE_CLOSED(c)) return;
    /* the device's allocation (k_cell_alloc) asks the block table of the fuse before first: whatever that names has to be what the
     * hash finds (gie_cell_prev_slot) */
    for (int cell = 0; cell < ncell; cell++) {
        const int bx = cell % c.tdim[0], by = (cell / c.tdim[0]) % c.tdim[1], bz = cell / (c.tdim[0] * c.tdim[1]);
        const int prev = gie_cell_prev_slot(c, bx, by, bz);
        if (prev >= 0 && prev != gie_hash_find(c, bx + c.tb0[0], by + c.tb0[1], bz + c.tb0[2])) {
            fprintf(stderr, "gie_emu: the block table of the fuse before names slot %d for block (%d, %d, %d), the hash %d\n", prev, bx + c.tb0[0], by + c.tb0[1], bz + c.tb0[2],
                    gie_hash_find(c, bx + c.tb0[0], by + c.tb0[1], bz + c.tb0[2]));
            abort();
        }
    }
    be_lin(b, c, op_cell_flag(), ncell);
    be_exclusive_scan(b, c.blk_new, rank, ncell);
    op_cell_insert ins; ins.flag = c.blk_new; ins.rank = rank;
    be_lin(b, c, ins, ncell);
    be_block_init(b, c, c.blk_new, rank, ncell);
    be_lin(b, c, op_cell_table(), ncell);
    for (int cell = 0; cell < ncell; cell++) if (c.blk_tab[cell] >= 0) gie_cell_mark_tiles(c, cell);
    if (fuse_list_ntile > 0) be_range(b, c, op_fuse_list(), fuse_list_ntile);     /* (the device builds the list in its block-initialisation launch) */
}
static void be_free_rays(be_state *, const gie_ctx &c, const float *g, int n) { for (int i = 0; i < n; i++) gie_free_ray(c, g, i); }
static void be_exclusive_scan(be_state *, const int32_t *flag, int32_t *rank, int n) { int s = 0; for (int i = 0; i < n; i++) { rank[i] = s; s += flag[i]; } }
static void be_block_init(be_state *, const gie_ctx &c, const int32_t *flag, const int32_t *rank, int ncell)
{
    for (int cell = 0; cell < ncell; cell++) {
        if (!flag[cell]) continue;
        const int slot = gie_emu_slot(c, rank[cell]);
        if (slot >= c.max_blocks) continue;
        for (int i = 0; i < GIE_VBSZ; i++) gie_init_voxel(c, slot, i);
    }
    for (int cell = 0; cell < ncell; cell++) {            /* the neighbour table (k_block_init_list: every key of the pass is in the hash by now) */
        if (!flag[cell]) continue;
        const int slot = gie_emu_slot(c, rank[cell]);
        if (slot >= c.max_blocks) continue;
        for (int k = 0; k < 6; k++) gie_nbr_link(c, slot, k);
    }
    const int total = rank[ncell - 1] + flag[ncell - 1];
    const int nfree = c.retain > 0 ? c.pool_count[1] : 0, from_free = total < nfree ? total : nfree;
    int pc = c.pool_count[0] + (total - from_free); if (pc > c.max_blocks) pc = c.max_blocks;
    c.pool_count[0] = pc; c.pool_count[1] -= c.retain > 0 ? from_free : 0; c.cnt[GIE_CNT_NEWBLK] = total;
}
/* plain restatement of the closed form the HIP EDT kernels implement (see
 * tests/test_oracle_edt.py::test_meijster_tie_rule) */
/* the list of tiles that hold a known voxel (the plane list and the reader masks are device-only shortcuts) */
static void be_edt_prep(be_state *, const gie_ctx &c)
{
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    for (int t = 0; t < ntile; t++) if (c.tknown[t]) c.tl_known[c.cnt[GIE_CNT_TL_KNOWN]++] = t;
}
static void be_coc_catchup(be_state *, const gie_ctx &c, const gie_catchup &p)
{   /* k_coc_catchup */
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    for (int t = 0; t < ntile; t++) if (gie_coc_catchup_tile(c, p, t)) for (int l = 0; l < 64; l++) gie_coc_catchup_column(c, p, t, l);
}
static void be_pair_materialise(be_state *, const gie_ctx &c, int stay)
{   /* k_pair_lazy_list + k_pair_lazy_run: the flagged tiles that will not be flagged again */
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    for (int t = 0; t < ntile; t++) {
        if (!c.tlazy[t] || (stay && c.tskip[t] == 2)) continue;
        for (int l = 0; l < 64; l++) gie_pair_materialise_column(c, t, l);
        c.tlazy[t] = 0;
    }
}
static void be_tile_oldskip(be_state *, const gie_ctx &c, const int pupvt[3])
{   /* k_tile_oldskip + k_coc_catchup_new */
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    int any = 0;
    for (int z = 0; z < c.Z; z++) any |= c.zocc[z];
    std::vector<int> list;
    for (int t = 0; t < ntile; t++) if (gie_tile_oldskip(c, t, any)) list.push_back(t);
    if (c.catchup_fast) for (int t : list) for (int l = 0; l < 64; l++) gie_coc_catchup_newcolumn(c, pupvt, t, l);
}
static void be_edt_z(be_state *, const gie_ctx &, int) {}          /* the emulation always computes every voxel */
static void be_edt(be_state *, const gie_ctx &c, int)
{
    const int X = c.X, Y = c.Y, Z = c.Z;
    for (int z = 0; z < Z; z++) for (int x = 0; x < X; x++) for (int y = 0; y < Y; y++) {
        int best = -1, bd = 1 << 30;
        for (int i = 0; i < Y; i++) if (c.glb_type[gie_lid(c, x, i, z)] == GIE_VOX_OCCUPIED) { const int d = abs(y - i); if (d <= bd) { bd = d; best = i; } }
        c.cy1[gie_lid(c, x, y, z)] = best < 0 ? 0xffff : (uint16_t)best;
    }
    for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) for (int u = 0; u < X; u++) {
        long long bf = 1ll << 60; int bs = -1;
        for (int i = 0; i < X; i++) { const uint16_t cy = c.cy1[gie_lid(c, i, y, z)]; if (cy == 0xffff) continue;
            const long long f = (long long)(u - i) * (u - i) + (long long)(y - cy) * (y - cy); if (f < bf) { bf = f; bs = i; } }
        c.cxy2[gie_lid(c, u, y, z)] = bs < 0 ? 0xffffffffu : ((uint32_t)bs | ((uint32_t)c.cy1[gie_lid(c, bs, y, z)] << 16));
    }
    for (int y = 0; y < Y; y++) for (int x = 0; x < X; x++) for (int u = 0; u < Z; u++) {
        long long bf = 1ll << 60; int bs = -1;
        for (int i = 0; i < Z; i++) { const uint32_t v = c.cxy2[gie_lid(c, x, y, i)]; if (v == 0xffffffffu) continue;
            const int dx = x - (int)(v & 0xffff), dy = y - (int)(v >> 16);
            const long long f = (long long)(u - i) * (u - i) + dx * dx + dy * dy; if (f < bf) { bf = f; bs = i; } }
        const int id = gie_lid(c, x, y, u);
        if (bs < 0) c.bcoc[id] = GIE_BCOC_NONE;
        else { const uint32_t v = c.cxy2[gie_lid(c, x, y, bs)]; c.bcoc[id] = gie_pack_bcoc((int)(v & 0xffff), (int)(v >> 16), bs); }
    }
    {   /* the planes with obstacles, ascending (k_edt_prep on the device): gie_batch_dist_direct walks them */
        int K = 0;
        for (int z = 0; z < Z; z++) if (c.cxy2[gie_lid(c, 0, 0, z)] != 0xffffffffu) c.zlist[K++] = (uint16_t)z;
        *c.zcount = K;
    }
}
/* wave A in the canonical checkerboard block-round schedule (oracle/gie_oracle.c wave_a): sequential statement on the
 * device data structures.  Proposals travel through host maps here (the device keeps those inside a block in LDS, those across
 * block borders in the two proposal planes of the global map). */
static void be_wave_a(be_state *, const gie_ctx &c)
{
    const int n0 = c.cnt[GIE_CNT_A] < c.qcap_ab ? c.cnt[GIE_CNT_A] : c.qcap_ab;
    c.cnt[GIE_CNT_SEED_A] = n0; c.cnt[GIE_CNT_SEED_B] = c.cnt[GIE_CNT_B];
    const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
    struct ent { gie_vaddr a; int g[3]; };
    auto colour = [](const int *g) { return ((g[0] >> 3) + (g[1] >> 3) + (g[2] >> 3)) & 1; };
    std::vector<ent> pend[2];
    for (int e = 0; e < n0; e++) {
        ent t; t.a = c.qa_a[e];
        if (t.a < 0) continue;
        gie_unpack_crd(c.qa[e], &t.g[0], &t.g[1], &t.g[2]);
        pend[colour(t.g)].push_back(t);
    }
    for (int h = 0; !pend[0].empty() || !pend[1].empty(); h++) {
        std::vector<ent> cur;
        cur.swap(pend[h & 1]);
        if (cur.empty()) continue;
        c.cnt[GIE_CNT_LVL_A] += 1;
        std::stable_sort(cur.begin(), cur.end(), [](const ent &p, const ent &q) { return (p.a >> 9) < (q.a >> 9); });
        std::vector<std::pair<ent, uint64_t>> xprops;               /* raises proposed into neighbouring blocks (min per voxel) */
        for (size_t b0 = 0; b0 < cur.size();) {
            const int slot = (int)(cur[b0].a >> 9);
            std::vector<ent> L;
            while (b0 < cur.size() && (cur[b0].a >> 9) == slot) L.push_back(cur[b0++]);
            while (!L.empty()) {                                    /* one level inside the block */
                c.cnt[GIE_CNT_VIS_A] += (int)L.size(); *reinterpret_cast<long long *>(&c.cnt[GIE_CNT_TOT_A]) += (long long)L.size();
                struct low { bool lowered; uint64_t coc; uint64_t pair; };
                std::vector<low> lw(L.size());
                std::vector<std::pair<ent, uint64_t>> props;
                for (size_t e = 0; e < L.size(); e++) {             /* phase 1: reads of the level-start state, proposals */
                    const gie_vaddr a = L[e].a; const int *g = L[e].g;
                    lw[e].lowered = false; lw[e].pair = GIE_NOPROP;
                    const uint64_t lcoc = c.g_coc[a] & ~GIE_COC_STALEPAIR;
                    int cd = gie_gdist(c, lcoc, g[0], g[1], g[2]);
                    if (cd > c.cutoff_sq) continue;
                    int lc[3];
                    gie_unpack_crd(lcoc, &lc[0], &lc[1], &lc[2]);
                    const uint64_t lpar = gie_pack_wr(lc[0] - c.upvt[0], lc[1] - c.upvt[1], lc[2] - c.upvt[2]);
                    for (int k = 0; k < 6; k++) {
                        const int ng[3] = { g[0] + dx[k], g[1] + dy[k], g[2] + dz[k] };
                        const int nb[3] = { ng[0] - c.pvt[0], ng[1] - c.pvt[1], ng[2] - c.pvt[2] };
                        if (gie_in_loc(c, nb[0], nb[1], nb[2]) || gie_in_whole(c, nb[0], nb[1], nb[2])) continue;
                        const gie_vaddr na = gie_gvox_hash(c, ng[0], ng[1], ng[2]);
                        if (na < 0 || c.g_type[na] == GIE_VOX_UNKNOWN) continue;
                        const uint64_t ncoc = c.g_coc[na] & ~GIE_COC_STALEPAIR;
                        int nc[3];
                        gie_unpack_crd(ncoc, &nc[0], &nc[1], &nc[2]);
                        if (gie_invalid_coc(nc[0], nc[1], nc[2]) || gie_invalid_dist(c, gie_gdist(c, ncoc, ng[0], ng[1], ng[2]))) continue;
                        if (c.g_wl[na] == -c.map_ct) continue;
                        if (nc[0] == lc[0] && nc[1] == lc[1] && nc[2] == lc[2]) continue;
                        const int nl[3] = { nc[0] - c.pvt[0], nc[1] - c.pvt[1], nc[2] - c.pvt[2] };
                        if (gie_in_loc(c, nl[0], nl[1], nl[2]) && c.glb_type[gie_lid(c, nl[0], nl[1], nl[2])] != GIE_VOX_OCCUPIED) {
                            const uint64_t key = gie_pair_make(gie_d2(lc[0], lc[1], lc[2], ng[0], ng[1], ng[2]), lpar);
                            ent t; t.a = na; t.g[0] = ng[0]; t.g[1] = ng[1]; t.g[2] = ng[2];
                            auto &into = ((na >> 9) == slot) ? props : xprops;
                            bool found = false;
                            for (auto &x : into) if (x.first.a == na) { if (key < x.second) x.second = key; found = true; break; }
                            if (!found) into.push_back(std::make_pair(t, key));
                        } else {
                            const int d = gie_d2(nc[0], nc[1], nc[2], g[0], g[1], g[2]);
                            if (cd > d) {
                                cd = d;
                                lw[e].lowered = true; lw[e].coc = ncoc;        /* the pair of an earlier in-range lowering stays (wave_core.cuh:199-221) */
                                const int nw[3] = { nc[0] - c.upvt[0], nc[1] - c.upvt[1], nc[2] - c.upvt[2] };
                                if (gie_in_wr(c, nw[0], nw[1], nw[2])) lw[e].pair = gie_pair_make(d, gie_pack_wr(nw[0], nw[1], nw[2]));
                            }
                        }
                    }
                }
                for (size_t e = 0; e < L.size(); e++) {             /* phase 2: apply */
                    if (!lw[e].lowered) continue;
                    const gie_vaddr a = L[e].a;
                    c.g_coc[a] = lw[e].coc; gie_touch(c, a); c.g_wl[a] = 1;
                    if (lw[e].pair != GIE_NOPROP) {
                        c.g_pair[a] = lw[e].pair;
                        gie_push64a(c, c.qb, c.qb_a, &c.cnt[GIE_CNT_B], c.qcap_ab, gie_pack_crd(L[e].g[0], L[e].g[1], L[e].g[2]), a);
                    }
                }
                std::vector<ent> Ln;
                for (auto &x : props) {
                    const gie_vaddr na = x.first.a;
                    int lw3[3];
                    gie_unpack_wr(gie_pair_par(x.second), &lw3[0], &lw3[1], &lw3[2]);
                    c.g_coc[na] = gie_pack_crd(lw3[0] + c.upvt[0], lw3[1] + c.upvt[1], lw3[2] + c.upvt[2]);
                    gie_touch(c, na); c.g_wl[na] = -c.map_ct; c.g_pair[na] = x.second;
                    Ln.push_back(x.first);
                }
                L.swap(Ln);
            }
        }
        for (auto &x : xprops) {                                    /* end of the round */
            const gie_vaddr na = x.first.a;
            int lw3[3];
            gie_unpack_wr(gie_pair_par(x.second), &lw3[0], &lw3[1], &lw3[2]);
            c.g_coc[na] = gie_pack_crd(lw3[0] + c.upvt[0], lw3[1] + c.upvt[1], lw3[2] + c.upvt[2]);
            gie_touch(c, na); c.g_wl[na] = -c.map_ct; c.g_pair[na] = x.second;
            pend[(h & 1) ^ 1].push_back(x.first);
        }
    }
}

/* wave B in the canonical block-round schedule (oracle/gie_oracle.c wave_b): sequential statement on the device data
 * structures.  Proposals inside a block and across block borders travel through host maps here (the device keeps the former in
 * LDS, the latter in the two proposal planes of the global map). */
static void be_wave_b(be_state *, const gie_ctx &c)
{
    const int n0 = c.cnt[GIE_CNT_B] < c.qcap_ab ? c.cnt[GIE_CNT_B] : c.qcap_ab;
    c.cnt[GIE_CNT_FRONT_B] = n0; c.cnt[GIE_CNT_SEED_C] = c.cnt[GIE_CNT_C];
    const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
    struct ent { gie_vaddr a; int g[3]; };
    std::vector<ent> cur;
    {
        std::vector<gie_vaddr> seen;
        for (int e = 0; e < n0; e++) {
            const gie_vaddr a = c.qb_a[e];
            if (a < 0 || std::find(seen.begin(), seen.end(), a) != seen.end()) continue;      /* the frontier is a set */
            seen.push_back(a);
            ent t; t.a = a; gie_unpack_crd(c.qb[e], &t.g[0], &t.g[1], &t.g[2]);
            cur.push_back(t);
        }
    }
    std::vector<int> inl;                                           /* voxels inside the volume that received a proposal */
    auto pdist = [](uint64_t p) { return gie_pair_dist(p); };
    while (!cur.empty()) {
        c.cnt[GIE_CNT_LVL_B] += 1;
        std::stable_sort(cur.begin(), cur.end(), [](const ent &p, const ent &q) { return (p.a >> 9) < (q.a >> 9); });
        std::vector<std::pair<ent, uint64_t>> xprops;               /* proposals across block borders of the running round (min per voxel) */
        auto xfind = [&](gie_vaddr a) { for (auto &x : xprops) if (x.first.a == a) return &x; return (std::pair<ent, uint64_t> *)nullptr; };
        for (size_t b0 = 0; b0 < cur.size();) {
            const int slot = (int)(cur[b0].a >> 9);
            std::vector<ent> L;
            while (b0 < cur.size() && (cur[b0].a >> 9) == slot) L.push_back(cur[b0++]);
            while (!L.empty()) {                                    /* one level inside the block */
                c.cnt[GIE_CNT_VIS_B] += (int)L.size(); *reinterpret_cast<long long *>(&c.cnt[GIE_CNT_TOT_B]) += (long long)L.size();
                struct snap { bool active; uint64_t par; int cc[3]; };
                std::vector<snap> sn(L.size());
                for (size_t e = 0; e < L.size(); e++) {
                    const gie_vaddr a = L[e].a;
                    sn[e].active = false;
                    if (gie_gdist(c, c.g_coc[a], L[e].g[0], L[e].g[1], L[e].g[2]) > c.cutoff_sq) continue;     /* (the distance stored BEFORE the commit) */
                    const uint64_t pr = c.g_pair[a] & ~GIE_PAIR_NEW;
                    int cw[3];
                    gie_unpack_wr(gie_pair_par(pr), &cw[0], &cw[1], &cw[2]);
                    sn[e].cc[0] = cw[0] + c.upvt[0]; sn[e].cc[1] = cw[1] + c.upvt[1]; sn[e].cc[2] = cw[2] + c.upvt[2];
                    c.g_coc[a] = gie_pack_crd(sn[e].cc[0], sn[e].cc[1], sn[e].cc[2]);
                    gie_touch(c, a);
                    sn[e].active = true; sn[e].par = gie_pair_par(pr);
                }
                std::vector<std::pair<ent, uint64_t>> props;
                for (size_t e = 0; e < L.size(); e++) {
                    if (!sn[e].active) continue;
                    const int *g = L[e].g;
                    for (int k = 0; k < 6; k++) {
                        const int ng[3] = { g[0] + dx[k], g[1] + dy[k], g[2] + dz[k] };
                        const int nb[3] = { ng[0] - c.pvt[0], ng[1] - c.pvt[1], ng[2] - c.pvt[2] };
                        const int cand = gie_d2(sn[e].cc[0], sn[e].cc[1], sn[e].cc[2], ng[0], ng[1], ng[2]);
                        if (!gie_in_loc(c, nb[0], nb[1], nb[2])) {
                            if (gie_in_whole(c, nb[0], nb[1], nb[2])) continue;
                            const gie_vaddr na = gie_gvox_hash(c, ng[0], ng[1], ng[2]);
                            if (na < 0 || c.g_type[na] == GIE_VOX_UNKNOWN) continue;
                            int nc[3];
                            gie_unpack_crd(c.g_coc[na], &nc[0], &nc[1], &nc[2]);
                            if (gie_invalid_coc(nc[0], nc[1], nc[2]) || cand >= c.empty_value) continue;
                            const uint64_t key = gie_pair_make(cand, sn[e].par);
                            ent t; t.a = na; t.g[0] = ng[0]; t.g[1] = ng[1]; t.g[2] = ng[2];
                            if ((na >> 9) == slot) {
                                bool found = false;
                                for (auto &x : props) if (x.first.a == na) { if (key < x.second) x.second = key; found = true; break; }
                                if (!found) props.push_back(std::make_pair(t, key));
                            } else {
                                auto *x = xfind(na);
                                if (x) { if (key < x->second) x->second = key; } else xprops.push_back(std::make_pair(t, key));
                            }
                        } else {
                            const int cl3[3] = { sn[e].cc[0] - c.pvt[0], sn[e].cc[1] - c.pvt[1], sn[e].cc[2] - c.pvt[2] };
                            if (gie_in_whole(c, cl3[0], cl3[1], cl3[2]) && !gie_in_loc(c, cl3[0], cl3[1], cl3[2])) continue;   /* (tiling: gie_frontier_outside) */
                            const int nid = gie_lid(c, nb[0], nb[1], nb[2]);
                            const int ref = (c.glb_type[nid] != GIE_VOX_UNKNOWN) ? pdist(gie_pair_get_id(c, nid)) : gie_batch_dist_direct(c, nb[0], nb[1], nb[2]);
                            if (ref > cand) {
                                uint64_t *lp = &c.lprop[gie_bdr_index(c, nb[0], nb[1], nb[2])];
                                if (*lp == GIE_NOPROP) inl.push_back(nid);
                                const uint64_t key = gie_pair_make(cand, sn[e].par);
                                if (key < *lp) *lp = key;
                            }
                        }
                    }
                }
                std::vector<ent> Ln;
                for (auto &x : props)
                    if (pdist(c.g_pair[x.first.a]) > pdist(x.second)) { c.g_pair[x.first.a] = x.second; Ln.push_back(x.first); }
                L.swap(Ln);
            }
        }
        std::vector<ent> next;
        for (auto &x : xprops)
            if (pdist(c.g_pair[x.first.a]) > pdist(x.second)) { c.g_pair[x.first.a] = x.second; next.push_back(x.first); }
        cur.swap(next);
    }
    /* the stores into the volume: the smallest proposal of the whole wave, stored unconditionally (wave_core.cuh:336-346) */
    for (int nid : inl) {
        const int x = nid % c.X, y = (nid / c.X) % c.Y, z = nid / (c.X * c.Y);
        uint64_t *lp = &c.lprop[gie_bdr_index(c, x, y, z)];
        c.cand[1][nid] = *lp; *lp = GIE_NOPROP;
        const uint32_t w = c.wl[nid];
        if (!(w == GIE_WL_SEED(c) || w == GIE_WL_PUSHED(c))) { c.wl[nid] = GIE_WL_PUSHED(c); gie_push32(c, c.qc[0], &c.cnt[GIE_CNT_C], c.qcap_c, nid); }
    }
}
/* wave C in the canonical tile-round schedule: sequential statement on the device data structures — the seeds are
 * the voxels of qc[0] with their pairs in cand[1]; proposals across tile borders travel through the candidate planes (parity of
 * the round) like on the device, proposals inside a tile through a scratch plane (the device keeps those in LDS) */
/* "lazy pairs": whoever writes a pair of a flagged tile first puts the tile's 512 pairs into the plane (gie_wave_c_tile's write-back) */
static void emu_pair_store(const gie_ctx &c, int id, uint64_t pr)
{
    const int x = id % c.X, y = (id / c.X) % c.Y, z = id / (c.X * c.Y), t = gie_tile_index(c, x, y, z);
    if (c.tlazy[t]) { for (int l = 0; l < 64; l++) gie_pair_materialise_column(c, t, l); c.tlazy[t] = 0; }
    c.pair[id] = pr;
}
static void be_wave_c(be_state *, const gie_ctx &c, int record_seeds, int)
{
    const int n0 = c.cnt[GIE_CNT_C] < c.qcap_c ? c.cnt[GIE_CNT_C] : c.qcap_c;
    c.cnt[GIE_CNT_FRONT_C] = n0;
    if (record_seeds) { c.cnt[GIE_CNT_SEED_C] = n0; c.cnt[GIE_CNT_SEED_A] = c.cnt[GIE_CNT_A]; c.cnt[GIE_CNT_SEED_B] = c.cnt[GIE_CNT_B]; }
    const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
    auto tile = [&](int id) { const int x = id % c.X, y = (id / c.X) % c.Y, z = id / (c.X * c.Y); return gie_tile_index(c, x, y, z); };
    auto set_pair = [&](int id, uint64_t pr) {
        if (c.glb_type[id] == GIE_VOX_UNKNOWN) gie_edt_unknown_touch(c, id, id % c.X, (id / c.X) % c.Y, id / (c.X * c.Y), gie_pair_get_id(c, id));
        emu_pair_store(c, id, pr);
        if (c.fused) {
            const int x = id % c.X, y = (id / c.X) % c.Y, z = id / (c.X * c.Y);
            gie_commit_merged(c, id, c.glb_type[id], c.blk_tab[gie_tab_index(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2])], x, y, z, pr);
        }
    };
    std::vector<uint64_t> lp((size_t)c.N, (uint64_t)GIE_NOPROP);
    std::vector<int> cur;
    for (int e = 0; e < n0; e++) {                                  /* round 0: the seeds' pairs are assignments */
        const int id = c.qc[0][e];
        const uint64_t pr = c.cand[1][id];
        c.cand[1][id] = GIE_NOPROP;
        set_pair(id, pr & ~GIE_PAIR_NEW);
        cur.push_back(id);
    }
    int round = 0;
    while (!cur.empty()) {
        c.cnt[GIE_CNT_LVL_C] += 1;
        uint64_t *xplane = c.cand[round & 1];                       /* what crosses a tile border in this round */
        std::vector<int> xtouched;
        std::stable_sort(cur.begin(), cur.end(), [&](int a, int b) { return tile(a) < tile(b); });
        for (size_t b = 0; b < cur.size();) {
            const int T = tile(cur[b]);
            std::vector<int> L;
            while (b < cur.size() && tile(cur[b]) == T) L.push_back(cur[b++]);
            while (!L.empty()) {
                c.cnt[GIE_CNT_VIS_C] += (int)L.size(); *reinterpret_cast<long long *>(&c.cnt[GIE_CNT_TOT_C]) += (long long)L.size();
                std::vector<uint64_t> par(L.size());
                for (size_t e = 0; e < L.size(); e++) par[e] = gie_pair_par(gie_pair_get_id(c, L[e]));
                std::vector<int> touched, Ln;
                for (size_t e = 0; e < L.size(); e++) {
                    const int id = L[e], x = id % c.X, y = (id / c.X) % c.Y, z = id / (c.X * c.Y);
                    int cw[3];
                    gie_unpack_wr(par[e], &cw[0], &cw[1], &cw[2]);
                    const int cl[3] = { cw[0] + c.upvt[0] - c.pvt[0], cw[1] + c.upvt[1] - c.pvt[1], cw[2] + c.upvt[2] - c.pvt[2] };
                    for (int k = 0; k < 6; k++) {
                        const int nx = x + dx[k], ny = y + dy[k], nz = z + dz[k];
                        if (!gie_in_loc(c, nx, ny, nz)) continue;
                        const int nid = gie_lid(c, nx, ny, nz);
                        const int cand = gie_d2(cl[0], cl[1], cl[2], nx, ny, nz);
                        if (cand >= c.empty_value) continue;
                        const uint64_t key = gie_pair_make(cand, par[e]);
                        if (gie_tile_index(c, nx, ny, nz) == T) { if (key < lp[(size_t)nid]) lp[(size_t)nid] = key; touched.push_back(nid); }
                        else { if (key < xplane[nid]) xplane[nid] = key; xtouched.push_back(nid); }
                    }
                }
                for (int nid : touched) {
                    if (lp[(size_t)nid] == GIE_NOPROP) continue;
                    if (gie_pair_dist(gie_pair_get_id(c, nid)) > gie_pair_dist(lp[(size_t)nid])) { set_pair(nid, lp[(size_t)nid]); Ln.push_back(nid); }
                    lp[(size_t)nid] = GIE_NOPROP;
                }
                L.swap(Ln);
            }
        }
        std::vector<int> next;
        for (int nid : xtouched) {
            if (xplane[nid] == GIE_NOPROP) continue;
            if (gie_pair_dist(gie_pair_get_id(c, nid)) > gie_pair_dist(xplane[nid])) { set_pair(nid, xplane[nid]); next.push_back(nid); }
            xplane[nid] = GIE_NOPROP;
        }
        cur.swap(next);
        round++;
    }
}
/* which statement of wave C the emulation runs (environment at load, or gie_emu_wave_c_model at any time) */
static int g_emu_wave_c_device = (getenv("GIE_EMU_WAVE_C") && !strcmp(getenv("GIE_EMU_WAVE_C"), "device")) ? 1 : (getenv("GIE_EMU_WAVE_C") && !strcmp(getenv("GIE_EMU_WAVE_C"), "device_live_halo")) ? 2 : 0;
static int g_emu_wave_c_r0filter = getenv("GIE_EMU_WAVE_C_R0FILTER") ? atoi(getenv("GIE_EMU_WAVE_C_R0FILTER")) : 0;
extern "C" void gie_emu_wave_c_model(int device, int r0filter) { g_emu_wave_c_device = device; g_emu_wave_c_r0filter = r0filter; }
/* Wave C the way the DEVICE schedules it (gie_wave_c_tile / gie_wave_c_run in gie_kernels.hip.h), stated sequentially: a second model
 * beside the canonical one above, selected with GIE_EMU_WAVE_C=device or gie_emu_wave_c_model(1, 0).  What it keeps of the kernel and the canonical statement does
 * not have: the seeds are assigned INSIDE round 0 (first sub-level of their tile), a tile's halo is what the pair plane held when the
 * round started, an expansion drops a proposal that does not beat the pair it sees for the neighbour (tile: live, halo: the
 * snapshot) — except across a tile border in round 0, where the neighbour may be a seed about to be assigned above its stale pair
 * (the fault the round-4 fuzz found on the GPU; GIE_EMU_WAVE_C_R0FILTER=1 / gie_emu_wave_c_model(1, 1) puts the old filter back: tests/test_host_logic.py shows
 * that this model then differs from the oracle on that scenario) —, proposals across borders wait in the candidate plane of the
 * round's parity and are merged by the neighbour's tile in the next round.  Tiles of a round are taken one after the other here; on
 * the device they run side by side and a halo read may also see a neighbour's write-back of the same round — a lower value, which
 * only drops proposals that could not have won either. */
static void be_wave_c_device(be_state *, const gie_ctx &c, int record_seeds, int)
{
    const int r0filter = g_emu_wave_c_r0filter;
    const int n0 = c.cnt[GIE_CNT_C] < c.qcap_c ? c.cnt[GIE_CNT_C] : c.qcap_c;
    c.cnt[GIE_CNT_FRONT_C] = n0;
    if (record_seeds) { c.cnt[GIE_CNT_SEED_C] = n0; c.cnt[GIE_CNT_SEED_A] = c.cnt[GIE_CNT_A]; c.cnt[GIE_CNT_SEED_B] = c.cnt[GIE_CNT_B]; }
    if (n0 == 0) return;
    const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
    const int ntile = c.tfd[0] * c.tfd[1] * c.tfd[2];
    std::vector<char> flag[2] = { std::vector<char>((size_t)ntile, 0), std::vector<char>((size_t)ntile, 0) };
    std::vector<int> list[2];
    for (int e = 0; e < n0; e++) {
        const int id = c.qc[0][e];
        const int t = gie_tile_index(c, id % c.X, (id / c.X) % c.Y, id / (c.X * c.Y));
        if (!flag[0][(size_t)t]) { flag[0][(size_t)t] = 1; list[0].push_back(t); }
    }
    int round = 0;
    while (!list[round & 1].empty()) {
        uint64_t *rd = c.cand[(round + 1) & 1], *wr = c.cand[round & 1];
        std::vector<uint64_t> snap((size_t)c.N);       /* the halos of the round */
        for (int i = 0; i < c.N; i++) snap[(size_t)i] = gie_pair_get_id(c, i);
        list[(round + 1) & 1].clear();
        long long vis = 0;
        for (const int t : list[round & 1]) {
            flag[round & 1][(size_t)t] = 0;
            const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
            const int x0 = tx * 8, y0 = ty * 8, z0 = tz * 8;
            uint64_t pair[512], prop[512], loaded[512];
            bool in[512];
            std::vector<int> pend;
            for (int v = 0; v < 512; v++) {
                const int x = x0 + (v & 7), y = y0 + ((v >> 3) & 7), z = z0 + (v >> 6);
                in[v] = gie_in_loc(c, x, y, z);
                pair[v] = 0; prop[v] = GIE_NOPROP;                   /* a position outside the volume: distance 0, never improved */
                if (in[v]) {
                    const int id = gie_lid(c, x, y, z);
                    pair[v] = gie_pair_get_id(c, id); prop[v] = rd[id];
                    if (prop[v] != GIE_NOPROP) { rd[id] = GIE_NOPROP; pend.push_back(v); }
                }
                loaded[v] = pair[v];
            }
            unsigned xmask = 0;
            for (int sub = 0;; sub++) {
                std::vector<int> ent, pend_next;
                for (const int v : pend) {
                    const uint64_t cd = prop[v];
                    prop[v] = GIE_NOPROP;
                    if ((round == 0 && sub == 0) || gie_pair_dist(cd) < gie_pair_dist(pair[v])) { pair[v] = cd & ~GIE_PAIR_NEW; vis++; ent.push_back(v); }
                }
                if (ent.empty()) break;
                for (const int v : ent) {
                    const int ex = v & 7, ey = (v >> 3) & 7, ez = v >> 6;
                    const uint64_t par = gie_pair_par(pair[v]);
                    int cw[3];
                    gie_unpack_wr(par, &cw[0], &cw[1], &cw[2]);
                    const int cl[3] = { cw[0] + c.upvt[0] - c.pvt[0], cw[1] + c.upvt[1] - c.pvt[1], cw[2] + c.upvt[2] - c.pvt[2] };
                    for (int k = 0; k < 6; k++) {
                        const int ux = ex + dx[k], uy = ey + dy[k], uz = ez + dz[k];
                        const int nx = x0 + ux, ny = y0 + uy, nz = z0 + uz;
                        if (!gie_in_loc(c, nx, ny, nz)) continue;
                        const int d = gie_d2(cl[0], cl[1], cl[2], nx, ny, nz);
                        if (d >= c.empty_value) continue;
                        const bool inside = (unsigned)ux < 8u && (unsigned)uy < 8u && (unsigned)uz < 8u;
                        const int nid = gie_lid(c, nx, ny, nz);
                        const uint64_t seen = inside ? pair[ux + 8 * uy + 64 * uz] : (g_emu_wave_c_device == 2 ? gie_pair_get_id(c, nid) : snap[(size_t)nid]);   /* (2: the other end of the device's race — a halo read sees the write-backs of the tiles taken before this one in the same round) */
                        if (!(d < gie_pair_dist(seen)) && (inside || round != 0 || r0filter)) continue;
                        const uint64_t key = gie_pair_make(d, par);
                        if (inside) {
                            const int nv = ux + 8 * uy + 64 * uz;
                            if (prop[nv] == GIE_NOPROP) pend_next.push_back(nv);
                            if (key < prop[nv]) prop[nv] = key;
                        } else { if (key < wr[nid]) wr[nid] = key; xmask |= 1u << k; }
                    }
                }
                pend.swap(pend_next);
            }
            for (int v = 0; v < 512; v++) {
                if (!in[v] || pair[v] == loaded[v]) continue;
                const int x = x0 + (v & 7), y = y0 + ((v >> 3) & 7), z = z0 + (v >> 6);
                const int id = gie_lid(c, x, y, z);
                if (c.glb_type[id] == GIE_VOX_UNKNOWN) gie_edt_unknown_touch(c, id, x, y, z, loaded[v]);
                emu_pair_store(c, id, pair[v]);
                if (c.fused) gie_commit_merged(c, id, c.glb_type[id], c.blk_tab[gie_tab_index(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2])], x, y, z, pair[v]);
            }
            const int dt[6] = { -1, 1, -c.tfd[0], c.tfd[0], -c.tfd[0] * c.tfd[1], c.tfd[0] * c.tfd[1] };
            for (int k = 0; k < 6; k++) if ((xmask >> k) & 1u) {
                const int nt = t + dt[k];
                if (!flag[(round + 1) & 1][(size_t)nt]) { flag[(round + 1) & 1][(size_t)nt] = 1; list[(round + 1) & 1].push_back(nt); }
            }
        }
        if (vis > 0) { c.cnt[GIE_CNT_LVL_C] += 1; c.cnt[GIE_CNT_VIS_C] += (int)vis; *reinterpret_cast<long long *>(&c.cnt[GIE_CNT_TOT_C]) += vis; }
        round++;
    }
}
static void be_waves(be_state *b, const gie_ctx &c, int with_ab, int record_seeds, int clear_first)
{
    if (GIE_GATE_CLOSED(c)) return;
    const int device_c = g_emu_wave_c_device;
    if (with_ab) { be_wave_a(b, c); be_wave_b(b, c); }
    if (device_c) be_wave_c_device(b, c, record_seeds, clear_first);
    else be_wave_c(b, c, record_seeds, clear_first);
}

#include "../../gie-mapping_amd/csrc/gie_api.inc.h"

/* test hooks: the byte-parallel occupancy filter of the block-row fuse kernel (gie_ops.h) and the per-voxel one it must equal */
extern "C" void gie_emu_fuse_row8(int thresh, uint64_t it8, uint64_t go8, uint64_t gy8, uint64_t *no8, uint64_t *ny8) { gie_fuse_row8_labels(thresh, it8, go8, gy8, no8, ny8); }
extern "C" void gie_emu_fuse_voxel(int thresh, int label, int occ_in, int ty_in, int *occ_out, int *ty_out)
{
    gie_ctx c; memset(&c, 0, sizeof(c)); c.pntcld_mode = 0; c.occ_thresh = thresh;
    uint8_t occ = (uint8_t)occ_in; int8_t ty = (int8_t)ty_in;
    gie_fuse_logic(c, 0, (int8_t)label, 0, &occ, &ty);
    *occ_out = occ; *ty_out = ty;
}
