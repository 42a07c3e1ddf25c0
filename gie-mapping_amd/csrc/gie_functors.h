/*
 * gie_functors.h — the per-voxel / per-item operations as functors, shared by the HIP kernels
 * (k_vox / k_lin in gie_kernels.hip.h) and by the test-only sequential emulation (tests/emu).
 */
#ifndef GIE_FUNCTORS_H
#define GIE_FUNCTORS_H

#include "gie_ops.h"


/* skip(c, id, x, y, z): cheap test (at most one small load, issued for a whole z-column up
 * front by k_voxz) that is true only when operator() would do nothing for the voxel. */
struct op_classify_depth { static constexpr bool rolled = false;
    GIE_DEVM bool tile_skip(const gie_ctx &, int, int, int) const { return false; } const float *img; gie_cam_param p;
    GIE_DEVM bool skip(const gie_ctx &, int, int, int, int) const { return false; }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const {
        const int t = gie_classify_depth(c, img, p, x, y, z);
        if (t != GIE_VOX_UNKNOWN) { c.inst_type[gie_lid(c, x, y, z)] = (int8_t)t; gie_mark_block_needed(c, x, y, z); } } };
struct op_classify_multiscan { static constexpr bool rolled = false;
    const float *img; gie_multiscan_param p; float tan_lo, tan_hi; int fov_test;
    /* the whole z-column (a straight segment in the sensor frame) lies above or below the field of
     * view when both its end voxels do: lz - hor*tan_hi is concave along a segment for tan_hi >= 0
     * (hor = distance from the sensor axis is convex), lz - hor*tan_lo convex for tan_lo <= 0, so
     * the end points bound the interior.  The per-voxel test's 1e-4 widening covers the rounding. */
    GIE_DEVM bool tile_skip(const gie_ctx &c, int x, int y, int z0) const {
        if (!fov_test || c.for_motion_planner || !(tan_hi >= 0.f && tan_lo <= 0.f)) return false;
        const float w = c.voxel_width;
        const int z1 = (z0 + 7 < c.Z) ? z0 + 7 : c.Z - 1;
        float ax, ay, az, bx, by, bz;
        gie_se3_apply(c.G2L, (float)(x + c.pvt[0]) * w, (float)(y + c.pvt[1]) * w, (float)(z0 + c.pvt[2]) * w, &ax, &ay, &az);
        gie_se3_apply(c.G2L, (float)(x + c.pvt[0]) * w, (float)(y + c.pvt[1]) * w, (float)(z1 + c.pvt[2]) * w, &bx, &by, &bz);
        const float ha = sqrtf(ay * ay + ax * ax), hb = sqrtf(by * by + bx * bx);
        return (az > ha * tan_hi && bz > hb * tan_hi) || (az < ha * tan_lo && bz < hb * tan_lo);
    }
    /* conservative field-of-view test: elevation surely outside [phi_min - inc/2, phi_max + inc/2]
     * (bounds widened by 1e-3 rad on the host, far above the 2-ulp error of gie_atan2f) */
    GIE_DEVM bool skip(const gie_ctx &c, int, int x, int y, int z) const {
        if (!fov_test || gie_robot_sphere(c, x, y, z)) return false;
        const float w = c.voxel_width;
        float lx, ly, lz;
        gie_se3_apply(c.G2L, (float)(x + c.pvt[0]) * w, (float)(y + c.pvt[1]) * w, (float)(z + c.pvt[2]) * w, &lx, &ly, &lz);
        const float hor = sqrtf(ly * ly + lx * lx);
        return lz > hor * tan_hi || lz < hor * tan_lo;
    }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const {
        const int t = gie_classify_multiscan(c, img, p, x, y, z);
        if (t != GIE_VOX_UNKNOWN) { c.inst_type[gie_lid(c, x, y, z)] = (int8_t)t; gie_mark_block_needed(c, x, y, z); } } };
struct op_classify_scan2d { static constexpr bool rolled = false;
    GIE_DEVM bool tile_skip(const gie_ctx &, int, int, int) const { return false; } const float *img; gie_scan_param p;
    GIE_DEVM bool skip(const gie_ctx &, int, int, int, int) const { return false; }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const {
        const int t = gie_classify_scan2d(c, img, p, x, y, z);
        if (t != GIE_VOX_UNKNOWN) { c.inst_type[gie_lid(c, x, y, z)] = (int8_t)t; gie_mark_block_needed(c, x, y, z); } } };
struct op_classify_labels { static constexpr bool rolled = false;
    GIE_DEVM bool tile_skip(const gie_ctx &, int, int, int) const { return false; } const int8_t *labels;
    GIE_DEVM bool skip(const gie_ctx &, int, int, int, int) const { return false; }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const {
        const int t = gie_classify_label(c, labels, x, y, z);
        c.inst_type[gie_lid(c, x, y, z)] = (int8_t)t;              /* the scan IS the label plane: unknown is written too */
        if (t != GIE_VOX_UNKNOWN) gie_mark_block_needed(c, x, y, z); } };
struct op_raycast_finalize { static constexpr bool rolled = false;
    /* no ray went through the tile (the robot sphere of for_motion_planner is written without rays) */
    GIE_DEVM bool tile_skip(const gie_ctx &c, int x, int y, int z0) const { return !c.for_motion_planner && !c.tray[gie_tile_index(c, x, y, z0)]; }
    GIE_DEVM bool skip(const gie_ctx &c, int id, int, int, int) const { return c.ray_count[id] == 0 && !c.for_motion_planner; }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const { gie_raycast_finalize(c, x, y, z); } };
/* staged ops (k_voxa<F, true>): st = per-voxel registers, load1/load2/finish as in gie_ops.h */
struct op_fuse { static constexpr bool rolled = false;
    typedef gie_fuse_st st;
    GIE_DEVM bool tile_skip(const gie_ctx &c, int x, int y, int z0) const { return gie_fuse_column_idle(c, x, y, z0) != 0; }
    /* per z-column (8 voxels of one 8x8x8 tile): which of them ended up known */
    GIE_DEVM void column(const gie_ctx &c, int x, int y, int z0, unsigned known, unsigned valid) const { gie_fuse_column_summary(c, x, y, z0, known, valid); }
    GIE_DEVM bool skip(const gie_ctx &, int, int, int, int) const { return false; }
    GIE_DEVM void load1(const gie_ctx &c, int id, int x, int y, int z, st &s) const { gie_fuse_load1(c, id, x, y, z, s); }
    GIE_DEVM void load2(const gie_ctx &c, int, int, int, int, st &s) const { gie_fuse_load2(c, s); }
    GIE_DEVM int finish(const gie_ctx &c, int id, int x, int y, int z, const st &s) const { return gie_fuse_finish(c, id, x, y, z, s); }
    GIE_DEVM int operator()(const gie_ctx &c, int x, int y, int z) const { return gie_fuse_voxel(c, x, y, z); } };
struct op_mark { static constexpr bool rolled = false;
    typedef gie_mark_st st;
    GIE_DEVM bool tile_skip(const gie_ctx &c, int x, int y, int z0) const { return !c.tknown[gie_tile_index(c, x, y, z0)]; }
    GIE_DEVM bool skip(const gie_ctx &c, int id, int, int, int) const { return c.glb_type[id] == GIE_VOX_UNKNOWN; }
    GIE_DEVM void load1(const gie_ctx &c, int id, int x, int y, int z, st &s) const { gie_mark_load1(c, id, x, y, z, s); }
    GIE_DEVM void load2(const gie_ctx &c, int, int, int, int, st &s) const { gie_mark_load2(c, s); }
    GIE_DEVM int finish(const gie_ctx &c, int id, int x, int y, int z, const st &s) const { gie_mark_finish(c, id, x, y, z, s); return 0; }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const { gie_mark_voxel(c, x, y, z); } };
struct op_markc { static constexpr bool rolled = false;
    typedef gie_markc_st st;
    GIE_DEVM bool tile_skip(const gie_ctx &c, int x, int y, int z0) const { return !c.tknown[gie_tile_index(c, x, y, z0)]; }
    GIE_DEVM bool skip(const gie_ctx &c, int id, int, int, int) const { return c.glb_type[id] == GIE_VOX_UNKNOWN; }
    GIE_DEVM void load1(const gie_ctx &c, int id, int x, int y, int z, st &s) const { gie_markc_load1(c, id, x, y, z, s); }
    GIE_DEVM void load2(const gie_ctx &c, int, int, int, int, st &s) const { gie_markc_load2(c, s); }
    GIE_DEVM int finish(const gie_ctx &c, int id, int x, int y, int z, const st &s) const { return gie_markc_finish(c, id, x, y, z, s); }
    /* per z-column: which voxels were committed and the largest value finish() returned (the tile's bound for the next map update) */
    GIE_DEVM void column_max(const gie_ctx &c, int x, int y, int z0, unsigned known, unsigned valid, int vmax) const { gie_markc_column(c, x, y, z0, known, valid, vmax); }
    GIE_DEVM int operator()(const gie_ctx &c, int x, int y, int z) const { return gie_markc_voxel(c, x, y, z); } };
struct op_tile_oldskip { GIE_DEVM void operator()(const gie_ctx &c, int t) const { (void)gie_tile_oldskip(c, t); } };
struct op_commit { static constexpr bool rolled = false;
    typedef gie_commit_st st;
    /* a map update whose waves were cut short by a barrier timeout commits nothing (GIE_ERR_TIMEOUT, include/gie.h): the flag of
     * THIS update (cleared with the frame), not the sticky one the host fetches — an enqueue-only pipeline that never
     * synchronises must not lose every later commit to one timeout */
    GIE_DEVM bool tile_skip(const gie_ctx &c, int x, int y, int z0) const { return c.cnt[GIE_CNT_BARFAIL] != 0 || !c.tknown[gie_tile_index(c, x, y, z0)]; }
    GIE_DEVM bool skip(const gie_ctx &c, int id, int, int, int) const { return c.glb_type[id] == GIE_VOX_UNKNOWN; }
    GIE_DEVM void load1(const gie_ctx &c, int id, int x, int y, int z, st &s) const { gie_commit_load1(c, id, x, y, z, s); }
    GIE_DEVM void load2(const gie_ctx &, int, int, int, int, st &) const {}
    GIE_DEVM int finish(const gie_ctx &c, int id, int, int, int, const st &s) const { gie_commit_finish(c, id, s); return 0; }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const { gie_commit_voxel(c, x, y, z); } };

/* obtainFrontiers with wave64 ballot compaction of the C seeds: one atomicAdd per wave */
struct op_frontier { static constexpr bool rolled = false;
    typedef gie_frontier_st st;
    /* the ballot inside finish() works on whatever lanes are active, so skipping is safe.
     * tsum == 0: nothing in or around this 8x8x8 tile can make obtainFrontiers act. */
    GIE_DEVM bool tile_skip(const gie_ctx &c, int x, int y, int z0) const { return c.tsum[gie_tile_index(c, x, y, z0)] == 0; }
    /* tsum == 2: a tile on a face of the volume with nothing else to look at — only its voxels ON the face can act */
    GIE_DEVM bool skip(const gie_ctx &c, int id, int x, int y, int z) const {
        if (c.glb_type[id] == GIE_VOX_UNKNOWN) return true;
        const bool on_face = x == 0 || y == 0 || z == 0 || x == c.X - 1 || y == c.Y - 1 || z == c.Z - 1;
        return !on_face && c.tsum[gie_tile_index(c, x, y, z)] == 2;
    }
    GIE_DEVM void load1(const gie_ctx &c, int id, int x, int y, int z, st &s) const { gie_frontier_load1(c, id, x, y, z, s); }
    GIE_DEVM void load2(const gie_ctx &, int, int, int, int, st &) const {}
    GIE_DEVM int finish(const gie_ctx &c, int id, int x, int y, int z, const st &s) const { push_seed(c, gie_frontier_finish(c, id, x, y, z, s), id); return 0; }
    GIE_DEVM void operator()(const gie_ctx &c, int x, int y, int z) const { push_seed(c, gie_frontier_voxel(c, x, y, z), gie_lid(c, x, y, z)); }
    GIE_DEVM void push_seed(const gie_ctx &c, const int push, const int id) const {
        const unsigned long long m = __ballot(push);
        if (m) {
            const int lane = __lane_id();
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = gie_aadd32(&c.cnt[GIE_CNT_C], __popcll(m));
            base = __shfl(base, leader);
            if (push) {
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < c.qcap_c) c.qc[0][slot] = id;
                else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
            }
        }
    }
};

/* tile summary for obtainFrontiers: a voxel of tile t can only act (C seed, A/B seed, FNT flip)
 * when t touches the volume faces, or t / a face-adjacent tile holds an unknown voxel or a
 * voxel whose Mark-time closest obstacle lies outside the volume */
struct op_tile_summary {
    /* tile holds an unknown voxel: fuse said so, or fuse never looked at it (then it is all-unknown) */
    GIE_DEVM static int unk(const gie_ctx &c, int t) { return c.tunk[t] | (c.tact[t] ^ 1); }
    /* 1: every voxel of the tile is looked at; 2: a tile on a face of the volume whose surroundings hold neither an unknown
     * voxel nor a closest obstacle outside the volume — only the voxels on the face itself can act; 0: nothing to look at */
    GIE_DEVM static uint8_t value(const gie_ctx &c, int t) {
        if (!c.tknown[t]) return 0;
        const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
        const bool face = tx == 0 || ty == 0 || tz == 0 || tx == c.tfd[0] - 1 || ty == c.tfd[1] - 1 || tz == c.tfd[2] - 1;
        const int sx = 1, sy = c.tfd[0], sz = c.tfd[0] * c.tfd[1];
        int w = unk(c, t) | c.tflag[t];
        if (tx > 0) w |= unk(c, t - sx) | c.tflag[t - sx];
        if (tx < c.tfd[0] - 1) w |= unk(c, t + sx) | c.tflag[t + sx];
        if (ty > 0) w |= unk(c, t - sy) | c.tflag[t - sy];
        if (ty < c.tfd[1] - 1) w |= unk(c, t + sy) | c.tflag[t + sy];
        if (tz > 0) w |= unk(c, t - sz) | c.tflag[t - sz];
        if (tz < c.tfd[2] - 1) w |= unk(c, t + sz) | c.tflag[t + sz];
        return w ? 1 : (face ? 2 : 0);
    }
    /* called for the tiles on tl_known (t = -1: a padding lane of the last wave) */
    GIE_DEVM void operator()(const gie_ctx &c, int t) const {
        const bool real = t >= 0;
        if (!real) t = 0;
        const uint8_t v = real ? value(c, t) : 0;
        if (real) c.tsum[t] = v;
        /* the tiles whose every voxel is looked at, as a list (order is irrelevant): k_frontier_tiles takes a tile per wave from it
         * — a list of its own, so that consecutive entries cost the same (with the plain face tiles in between, the few waves
         * whose stride met the tx = 0 tiles did all the work).  The voxels on the faces go by patches of the faces (tsum != 0). */
        const int slot = gie_wg_reserve(&c.cnt[GIE_CNT_TL_FRONT], v == 1);
        if (slot >= 0) c.tl_front[slot] = t | (c.tlazy[t] ? (int32_t)0x80000000 : 0);      /* (bit 31: the tile is lazy — the entry's reader need not ask the flag before it knows which plane to load) */
    } };
struct op_halo_export { int face; gie_halo_voxel *out; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_halo_export_voxel(c, face, i, out); } };
struct op_halo_need { int face; const gie_halo_voxel *in; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_halo_need_voxel(c, face, i, in); } };
struct op_halo_import { int face; const gie_halo_voxel *in; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_halo_import_voxel(c, face, i, in); } };
struct op_halo_export_sparse { int face; gie_halo_entry *out; int32_t *count; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_halo_export_sparse_voxel(c, face, i, out, count); } };
struct op_halo_need_sparse { int face, nface; const gie_halo_entry *in; const int32_t *count; GIE_DEVM void operator()(const gie_ctx &c, int j) const { gie_halo_need_entry(c, face, j, in, count, nface); } };
struct op_halo_import_sparse { int face, nface; const gie_halo_entry *in; const int32_t *count; GIE_DEVM void operator()(const gie_ctx &c, int j) const { gie_halo_import_entry(c, face, j, in, count, nface); } };
/* the same three operations for several faces in one launch: item i belongs to the face whose
 * [off[f], off[f+1]) range holds it (faces without a buffer have an empty range) */
struct gie_face_set { int off[7]; GIE_DEVM int face_of(int i) const { int f = 0; while (f < 5 && i >= off[f + 1]) f++; return f; } };
struct op_halo_export_all { gie_face_set fs; gie_halo_voxel *out[6];
    GIE_DEVM void operator()(const gie_ctx &c, int i) const { const int f = fs.face_of(i); gie_halo_export_voxel(c, f, i - fs.off[f], out[f]); } };
struct op_halo_need_all { gie_face_set fs; const gie_halo_voxel *in[6];
    GIE_DEVM void operator()(const gie_ctx &c, int i) const { const int f = fs.face_of(i); gie_halo_need_voxel(c, f, i - fs.off[f], in[f]); } };
struct op_halo_import_all { gie_face_set fs; const gie_halo_voxel *in[6];
    GIE_DEVM void operator()(const gie_ctx &c, int i) const { const int f = fs.face_of(i); gie_halo_import_voxel(c, f, i - fs.off[f], in[f]); } };
struct op_refine { GIE_DEVM void operator()(const gie_ctx &c, int j) const {
        const int id = gie_refine_entry(c, j);
        if (id >= 0 && gie_refine_voxel(c, id)) gie_push32(c, c.qc[0], &c.cnt[GIE_CNT_C], c.qcap_c, id);
    } };
struct op_register_point { const float *xyz; float *g; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_register_point(c, xyz, g, i); } };
struct op_query { const int32_t *xyz; gie_voxel *out; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_query_voxel(c, xyz, i, out); } };
/* the fuse tile list (thread per tile; device: one atomic per wave) */
/* (called for every thread of a workgroup: t = -1 past the end) */
struct op_fuse_list { GIE_DEVM void operator()(const gie_ctx &c, int t) const {
        const int v = t >= 0 ? gie_fuse_tile_listed(c, t) : 0;
        const int slot = gie_wg_reserve(&c.cnt[GIE_CNT_TL_FUSE], v != 0);
        if (slot >= 0) c.tl_front[slot] = t;
    } };
struct op_pair_flush { gie_flush_boxes b; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_pair_flush_voxel(c, b, i); } };
/* test hook (gie_debug_nbr_check): item i = (slot, direction); counts the rows of the neighbour table that do not say what the hash says —
 * a row naming a slot whose key is not the neighbour's counts as "no neighbour" (the readers' rule, gie_wave_a_block) */
struct op_nbr_check { int32_t *bad; GIE_DEVM void operator()(const gie_ctx &c, int i) const {
        const int slot = i / 6, k = i - 6 * slot;
        if (slot >= c.pool_count[0] || c.g_key[slot] == GIE_KEY_EMPTY) return;
        int b[3];
        gie_unpack_crd(c.g_key[slot], &b[0], &b[1], &b[2]);
        b[k >> 1] += (k & 1) ? 1 : -1;
        const int row = c.g_nbr[8 * (size_t)slot + k];
        const int named = (row >= 0 && row < c.max_blocks && c.g_key[row] == gie_pack_crd(b[0], b[1], b[2])) ? row : -1;
        if (named != gie_hash_find(c, b[0], b[1], b[2])) {
            gie_aadd32(bad, 1);
        }
    } };
struct op_evict { GIE_DEVM void operator()(const gie_ctx &c, int slot) const { if (slot < c.pool_count[0]) gie_evict_slot(c, slot); } };
struct op_rehash { GIE_DEVM void operator()(const gie_ctx &c, int slot) const { if (slot < c.pool_count[0]) gie_rehash_slot(c, slot); } };
struct op_stream_list { const int32_t *rank; int32_t *list; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_stream_list(c, rank, list, i); } };
struct op_stream_gather { const int32_t *list; int first; int32_t *keys; gie_voxel *out;
    GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_stream_gather(c, list, first, keys, out, i); } };
struct op_stream_clear { const int32_t *list; int first; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_stream_clear(c, list, first, i); } };
struct op_export_pair { int32_t *d; int32_t *coc; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_export_pair(c, i, d, coc); } };
struct op_export_bcoc { int32_t *d; int32_t *coc; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_export_bcoc(c, i, d, coc); } };
/* `bool o = char` (local_batch.h:19-24,389): 1 for FREE / OCCUPIED / FNT, 0 for UNKNOWN -- not the raw type */
struct op_export_edt { float *out; GIE_DEVM void operator()(const gie_ctx &c, int i) const { out[i] = gie_edt_value(c, i); } };
struct op_costmap { gie_seendist *out; GIE_DEVM void operator()(const gie_ctx &c, int i) const {
        gie_seendist s; s.d = gie_edt_value(c, i); s.s = 0; s.o = (uint8_t)(c.glb_type[i] != 0); s.pad[0] = s.pad[1] = 0; out[i] = s; } };



#endif /* GIE_FUNCTORS_H */
