/*
 * gie_emu_ops.h — TEST-ONLY pieces of the sequential emulation (tests/emu/gie_emu.cpp): the one-thread-per-ray
 * walk and the scan-based block allocation.  They restate stages the HIP product
 * runs with different kernels (k_free_rays, k_cell_alloc, k_waves) and live here, not in the product headers.
 */
#ifndef GIE_EMU_OPS_H
#define GIE_EMU_OPS_H

#include "../../gie-mapping_amd/csrc/gie_functors.h"

/* clearRayLoc, pntcld_raycast.cu:9-18 */
GIE_DEV int gie_clear_ray(const gie_ctx &c, int lx, int ly, int lz)
{
    if (!gie_in_loc(c, lx, ly, lz)) return 1;
    const int id = gie_lid(c, lx, ly, lz);
    if (c.inst_type[id] != GIE_VOX_OCCUPIED) { gie_ray_marks lt = { -1, -1 }; gie_ray_touch(c, lx, ly, lz, &lt); gie_aadd32(&c.ray_count[id], -1); return 1; }
    return 0;
}

/* sequential walk of one ray (one thread per ray) */
GIE_DEV void gie_free_ray(const gie_ctx &c, const float *g, int i)
{
    gie_dda d;
    int s0[3];
    gie_ray_marks last_tile = { -1, -1 };
    if (!gie_point_ok(g[3 * i], g[3 * i + 1], g[3 * i + 2])) return;
    const int walk = gie_dda_init(c, g, i, d, s0);
    {   /* clearRayLoc on the sensor's own cell */
        const int id0 = gie_in_loc(c, s0[0], s0[1], s0[2]) ? gie_lid(c, s0[0], s0[1], s0[2]) : -1;
        if (id0 >= 0) gie_ray_touch(c, s0[0], s0[1], s0[2], &last_tile);
        gie_wave_add(c, (id0 >= 0 && c.inst_type[id0] != GIE_VOX_OCCUPIED) ? id0 : -1, -1);
    }
    if (!walk) return;
    /* The traversal itself (ray_cast.h:102-142) is a dependent chain "step → read the cell's
     * type → stop or decrement"; the cells do not depend on what is read, so GIE_RAY_BATCH steps
     * are generated ahead, their types are fetched together, and the effects are then applied in
     * the reference's order (a speculative cell beyond the stopping point is simply dropped). */
    for (;;) {
        int ids[GIE_RAY_BATCH];            /* local voxel id, -1 = outside the volume */
        int stop_after[GIE_RAY_BATCH];
        int loc[GIE_RAY_BATCH][3];
        GIE_UNROLL_BATCH
        for (int j = 0; j < GIE_RAY_BATCH; j++) {
            stop_after[j] = gie_dda_step(d);
            const int lx = d.cur[0] - c.pvt[0], ly = d.cur[1] - c.pvt[1], lz = d.cur[2] - c.pvt[2];
            ids[j] = gie_in_loc(c, lx, ly, lz) ? gie_lid(c, lx, ly, lz) : -1;
            loc[j][0] = lx; loc[j][1] = ly; loc[j][2] = lz;
        }
        int8_t ty[GIE_RAY_BATCH];
        GIE_UNROLL_BATCH
        for (int j = 0; j < GIE_RAY_BATCH; j++) ty[j] = ids[j] >= 0 ? c.inst_type[ids[j]] : (int8_t)GIE_VOX_UNKNOWN;
        GIE_UNROLL_BATCH
        for (int j = 0; j < GIE_RAY_BATCH; j++) {
            if (ty[j] == GIE_VOX_OCCUPIED) return;                 /* clearRayLoc returned false */
            if (ids[j] >= 0) gie_ray_touch(c, loc[j][0], loc[j][1], loc[j][2], &last_tile);   /* only cells that are really cleared */
            gie_wave_add(c, ids[j], -1);
            if (stop_after[j]) return;
        }
    }
}

/* RequiresAllocation (alloc_helper.cuh:13-21): table cell needs a block that does not exist */
GIE_DEV int gie_cell_needs_new(const gie_ctx &c, int cell)
{
    if (!c.blk_need[cell]) return 0;
    const int bx = cell % c.tdim[0], by = (cell / c.tdim[0]) % c.tdim[1], bz = cell / (c.tdim[0] * c.tdim[1]);
    return gie_hash_find(c, bx + c.tb0[0], by + c.tb0[1], bz + c.tb0[2]) < 0;
}

/* slot of the r-th new block of a map update: from the free list while it lasts, then from the bump allocator
 * (the counters move once, after the whole batch: be_block_init) */
GIE_DEV int gie_emu_slot(const gie_ctx &c, int r)
{ const int nfree = c.retain > 0 ? c.pool_count[1] : 0; return gie_alloc_slot(c, r, nfree, nfree, c.pool_count[0]); }

struct op_free_ray { const float *g; GIE_DEVM void operator()(const gie_ctx &c, int i) const { gie_free_ray(c, g, i); } };
/* block allocation (allocHashTB, glb_hash_map.cu:58-113) */
struct op_cell_flag { GIE_DEVM void operator()(const gie_ctx &c, int i) const { c.blk_new[i] = gie_cell_needs_new(c, i); } };
struct op_cell_insert { const int32_t *flag; const int32_t *rank;
    GIE_DEVM void operator()(const gie_ctx &c, int i) const {
        if (!flag[i]) return;
        const int slot = gie_emu_slot(c, rank[i]);
        if (slot >= c.max_blocks) { gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_POOL); return; }
        gie_cell_insert(c, i, slot);
    } };
struct op_cell_table { GIE_DEVM void operator()(const gie_ctx &c, int i) const {
        const int bx = i % c.tdim[0], by = (i / c.tdim[0]) % c.tdim[1], bz = i / (c.tdim[0] * c.tdim[1]);
        c.blk_tab[i] = gie_hash_find(c, bx + c.tb0[0], by + c.tb0[1], bz + c.tb0[2]);
        c.blk_need[i] = 0;
    } };

#endif /* GIE_EMU_OPS_H */
