/*
 * gie_ops.h — per-voxel and per-frontier-entry logic of the map update, written once and used
 * by the HIP kernels (gie_kernels.hip.h).  Each function cites the reference code it replaces.
 * Order-free by construction: conflicting writers go through 64-bit atomicMin / CAS, frontier
 * entries read a level-start snapshot (DESIGN.md "Canonical wave schedule").
 */
#ifndef GIE_OPS_H
#define GIE_OPS_H

#include "gie_types.h"

/* memory primitives, function qualifiers, loop pragmas: gie_platform.h (gfx950) — or the test backend's own, included before this file */
#ifndef GIE_PLATFORM_DEFINED
#include "gie_platform.h"
#endif

/* append to a frontier queue; overflow raises the sticky error flag */
GIE_DEV void gie_push64(const gie_ctx &c, uint64_t *q, int32_t *counter, int cap, uint64_t v)
{
    const int i = gie_aadd32(counter, 1);
    if (i < cap) gie_st(&q[i], v); else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
}
/* frontiers A / B carry the voxel's address (slot * 512 + in-block index) next to its coordinate: whoever
 * appends an entry has just touched that voxel, and the phases that expand it need no hash probe for it */
GIE_DEV void gie_push64a(const gie_ctx &c, uint64_t *q, gie_vaddr *qaddr, int32_t *counter, int cap, uint64_t v, gie_vaddr a)
{
    const int i = gie_aadd32(counter, 1);
    if (i < cap) { gie_st(&q[i], v); gie_st(&qaddr[i], a); } else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
}
/* the same append for every lane of the wave that has `push` set, with ONE atomic on the counter per wave (tens of
 * thousands of per-lane atomics on one word were most of obtainFrontiers' time when a whole face of the volume seeds waves
 * A / B).  Only executing lanes are looked at, so it is safe in divergent code. */
GIE_DEV void gie_push64a_wave(const gie_ctx &c, uint64_t *q, gie_vaddr *qaddr, int32_t *counter, int cap, bool push, uint64_t v, gie_vaddr a)
{
    const unsigned long long m = __ballot(push);
    if (!m) return;
    const int lane = __lane_id(), leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = gie_aadd32(counter, __popcll(m));
    base = __shfl(base, leader);
    if (push) {
        const int i = base + __popcll(m & ((1ull << lane) - 1ull));
        if (i < cap) { gie_st(&q[i], v); gie_st(&qaddr[i], a); } else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
    }
}
GIE_DEV void gie_push32(const gie_ctx &c, int32_t *q, int32_t *counter, int cap, int32_t v)
{
    const int i = gie_aadd32(counter, 1);
    if (i < cap) gie_st(&q[i], v); else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
}

GIE_DEV void gie_push32_wave(const gie_ctx &c, int32_t *q, int32_t *counter, int cap, bool push, int32_t v)
{
    const unsigned long long m = __ballot(push);
    if (!m) return;
    const int lane = __lane_id(), leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = gie_aadd32(counter, __popcll(m));
    base = __shfl(base, leader);
    if (push) {
        const int i = base + __popcll(m & ((1ull << lane) - 1ull));
        if (i < cap) gie_st(&q[i], v); else gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_QUEUE);
    }
}

/* ray_count[id] += val for every lane with id >= 0, with equal targets of NEIGHBOURING lanes merged
 * into one atomic: lanes are rays adjacent in the scan, which cross the same cells at the same
 * step near the sensor, so equal targets come as contiguous runs of lanes (unmerged, thousands of
 * wave-wide same-address atomics serialise at one L2 slice).  One shuffle and three ballots: a
 * lane heads a run when the lane below it does not take part or has another target; the head
 * adds for the whole run.  Equal targets that are not neighbours simply get their own atomics.
 * Safe in divergent code: only executing lanes are looked at. */
GIE_DEV void gie_wave_add(const gie_ctx &c, int id, int val)
{
    const int lane = __lane_id();
    const unsigned long long exec = __ballot(1);
    const unsigned long long valid = __ballot(id >= 0);
    if (!valid) return;
    const int below = __builtin_amdgcn_update_dpp(id, id, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);   /* lane-1's id without the LDS crossbar */
    const bool below_in = lane > 0 && ((exec & valid) >> (lane - 1)) & 1ull;
    const bool head = id >= 0 && !(below_in && below == id);
    const unsigned long long heads = __ballot(head);
    if (head) {
        const unsigned long long stops = (heads | ~valid) & ~((2ull << lane) - 1ull);   /* where the run above me ends */
        const int end = (lane == 63 || !stops) ? 64 : __ffsll((long long)stops) - 1;
        gie_aadd32(&c.ray_count[id], val * (end - lane));
    }
}

/* every cell whose _ray_count changes flags its tile, so that getAllocKeys (ray_finalize) only
 * looks at tiles a ray went through; `last` spares the store while the ray stays in one tile */
struct gie_ray_marks { int tile, blk; };
GIE_DEV void gie_ray_touch(const gie_ctx &c, int lx, int ly, int lz, gie_ray_marks *last)
{
    const int t = gie_tile_index(c, lx, ly, lz);
    if (t != last->tile) { c.tray[t] = 1; last->tile = t; }
    /* a touched cell needs its block (getAllocKeys, pntcld_raycast.cu:21-63: every cell whose count
     * is not zero — hits and misses never cancel: a free ray stops in front of a hit cell) */
    const int b = gie_tab_index(c, lx + c.pvt[0], ly + c.pvt[1], lz + c.pvt[2]);
    if (b != last->blk) { c.blk_need[b] = 1; last->blk = b; }
}

/* global voxel address: block slot through the frame's block table (volume +-1 voxel) */
GIE_DEV gie_vaddr gie_gvox_tab(const gie_ctx &c, int gx, int gy, int gz)
{
    const int s = c.blk_tab[gie_tab_index(c, gx, gy, gz)];
    return s < 0 ? (gie_vaddr)-1 : (gie_vaddr)s * GIE_VBSZ + gie_vox_in_blk(gx, gy, gz);
}
/* stream_VB_keys_D bookkeeping as one flag per block (all writers store 1) */
GIE_DEV void gie_touch(const gie_ctx &c, gie_vaddr a) { if (c.track) gie_st(&c.g_dirty[a >> 9], (int32_t)1); }
/* … or through the hash (anywhere) */
GIE_DEV gie_vaddr gie_gvox_hash(const gie_ctx &c, int gx, int gy, int gz)
{
    const int s = gie_hash_find(c, gx >> 3, gy >> 3, gz >> 3);
    return s < 0 ? (gie_vaddr)-1 : (gie_vaddr)s * GIE_VBSZ + gie_vox_in_blk(gx, gy, gz);
}

/* The stored squared distance of a global voxel is not a plane of its own: every writer of the reference stores dist_sq and
 * coc_glb together, and dist_sq is |coc_glb - voxel|^2 whenever the obstacle is valid (MarkLimitedObserve keeps an old pair
 * whole, obtainFrontiers / raise_outside / lower_outside / UpdateHashBatch compute the distance from the obstacle they store;
 * tests: every stored record is a witness), EMPTY_VALUE next to EMPTY_KEY otherwise.  So GlbVoxel.dist_sq is derived from the
 * stored obstacle where it is read — 4 bytes per voxel less to write in the commit sweep, which is bound by its stores. */
#define GIE_COC_STALEPAIR (1ull << 63)            /* bit 63 of a stored obstacle: see gie_commit_pair */
#define GIE_COC_EMPTY ((uint64_t)(uint32_t)(GIE_EMPTY_VALUE + GIE_CRD_OFF) | ((uint64_t)(uint32_t)(GIE_EMPTY_VALUE + GIE_CRD_OFF) << 21) | ((uint64_t)(uint32_t)(GIE_EMPTY_VALUE + GIE_CRD_OFF) << 42))
GIE_DEV int gie_gdist(const gie_ctx &c, uint64_t coc, int gx, int gy, int gz)
{
    coc &= ~GIE_COC_STALEPAIR;
    if (coc == GIE_COC_EMPTY) return c.empty_value;
    int cx, cy, cz;
    gie_unpack_crd(coc, &cx, &cy, &cz);
    return gie_d2(cx, cy, cz, gx, gy, gz);
}

/* ================================================================== OGM: projective */
GIE_DEV int gie_robot_sphere(const gie_ctx &c, int x, int y, int z)
{   /* pntcld_raycast.cu:33-41, vlp16_fast.cu:30-40: |crd - _half_shift|² <= rbt_r2_grids */
    if (!c.for_motion_planner) return 0;
    const int cx = x - (c.X / 2 - c.tile_off[0]), cy = y - (c.Y / 2 - c.tile_off[1]), cz = z - (c.Z / 2 - c.tile_off[2]);
    return cx * cx + cy * cy + cz * cz <= c.robot_r2;
}
GIE_DEV int gie_pos_mod(int i, int n) { return (i % n + n) % n; }

/* REALSENSE_FAST::setLocalOccupancy (realsense_fast.cu:9-94) + CAM_HELPER::G2L
 * (camera_helper.h:11-23).  Returns the scan label of local voxel (x,y,z). */
GIE_DEV int gie_classify_depth(const gie_ctx &c, const float *depth, const gie_cam_param &p, int x, int y, int z)
{
    if (gie_robot_sphere(c, x, y, z)) return GIE_VOX_FREE;
    const float w = c.voxel_width;
    const float gx = (float)(x + c.pvt[0]) * w, gy = (float)(y + c.pvt[1]) * w, gz = (float)(z + c.pvt[2]) * w;
    float lx, ly, lz;
    gie_se3_apply(c.G2L, gx, gy, gz, &lx, &ly, &lz);
    const float ideal = lx;
    if (ideal <= 0.3f || ideal > 6.0f) return GIE_VOX_UNKNOWN;
    const float fpx = floorf(-ly * p.fx / ideal + p.cx + 0.5f);
    const float fpy = floorf(-lz * p.fy / ideal + p.cy + 0.5f);
    if (!(fpx >= 0.0f && fpx < (float)p.cols && fpy >= 0.0f && fpy < (float)p.rows)) return GIE_VOX_UNKNOWN;
    float real = depth[p.cols * (int)fpy + (int)fpx];
    if (real <= 0.21f) return GIE_VOX_UNKNOWN;
    if (real != real) { if (p.valid_nan) real = 1000.f; else return GIE_VOX_UNKNOWN; }
    if (ideal < real - w) return GIE_VOX_FREE;
    if (ideal > real + w) return GIE_VOX_UNKNOWN;
    if (gz >= c.min_h && gz <= c.max_h) return GIE_VOX_OCCUPIED;
    return GIE_VOX_UNKNOWN;
}

/* VLP_FAST::setLocalOccupancy (vlp16_fast.cu:8-87) + VLP_HELPER::G2L (vlp16_helper.h:35-65) */
GIE_DEV int gie_classify_multiscan(const gie_ctx &c, const float *ranges, const gie_multiscan_param &p, int x, int y, int z)
{
    if (gie_robot_sphere(c, x, y, z)) return GIE_VOX_FREE;
    const float w = c.voxel_width;
    const float gx = (float)(x + c.pvt[0]) * w, gy = (float)(y + c.pvt[1]) * w, gz = (float)(z + c.pvt[2]) * w;
    float lx, ly, lz;
    gie_se3_apply(c.G2L, gx, gy, gz, &lx, &ly, &lz);
    /* (the ring first: two thirds of a cubic volume lie above or below a 16-ring lidar's field of view, whole wavefronts of 64
     * x-adjacent voxels at a time, and leave before the azimuth's atan2 and division; the values are the same in either order) */
    const float range_hor = sqrtf(ly * ly + lx * lx);
    const float phi = gie_atan2f(lz, range_hor);
    const int phi_idx = (int)floorf((phi - p.phi_min) / p.phi_inc + 0.5f);
    if (phi_idx < 0 || phi_idx >= p.ring_num) return GIE_VOX_UNKNOWN;
    const float theta = gie_atan2f(ly, lx);
    int theta_idx = (int)floorf((theta - p.theta_min) / p.theta_inc + 0.5f);
    theta_idx = gie_pos_mod(theta_idx, p.scan_num);
    const float ideal = range_hor;                    /* (sqrtf(lx * lx + ly * ly): the same sum, the same root) */
    if (ideal < 0 || theta_idx < 0 || theta_idx >= p.scan_num) return GIE_VOX_UNKNOWN;
    const float real = ranges[phi_idx * p.scan_num + theta_idx];
    if (real != real || real <= 0.3f) return GIE_VOX_UNKNOWN;
    if (ideal < real - 0.1f) return (ideal < real - 0.3f) ? GIE_VOX_FREE : GIE_VOX_UNKNOWN;
    if ((double)ideal > (double)real + 0.1) return GIE_VOX_UNKNOWN;
    if (gz >= c.min_h && gz <= c.max_h) return GIE_VOX_OCCUPIED;
    return GIE_VOX_UNKNOWN;
}

/* HOKUYO_FAST::setLocalOccupancy (hokuyo_fast.cu:9-81) + SCAN_HELPER::G2L (hokuyo_helper.h:17-33) */
GIE_DEV int gie_classify_scan2d(const gie_ctx &c, const float *ranges, const gie_scan_param &p, int x, int y, int z)
{
    if (gie_robot_sphere(c, x, y, z)) return GIE_VOX_FREE;
    const float w = c.voxel_width;
    const float gx = (float)(x + c.pvt[0]) * w, gy = (float)(y + c.pvt[1]) * w, gz = (float)(z + c.pvt[2]) * w;
    float lx, ly, lz;
    gie_se3_apply(c.G2L, gx, gy, gz, &lx, &ly, &lz);
    if (!(fabsf(lz) < w)) return GIE_VOX_UNKNOWN;        /* (the slab test first: nearly every voxel leaves here, before the atan2) */
    const float theta = gie_atan2f(ly, lx);
    int theta_idx = (int)floorf((theta - p.theta_min) / p.theta_inc + 0.5f);
    theta_idx = gie_pos_mod(theta_idx, p.scan_num);
    const float ideal = sqrtf(lx * lx + ly * ly);
    const float real = ranges[theta_idx];
    if (real != real || real <= 0.3f) return GIE_VOX_UNKNOWN;
    if (ideal < real - 0.3f) return GIE_VOX_FREE;
    if ((double)ideal > (double)real + 0.3) return GIE_VOX_UNKNOWN;
    if (gz >= c.min_h && gz <= c.max_h) return GIE_VOX_OCCUPIED;
    return GIE_VOX_UNKNOWN;
}

/* the block of an observed voxel needs to exist (the reference's per-voxel VB_keys_loc_D entry) */
GIE_DEV void gie_mark_block_needed(const gie_ctx &c, int x, int y, int z)
{
    c.blk_need[gie_tab_index(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2])] = 1;
}

/* a pre-classified scan (gie_ogm_labels): the label a projective kernel would have left */
GIE_DEV int gie_classify_label(const gie_ctx &c, const int8_t *labels, int x, int y, int z)
{
    if (gie_robot_sphere(c, x, y, z)) return GIE_VOX_FREE;
    const int8_t l = labels[gie_lid(c, x, y, z)];
    return (l == GIE_VOX_FREE || l == GIE_VOX_OCCUPIED) ? l : GIE_VOX_UNKNOWN;
}

/* ================================================================== OGM: ray casting */
/* registerLocObs, pntcld_raycast.cu:83-102.  g_out receives the global-frame point. */
GIE_DEV void gie_register_point(const gie_ctx &c, const float *xyz, float *g_out, int i)
{
    float gx, gy, gz;
    gie_se3_apply(c.L2G, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &gx, &gy, &gz);
    g_out[3 * i] = gx; g_out[3 * i + 1] = gy; g_out[3 * i + 2] = gz;
    if (gie_point_ok(gx, gy, gz) && gz >= c.min_h && gz <= c.max_h) {
        const int lx = gie_pos2coord(gx, c.voxel_width) - c.pvt[0];
        const int ly = gie_pos2coord(gy, c.voxel_width) - c.pvt[1];
        const int lz = gie_pos2coord(gz, c.voxel_width) - c.pvt[2];
        if (gie_in_loc(c, lx, ly, lz)) {
            const int id = gie_lid(c, lx, ly, lz);
            c.inst_type[id] = GIE_VOX_OCCUPIED;      /* all writers store the same value */
            gie_ray_marks lt = { -1, -1 };
            gie_ray_touch(c, lx, ly, lz, &lt);
            gie_wave_add(c, id, 1);
        }
    }
}

/* freeLocObs (pntcld_raycast.cu:67-80) → RAY::rayCastLoc (ray_cast.h:57-144).
 * The 3-D DDA itself, shared by the sequential walk below and by the segmented kernel
 * (k_free_rays): same float operations in the same order, so a replayed walk visits the same cells. */
struct gie_dda { int cur[3], step[3], i1[3]; float tMax[3], tDelta[3]; float len, max_length; };
/* returns 0 when origin and end point share a cell (nothing to walk); sensor cell (local) in s0 */
GIE_DEV int gie_dda_init(const gie_ctx &c, const float *g, int i, gie_dda &d, int s0[3])
{
    const float w = c.voxel_width;
    const float p0[3] = { c.origin[0], c.origin[1], c.origin[2] };
    const float p1[3] = { g[3 * i], g[3 * i + 1], g[3 * i + 2] };
    d.max_length = 0.707f * (float)c.X * w;
    int i0[3];
    for (int k = 0; k < 3; k++) { i0[k] = gie_pos2coord(p0[k], w); d.i1[k] = gie_pos2coord(p1[k], w); d.cur[k] = i0[k]; s0[k] = i0[k] - c.pvt[k]; }
    if (i0[0] == d.i1[0] && i0[1] == d.i1[1] && i0[2] == d.i1[2]) return 0;
    float dir[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] };
    d.len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    for (int k = 0; k < 3; k++) dir[k] = dir[k] / d.len;
    for (int k = 0; k < 3; k++) {
        if (dir[k] > 0.0f) d.step[k] = 1; else if (dir[k] < 0.0f) d.step[k] = -1; else d.step[k] = 0;
        if (d.step[k] != 0) {
            const float border = (float)d.cur[k] * w + (float)d.step[k] * w * 0.5f;
            d.tMax[k] = (border - p0[k]) / dir[k];
            d.tDelta[k] = w / fabsf(dir[k]);
        } else { d.tMax[k] = 3.402823466e+38f; d.tDelta[k] = 3.402823466e+38f; }
    }
    return 1;
}
/* one cell forward; returns the reference's stop test for the NEW cell (end cell reached or
 * past the ray / the maximum length): that cell is still cleared, the walk ends after it */
GIE_DEV int gie_dda_step(gie_dda &d)
{
    /* the axis whose boundary comes first (ray_cast.h:104-118: x if tMax.x < tMax.y and < tMax.z, else y if
     * tMax.y < tMax.z, else z), written with selects only: as branches this loop body costs several
     * exec-mask round trips per step, and the walk is a chain of a few hundred dependent steps */
    const bool c01 = d.tMax[0] < d.tMax[1], c02 = d.tMax[0] < d.tMax[2], c12 = d.tMax[1] < d.tMax[2];
    const bool is0 = c01 & c02, is1 = (!c01) & c12, is2 = !(is0 | is1);
    d.cur[0] = is0 ? d.cur[0] + d.step[0] : d.cur[0]; d.tMax[0] = is0 ? d.tMax[0] + d.tDelta[0] : d.tMax[0];
    d.cur[1] = is1 ? d.cur[1] + d.step[1] : d.cur[1]; d.tMax[1] = is1 ? d.tMax[1] + d.tDelta[1] : d.tMax[1];
    d.cur[2] = is2 ? d.cur[2] + d.step[2] : d.cur[2]; d.tMax[2] = is2 ? d.tMax[2] + d.tDelta[2] : d.tMax[2];
    const float m01 = d.tMax[0] < d.tMax[1] ? d.tMax[0] : d.tMax[1];
    const float dist = m01 < d.tMax[2] ? m01 : d.tMax[2];
    const int at_end = (d.cur[0] == d.i1[0]) & (d.cur[1] == d.i1[1]) & (d.cur[2] == d.i1[2]);
    return at_end | (int)(dist > d.max_length) | (int)(dist > d.len);
}
/* upper bound of the cells one ray can visit: max_length / w voxels along the ray, at most
 * |dx|+|dy|+|dz| <= sqrt(3) cell changes per voxel of length */
GIE_HD int gie_ray_max_steps(const gie_ctx &c) { return (int)(0.707f * (float)c.X * 1.7321f) + 8; }

#define GIE_RAY_BATCH 8     /* cells whose types a ray walk fetches together (k_free_rays) */

/* getAllocKeys, pntcld_raycast.cu:21-63 */
GIE_DEV void gie_raycast_finalize(const gie_ctx &c, int x, int y, int z)
{
    const int id = gie_lid(c, x, y, z);
    if (gie_robot_sphere(c, x, y, z)) c.ray_count[id] = -1;
    const int cnt = c.ray_count[id];
    if (cnt == 0) return;
    c.inst_type[id] = cnt > 0 ? GIE_VOX_OCCUPIED : GIE_VOX_FREE;
    gie_mark_block_needed(c, x, y, z);
}

/* ================================================================== block allocation */
/* TryAllocateKernel / insert_to_id (alloc_helper.cuh:30-50, vhashing.h:387-455) without locks:
 * the slot comes from an exclusive scan (deterministic), the key is claimed by CAS. */
GIE_DEV void gie_key_insert(const gie_ctx &c, uint64_t key, int bx, int by, int bz, int slot)
{
    uint32_t h = gie_hash_key(bx, by, bz) & c.hmask;
    for (uint32_t probes = 0; probes <= c.hmask; probes++) {
        /* a free cell or the cell of an erased block (the caller has looked the key up: it is not further down the chain) */
        const uint64_t k = gie_ld(&c.hkeys[h]);
        if (k == GIE_KEY_EMPTY || k == GIE_KEY_TOMB) {
            if (gie_acas64(&c.hkeys[h], k, key) == k) { c.hvals[h] = slot; c.g_key[slot] = key; return; }
        }
        h = (h + 1) & c.hmask;
    }
    gie_aor32(&c.cnt[GIE_CNT_ERR], GIE_ERRF_HASH);
}
GIE_DEV void gie_cell_insert(const gie_ctx &c, int cell, int slot)
{
    const int bx = cell % c.tdim[0] + c.tb0[0], by = (cell / c.tdim[0]) % c.tdim[1] + c.tb0[1], bz = cell / (c.tdim[0] * c.tdim[1]) + c.tb0[2];
    gie_key_insert(c, gie_pack_crd(bx, by, bz), bx, by, bz, slot);
}

/* The slot the block table of the fuse before (c.tab_prev, shifted by c.tab_prev_d cells) holds for table cell (bx, by, bz) of this
 * update, -1: none / not in that table — then the hash answers.  INVARIANT: a slot returned here is the one the hash would find.
 * With block erasure (retain_radius_blocks) the table before may name a block that this update's erasure — it runs before the
 * allocation — has just taken away: the table reaches a cell or two beyond the retention box, and an entry there would be handed
 * from table to table for as long as the cell stays in range (found by the round-4 fuzz, seeds 51 / 53: a robot that turns round
 * met its erased blocks again, with their old contents).  The slot's own key says whether it still holds this block.  (The
 * emulation's allocation never looks at the table; it checks the invariant for every cell: tests/emu/gie_emu.cpp.) */
GIE_DEV int gie_cell_prev_slot(const gie_ctx &c, int bx, int by, int bz)
{
    if (!c.tab_prev) return -1;
    const int px = bx + c.tab_prev_d[0], py = by + c.tab_prev_d[1], pz = bz + c.tab_prev_d[2];
    if (!((unsigned)px < (unsigned)c.tdim[0] && (unsigned)py < (unsigned)c.tdim[1] && (unsigned)pz < (unsigned)c.tdim[2])) return -1;
    const int found = c.tab_prev[(pz * c.tdim[1] + py) * c.tdim[0] + px];
#if !defined(GIE_EMU_BREAK_PREV_CHECK)
    if (found >= 0 && c.retain > 0 && c.g_key[found] != gie_pack_crd(bx + c.tb0[0], by + c.tb0[1], bz + c.tb0[2])) return -1;
#endif
    return found;
}

/* ---- block-pool lifecycle (gie_config.retain_radius_blocks > 0; the reference never erases: blockalloc.h:50-67) */
/* Slot `r` of `cnt` blocks a caller allocates at once: the first `nf` come from the top of the free list (entries
 * [ftop - nf, ftop)), the rest from the bump allocator starting at `base`. */
GIE_DEV int gie_alloc_slot(const gie_ctx &c, int r, int nf, int ftop, int base)
{ return r < nf ? c.free_list[ftop - 1 - r] : base + (r - nf); }
/* erase block `slot` when it lies more than `retain` blocks outside the block box of the volume: its hash cell becomes a
 * tombstone, the slot goes onto the free list, its voxels are re-initialised when the slot is handed out again */
GIE_DEV int gie_evict_slot(const gie_ctx &c, int slot)
{
    const uint64_t key = c.g_key[slot];
    if (key == GIE_KEY_EMPTY) return 0;                             /* already on the free list */
    int b[3];
    gie_unpack_crd(key, &b[0], &b[1], &b[2]);
    int far = 0;
    for (int i = 0; i < 3; i++) far |= (b[i] < c.vb_lo[i] - c.retain) | (b[i] > c.vb_hi[i] + c.retain);
    if (!far) return 0;
    uint32_t h = gie_hash_key(b[0], b[1], b[2]) & c.hmask;
    for (uint32_t probes = 0; probes <= c.hmask; probes++) {
        const uint64_t k = c.hkeys[h];
        if (k == key) { c.hkeys[h] = GIE_KEY_TOMB; break; }
        if (k == GIE_KEY_EMPTY) break;
        h = (h + 1) & c.hmask;
    }
    c.g_key[slot] = GIE_KEY_EMPTY;
    c.g_dirty[slot] = 0;                                            /* erased blocks are not streamed */
    c.free_list[gie_aadd32(&c.pool_count[1], 1)] = slot;
    return 1;
}
/* re-insert a live slot into a freshly cleared table (drops the tombstones) */
GIE_DEV void gie_rehash_slot(const gie_ctx &c, int slot)
{
    const uint64_t key = c.g_key[slot];
    if (key == GIE_KEY_EMPTY) return;
    int b[3];
    gie_unpack_crd(key, &b[0], &b[1], &b[2]);
    gie_key_insert(c, key, b[0], b[1], b[2], slot);
}

/* row `slot` of the neighbour table (c.g_nbr), direction k (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z): the slot of the block across that face,
 * and this block's slot into that neighbour's row.  Called for every block that is initialised (new, or a slot handed out again),
 * after the launch that inserted the keys: two blocks that are new together write the same two words.  This library's own
 * structure — the reference probes its hash per voxel and direction (wave_core.cuh:139-150, 283-296). */
GIE_DEV void gie_nbr_link(const gie_ctx &c, int slot, int k)
{
    int b[3];
    gie_unpack_crd(c.g_key[slot], &b[0], &b[1], &b[2]);
    b[k >> 1] += (k & 1) ? 1 : -1;
    const int ns = gie_hash_find(c, b[0], b[1], b[2]);
    c.g_nbr[8 * (size_t)slot + k] = ns;
    if (ns >= 0) c.g_nbr[8 * (size_t)ns + (k ^ 1)] = slot;
}

/* GlbVoxel defaults (voxmap_utils.cuh:30-43) for voxel i of a fresh block */
GIE_DEV void gie_init_voxel(const gie_ctx &c, int slot, int i)
{
    const gie_vaddr a = (gie_vaddr)slot * GIE_VBSZ + i;
    c.g_occ[a] = 0; c.g_type[a] = GIE_VOX_UNKNOWN;
    c.g_coc[a] = gie_pack_crd(GIE_EMPTY_VALUE, GIE_EMPTY_VALUE, GIE_EMPTY_VALUE);
    c.g_pair[a] = 0; c.g_prop[a] = GIE_NOPROP; c.g_prop2[a] = GIE_NOPROP; c.g_wl[a] = -1;
}

/* an existing block makes the (up to eight) local tiles it overlaps interesting for fuse */
GIE_DEV void gie_cell_mark_tiles(const gie_ctx &c, int cell)
{
    const int bx = cell % c.tdim[0] + c.tb0[0], by = (cell / c.tdim[0]) % c.tdim[1] + c.tb0[1], bz = cell / (c.tdim[0] * c.tdim[1]) + c.tb0[2];
    const int l0[3] = { bx * 8 - c.pvt[0], by * 8 - c.pvt[1], bz * 8 - c.pvt[2] };
    int t0[3], t1[3];
    for (int a = 0; a < 3; a++) {
        t0[a] = l0[a] >> 3; t1[a] = (l0[a] + 7) >> 3;              /* arithmetic shifts: floor */
        if (t0[a] < 0) t0[a] = 0;
        if (t1[a] > c.tfd[a] - 1) t1[a] = c.tfd[a] - 1;
        if (t0[a] > t1[a]) return;                                  /* the block lies outside the volume */
    }
    for (int tz = t0[2]; tz <= t1[2]; tz++) for (int ty = t0[1]; ty <= t1[1]; ty++) for (int tx = t0[0]; tx <= t1[0]; tx++)
        c.tact[(tz * c.tfd[1] + ty) * c.tfd[0] + tx] = 1;          /* all writers store 1 */
}
/* fuse looks at a tile when a block overlaps it or its _glb_type is not all-UNKNOWN yet; a tile
 * that is not listed stays all-unknown */
GIE_DEV int gie_fuse_tile_listed(const gie_ctx &c, int t)
{
    const int v = c.tact[t] | c.tknown_prev[t];
    c.tact[t] = (uint8_t)v;                                         /* from here on: "fuse looked at this tile" */
    return v;
}

/* ================================================================== fuse */
GIE_DEV int gie_inside_aabb(float px, float py, float pz, const float *ll, const float *ur)
{ return px >= ll[0] && py >= ll[1] && pz >= ll[2] && px <= ur[0] && py <= ur[1] && pz <= ur[2]; }

/* set_hashvoxel_occ_val, voxmap_utils.cuh:181-200 */
GIE_DEV void gie_set_occ(uint8_t *occ, int8_t *type, float val, float a, int thresh)
{
    if (*type != GIE_VOX_UNKNOWN) val = a * val + (1.0f - a) * (float)(*occ);
    else val = a * val + (1.0f - a) * 0.0f;
    if (val > 254.0f) val = 254.0f;
    if (val < 1.0f) val = 1.0f;
    *occ = (uint8_t)val;
    *type = (*occ > thresh) ? GIE_VOX_OCCUPIED : GIE_VOX_FREE;
}


/* The occupancy filter of a labelled scan for EIGHT voxels at once: bytes i of `it8` / `go8` / `gy8` = scan label / stored occupancy /
 * stored type of voxel i of a row, results in *no8 / *ny8.  Nearly all voxels are free space seen free, or not seen at all, and for
 * those set_hashvoxel_occ_val (voxmap_utils.cuh:181-200) is integer arithmetic on bytes:
 *   label FREE: val = 0.5 * occ (occ = 0 for an UNKNOWN voxel), clamped up to 1, truncated = max(1, occ >> 1); the type becomes FREE
 *     because occ >> 1 <= 127 <= the occupancy threshold (the caller checks thresh >= 127);
 *   no label (or any other value): nothing changes;
 *   label OCCUPIED (1 % of the voxels): the fp32 filter, voxel by voxel — a wavefront pays for the largest number of such voxels
 *     one of its rows holds (one or two), not for eight.
 * Equal to gie_fuse_logic byte by byte (tests/test_host_logic.py::test_byte_parallel_occupancy_filter, exhaustive). */
GIE_DEV void gie_fuse_row8_labels(int thresh, uint64_t it8, uint64_t go8, uint64_t gy8, uint64_t *no8, uint64_t *ny8)
{
    const uint64_t ONES = 0x0101010101010101ull, L7 = 0x7f7f7f7f7f7f7f7full, H8 = 0x8080808080808080ull;
    const uint64_t tf = it8 ^ ONES, to = it8 ^ (2ull * ONES);
    const uint64_t mF = (((((tf & L7) + L7) | tf) & H8) ^ H8) >> 7;           /* 1 in the bytes whose label is FREE */
    const uint64_t mO = (((((to & L7) + L7) | to) & H8) ^ H8) >> 7;           /* ... OCCUPIED */
    const uint64_t fF = mF * 0xffull;
    const uint64_t mK = ((gy8 | (gy8 >> 1)) & ONES) * 0xffull;                /* 0xff in the bytes whose stored type is known (types are 0..3) */
    uint64_t half = (go8 >> 1) & L7 & mK;
    half += ONES ^ (((half + L7) & H8) >> 7);                                 /* 0 -> 1 */
    uint64_t o8 = (go8 & ~fF) | (half & fF), y8 = (gy8 & ~fF) | (ONES & fF);
    uint32_t ob = (uint32_t)((mO * 0x0102040810204080ull) >> 56);             /* bit i: voxel i carries an OCCUPIED label */
    while (ob) {
        const int i = __ffs((int)ob) - 1;
        ob &= ob - 1u;
        uint8_t occ = (uint8_t)(go8 >> (8 * i));
        int8_t ty = (int8_t)(gy8 >> (8 * i));
        gie_set_occ(&occ, &ty, 250.f, 0.8f, thresh);
        o8 = (o8 & ~(0xffull << (8 * i))) | ((uint64_t)occ << (8 * i));
        y8 = (y8 & ~(0xffull << (8 * i))) | ((uint64_t)(uint8_t)ty << (8 * i));
    }
    *no8 = o8; *ny8 = y8;
}

struct gie_fuse_st { int count; int8_t nt, gt0; gie_vaddr a; uint8_t occ; int8_t ty; };

GIE_DEV void gie_fuse_load1(const gie_ctx &c, int id, int x, int y, int z, gie_fuse_st &s)
{
    s.count = c.pntcld_mode ? c.ray_count[id] : 0;
    s.nt = c.scan_labels ? c.scan_labels[id] : c.inst_type[id];      /* (labels other than FREE / OCCUPIED count as "not observed": gie_fuse_logic) */
    s.gt0 = c.glb_type[id];
    s.a = gie_gvox_tab(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
}
GIE_DEV void gie_fuse_load2(const gie_ctx &c, gie_fuse_st &s)
{
    s.occ = 0; s.ty = GIE_VOX_UNKNOWN;
    if (s.a < 0) return;
    s.occ = c.g_occ[s.a];
    s.ty = c.g_type[s.a];
}
/* ================================================================== lazy pairs (round 6)
 * In a tile the fused sweep handles by its short way — tskip 2 with deferred records, the volume inside the wave range: 82 % of the
 * headline volume — MarkLimitedObserve's answer is a function of the batch obstacle alone: (|voxel - obstacle|^2, obstacle in
 * wave-range coordinates), or (EMPTY, none) without one.  The sweep wrote those 8 bytes per voxel for somebody to read them back
 * unchanged; since round 6 it does not: it flags the tile (`tlazy`), and whoever wants the pair of a voxel of a flagged tile derives
 * it from the batch-obstacle plane the flags refer to (`bcoc_lazy`, which alternates with the plane the next batch EDT writes) at the
 * pivots of the merge that left it (pp_pvt / pp_upvt).  A flagged tile gets its pairs into the plane when
 *   - wave C changes a pair of it (the tile's wave writes all 512 and takes the flag away),
 *   - the next gie_fuse finds that it will not be flagged again (gie_pair_materialise_tile: before the next Mark reads old pairs by
 *     local index), or every flag falls (the reference's order of kernels, a ray-cast scan, an update without the bound).
 * The flags are ONE plane that is never swapped or cleared with the frame: it says what the pair plane holds now.
 * Readers: obtainFrontiers and wave C (tile loads and halos), the pair flush and the catch-up of stored records in gie_fuse, the
 * exports and the single-voxel readers. */
GIE_DEV uint64_t gie_pair_of_bcoc(const gie_ctx &c, uint32_t bc, int x, int y, int z)
{
    if (bc == GIE_BCOC_NONE) return gie_pair_make(c.empty_value, GIE_PAR_NONE);
    const int cx = (int)(bc & 1023u), cy = (int)((bc >> 10) & 1023u), cz = (int)(bc >> 20);
    const int dx = x - cx, dy = y - cy, dz = z - cz;
    return gie_pair_make(dx * dx + dy * dy + dz * dz, gie_pack_wr(cx + c.pp_pvt[0] - c.pp_upvt[0], cy + c.pp_pvt[1] - c.pp_upvt[1], cz + c.pp_pvt[2] - c.pp_upvt[2]));
}
/* the pair of local voxel (x, y, z) = index id of the update the pair plane is from */
GIE_DEV uint64_t gie_pair_get(const gie_ctx &c, int id, int x, int y, int z)
{
    return c.tlazy[gie_tile_index(c, x, y, z)] ? gie_pair_of_bcoc(c, c.bcoc_lazy[id], x, y, z) : c.pair[id];
}
GIE_DEV uint64_t gie_pair_get_id(const gie_ctx &c, int id)
{
    const int plane = c.X * c.Y, z = id / plane, r = id - z * plane, y = r / c.X, x = r - y * c.X;
    return gie_pair_get(c, id, x, y, z);
}
/* all 512 pairs of a flagged tile into the pair plane (one lane per z-column: l = 0 .. 63); the caller takes the flag away */
GIE_DEV void gie_pair_materialise_column(const gie_ctx &c, int t, int l)
{
    const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
    const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
    if (x >= c.X || y >= c.Y) return;
    for (int k = 0; k < 8; k++) {
        const int z = tz * 8 + k;
        if (z >= c.Z) break;
        const int id = gie_lid(c, x, y, z);
        c.pair[id] = gie_pair_of_bcoc(c, c.bcoc_lazy[id], x, y, z);
    }
}

/* ---- `_edt_D` (the float distance plane of the CostMap) is DERIVED, not stored (round 4).  UpdateHashBatch writes it for every
 * known voxel from the voxel's final pair — sqrtf(dist), or X^2+Y^2+Z^2 for "see nothing" (EMPTY, 0xffffffff) — except where the
 * pair is (EMPTY, some parent): obstacle outside the wave range, "don't update" (unify_helper.cuh:467-475); UNKNOWN voxels keep
 * theirs, by LOCAL index, and so does `_dist_id_pair`.  Hence the plane is a function of the pair plane wherever the pair at an
 * index is the one its last commit saw and is not of the keeping kind, and the 4-byte store per voxel (0.54 GB of the 2.7 GB the
 * C5 sweep wrote, ~0.1 ms) is left out; readers (gie_read_local, gie_read_costmap) evaluate gie_edt_value.  The plane `edt`
 * holds the value only where that does not work, written at the moment it stops working:
 *   - Mark turns a known voxel's pair into (EMPTY, parent): edt <- value of the pair it overwrites (gie_edt_before_keep);
 *   - a wave changes the pair of an UNKNOWN voxel (lower_inside and lower_outside's stores have no type test, wave_core.cuh:
 *     336-346, 353-393; UpdateHashBatch skips the voxel): edt <- value of the pair before the change, and the index is marked
 *     (gie_edt_unknown_touch) — one bit per index (`ucol`: a byte per z-column of eight voxels, bit = z & 7), cleared when Mark
 *     next writes the index's pair.
 * An index that merely changes between known and unknown (the volume moves) needs nothing: neither plane is touched. */
GIE_DEV bool gie_pair_keeps_edt(const gie_ctx &c, uint64_t pr) { return gie_pair_dist(pr) == c.empty_value && gie_pair_par(pr) != GIE_PAR_NONE; }
GIE_DEV float gie_edt_of_pair(const gie_ctx &c, uint64_t pr) { const int d = gie_pair_dist(pr); return d == c.empty_value ? (float)c.max_loc_dist_sq : sqrtf((float)d); }
GIE_DEV size_t gie_ucol_index(const gie_ctx &c, int x, int y, int z) { return ((size_t)(z >> 3) * c.Y + y) * c.X + x; }
GIE_DEV float gie_edt_value(const gie_ctx &c, int id)
{
    const int plane = c.X * c.Y, z = id / plane, r = id - z * plane;
    const uint64_t pr = gie_pair_get_id(c, id);
    const bool marked = (c.ucol[(size_t)(z >> 3) * plane + r] >> (z & 7)) & 1u;
    return (!marked && !gie_pair_keeps_edt(c, pr)) ? gie_edt_of_pair(c, pr) : c.edt[id];
}
/* Mark is about to store `pr_new` for the known voxel `id`; `marked` = the index's ucol bit */
GIE_DEV void gie_edt_before_keep(const gie_ctx &c, int id, uint64_t pr_new, bool marked)
{
    if (marked || !gie_pair_keeps_edt(c, pr_new)) return;
    const uint64_t old = c.pair[id];
    if (!gie_pair_keeps_edt(c, old)) c.edt[id] = gie_edt_of_pair(c, old);
}
/* a wave is about to change the pair of the UNKNOWN voxel (x, y, z) from `old`; the caller owns the voxel's z-column (a lane of
 * wave C's tile routine), other wavefronts read the byte in later rounds: agent-scope accesses */
GIE_DEV void gie_edt_unknown_touch(const gie_ctx &c, int id, int x, int y, int z, uint64_t old)
{
    uint8_t *const u = &c.ucol[gie_ucol_index(c, x, y, z)];
    const uint8_t ub = gie_ld(u);
    if ((ub >> (z & 7)) & 1u) return;
    if (!gie_pair_keeps_edt(c, old)) gie_st(&c.edt[id], gie_edt_of_pair(c, old));
    gie_st(u, (uint8_t)(ub | (1u << (z & 7))));
}

/* external obstacle boxes (unify_helper.cuh:60-78, voxmap_utils.cuh:203-207): box 0 is a fence ("outside => occupied"),
 * boxes >= 1 are obstacles ("inside => occupied"), each only while activated */
GIE_DEV int gie_fuse_occ_flag(const gie_ctx &c, int gx, int gy, int gz)
{
    if (c.nbox <= 0) return 0;
    const float w = c.voxel_width;
    const float px = (float)gx * w, py = (float)gy * w, pz = (float)gz * w;
    if (c.box_act[0] && !gie_inside_aabb(px, py, pz, c.box_ll, c.box_ur)) return 1;
    for (int i = 1; i < c.nbox; i++)
        if (c.box_act[i] && gie_inside_aabb(px, py, pz, c.box_ll + 3 * i, c.box_ur + 3 * i)) return 1;
    return 0;
}
/* the occupancy filter of one voxel on values in registers: scan (count / label) + stored (occ, ty) -> new (occ, ty) */
GIE_DEV void gie_fuse_logic(const gie_ctx &c, int count, int8_t nt, int occ_flag, uint8_t *occ, int8_t *ty)
{
    if (c.pntcld_mode) {
        if (count > 0 || occ_flag) gie_set_occ(occ, ty, 250.f, 1.f, c.occ_thresh);
        else if (count < 0) { float pb = (float)(-count) / 10.f; if (pb > 1.f) pb = 1.f; gie_set_occ(occ, ty, 0.f, pb, c.occ_thresh); }
    } else {
        if (nt == GIE_VOX_OCCUPIED || occ_flag) gie_set_occ(occ, ty, 250.f, 0.8f, c.occ_thresh);
        else if (nt == GIE_VOX_FREE) gie_set_occ(occ, ty, 0.f, 0.5f, c.occ_thresh);
    }
}
/* returns 1 when the voxel ends up known (feeds the per-tile known/unknown summaries) */
GIE_DEV int gie_fuse_finish(const gie_ctx &c, int id, int x, int y, int z, const gie_fuse_st &s)
{
    const int count = s.count;
    if (count != 0) c.ray_count[id] = 0;                                 /* write only what changes */
    const int8_t nt = s.nt;
    if (nt != GIE_VOX_UNKNOWN && !c.scan_labels) c.inst_type[id] = GIE_VOX_UNKNOWN;
    const int8_t gt0 = s.gt0;
    const gie_vaddr a = s.a;
    if (a < 0) { if (gt0 != GIE_VOX_UNKNOWN) c.glb_type[id] = GIE_VOX_UNKNOWN; return 0; }
    const int occ_flag = gie_fuse_occ_flag(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
    uint8_t occ = s.occ;
    int8_t ty = s.ty;
    const int8_t ty0 = ty;
    const uint8_t occ0 = occ;
    gie_fuse_logic(c, count, nt, occ_flag, &occ, &ty);
    if (occ != occ0) c.g_occ[a] = occ;
    if (ty != ty0) { c.g_type[a] = ty; gie_touch(c, a); }
    if (gt0 != ty) c.glb_type[id] = ty;
    if (ty == GIE_VOX_OCCUPIED) c.zocc[z] = 1;                           /* all writers store 1 */
    return ty != GIE_VOX_UNKNOWN;
}
/* A z-column of 8 voxels (x, y, z0..z0+7: one tile, at most two blocks) has nothing to fuse when
 * neither block exists — a voxel the scan observed always has its block by now — and the tile's
 * _glb_type is still all UNKNOWN from the frames before.  Such a column is all-unknown. */
GIE_DEV int gie_fuse_column_idle(const gie_ctx &c, int x, int y, int z0)
{
    const int t = gie_tile_index(c, x, y, z0);
    if (c.tknown_prev[t]) return 0;
    const int z1 = (z0 + 7 < c.Z) ? z0 + 7 : c.Z - 1;
    const int gx = x + c.pvt[0], gy = y + c.pvt[1];
    if (c.blk_tab[gie_tab_index(c, gx, gy, z0 + c.pvt[2])] >= 0 || c.blk_tab[gie_tab_index(c, gx, gy, z1 + c.pvt[2])] >= 0) return 0;
    c.tunk[t] = 1;
    return 1;
}
GIE_DEV void gie_fuse_column_summary(const gie_ctx &c, int x, int y, int z0, unsigned known, unsigned valid)
{
    const int t = gie_tile_index(c, x, y, z0);
    if (known) c.tknown[t] = 1;                 /* all writers store 1 */
    if (known != valid) c.tunk[t] = 1;
}
/* updateHashOGMWithPntCld / updateHashOGMWithSensor, unify_helper.cuh:35-197 */
GIE_DEV int gie_fuse_voxel(const gie_ctx &c, int x, int y, int z)
{
    const int id = gie_lid(c, x, y, z);
    gie_fuse_st s;
    gie_fuse_load1(c, id, x, y, z, s);
    gie_fuse_load2(c, s);
    return gie_fuse_finish(c, id, x, y, z, s);
}

/* ================================================================== Mark */
/* MarkLimitedObserve, unify_helper.cuh:201-273 */
/* Split in stages so that a thread can keep the loads of several voxels in flight (the sweep is
 * latency-bound): load1 = independent reads, load2 = reads that need load1's block slot,
 * finish = arithmetic + writes.  gie_mark_voxel chains them for one voxel. */
struct gie_mark_st { uint32_t bc; uint64_t pr; gie_vaddr a; uint64_t ococ; };

GIE_DEV void gie_mark_load1(const gie_ctx &c, int id, int x, int y, int z, gie_mark_st &s)
{
    s.bc = c.bcoc[id];
    s.pr = c.pair[id];
    s.a = gie_gvox_tab(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
}
GIE_DEV void gie_mark_load2(const gie_ctx &c, gie_mark_st &s)
{
    if (s.a < 0) return;
    s.ococ = c.g_coc[s.a];
}
GIE_DEV void gie_mark_finish(const gie_ctx &c, int id, int x, int y, int z, const gie_mark_st &s)
{
    if (s.a < 0) return;
    const uint32_t bc = s.bc;
    int cn[3];
    const int dn = gie_bcoc_dist(bc, x, y, z, c.max_width * c.max_width);   /* `_aux` after the batch EDT */
    int auxv = dn;
    uint64_t pr = s.pr;
    const int batch_invalid = (bc == GIE_BCOC_NONE);
    if (batch_invalid) {
        pr = gie_pair_make(c.empty_value, GIE_PAR_NONE);
        auxv = c.empty_value;
        cn[0] = cn[2] = 0; cn[1] = 16383;          /* the oracle's invalid marker: outside every wave range */
    } else { cn[0] = (int)(bc & 1023u); cn[1] = (int)((bc >> 10) & 1023u); cn[2] = (int)(bc >> 20); }
    const int dold = gie_gdist(c, s.ococ, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
    int ox, oy, oz;
    gie_unpack_crd(s.ococ, &ox, &oy, &oz);
    const int ol[3] = { ox - c.pvt[0], oy - c.pvt[1], oz - c.pvt[2] };
    /* limited observation: the old closest obstacle is out of sight (outside the volume; with
     * tiling: outside the union of all tiles) */
    if (dn > dold && !gie_in_whole(c, ol[0], ol[1], ol[2])) { cn[0] = ol[0]; cn[1] = ol[1]; cn[2] = ol[2]; auxv = dold; }
    const long long wx = (long long)cn[0] + c.pvt[0] - c.upvt[0];
    const long long wy = (long long)cn[1] + c.pvt[1] - c.upvt[1];
    const long long wz = (long long)cn[2] + c.pvt[2] - c.upvt[2];
    int flag_tile = 1;                       /* conservative: the kept (stale) parent may point anywhere */
    if (!(wx >= 0 && wx < c.wr[0] && wy >= 0 && wy < c.wr[1] && wz >= 0 && wz < c.wr[2])) {
        pr = gie_pair_make(c.empty_value, gie_pair_par(pr));       /* parent id left as is */
        auxv = c.empty_value;
        if (gie_pair_par(pr) == GIE_PAR_NONE) flag_tile = 0;        /* NONE is outside every wave range */
    } else {
        pr = gie_pair_make(auxv, gie_pack_wr((int)wx, (int)wy, (int)wz));
        flag_tile = !gie_in_loc(c, cn[0], cn[1], cn[2]);
    }
    if (flag_tile) c.tflag[gie_tile_index(c, x, y, z)] = 1;         /* all writers store 1 */
    {
        uint8_t *const u = &c.ucol[gie_ucol_index(c, x, y, z)];
        const uint8_t ub = *u;
        gie_edt_before_keep(c, id, pr, (ub >> (z & 7)) & 1u);
        if ((ub >> (z & 7)) & 1u) *u = (uint8_t)(ub & ~(1u << (z & 7)));     /* (a column's eight voxels belong to one thread) */
    }
    c.pair[id] = pr;
}
GIE_DEV void gie_mark_voxel(const gie_ctx &c, int x, int y, int z)
{
    const int id = gie_lid(c, x, y, z);
    if (c.glb_type[id] == GIE_VOX_UNKNOWN) return;
    gie_mark_st s;
    gie_mark_load1(c, id, x, y, z, s);
    gie_mark_load2(c, s);
    gie_mark_finish(c, id, x, y, z, s);
}

/* ================================================================== obtainFrontiers */
/* obtainFrontiers, unify_helper.cuh:275-446.  Returns a bit mask of what this voxel did so
 * that the kernel can compact the C-queue append with a wave ballot: bit0 = push to C. */
/* obtainFrontiers' branch for a 6-neighbour outside the local volume (unify_helper.cuh:346-438).
 * Returns bit0 = the voxel became a C seed (*seed set), bit1 = the neighbour is unknown. */
/* *push = 1: the neighbour joins frontier B, 2: frontier A (appended by the caller, wave-aggregated), *pa = its address */
/* the record of the outside neighbour in direction k, fetched ahead by the caller (k_frontier_faces: the neighbour across the
 * patch's own face is known before anything of the voxel has been read — its table lookup and record go out with the voxel's own
 * loads instead of behind them); nobody else touches that record during the launch (an outside voxel next to a face has ONE
 * neighbour inside the volume) */
struct gie_out_pre { int k; gie_vaddr a; int8_t nty; uint64_t ncoc; };
GIE_DEV_COLD int gie_frontier_outside(const gie_ctx &c, int x, int y, int z, int nx, int ny, int nz,
                                      const int cl[3], const int cw[3], int cd, uint64_t *seed, int *push, gie_vaddr *pa,
                                      int k = -1, const gie_out_pre *pre = nullptr)
{
    *push = 0; *pa = -1;
    int cur_in_q = 0;
    const int ng[3] = { nx + c.pvt[0], ny + c.pvt[1], nz + c.pvt[2] };
    const bool have = pre != nullptr && pre->k == k;
    const gie_vaddr a = have ? pre->a : gie_gvox_tab(c, ng[0], ng[1], ng[2]);
    if (a < 0) return 2;
    /* the neighbour's record in one batch of loads (a face voxel is a chain of dependent round trips) */
    const int8_t nty = have ? pre->nty : c.g_type[a];
    const uint64_t ncoc = have ? pre->ncoc : c.g_coc[a];
    const int nd = gie_gdist(c, ncoc, ng[0], ng[1], ng[2]);
    if (nty == GIE_VOX_UNKNOWN) return 2;
    if (gie_invalid_dist(c, nd)) return 0;
    int ncx, ncy, ncz;
    gie_unpack_crd(ncoc, &ncx, &ncy, &ncz);
    if (gie_invalid_coc(ncx, ncy, ncz)) return 0;
    const int nw[3] = { ncx - c.upvt[0], ncy - c.upvt[1], ncz - c.upvt[2] };
    const int nl[3] = { ncx - c.pvt[0], ncy - c.pvt[1], ncz - c.pvt[2] };
    const int n_valid = gie_in_wr(c, nw[0], nw[1], nw[2]);
    const int n_local = gie_in_loc(c, nl[0], nl[1], nl[2]);
    /* tiling: a ghost (another tile's voxel, refreshed this update) vouches for its obstacle wherever it lies outside this tile; a
     * remembered voxel outside the WHOLE volume only for an obstacle outside the whole volume too — one inside it belongs to
     * another tile and may have vanished since (only the owner knows).  Without tiling whole = local: the reference's test. */
    const int n_hidden = gie_in_whole(c, nx, ny, nz) ? !n_local : !gie_in_whole(c, nl[0], nl[1], nl[2]);
    if (n_hidden && n_valid) {
        const int d = gie_d2(nl[0], nl[1], nl[2], x, y, z);
        if (d < cd) {
            *seed = gie_pair_make(d, gie_pack_wr(nw[0], nw[1], nw[2]));
            cur_in_q = 1;
        }
    }
    /* tiling: a neighbour inside the WHOLE volume is another tile's voxel, known here only as a ghost layer that the owner
     * refreshes every map update — it can seed wave C (above) but is never raised or lowered from here, and waves A / B do
     * not walk into other tiles' territory (this tile's copies of it are stale).  Without tiling whole = local: no effect. */
    if (c.fast_mode || gie_in_whole(c, nx, ny, nz)) return cur_in_q;
    const int c2n = gie_d2(nx, ny, nz, cl[0], cl[1], cl[2]);
    if (c2n < nd) {                                       /* lower out → frontier B */
        c.g_wl[a] = 1;
        c.g_pair[a] = gie_pair_make(c2n, gie_pack_wr(cw[0], cw[1], cw[2]));
        *push = 1; *pa = a;
    } else if (c2n > nd && n_local) {                     /* raise out → frontier A */
        /* the reference reads the live _glb_type here; FNT never aliases OCCUPIED */
        if (c.glb_type[gie_lid(c, nl[0], nl[1], nl[2])] != GIE_VOX_OCCUPIED) {
            c.g_coc[a] = gie_pack_crd(cl[0] + c.pvt[0], cl[1] + c.pvt[1], cl[2] + c.pvt[2]);     /* (its distance to the neighbour is c2n) */
            gie_touch(c, a);
            c.g_wl[a] = -c.map_ct;
            c.g_pair[a] = gie_pair_make(c2n, gie_pack_wr(cw[0], cw[1], cw[2]));
            *push = 2; *pa = a;
        }
    }
    return cur_in_q;
}

struct gie_frontier_st { uint64_t p0; uint32_t tys; uint32_t tfm; };   /* tys: own type | six neighbour types << 4 (k + 1); tfm: bit k = tile flag of neighbour k */

/* every read that does not depend on another one is issued together — own pair + type, six
 * neighbour types, six tile flags: one memory round trip per voxel (the sweep is latency-bound) */
/* PLANE: the voxel lies in a tile that touches a face of the whole volume — such a tile is never cleared, so never lazy: its pair is in
 * the plane, no flag to ask first (k_frontier_faces: one dependent round trip less per face voxel) */
template <bool PLANE = false>
GIE_DEV void gie_frontier_load1(const gie_ctx &c, int id, int x, int y, int z, gie_frontier_st &s)
{
    int8_t ty[7];
    uint8_t tf[6];
    ty[0] = c.glb_type[id];
    s.p0 = PLANE ? c.pair[id] : gie_pair_get(c, id, x, y, z);      /* Mark-time value: this kernel never writes `pair` (seeds go to cand[1]) */
    const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
    GIE_UNROLL6
    for (int k = 0; k < 6; k++) {
        const int nx = x + dx[k], ny = y + dy[k], nz = z + dz[k];
        const int inl = gie_in_loc(c, nx, ny, nz);
        ty[k + 1] = c.glb_type[inl ? gie_lid(c, nx, ny, nz) : id];
        tf[k] = c.tflag[inl ? gie_tile_index(c, nx, ny, nz) : 0];
    }
    s.tys = (uint32_t)(ty[0] & 15); s.tfm = 0;
    GIE_UNROLL6
    for (int k = 0; k < 6; k++) { s.tys |= (uint32_t)(ty[k + 1] & 15) << (4 * (k + 1)); s.tfm |= (tf[k] ? 1u : 0u) << k; }
}

/* `nb(k, nid)` = the Mark-time pair of the in-volume neighbour k (local index nid): from memory (gie_nbpair_mem), or from
 * the tile a wave has staged in LDS (k_frontier_tiles) */
struct gie_nbpair_mem { const gie_ctx *c; GIE_DEV_MEMBER uint64_t operator()(int, int nid) const { return gie_pair_get_id(*c, nid); } };
/* ... of a voxel ON a face of the whole volume of an untiled mapper: its neighbours lie in tiles that touch the face too — never lazy */
struct gie_nbpair_plane { const uint64_t *pair; GIE_DEV_MEMBER uint64_t operator()(int, int nid) const { return pair[nid]; } };
/* `sink.ab(c, push, crd, a)` = the outside neighbour at global coordinate crd (address a) joins frontier B (push == 1) or
 * frontier A (push == 2); called by every executing lane for every direction: straight into the queues with one atomic per
 * wave and call (gie_absink_queues), or collected per tile in LDS (k_frontier_tiles) */
struct gie_absink_queues {
    static constexpr bool outside = true;                  /* false: the caller only hands over voxels off the faces (no outside neighbour: that branch is not compiled) */
    GIE_DEV_MEMBER void ab(const gie_ctx &c, int push, uint64_t crd, gie_vaddr a) const {
        gie_push64a_wave(c, c.qb, c.qb_a, &c.cnt[GIE_CNT_B], c.qcap_ab, push == 1, crd, a);
        gie_push64a_wave(c, c.qa, c.qa_a, &c.cnt[GIE_CNT_A], c.qcap_ab, push == 2, crd, a);
    } };
template <class NB, class SINK>
GIE_DEV int gie_frontier_finish_nb(const gie_ctx &c, int id, int x, int y, int z, const gie_frontier_st &s, const NB &nb, const SINK &sink, const gie_out_pre *pre = nullptr)
{
    const int8_t ty = (int8_t)(s.tys & 15u);
    if (ty == GIE_VOX_UNKNOWN) return 0;
    const uint64_t p0 = s.p0;
    int cw[3];
    gie_unpack_wr(gie_pair_par(p0), &cw[0], &cw[1], &cw[2]);
    const int cl[3] = { cw[0] + c.upvt[0] - c.pvt[0], cw[1] + c.upvt[1] - c.pvt[1], cw[2] + c.upvt[2] - c.pvt[2] };
    const int cd = gie_pair_dist(p0);
    if (!gie_in_loc(c, cl[0], cl[1], cl[2])) return 0;
    int cur_in_q = 0, has_unknown = 0;
    uint64_t seed = 0;
    const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
    GIE_UNROLL6
    for (int k = 0; k < 6; k++) {
        const int nx = x + dx[k], ny = y + dy[k], nz = z + dz[k];
        if (gie_in_loc(c, nx, ny, nz)) {
            const int nid = gie_lid(c, nx, ny, nz);
            const int8_t nty = (int8_t)((s.tys >> (4 * (k + 1))) & 15u);
            if (nty == GIE_VOX_UNKNOWN) { has_unknown = 1; continue; }
            /* only a neighbour whose closest obstacle is outside the volume can seed C; Mark
             * flagged the 8x8x8 tiles that contain one */
            if (!((s.tfm >> k) & 1u)) continue;
            int nw[3];
            gie_unpack_wr(gie_pair_par(nb(k, nid)), &nw[0], &nw[1], &nw[2]);
            const int nl[3] = { nw[0] + c.upvt[0] - c.pvt[0], nw[1] + c.upvt[1] - c.pvt[1], nw[2] + c.upvt[2] - c.pvt[2] };
            if (!gie_in_loc(c, nl[0], nl[1], nl[2]) && gie_in_wr(c, nw[0], nw[1], nw[2])) {
                const int d = gie_d2(nl[0], nl[1], nl[2], x, y, z);
                if (d < cd) {
                    seed = gie_pair_make(d, gie_pack_wr(nw[0], nw[1], nw[2]));
                    cur_in_q = 1;
                }
            }
        } else if (SINK::outside) {
            /* a neighbour outside the volume (only voxels on the six faces get here): kept out of
             * line so that the hot interior path stays small */
            int opush = 0;
            gie_vaddr oaddr = -1;
            const int r = gie_frontier_outside(c, x, y, z, nx, ny, nz, cl, cw, cd, &seed, &opush, &oaddr, k, pre);
            cur_in_q |= r & 1; has_unknown |= (r >> 1) & 1;
            /* a neighbour outside the volume that seeds wave B / wave A (the sinks only look at executing lanes) */
            sink.ab(c, opush, gie_pack_crd(nx + c.pvt[0], ny + c.pvt[1], nz + c.pvt[2]), oaddr);
        }
    }
    if (cur_in_q) { c.wl[id] = GIE_WL_SEED(c); c.cand[1][id] = seed; }   /* the pair the seed enters wave C with */
    if (ty == GIE_VOX_FREE && has_unknown) {
        c.glb_type[id] = GIE_VOX_FNT;
        if (c.fused && cd != c.empty_value) {       /* UpdateHashBatch's FNT store (unify_helper.cuh:448-523); a pair wave C lowers from EMPTY brings its own */
            const gie_vaddr a = gie_gvox_tab(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
            if (a >= 0) c.g_type[a] = GIE_VOX_FNT;
        }
    }
    return cur_in_q;
}
GIE_DEV int gie_frontier_finish(const gie_ctx &c, int id, int x, int y, int z, const gie_frontier_st &s)
{
    const gie_nbpair_mem nb = { &c };
    return gie_frontier_finish_nb(c, id, x, y, z, s, nb, gie_absink_queues());
}
GIE_DEV int gie_frontier_voxel(const gie_ctx &c, int x, int y, int z)
{
    const int id = gie_lid(c, x, y, z);
    if (c.glb_type[id] == GIE_VOX_UNKNOWN) return 0;
    gie_frontier_st s;
    gie_frontier_load1(c, id, x, y, z, s);
    return gie_frontier_finish(c, id, x, y, z, s);
}

/* ================================================================== waves A / B */
/* raise_outside / lower_outside (wave_core.cuh:103-350) run in block rounds out of LDS: gie_wave_a_block / gie_wave_b_block in
 * gie_kernels.hip.h (the emulation states the same schedule sequentially: tests/emu/gie_emu.cpp be_wave_a / be_wave_b). */
/* batch EDT distance of ONE voxel straight from the pass-X planes: min over the planes with
 * obstacles of (in-plane distance)² + (z - plane)².  A few dozen reads — for rare lookups only. */
GIE_DEV int gie_batch_dist_direct(const gie_ctx &c, int x, int y, int z)
{
    const int K = *c.zcount;
    const size_t plane = (size_t)c.X * c.Y, o = (size_t)y * c.X + x;
    int best = c.max_width * c.max_width;
    /* eight planes per trip, their loads in flight together (round 6: one load per trip made a lookup K dependent round trips — a lidar
     * scene has obstacles in 60-100 planes, and a block-run of wave B at a face of the volume waited 30-50 us for its unknown
     * neighbours' distances); a repeated plane does not change a minimum */
    for (int j = 0; j < K; j += 8) {
        int zj[8]; uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) zj[u] = c.zlist[j + u < K ? j + u : K - 1];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = c.cxy2[(size_t)zj[u] * plane + o];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int dx = x - (int)(v[u] & 0xffffu), dy = y - (int)(v[u] >> 16), dz = z - zj[u];
            const int d = dx * dx + dy * dy + dz * dz;
            best = d < best ? d : best;
        }
    }
    return best;
}

/* UpdateHashBatch for one voxel whose final pair is `pr` (type FNT is handled by the callers).
 * AGENT: the stores go through agent-scope (write-through) accesses — wave C may commit the same
 * voxel again one BFS level later from a workgroup on another XCD, and the per-XCD L2s are not
 * coherent: two plain stores would reach memory in either order. */
/* DEFERRED dist_id_pair (fused mode only — this function's only callers): UpdateHashBatch also stores the pair in the
 * GlbVoxel (`C.pair = pair[v]`, unify_helper.cuh:448-523).  The stored copy of a voxel INSIDE the volume is never read — the
 * waves work on the local plane there, and Mark rewrites it next map update — it only matters once the voxel has left the
 * volume (wave B's atomic minimum compares a proposal with it, wave_core.cuh:300-330).  So the 8-byte store per voxel and map
 * update is left out here and made up for when the voxel leaves: the next gie_fuse writes the pairs of the voxels that
 * are in this map update's volume and not in the next one's (gie_pair_flush_voxel below; 8 of 512 layers on the C5 drive).
 * Bit 63 of the stored closest obstacle (spare: the packed coordinate has 63 bits, gie_unpack_crd masks it) says "the stored pair
 * is older than this record": set by this commit, cleared by whoever stores a pair for real.  It settles the one case the
 * local plane cannot — a voxel whose pair was EMPTY (not committed) when it left: flag set = its last commit happened
 * during this stay in the volume and the pair of that commit is (stored distance, stored closest obstacle). */
template <bool AGENT>
GIE_DEV void gie_commit_pair(const gie_ctx &c, int id, gie_vaddr a, uint64_t pr)
{
    const int d = gie_pair_dist(pr);
    if (d == c.empty_value) return;
    if (a < 0) return;
    int cw[3];
    gie_unpack_wr(gie_pair_par(pr), &cw[0], &cw[1], &cw[2]);
    const uint64_t ncoc = gie_pack_crd(cw[0] + c.upvt[0], cw[1] + c.upvt[1], cw[2] + c.upvt[2]) | GIE_COC_STALEPAIR;
    /* (tried in round 4: leaving the store out when the record read before already holds this value — a settled scene would write
     * nothing — makes the dense sweep 12 % SLOWER on the C5 workload, 0.82 -> 0.93 ms: the test splits every wavefront's stores) */
    if (AGENT) gie_st(&c.g_coc[a], ncoc); else c.g_coc[a] = ncoc;     /* (nontemporal stores: no gain measured) */
}
/* The voxels of the LAST fused map update's volume (pivot opvt, wave-range pivot oupvt, block table still that update's)
 * that the next volume (pivot c.pvt) no longer holds, enumerated as three slabs: item i -> old local coordinate. */
struct gie_flush_boxes { int opvt[3], oupvt[3], otb0[3]; int lo[3], hi[3]; int n0, n1, n2; int rehash; };   /* rehash: types and block table are no longer that update's (a fuse without a merge came in between): find the block through the hash */   /* [lo, hi) = old local range that stays, per axis */
GIE_DEV int gie_flush_skipax(int i, int lo, int hi) { return i < lo ? i : i + (hi - lo); }        /* i-th coordinate outside [lo, hi) */
GIE_DEV void gie_pair_flush_voxel(const gie_ctx &c, const gie_flush_boxes &b, int i)
{
    int x, y, z;
    const int w0 = b.hi[0] - b.lo[0], w1 = b.hi[1] - b.lo[1];
    if (i < b.n0) {                                   /* x leaves: every y, z */
        const int nx = c.X - w0;
        x = gie_flush_skipax(i % nx, b.lo[0], b.hi[0]); y = (i / nx) % c.Y; z = i / (nx * c.Y);
    } else if (i < b.n0 + b.n1) {                     /* x stays, y leaves */
        const int j = i - b.n0, ny = c.Y - w1;
        x = b.lo[0] + j % w0; y = gie_flush_skipax((j / w0) % ny, b.lo[1], b.hi[1]); z = j / (w0 * ny);
    } else {                                          /* x, y stay, z leaves */
        const int j = i - b.n0 - b.n1;
        x = b.lo[0] + j % w0; y = b.lo[1] + (j / w0) % w1; z = gie_flush_skipax(j / (w0 * w1), b.lo[2], b.hi[2]);
    }
    const int id = gie_lid(c, x, y, z);
    const int gx = x + b.opvt[0], gy = y + b.opvt[1], gz = z + b.opvt[2];
    int slot;
    if (!b.rehash) {
        if (c.glb_type[id] == GIE_VOX_UNKNOWN) return;    /* (still the last update's types: fuse has not run yet) */
        const int cell = (((gz >> 3) - b.otb0[2]) * c.tdim[1] + ((gy >> 3) - b.otb0[1])) * c.tdim[0] + ((gx >> 3) - b.otb0[0]);
        slot = c.blk_tab[cell];
    } else slot = gie_hash_find(c, gx >> 3, gy >> 3, gz >> 3);
    if (slot < 0) return;
    const gie_vaddr a = (gie_vaddr)slot * GIE_VBSZ + gie_vox_in_blk(gx, gy, gz);
    if (b.rehash && !(c.g_coc[a] & GIE_COC_STALEPAIR)) return;    /* never committed in this stay, or flushed by an earlier fuse (and waves A / B may have rewritten it since);
                                                                   * (a voxel of a cleared tile whose record was left to the pair plane carries no mark either: those are
                                                                   * brought up to date by the catch-up over the OLD tiles, which gie_fuse runs whenever this form of the
                                                                   * flush does — gie_coc_catchup_column stores the ones outside the new volume too) */
    const uint64_t pr = gie_pair_get(c, id, x, y, z);
    if (gie_pair_dist(pr) != c.empty_value) {         /* committed by that update: the pair it would have stored */
        int cw[3];
        gie_unpack_wr(gie_pair_par(pr), &cw[0], &cw[1], &cw[2]);
        c.g_pair[a] = pr;
        c.g_coc[a] = gie_pack_crd(cw[0] + b.oupvt[0], cw[1] + b.oupvt[1], cw[2] + b.oupvt[2]);
    } else {
        const uint64_t cc = c.g_coc[a];
        if (cc & GIE_COC_STALEPAIR) {                 /* not committed by that update, but by an earlier one of this stay */
            int ox, oy, oz;
            gie_unpack_crd(cc, &ox, &oy, &oz);
            c.g_pair[a] = gie_pair_make(gie_gdist(c, cc, gx, gy, gz), gie_pack_wr((ox - b.oupvt[0]) & 0x3fff, (oy - b.oupvt[1]) & 0x3fff, (oz - b.oupvt[2]) & 0x1fff));   /* (only the distance of a stored pair is ever compared) */
            c.g_coc[a] = cc & ~GIE_COC_STALEPAIR;
        }
    }
}
/* wave C merged `pr` into pair[id] (fused mode): commit it on the spot */
/* (type and block slot of the voxel are read by the caller together with everything else the entry needs: one round trip) */
GIE_DEV void gie_commit_merged(const gie_ctx &c, int id, int8_t ty, int slot, int x, int y, int z, uint64_t pr)
{
    if (ty == GIE_VOX_UNKNOWN) return;       /* lower_inside has no type test (wave_core.cuh:353-393), UpdateHashBatch has (unify_helper.cuh:459) */
    const gie_vaddr a = slot < 0 ? (gie_vaddr)-1 : (gie_vaddr)slot * GIE_VBSZ + gie_vox_in_blk(x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
    gie_commit_pair<true>(c, id, a, pr);
    if (a >= 0 && gie_pair_dist(pr) != c.empty_value && ty == GIE_VOX_FNT) gie_st(&c.g_type[a], (int8_t)GIE_VOX_FNT);
}

/* ================================================================== wave C (lower_inside) */
/* wave_core.cuh:353-393 in the canonical tile-round schedule: the kernel is gie_wave_c_tile (gie_kernels.hip.h) — a wave per
 * active 8x8x8 tile, BFS inside the tile out of LDS; the sequential statements are oracle/gie_oracle.c wave_c and
 * tests/emu/gie_emu.cpp be_wave_c.  What the three share is gie_commit_merged above. */

/* ================================================================== commit */
/* UpdateHashBatch, unify_helper.cuh:448-523 */
struct gie_commit_st { int8_t ty; uint64_t pr; gie_vaddr a; };

GIE_DEV void gie_commit_load1(const gie_ctx &c, int id, int x, int y, int z, gie_commit_st &s)
{
    s.ty = c.glb_type[id];
    s.pr = c.pair[id];
    s.a = gie_gvox_tab(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
}
GIE_DEV void gie_commit_finish(const gie_ctx &c, int id, const gie_commit_st &s)
{
    const int8_t ty = s.ty;
    if (ty == GIE_VOX_UNKNOWN) return;
    const uint64_t pr = s.pr;
    const int d = gie_pair_dist(pr);
    const gie_vaddr a = s.a;
    if (d == c.empty_value) {
        if (a >= 0) {
            /* nothing to commit -- but a FUSED update before this one may have left this record's stored pair out (bit 63 of the
             * stored obstacle, gie_commit_pair): the mapper changed to the reference's order in between (gie_stream_enable), and
             * nobody will flush it later */
            const uint64_t cc = c.g_coc[a];
            if (cc & GIE_COC_STALEPAIR) {
                int ox, oy, oz;
                gie_unpack_crd(cc, &ox, &oy, &oz);
                const int gx = id % c.X + c.pvt[0], gy = (id / c.X) % c.Y + c.pvt[1], gz = id / (c.X * c.Y) + c.pvt[2];
                c.g_pair[a] = gie_pair_make(gie_gdist(c, cc, gx, gy, gz), gie_pack_wr((ox - c.upvt[0]) & 0x3fff, (oy - c.upvt[1]) & 0x3fff, (oz - c.upvt[2]) & 0x1fff));
                c.g_coc[a] = cc & ~GIE_COC_STALEPAIR;
            }
        }
        return;
    }
    if (a < 0) return;
    int cw[3];
    gie_unpack_wr(gie_pair_par(pr), &cw[0], &cw[1], &cw[2]);
    const uint64_t ncoc = gie_pack_crd(cw[0] + c.upvt[0], cw[1] + c.upvt[1], cw[2] + c.upvt[2]);
    if (c.track && ((c.g_coc[a] & ~GIE_COC_STALEPAIR) != ncoc || (ty == GIE_VOX_FNT && c.g_type[a] != GIE_VOX_FNT))) gie_touch(c, a);   /* (same obstacle: same distance) */
    c.g_coc[a] = ncoc;
    c.g_pair[a] = pr;
    if (ty == GIE_VOX_FNT) c.g_type[a] = GIE_VOX_FNT;
}
GIE_DEV void gie_commit_voxel(const gie_ctx &c, int x, int y, int z)
{
    const int id = gie_lid(c, x, y, z);
    gie_commit_st s;
    gie_commit_load1(c, id, x, y, z, s);
    gie_commit_finish(c, id, s);
}

/* ================================================================== Mark + commit in one sweep */
/* MarkLimitedObserve (unify_helper.cuh:201-273) and UpdateHashBatch (:448-523) for the same voxel in
 * one pass.  Between the two the reference runs obtainFrontiers and the waves, which change the pair
 * of a few voxels only: wave C (the only writer of `pair` after Mark) commits every pair it merges on
 * the spot (gie_commit_merged), obtainFrontiers stores its own FNT flips in the global map, and what
 * this sweep commits is the Mark-time pair — the value UpdateHashBatch finds for every voxel the waves
 * never visit.  Waves A / B only touch global voxels OUTSIDE the volume, this sweep only the ones
 * inside, so the order of the two does not matter.  Saves one read + one write of `pair` and one
 * block-table lookup per voxel, and the previous frame's pair is read only in the one branch that
 * needs it.  Not used while the changed-block flags are on (gie_stream_enable): a pair that is
 * committed twice could flag a block the reference's single commit would not. */
struct gie_markc_st { uint32_t bc; gie_vaddr a; int dold; uint64_t ococ; int skipold; int g[3]; };

/* ---- which stored global records Mark has to read at all.  MarkLimitedObserve compares the batch distance with the voxel's
 * stored (distance, closest obstacle) and keeps the stored one when it is smaller AND its obstacle lies outside the (whole)
 * volume.  A stored obstacle is at most sqrt(stored distance) away from its voxel, so for a tile that lies deeper inside the
 * whole volume than the square root of the largest distance stored in it, no stored obstacle can be outside: the 12 bytes per
 * voxel need not be read (every voxel of the C5 volume more than ~15 voxels from the faces).  What is stored in the tile's voxels is what
 * the map update before this one committed there — the sweep below records, per tile, 1 + the largest committed distance
 * (wave C only lowers what it commits later; nothing else writes voxels inside the volume) — so the bound is only used when
 * that update was the previous one, ran fused, and covered every voxel of the tile (a voxel it did not commit may hold
 * anything: its tile's bound is "infinite"). */
#define GIE_TMAX_INF 0x7fffffff
/* returns 1 when the tile is NOT cleared but holds voxels whose records the update before left to its pair plane (its tiles flagged
 * 2 in tskip_prev; only asked for with c.catchup_fast): gie_fuse brings exactly those tiles' stored copies up to date */
GIE_DEV int gie_tile_oldskip(const gie_ctx &c, int t, int allow = 1)      /* allow = 0: no tile is cleared (an update without obstacles) */
{
    const int tc[3] = { t % c.tfd[0], (t / c.tfd[0]) % c.tfd[1], t / (c.tfd[0] * c.tfd[1]) };
    const int sz[3] = { c.X, c.Y, c.Z };
    int o0[3], o1[3];
    long long reach2 = 0x7fffffffffffll;              /* (smallest distance to a face of the whole volume)^2, conservative per axis */
    bool can = true;
    for (int a = 0; a < 3; a++) {
        const int v0 = tc[a] * 8, v1 = (v0 + 7 < sz[a] ? v0 + 7 : sz[a] - 1);
        o0[a] = (v0 + c.prev_shift[a]) >> 3; o1[a] = (v1 + c.prev_shift[a]) >> 3;      /* the previous update's tiles that hold these voxels */
        if (v0 + c.prev_shift[a] < 0 || v1 + c.prev_shift[a] >= sz[a]) can = false;   /* (partly) new in the volume */
        const long long m = (long long)(v0 - c.whole_lo[a] < c.whole_hi[a] - 1 - v1 ? v0 - c.whole_lo[a] : c.whole_hi[a] - 1 - v1);
        if (m < 0) can = false;
        else if (m * m < reach2) reach2 = m * m;
    }
    int v = 0;
    if (can && allow) {
        long long d = 0, dl = 0;
        for (int z = o0[2]; z <= o1[2] && can; z++) for (int y = o0[1]; y <= o1[1] && can; y++) for (int x = o0[0]; x <= o1[0]; x++) {
            const int ot = (z * c.tfd[1] + y) * c.tfd[0] + x;
            const int w = c.tmax_prev[ot];
            if (w <= 0 || w == GIE_TMAX_INF) { can = false; break; }
            if (w - 1 > d) d = w - 1;
            /* what the bound of a LAZY tile would be ("lazy pairs" below: the sweep bounds such a tile from a few samples, up to
             * (sqrt(d) + 5)^2): a tile only becomes one if it would also stay one — a tile of the layer where the exact bound clears
             * and the loose one does not would be lazy in one update, swept in the next, and every turn costs a catch-up of its
             * 512 records and 512 pairs written out (round 6: 17 K tiles caught up and 13 K written out per update of the headline) */
            long long wl = w - 1;
            if (!c.tlazy[ot] && !c.cnt[GIE_CNT_LAZY_EXACT]) { long long r = (long long)sqrtf((float)wl); while (r * r < wl) r++; wl = (r + 5) * (r + 5); }      /* (no margin while the lazy tiles' bounds are pass Z's exact ones) */
            if (wl > dl) dl = wl;
        }
        /* |obstacle - voxel| <= sqrt(d) < distance to the nearest face, on every axis.  2 = ... and the voxels' tiles of the update before
         * were such tiles too: only then does the sweep leave the tile's records to the pair plane ("deferred records").  A tile at the
         * rim of the cleared region flips from update to update (a lidar's flood waves move the bounds), and every flip back costs a
         * catch-up of its 512 records — more than the stores it saves (round 5, the projective lidar workload: 0.07 ms per update). */
        if (can && d < reach2) {
            v = 1;
            if (c.skip2_ok && (dl < reach2 || !c.lazy_ok)) {
                v = 2;
                for (int z = o0[2]; z <= o1[2]; z++) for (int y = o0[1]; y <= o1[1]; y++) for (int x = o0[0]; x <= o1[0]; x++)
                    if (!c.tskip_prev[(z * c.tfd[1] + y) * c.tfd[0] + x]) v = 1;
            }
            GIE_COUNT_TSKIP(c);
        }
    }
    c.tskip[t] = (uint8_t)v;
    if (v != 0 || !c.catchup_fast) return 0;
    /* a tile that straddles the old volume's face is looked at too (round 6): with a side that is no multiple of 8 its
     * voxels reach past the old face tile (never a cleared one) into the old tile before it, which may be flagged 2.  Old tile
     * coordinates are clamped to the old volume; gie_coc_catchup_newcolumn tests every old coordinate itself. */
    for (int a = 0; a < 3; a++) {
        if (o0[a] < 0) o0[a] = 0;
        if (o1[a] > c.tfd[a] - 1) o1[a] = c.tfd[a] - 1;
    }
    for (int z = o0[2]; z <= o1[2]; z++) for (int y = o0[1]; y <= o1[1]; y++) for (int x = o0[0]; x <= o1[0]; x++)
        if (c.tskip_prev[(z * c.tfd[1] + y) * c.tfd[0] + x] == 2) return 1;
    return 0;
}
/* column l of tile t of THIS update's volume (a tile gie_tile_oldskip returned 1 for): the records of its voxels that lay in a
 * tile flagged 2 of the update before, from that update's pair plane (indices by prev_shift, wave-range pivot pupvt) */
GIE_DEV void gie_coc_catchup_newcolumn(const gie_ctx &c, const int pupvt[3], int t, int l)
{
    const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
    const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
    if (x >= c.X || y >= c.Y) return;
    const int px = x + c.prev_shift[0], py = y + c.prev_shift[1];
    const int gx = x + c.pvt[0], gy = y + c.pvt[1];
    const int z0 = tz * 8, nz = c.Z - z0 < 8 ? c.Z - z0 : 8;
    /* every load of the column in one batch (flags, pairs, the column's two block slots), then the stores: the launch is a few
     * thousand tiles, bound by how many round trips a lane makes one after the other */
    const bool cin = px >= 0 && px < c.X && py >= 0 && py < c.Y;
    uint8_t fl[8]; uint64_t pr[8];
GIE_UNROLL
    for (int k = 0; k < 8; k++) {
        const int pz = z0 + k + c.prev_shift[2];
        const bool in = k < nz && cin && pz >= 0 && pz < c.Z;
        fl[k] = in ? c.tskip_prev[gie_tile_index(c, px, py, pz)] : (uint8_t)0;
        pr[k] = in ? gie_pair_get(c, gie_lid(c, px, py, pz), px, py, pz) : gie_pair_make(c.empty_value, GIE_PAR_NONE);
    }
    const int gz0 = z0 + c.pvt[2];
    const int slot_lo = c.blk_tab[gie_tab_index(c, gx, gy, gz0)], slot_hi = c.blk_tab[gie_tab_index(c, gx, gy, gz0 + nz - 1)];
GIE_UNROLL
    for (int k = 0; k < 8; k++) {
        if (fl[k] != 2 || gie_pair_dist(pr[k]) == c.empty_value) continue;   /* (EMPTY cannot happen: an update without obstacles clears no tile) */
        const int gz = gz0 + k;
        const int slot = ((gz >> 3) == (gz0 >> 3)) ? slot_lo : slot_hi;
        if (slot < 0) continue;
        const gie_vaddr a = (gie_vaddr)slot * GIE_VBSZ + gie_vox_in_blk(gx, gy, gz);
        int cw[3];
        gie_unpack_wr(gie_pair_par(pr[k]), &cw[0], &cw[1], &cw[2]);
        c.g_pair[a] = pr[k];
        c.g_coc[a] = gie_pack_crd(cw[0] + pupvt[0], cw[1] + pupvt[1], cw[2] + pupvt[2]);
    }
}
/* the column's share of its tile's bound (known != valid: a voxel of the column was not committed) */
GIE_DEV void gie_markc_column(const gie_ctx &c, int x, int y, int z0, unsigned known, unsigned valid, int vmax)
{
    const int t = gie_tile_index(c, x, y, z0);
    int v = (known != valid) ? GIE_TMAX_INF : vmax;
    /* the eight lanes of a wave that share (lane >> 3) hold one row of one tile in every form of the sweep (inactive ones are the
     * row's upper end, past the volume); lane ^ 32 is the same tile two (sweep, 32 lanes along x) or four (list) rows on */
    const int lane = __lane_id();
    const unsigned long long ex = __ballot(1);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { const int ov = __shfl_xor(v, o); if ((ex >> (lane ^ o)) & 1ull) v = v > ov ? v : ov; }
    const int ov = __shfl_xor(v, 32), ot = __shfl_xor(t, 32);
    const bool paired = ((ex >> (lane ^ 32)) & 1ull) && ot == t;
    if (paired) v = v > ov ? v : ov;
    if ((lane & 7) == 0 && !(paired && lane >= 32) && v > 0) __hip_atomic_fetch_max(&c.tmax[t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

GIE_DEV void gie_markc_load1(const gie_ctx &c, int id, int x, int y, int z, gie_markc_st &s)
{
    s.bc = c.bcoc[id];
    s.g[0] = x + c.pvt[0]; s.g[1] = y + c.pvt[1]; s.g[2] = z + c.pvt[2];
    s.a = gie_gvox_tab(c, s.g[0], s.g[1], s.g[2]);
    s.skipold = c.tskip[gie_tile_index(c, x, y, z)];
}
GIE_DEV void gie_markc_load2(const gie_ctx &c, gie_markc_st &s)
{
    if (s.a < 0) return;
    if (s.skipold) { s.dold = GIE_TMAX_INF; s.ococ = 0; return; }     /* no stored record can win (gie_tile_oldskip): never "dn > dold" */
    s.ococ = c.g_coc[s.a];
    s.dold = gie_gdist(c, s.ococ, s.g[0], s.g[1], s.g[2]);
}
/* MarkLimitedObserve for one voxel on values in registers: batch closest obstacle `bc`, the stored global
 * (dold, ococ); last frame's pair is read through `old_pair` only in the one branch that needs it.
 * Returns the new pair; *flag_tile = the voxel's closest obstacle lies (or may lie) outside the volume.
 * (The same decisions as gie_mark_finish, in 32-bit arithmetic: every coordinate is within +-2^21.) */
GIE_DEV uint64_t gie_mark_logic(const gie_ctx &c, int x, int y, int z, uint32_t bc, int dold, uint64_t ococ, const uint64_t *old_pair, int *flag_tile)
{
    int cn0, cn1, cn2;
    const int batch_invalid = (bc == GIE_BCOC_NONE);
    int auxv;
    if (batch_invalid) { auxv = c.empty_value; cn0 = cn2 = 0; cn1 = 16383; }
    else {
        cn0 = (int)(bc & 1023u); cn1 = (int)((bc >> 10) & 1023u); cn2 = (int)(bc >> 20);
        const int dx = x - cn0, dy = y - cn1, dz = z - cn2;
        auxv = dx * dx + dy * dy + dz * dz;
    }
    const int dn = batch_invalid ? c.max_width * c.max_width : auxv;
    if (dn > dold) {                        /* limited observation: the old closest obstacle is out of sight */
        int ox, oy, oz;
        gie_unpack_crd(ococ, &ox, &oy, &oz);
        const int o0 = ox - c.pvt[0], o1 = oy - c.pvt[1], o2 = oz - c.pvt[2];
        if (!gie_in_whole(c, o0, o1, o2)) { cn0 = o0; cn1 = o1; cn2 = o2; auxv = dold; }
    }
    const int wx = cn0 + c.pvt[0] - c.upvt[0], wy = cn1 + c.pvt[1] - c.upvt[1], wz = cn2 + c.pvt[2] - c.upvt[2];
    if (!gie_in_wr(c, wx, wy, wz)) {
        /* outside the wave range: distance EMPTY, "parent id left as is" — the only reader of last frame's pair */
        const uint64_t oldpar = batch_invalid ? GIE_PAR_NONE : gie_pair_par(*old_pair);
        *flag_tile = (oldpar != GIE_PAR_NONE);
        return gie_pair_make(c.empty_value, oldpar);
    }
    *flag_tile = !gie_in_loc(c, cn0, cn1, cn2);
    return gie_pair_make(auxv, gie_pack_wr(wx, wy, wz));
}
/* returns the voxel's share of its tile's bound: 1 + the committed distance, GIE_TMAX_INF when nothing was committed */
GIE_DEV int gie_markc_finish(const gie_ctx &c, int id, int x, int y, int z, const gie_markc_st &s)
{
    if (s.a < 0) { gie_commit_pair<false>(c, id, -1, c.pair[id]); return GIE_TMAX_INF; }   /* (a known voxel always has its block) */
    int flag_tile;
    const uint64_t pr = gie_mark_logic(c, x, y, z, s.bc, s.dold, s.ococ, &c.pair[id], &flag_tile);
    if (flag_tile) c.tflag[gie_tile_index(c, x, y, z)] = 1;
    {
        uint8_t *const u = &c.ucol[gie_ucol_index(c, x, y, z)];
        const uint8_t ub = *u;
        gie_edt_before_keep(c, id, pr, (ub >> (z & 7)) & 1u);
        if ((ub >> (z & 7)) & 1u) *u = (uint8_t)(ub & ~(1u << (z & 7)));
    }
    if (c.coc_defer && s.skipold == 2 && c.lazy_ok) c.tlazy[gie_tile_index(c, x, y, z)] = 1;      /* "lazy pairs": pr is gie_pair_of_bcoc(s.bc) here, and is not stored */
    else c.pair[id] = pr;
    if (!(c.coc_defer && s.skipold == 2)) gie_commit_pair<false>(c, id, s.a, pr);       /* (a tskip tile of the second update running: the pair plane is the record — "deferred records" below) */
    const int d = gie_pair_dist(pr);
    return d == c.empty_value ? GIE_TMAX_INF : d + 1;
}
GIE_DEV int gie_markc_voxel(const gie_ctx &c, int x, int y, int z)
{
    const int id = gie_lid(c, x, y, z);
    if (c.glb_type[id] == GIE_VOX_UNKNOWN) return 0;
    gie_markc_st s;
    gie_markc_load1(c, id, x, y, z, s);
    gie_markc_load2(c, s);
    return gie_markc_finish(c, id, x, y, z, s);
}
/* ================================================================== deferred records (fused Mark + commit)
 * UpdateHashBatch stores (dist², closest obstacle, pair) in the GlbVoxel of every observed voxel, every update
 * (unify_helper.cuh:448-523).  For a voxel INSIDE the volume the stored copy is only ever read
 *   (a) by the next update's MarkLimitedObserve — and not at all in the tiles gie_tile_oldskip clears (tskip: deeper inside the
 *       volume than any stored distance reaches, 84 % of the C5 volume),
 *   (b) by readers of single voxels (gie_query_global, the changed-block gather, halo export of a face),
 *   (c) once the voxel has left the volume (waves A / B, obtainFrontiers' outside branch).
 * So in a tskip tile the fused sweep writes the pair plane only (8 of its 16 written bytes per voxel; rounds 3-4 had already
 * deferred the stored PAIR this way).  The record of such a voxel IS its pair-plane entry — every voxel of a tskip tile was
 * committed by the update before (gie_tile_oldskip's condition), and an update that cannot commit (no obstacle in the volume:
 * every pair EMPTY) clears no tile — until one of these brings the stored copy up to date:
 *   - gie_pair_flush_voxel when the voxel leaves the volume (as for the pair);
 *   - gie_coc_catchup_column, in the next gie_fuse, when its tile is not a tskip tile any more (the robot came closer to it; a
 *     fuse without a merge in between, which trusts no bound; an update without obstacles);
 *   - wave C, which commits what it merges on the spot.
 * (b) looks into the pair plane for a voxel of a tskip tile (gie_deferred_coc: gie_query_voxel, and gie_halo_record — a face of a
 * TILE can lie deep inside the whole volume); the reference's order of kernels (changed-block flags on) does not defer. */
struct gie_catchup { const uint8_t *flags; int fpvt[3]; int ppvt[3], pupvt[3]; int all; };   /* flags: the tskip plane that marks the deferred tiles, at pivot fpvt; all: none of them stays deferred */
/* does tile t (of the flags' plane) hold a voxel whose record has to be stored now?  Not when every tile of THIS update's volume
 * that its voxels lie in is a tskip tile again (the usual case: one comparison per old tile, a handful of byte loads) */
GIE_DEV int gie_coc_catchup_tile(const gie_ctx &c, const gie_catchup &p, int t)
{
    if (p.all ? p.flags[t] == 0 : p.flags[t] != 2) return 0;   /* (1: the sweep stored the tile's records itself — in the update the flags are from; with `all`
                                                                 * the flags are THIS update's, whose sweep may not have run yet: a tile flagged 1 may hold voxels the
                                                                 * update before left to its pair plane) */
    if (p.all) return 1;
    const int tc[3] = { t % c.tfd[0], (t / c.tfd[0]) % c.tfd[1], t / (c.tfd[0] * c.tfd[1]) };
    const int sz[3] = { c.X, c.Y, c.Z };
    int n0[3], n1[3];
    for (int a = 0; a < 3; a++) {
        const int v0 = tc[a] * 8 + p.fpvt[a] - c.pvt[a], v1 = (tc[a] * 8 + 7 < sz[a] ? tc[a] * 8 + 7 : sz[a] - 1) + p.fpvt[a] - c.pvt[a];   /* this update's local coordinates */
        if (v0 < 0 || v1 >= sz[a]) return 1;              /* (partly) outside this update's volume */
        n0[a] = v0 >> 3; n1[a] = v1 >> 3;
    }
    for (int z = n0[2]; z <= n1[2]; z++) for (int y = n0[1]; y <= n1[1]; y++) for (int x = n0[0]; x <= n1[0]; x++)
        if (!c.tskip[(z * c.tfd[1] + y) * c.tfd[0] + x]) return 1;
    return 0;
}
/* column l (0..63) of tile t: the records of the column's voxels, if they do not stay deferred */
GIE_DEV void gie_coc_catchup_column(const gie_ctx &c, const gie_catchup &p, int t, int l)
{
    const int tx = t % c.tfd[0], ty = (t / c.tfd[0]) % c.tfd[1], tz = t / (c.tfd[0] * c.tfd[1]);
    const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
    if (x >= c.X || y >= c.Y) return;
    const int gx = x + p.fpvt[0], gy = y + p.fpvt[1];
    int slot = -1, slot_bz = 0x7fffffff;
    for (int k = 0; k < 8; k++) {
        const int z = tz * 8 + k;
        if (z >= c.Z) break;
        const int gz = z + p.fpvt[2];
        bool in_new = false;                               /* inside THIS update's volume (whose block table is built: gie_fuse's catch-up) */
        if (!p.all) {                                      /* the voxel's tile of THIS update is a tskip tile again: the new pair plane takes over */
            const int nx = gx - c.pvt[0], ny = gy - c.pvt[1], nz = gz - c.pvt[2];
            in_new = gie_in_loc(c, nx, ny, nz);
            if (in_new && c.tskip[gie_tile_index(c, nx, ny, nz)]) continue;
        }
        const int px = gx - p.ppvt[0], py = gy - p.ppvt[1], pz = gz - p.ppvt[2];
        if (!gie_in_loc(c, px, py, pz)) continue;          /* (cannot happen: a tskip tile lies inside the volume of the merge before) */
        const uint64_t pr = gie_pair_get(c, gie_lid(c, px, py, pz), px, py, pz);
        if (gie_pair_dist(pr) == c.empty_value) continue;  /* (cannot happen either: such an update clears no tile) */
        if ((gz >> 3) != slot_bz) {                        /* (one coalesced table read instead of a chain of hash probes where the table covers the voxel) */
            slot_bz = gz >> 3;
            slot = in_new ? c.blk_tab[gie_tab_index(c, gx, gy, gz)] : gie_hash_find(c, gx >> 3, gy >> 3, gz >> 3);
        }
        if (slot < 0) continue;
        const gie_vaddr a = (gie_vaddr)slot * GIE_VBSZ + gie_vox_in_blk(gx, gy, gz);
        int cw[3];
        gie_unpack_wr(gie_pair_par(pr), &cw[0], &cw[1], &cw[2]);
        c.g_pair[a] = pr;
        c.g_coc[a] = gie_pack_crd(cw[0] + p.pupvt[0], cw[1] + p.pupvt[1], cw[2] + p.pupvt[2]);
    }
}

/* the stored closest obstacle of global voxel g as a reader of single voxels must see it: for a voxel of a tskip tile whose record
 * was left to the pair plane, what the commit would have stored (true, *cc); false: the stored copy is the record */
GIE_DEV bool gie_deferred_coc(const gie_ctx &c, int gx, int gy, int gz, uint64_t *cc)
{
    if (!c.qdefer) return false;
    const int lx = gx - c.ts_pvt[0], ly = gy - c.ts_pvt[1], lz = gz - c.ts_pvt[2];
    if (!gie_in_loc(c, lx, ly, lz) || !c.tskip[gie_tile_index(c, lx, ly, lz)]) return false;
    const int px = gx - c.pp_pvt[0], py = gy - c.pp_pvt[1], pz = gz - c.pp_pvt[2];
    if (!gie_in_loc(c, px, py, pz)) return false;
    const uint64_t pr = gie_pair_get(c, gie_lid(c, px, py, pz), px, py, pz);
    if (gie_pair_dist(pr) == c.empty_value) return false;
    int cw[3];
    gie_unpack_wr(gie_pair_par(pr), &cw[0], &cw[1], &cw[2]);
    *cc = gie_pack_crd(cw[0] + c.pp_upvt[0], cw[1] + c.pp_upvt[1], cw[2] + c.pp_upvt[2]);
    return true;
}

/* ================================================================== halo exchange between tiles */
/* face f: axis f/2, side f%2.  Layer index i ↔ the two remaining axes (a fastest). */
GIE_DEV void gie_face_coord(const gie_ctx &c, int face, int i, int depth_off, int *x, int *y, int *z)
{
    const int axis = face >> 1, hi = face & 1;
    const int sz[3] = { c.X, c.Y, c.Z };
    const int along = hi ? sz[axis] - 1 + depth_off : -depth_off;   /* depth_off 0: own face layer, 1: just outside */
    if (axis == 0) { *x = along; *y = i % c.Y; *z = i / c.Y; }
    else if (axis == 1) { *x = i % c.X; *y = along; *z = i / c.X; }
    else { *x = i % c.X; *y = i / c.X; *z = along; }
}
GIE_HD int gie_face_count(const gie_ctx &c, int face)
{ const int axis = face >> 1; return axis == 0 ? c.Y * c.Z : (axis == 1 ? c.X * c.Z : c.X * c.Y); }

/* committed state of the voxels on one face of the local volume */
GIE_DEV gie_halo_voxel gie_halo_record(const gie_ctx &c, int face, int i)
{
    int x, y, z;
    gie_face_coord(c, face, i, 0, &x, &y, &z);
    gie_halo_voxel h;
    h.pad[0] = h.pad[1] = 0; h.occ_val = 0;
    const gie_vaddr a = gie_gvox_tab(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
    if (a < 0 || c.g_type[a] == GIE_VOX_UNKNOWN) {
        h.vox_type = GIE_VOX_UNKNOWN; h.dist_sq = c.empty_value; h.coc[0] = h.coc[1] = h.coc[2] = GIE_EMPTY_VALUE;
    } else {
        h.vox_type = c.g_type[a]; h.occ_val = c.g_occ[a];
        uint64_t cc = c.g_coc[a];
        (void)gie_deferred_coc(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2], &cc);      /* (a face of a TILE may lie in a tskip tile: deep inside the whole volume) */
        h.dist_sq = gie_gdist(c, cc, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
        gie_unpack_crd(cc, &h.coc[0], &h.coc[1], &h.coc[2]);
    }
    return h;
}
GIE_DEV void gie_halo_export_voxel(const gie_ctx &c, int face, int i, gie_halo_voxel *out) { out[i] = gie_halo_record(c, face, i); }
/* the sparse form of a layer: the KNOWN voxels only, as (index in the layer, record), in no particular order (one counter update
 * per wave).  An unknown ghost changes nothing on import, so the two forms are interchangeable. */
GIE_DEV void gie_halo_export_sparse_voxel(const gie_ctx &c, int face, int i, gie_halo_entry *out, int32_t *count)
{
    const gie_halo_voxel h = gie_halo_record(c, face, i);
    const bool known = h.vox_type != GIE_VOX_UNKNOWN;
    const unsigned long long m = __ballot(known);
    if (!m) return;
    const int lane = __lane_id(), leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = gie_aadd32(count, __popcll(m));
    base = __shfl(base, leader);
    if (known) { const int s = base + __popcll(m & ((1ull << lane) - 1ull)); out[s].index = i; out[s].v = h; }
}
/* ghost voxels just outside `face`: mark the blocks they need … */
GIE_DEV void gie_halo_need_rec(const gie_ctx &c, int face, int i, const gie_halo_voxel &v)
{
    if (v.vox_type == GIE_VOX_UNKNOWN) return;
    int x, y, z;
    gie_face_coord(c, face, i, 1, &x, &y, &z);
    c.blk_need[gie_tab_index(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2])] = 1;
}
GIE_DEV void gie_halo_need_voxel(const gie_ctx &c, int face, int i, const gie_halo_voxel *in) { gie_halo_need_rec(c, face, i, in[i]); }
/* … and store them (type / dist² / coc of the owning tile) */
GIE_DEV void gie_halo_import_rec(const gie_ctx &c, int face, int i, const gie_halo_voxel &v)
{
    if (v.vox_type == GIE_VOX_UNKNOWN) return;
    int x, y, z;
    gie_face_coord(c, face, i, 1, &x, &y, &z);
    const gie_vaddr a = gie_gvox_tab(c, x + c.pvt[0], y + c.pvt[1], z + c.pvt[2]);
    if (a < 0) return;
    c.g_type[a] = v.vox_type;
    c.g_occ[a] = v.occ_val;
    c.g_coc[a] = gie_pack_crd(v.coc[0], v.coc[1], v.coc[2]);      /* (v.dist_sq is this obstacle's distance: the owner's record is a witness) */
    gie_touch(c, a);
}
GIE_DEV void gie_halo_import_voxel(const gie_ctx &c, int face, int i, const gie_halo_voxel *in) { gie_halo_import_rec(c, face, i, in[i]); }
/* the sparse form: entry j of `*count` (an index outside the layer is ignored) */
GIE_DEV void gie_halo_need_entry(const gie_ctx &c, int face, int j, const gie_halo_entry *in, const int32_t *count, int nface)
{
    if (j >= *count) return;
    const gie_halo_entry e = in[j];
    if ((unsigned)e.index < (unsigned)nface) gie_halo_need_rec(c, face, e.index, e.v);
}
GIE_DEV void gie_halo_import_entry(const gie_ctx &c, int face, int j, const gie_halo_entry *in, const int32_t *count, int nface)
{
    if (j >= *count) return;
    const gie_halo_entry e = in[j];
    if ((unsigned)e.index < (unsigned)nface) gie_halo_import_rec(c, face, e.index, e.v);
}
/* obtainFrontiers' C-seed rule (unify_helper.cuh:365-399) for a face voxel against its ghost
 * neighbours, on the committed state: a ghost whose closest obstacle lies outside this tile and
 * is closer than the voxel's own makes the voxel a wave-C seed.  Voxel index v = local id. */
GIE_DEV int gie_refine_voxel(const gie_ctx &c, int id)
{
    const int x = id % c.X, y = (id / c.X) % c.Y, z = id / (c.X * c.Y);
    if (c.glb_type[id] == GIE_VOX_UNKNOWN) return 0;
    const int cd = gie_pair_dist(c.pair[id]);
    uint64_t seed = 0;
    int hit = 0;
    const int dx[6] = { -1, 1, 0, 0, 0, 0 }, dy[6] = { 0, 0, -1, 1, 0, 0 }, dz[6] = { 0, 0, 0, 0, -1, 1 };
    GIE_UNROLL6
    for (int k = 0; k < 6; k++) {
        const int nx = x + dx[k], ny = y + dy[k], nz = z + dz[k];
        if (gie_in_loc(c, nx, ny, nz)) continue;
        const gie_vaddr a = gie_gvox_tab(c, nx + c.pvt[0], ny + c.pvt[1], nz + c.pvt[2]);
        if (a < 0 || c.g_type[a] == GIE_VOX_UNKNOWN) continue;
        if (gie_invalid_dist(c, gie_gdist(c, c.g_coc[a], nx + c.pvt[0], ny + c.pvt[1], nz + c.pvt[2]))) continue;
        int ncx, ncy, ncz;
        gie_unpack_crd(c.g_coc[a], &ncx, &ncy, &ncz);
        if (gie_invalid_coc(ncx, ncy, ncz)) continue;
        const int nw[3] = { ncx - c.upvt[0], ncy - c.upvt[1], ncz - c.upvt[2] };
        const int nl[3] = { ncx - c.pvt[0], ncy - c.pvt[1], ncz - c.pvt[2] };
        /* a ghost vouches for an obstacle anywhere outside this tile, a remembered voxel outside the whole volume only for one
         * outside the whole volume (gie_frontier_outside) */
        const int hidden = gie_in_whole(c, nx, ny, nz) ? !gie_in_loc(c, nl[0], nl[1], nl[2]) : !gie_in_whole(c, nl[0], nl[1], nl[2]);
        if (!hidden || !gie_in_wr(c, nw[0], nw[1], nw[2])) continue;
        const int d = gie_d2(nl[0], nl[1], nl[2], x, y, z);
        if (d < cd) { seed = gie_pair_make(d, gie_pack_wr(nw[0], nw[1], nw[2])); hit = 1; }
    }
    if (hit) c.cand[1][id] = seed;
    return hit;
}
/* boundary voxel enumeration: j in [0, 2(XY+YZ+XZ)) → face voxel, each voxel taken once (at its
 * first face in the order -x,+x,-y,+y,-z,+z) */
GIE_DEV int gie_refine_entry(const gie_ctx &c, int j)
{
    int face = 0, i = j;
    for (; face < 6; face++) { const int n = gie_face_count(c, face); if (i < n) break; i -= n; }
    if (face >= 6) return -1;
    int x, y, z;
    gie_face_coord(c, face, i, 0, &x, &y, &z);
    /* skip voxels that an earlier face already covers */
    if (face >= 1 && x == 0) return -1;
    if (face >= 2 && x == c.X - 1) return -1;
    if (face >= 3 && y == 0) return -1;
    if (face >= 4 && y == c.Y - 1) return -1;
    if (face >= 5 && z == 0) return -1;
    return gie_lid(c, x, y, z);
}

/* ================================================================== export for the readers */
GIE_DEV void gie_export_pair(const gie_ctx &c, int id, int32_t *dist_sq, int32_t *coc_xyz)
{
    const uint64_t pr = gie_pair_get_id(c, id);
    const int d = gie_pair_dist(pr);
    if (dist_sq) dist_sq[id] = d;
    if (coc_xyz) {
        if (gie_pair_par(pr) == GIE_PAR_NONE || d >= c.empty_value) {
            coc_xyz[3 * id] = coc_xyz[3 * id + 1] = coc_xyz[3 * id + 2] = GIE_EMPTY_VALUE;
        } else {
            int w[3];
            gie_unpack_wr(gie_pair_par(pr), &w[0], &w[1], &w[2]);
            coc_xyz[3 * id] = w[0] + c.upvt[0]; coc_xyz[3 * id + 1] = w[1] + c.upvt[1]; coc_xyz[3 * id + 2] = w[2] + c.upvt[2];
        }
    }
}
GIE_DEV void gie_export_bcoc(const gie_ctx &c, int id, int32_t *dist_sq, int32_t *coc_xyz)
{
    const uint32_t bc = c.bcoc[id];
    if (dist_sq) dist_sq[id] = gie_bcoc_dist(bc, id % c.X, (id / c.X) % c.Y, id / (c.X * c.Y), c.max_width * c.max_width);
    if (!coc_xyz) return;
    if (bc == GIE_BCOC_NONE) { coc_xyz[3 * id] = coc_xyz[3 * id + 1] = coc_xyz[3 * id + 2] = -1; }
    else { coc_xyz[3 * id] = (int)(bc & 1023u); coc_xyz[3 * id + 1] = (int)((bc >> 10) & 1023u); coc_xyz[3 * id + 2] = (int)(bc >> 20); }
}
/* ---- changed-block streaming (streamPipeline / getUpdatedAddr / streamD2H, glb_hash_map.cu:209-247) */
/* slot list of the flagged blocks in slot order */
GIE_DEV void gie_stream_list(const gie_ctx &c, const int32_t *rank, int32_t *list, int slot)
{ if (c.g_dirty[slot]) list[rank[slot]] = slot; }
/* voxel j (reference in-block order x*64 + y*8 + z) of list entry first + i/512 → staging */
GIE_DEV void gie_stream_gather(const gie_ctx &c, const int32_t *list, int first, int32_t *keys, gie_voxel *out, int i)
{
    const int slot = list[first + (i >> 9)], j = i & 511;
    const gie_vaddr a = (gie_vaddr)slot * GIE_VBSZ + ((j >> 6) | (((j >> 3) & 7) << 3) | ((j & 7) << 6));
    gie_voxel v;
    v.occ_val = c.g_occ[a]; v.vox_type = c.g_type[a]; v.pad = 0;
    {   /* the voxel's global coordinate: block key * 8 + in-block position */
        int k[3];
        gie_unpack_crd(c.g_key[slot], &k[0], &k[1], &k[2]);
        const int ib = (int)(a & (GIE_VBSZ - 1));
        v.dist_sq = gie_gdist(c, c.g_coc[a], k[0] * 8 + (ib & 7), k[1] * 8 + ((ib >> 3) & 7), k[2] * 8 + (ib >> 6));
    }
    gie_unpack_crd(c.g_coc[a], &v.coc[0], &v.coc[1], &v.coc[2]);
    out[i] = v;
    if (j < 3) {
        int k[3];
        gie_unpack_crd(c.g_key[slot], &k[0], &k[1], &k[2]);
        keys[3 * (i >> 9) + j] = k[j];
    }
}
GIE_DEV void gie_stream_clear(const gie_ctx &c, const int32_t *list, int first, int i) { c.g_dirty[list[first + i]] = 0; }

GIE_DEV void gie_query_voxel(const gie_ctx &c, const int32_t *xyz, int i, gie_voxel *out)
{
    const gie_vaddr a = gie_gvox_hash(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    out[i].pad = 0;
    uint64_t dcc;
    if (a >= 0 && gie_deferred_coc(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &dcc)) {      /* a voxel of a tskip tile: its record is its pair-plane entry ("deferred records") */
        out[i].occ_val = c.g_occ[a]; out[i].vox_type = c.g_type[a]; out[i].dist_sq = gie_gdist(c, dcc, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        gie_unpack_crd(dcc, &out[i].coc[0], &out[i].coc[1], &out[i].coc[2]);
        return;
    }
    if (a < 0) {
        out[i].occ_val = 0; out[i].vox_type = GIE_VOX_UNKNOWN; out[i].dist_sq = c.empty_value;
        out[i].coc[0] = out[i].coc[1] = out[i].coc[2] = GIE_EMPTY_VALUE;
    } else {
        out[i].occ_val = c.g_occ[a]; out[i].vox_type = c.g_type[a]; out[i].dist_sq = gie_gdist(c, c.g_coc[a], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        gie_unpack_crd(c.g_coc[a], &out[i].coc[0], &out[i].coc[1], &out[i].coc[2]);
    }
}

#endif /* GIE_OPS_H */
