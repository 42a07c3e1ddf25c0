#define _GNU_SOURCE   /* qsort_r */
/*
 * gie_oracle.c — CPU restatement of the GIE-mapping per-frame map update.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  The product path (gie-mapping_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and cannot be built here (needs nvcc, CUDA runtime, thrust, ROS, PCL and the
 * binary-only cuTT blob; writing stand-ins for those is not a reference build).  This file is a
 * from-scratch restatement of the reference's algorithm, function by function, with the
 * file:line each part follows.  The batch-EDT stage is additionally pinned by the mathematical
 * definition (go_brute_force_edt below: exact Euclidean distance transform).
 *
 * Scalar, single-threaded C99.  Semantics that the reference leaves to thread scheduling
 * (CAS tie order, plain stores racing in the BFS waves) are fixed here to ONE canonical,
 * order-independent schedule — level-synchronous, every frontier entry reads the state at the
 * start of its level, conflicting writers are resolved by lexicographic (dist, parent-id) min —
 * which is a legal schedule of the reference wherever the reference is race-free, and which the
 * HIP path reproduces exactly (see DESIGN.md "Canonical wave schedule").
 *
 * Extension outside the reference's envelope: the reference packs closest-obstacle ids in
 * 11/11/10 bits and asserts X+Y+Z < 1022 (local_batch.h:12-17,51-58).  For larger volumes
 * ("wide" mode) the wave range becomes (16382,16382,8190) and the distance sentinel 4194303;
 * below that size every constant is the reference's.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include "../include/gie.h"
#include "oracle_math.h"      /* the oracle's own statement of the fp32 geometry: shares nothing with the product's include/gie_math.h */

#define VB 8
#define VBSZ 512
#define GRAY0 16677219 /* voxmap_utils.cuh:17-22 */
#define GRAY1 16677220
#define BLACK 16677223
#define EMPTY_KEY_C 999999 /* voxmap_utils.cuh:8-9 */
#define PAR_NONE ((int64_t)0x3ffffffffffLL) /* 42 one-bits; the reference's 0xffffffff "see nothing" id */

typedef struct { int x, y, z; } i3;

/* GlbVoxel, voxmap_utils.cuh:29-44 (+ scratch words for the canonical level schedule) */
typedef struct {
    uint8_t occ_val;
    int8_t vox_type;
    int32_t update_ct;
    int32_t coc[3];
    int32_t dist_sq;
    int32_t wave_layer;
    int32_t pair_dist;
    int64_t pair_par;
    int32_t prop_dist; /* best proposal of the running level (scratch) */
    int32_t blk;       /* slot of the owning block (for the changed-block flags) */
    int64_t prop_par;
    int32_t xprop_dist; /* best proposal from ANOTHER block in the running round (scratch of the block-round schedule) */
    int64_t xprop_par;
} ovox;

typedef struct { int32_t key[3]; int32_t dirty; ovox v[VBSZ]; } oblock;

typedef struct { i3 *d; int n, cap; } queue;

typedef struct gie_oracle {
    gie_config cfg;
    int X, Y, Z, N;
    int max_width, max_loc_dist_sq;
    int wr[3];          /* _wave_range */
    int inv_coc;        /* component of INVALID_LOC_COC (>= every wave range) */
    int empty_value;    /* EMPTY_VALUE (999999) or the wide sentinel */
    int invalid_dist_min; /* invalid_dist_glb threshold (900000) */
    int wide;
    int map_ct;
    om_pose L2G, G2L;
    float origin[3];
    int pvt[3], upvt[3];
    int tile_off[3], next_off[3], next_whole[3], whole_lo[3], whole_hi[3];
    float msg_origin[3];
    int pntcld_mode;    /* last OGM call was the ray-cast one */
    /* LocMap arrays, local_batch.h:541-561 */
    int32_t *ray_count;
    int8_t *inst_type, *glb_type;
    float *edt;
    int32_t *aux;            /* _aux: batch dist², then Mark-edited */
    int32_t *bcoc;           /* batch coc (local), 3 ints per voxel, -1 = none */
    int32_t *bdist;          /* copy of the batch dist² before Mark (for gie_read_batch_edt) */
    int32_t *pair_dist; int64_t *pair_par; /* _dist_id_pair */
    int32_t *g_bak; int64_t *coc_bak;      /* _g / _coc_idx after Mark ("read-only backups") */
    int32_t *wave_layer;     /* _loc_wave_layer */
    int32_t *lprop_dist; int64_t *lprop_par; /* scratch for the canonical level schedule */
    /* scratch for EDT */
    int32_t *g1, *cy1, *d2, *cx2, *cy2;
    /* global map */
    oblock **blocks; int nblocks, blocks_cap;
    int track;         /* changed-block flags on (display_glb_edt / display_glb_ogm) */
    int32_t *htab; int hcap; /* open addressing: index into blocks or -1 */
    /* ext boxes */
    int nbox; float *box_ll, *box_ur; uint8_t *box_act;
    gie_frame_stats st;
    long long tot_vis[3];
} gie_oracle;

/* ------------------------------------------------------------------ small helpers */
static int fdiv8(int a) { return a >> 3; }                 /* get_VB_key, voxmap_utils.cuh:94-101 */
static int vox_in_blk(int gx, int gy, int gz) { return ((gz & 7) << 6) | ((gy & 7) << 3) | (gx & 7); }
static int in_whole(const gie_oracle *o, int x, int y, int z);
static int in_loc(const gie_oracle *o, int x, int y, int z)   /* local_batch.h:113-126 */
{ return x >= 0 && x < o->X && y >= 0 && y < o->Y && z >= 0 && z < o->Z; }
static int in_wr(const gie_oracle *o, int x, int y, int z)    /* local_batch.h:144-157 */
{ return x >= 0 && x < o->wr[0] && y >= 0 && y < o->wr[1] && z >= 0 && z < o->wr[2]; }
static int lid(const gie_oracle *o, int x, int y, int z) { return (z * o->Y + y) * o->X + x; }
static int64_t pack_wr(int x, int y, int z) { return (int64_t)x | ((int64_t)y << 14) | ((int64_t)z << 28); }
static void unpack_wr(int64_t id, int *x, int *y, int *z)
{ *x = (int)(id & 0x3fff); *y = (int)((id >> 14) & 0x3fff); *z = (int)((id >> 28) & 0x3fff); }
static int d2i(int ax, int ay, int az, int bx, int by, int bz)  /* get_squred_dist, voxmap_utils.cuh:135-145 (no trap) */
{ long long dx = ax - bx, dy = ay - by, dz = az - bz; long long r = dx * dx + dy * dy + dz * dz; return r > 0x7fffffffLL ? 0x7fffffff : (int)r; }
static int invalid_dist_glb(const gie_oracle *o, int d) { return d < 0 || d >= o->invalid_dist_min; } /* :161-165 */
static int invalid_coc_glb(const int32_t c[3]) { return c[0] > 900000 || c[1] > 900000 || c[2] > 900000; } /* :167-172 */
static int pair_less(int d1, int64_t p1, int d2, int64_t p2) { return d1 < d2 || (d1 == d2 && p1 < p2); }

static void q_push(queue *q, int x, int y, int z)
{
    if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 1024; q->d = (i3 *)realloc(q->d, sizeof(i3) * (size_t)q->cap); }
    q->d[q->n].x = x; q->d[q->n].y = y; q->d[q->n].z = z; q->n++;
}
static void q_free(queue *q) { free(q->d); q->d = NULL; q->n = q->cap = 0; }

/* ------------------------------------------------------------------ hash of voxel blocks */
static uint64_t hash3(int x, int y, int z)  /* BlockHasher, voxmap_utils.cuh:69-81 */
{ return ((uint64_t)(int64_t)x * 73856093ull) ^ ((uint64_t)(int64_t)y * 19349669ull) ^ ((uint64_t)(int64_t)z * 83492791ull); }

static oblock *blk_find(const gie_oracle *o, int bx, int by, int bz)
{
    uint64_t h = hash3(bx, by, bz) & (uint64_t)(o->hcap - 1);
    for (;;) {
        int32_t s = o->htab[h];
        if (s < 0) return NULL;
        oblock *b = o->blocks[s];
        if (b->key[0] == bx && b->key[1] == by && b->key[2] == bz) return b;
        h = (h + 1) & (uint64_t)(o->hcap - 1);
    }
}
static void htab_insert(gie_oracle *o, int slot)
{
    oblock *b = o->blocks[slot];
    uint64_t h = hash3(b->key[0], b->key[1], b->key[2]) & (uint64_t)(o->hcap - 1);
    while (o->htab[h] >= 0) h = (h + 1) & (uint64_t)(o->hcap - 1);
    o->htab[h] = slot;
}
static oblock *blk_get_or_alloc(gie_oracle *o, int bx, int by, int bz)
{
    oblock *b = blk_find(o, bx, by, bz);
    if (b) return b;
    if (o->nblocks == o->blocks_cap) {
        o->blocks_cap *= 2;
        o->blocks = (oblock **)realloc(o->blocks, sizeof(oblock *) * (size_t)o->blocks_cap);
    }
    if ((o->nblocks + 1) * 2 > o->hcap) {
        o->hcap *= 2;
        o->htab = (int32_t *)realloc(o->htab, sizeof(int32_t) * (size_t)o->hcap);
        for (int i = 0; i < o->hcap; i++) o->htab[i] = -1;
        for (int i = 0; i < o->nblocks; i++) htab_insert(o, i);
    }
    b = (oblock *)malloc(sizeof(oblock));
    b->key[0] = bx; b->key[1] = by; b->key[2] = bz; b->dirty = 0;
    for (int i = 0; i < VBSZ; i++) { /* GlbVoxel defaults, voxmap_utils.cuh:30-43 */
        ovox *v = &b->v[i];
        v->occ_val = 0; v->vox_type = GIE_VOX_UNKNOWN; v->update_ct = 0;
        v->coc[0] = v->coc[1] = v->coc[2] = EMPTY_KEY_C;
        v->dist_sq = GIE_EMPTY_VALUE; v->wave_layer = -1;
        v->pair_dist = 0; v->pair_par = 0;
        v->prop_dist = 0x7fffffff; v->prop_par = 0; v->blk = o->nblocks;
        v->xprop_dist = 0x7fffffff; v->xprop_par = 0;
    }
    if (o->wide) for (int i = 0; i < VBSZ; i++) b->v[i].dist_sq = o->empty_value;
    o->blocks[o->nblocks] = b;
    htab_insert(o, o->nblocks);
    o->nblocks++;
    o->st.blocks_new++;
    return b;
}
/* stream_VB_keys_D bookkeeping (unify_helper.cuh:103-110,510-520, wave_core.cuh:129-134) as one
 * flag per block: set whenever a store changes the type, distance or closest obstacle of a voxel */
static void touch(gie_oracle *o, const ovox *v) { if (o->track) o->blocks[v->blk]->dirty = 1; }

static ovox *vox_find(const gie_oracle *o, int gx, int gy, int gz)  /* hash lookup + retrive_vox_D */
{
    oblock *b = blk_find(o, fdiv8(gx), fdiv8(gy), fdiv8(gz));
    return b ? &b->v[vox_in_blk(gx, gy, gz)] : NULL;
}

/* ------------------------------------------------------------------ create / destroy */
gie_oracle *go_create(const gie_config *cfg)
{
    gie_oracle *o = (gie_oracle *)calloc(1, sizeof(gie_oracle));
    o->cfg = *cfg;
    o->X = cfg->local_size[0]; o->Y = cfg->local_size[1]; o->Z = cfg->local_size[2];
    o->N = o->X * o->Y * o->Z;
    o->max_width = o->X + o->Y + o->Z;                         /* local_batch.h:46 */
    o->max_loc_dist_sq = o->X * o->X + o->Y * o->Y + o->Z * o->Z; /* :47 */
    if (o->max_width < 1022) {                                 /* :51-58 the reference's envelope */
        o->wide = 0; o->wr[0] = 2046; o->wr[1] = 2046; o->wr[2] = 1022;
        o->empty_value = GIE_EMPTY_VALUE; o->invalid_dist_min = 900000;
    } else {
        o->wide = 1; o->wr[0] = 16382; o->wr[1] = 16382; o->wr[2] = 8190;
        o->empty_value = 4194303; o->invalid_dist_min = 4000000;
    }
    o->inv_coc = 16383;
    size_t n = (size_t)o->N;
    o->ray_count = (int32_t *)calloc(n, 4);
    o->inst_type = (int8_t *)calloc(n, 1);
    o->glb_type = (int8_t *)calloc(n, 1);
    o->edt = (float *)calloc(n, 4);
    o->aux = (int32_t *)calloc(n, 4);
    o->bdist = (int32_t *)calloc(n, 4);
    o->bcoc = (int32_t *)calloc(n * 3, 4);
    o->pair_dist = (int32_t *)calloc(n, 4);  /* zero-initialised: SURVEY App. B #3 */
    o->pair_par = (int64_t *)calloc(n, 8);
    o->g_bak = (int32_t *)calloc(n, 4);
    o->coc_bak = (int64_t *)calloc(n, 8);
    o->wave_layer = (int32_t *)calloc(n, 4);
    o->lprop_dist = (int32_t *)malloc(n * 4);
    o->lprop_par = (int64_t *)calloc(n, 8);
    for (size_t i = 0; i < n; i++) o->lprop_dist[i] = 0x7fffffff;
    o->g1 = (int32_t *)malloc(n * 4); o->cy1 = (int32_t *)malloc(n * 4);
    o->d2 = (int32_t *)malloc(n * 4); o->cx2 = (int32_t *)malloc(n * 4); o->cy2 = (int32_t *)malloc(n * 4);
    o->blocks_cap = 1024; o->blocks = (oblock **)malloc(sizeof(oblock *) * 1024);
    o->hcap = 4096; o->htab = (int32_t *)malloc(sizeof(int32_t) * 4096);
    for (int i = 0; i < o->hcap; i++) o->htab[i] = -1;
    { const float q1[4] = { 1, 0, 0, 0 }, t0[3] = { 0, 0, 0 }; o->L2G = om_from_quat(q1, t0); o->G2L = om_inverse(o->L2G); }
    o->next_whole[0] = o->X; o->next_whole[1] = o->Y; o->next_whole[2] = o->Z;
    return o;
}

void go_destroy(gie_oracle *o)
{
    if (!o) return;
    for (int i = 0; i < o->nblocks; i++) free(o->blocks[i]);
    free(o->blocks); free(o->htab);
    free(o->ray_count); free(o->inst_type); free(o->glb_type); free(o->edt); free(o->aux); free(o->bdist);
    free(o->bcoc); free(o->pair_dist); free(o->pair_par); free(o->g_bak); free(o->coc_bak);
    free(o->wave_layer); free(o->lprop_dist); free(o->lprop_par);
    free(o->g1); free(o->cy1); free(o->d2); free(o->cx2); free(o->cy2);
    free(o->box_ll); free(o->box_ur); free(o->box_act);
    free(o);
}

/* trans2proj (projection.h:14-33) + calculate_pivot_origin / calculate_update_pivot
 * (local_batch.h:128-166) + _time++ (volumetric_mapper.cpp:144). */
int go_set_pose(gie_oracle *o, const float pos[3], const float q[4])
{
    o->map_ct++;
    memset(&o->st, 0, sizeof(o->st));
    o->st.frame = o->map_ct;
    o->L2G = om_from_quat(q, pos);
    o->G2L = om_inverse(o->L2G);
    const float w = o->cfg.voxel_width;
    const int sz[3] = { o->X, o->Y, o->Z };
    for (int i = 0; i < 3; i++) {
        o->origin[i] = pos[i];
        const int c = om_voxel_of(pos[i], w);
        o->tile_off[i] = o->next_off[i];
        o->whole_lo[i] = -(o->next_whole[i] / 2) + sz[i] / 2 - o->next_off[i];
        o->whole_hi[i] = o->whole_lo[i] + o->next_whole[i];
        o->pvt[i] = c - sz[i] / 2 + o->tile_off[i];
        o->msg_origin[i] = (float)o->pvt[i] * w;   /* coord2pos, local_batch.h:259-267 */
        o->upvt[i] = c - o->wr[i] / 2;
    }
    return 0;
}

/* ------------------------------------------------------------------ OGM: ray casting */
/* clearRayLoc, pntcld_raycast.cu:9-18 (+ get_vox_type / atom_add_type_count bounds rules,
 * local_batch.h:302-339: outside the volume the type reads UNKNOWN and the add is dropped). */
static int clear_ray(gie_oracle *o, int lx, int ly, int lz)
{
    if (!in_loc(o, lx, ly, lz)) return 1;            /* UNKNOWN != OCCUPIED → true, add dropped */
    const int id = lid(o, lx, ly, lz);
    if (o->inst_type[id] != GIE_VOX_OCCUPIED) { o->ray_count[id] -= 1; return 1; }
    return 0;
}

/* RAY::rayCastLoc, ray_cast.h:57-144 */
static void ray_cast(gie_oracle *o, const float p0[3], const float p1[3], float max_length)
{
    const float w = o->cfg.voxel_width;
    int i0[3], i1[3];
    for (int i = 0; i < 3; i++) { i0[i] = om_voxel_of(p0[i], w); i1[i] = om_voxel_of(p1[i], w); }
    clear_ray(o, i0[0] - o->pvt[0], i0[1] - o->pvt[1], i0[2] - o->pvt[2]);
    if (i0[0] == i1[0] && i0[1] == i1[1] && i0[2] == i1[2]) return;
    float dir[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] };
    const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]); /* helper_math length(): sqrtf(dot) */
    for (int i = 0; i < 3; i++) dir[i] = dir[i] / len;
    int step[3]; float tMax[3], tDelta[3];
    int cur[3] = { i0[0], i0[1], i0[2] };
    for (int i = 0; i < 3; i++) {
        if (dir[i] > 0.0f) step[i] = 1; else if (dir[i] < 0.0f) step[i] = -1; else step[i] = 0;
        if (step[i] != 0) {
            const float border = (float)cur[i] * w + (float)step[i] * w * 0.5f;
            tMax[i] = (border - p0[i]) / dir[i];
            tDelta[i] = w / fabsf(dir[i]);
        } else { tMax[i] = 3.402823466e+38f; tDelta[i] = 3.402823466e+38f; }
    }
    for (;;) {
        int dim;
        if (tMax[0] < tMax[1]) dim = (tMax[0] < tMax[2]) ? 0 : 2;
        else dim = (tMax[1] < tMax[2]) ? 1 : 2;
        cur[dim] += step[dim];
        tMax[dim] += tDelta[dim];
        if (!clear_ray(o, cur[0] - o->pvt[0], cur[1] - o->pvt[1], cur[2] - o->pvt[2])) break;
        if (cur[0] == i1[0] && cur[1] == i1[1] && cur[2] == i1[2]) break;
        const float m01 = tMax[0] < tMax[1] ? tMax[0] : tMax[1];
        const float dist = m01 < tMax[2] ? m01 : tMax[2];
        if (dist > max_length || dist > len) break;
    }
}

/* PNTCLD_RAYCAST::localOGMKernels, pntcld_raycast.cu:105-117:
 * registerLocObs (:83-102) for every point, then freeLocObs (:67-80) for every point, then
 * getAllocKeys (:21-63). */
int go_ogm_pointcloud(gie_oracle *o, const float *xyz, int n)
{
    const float w = o->cfg.voxel_width;
    o->pntcld_mode = 1;
    float *g = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        om_transform(o->L2G, &xyz[3 * i], &g[3 * i]);
        const float gz = g[3 * i + 2];
        if (om_point_usable(&g[3 * i]) && gz >= o->cfg.ogm_min_h && gz <= o->cfg.ogm_max_h) {
            const int lx = om_voxel_of(g[3 * i], w) - o->pvt[0];
            const int ly = om_voxel_of(g[3 * i + 1], w) - o->pvt[1];
            const int lz = om_voxel_of(gz, w) - o->pvt[2];
            if (in_loc(o, lx, ly, lz)) { const int id = lid(o, lx, ly, lz); o->inst_type[id] = GIE_VOX_OCCUPIED; o->ray_count[id] += 1; }
        }
    }
    const float max_len = 0.707f * (float)o->X * w;   /* pntcld_raycast.cu:79 */
    for (int i = 0; i < n; i++)
        if (om_point_usable(&g[3 * i])) ray_cast(o, o->origin, &g[3 * i], max_len);   /* (non-finite points are ignored: oracle_math.h) */
    free(g);
    /* getAllocKeys: robot sphere → count = -1; count>0 OCC, <0 FREE (the block key it also
     * writes is "this voxel was observed", which is inst_type != UNKNOWN here). */
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        if (o->cfg.for_motion_planner) {
            const int cx = x - (o->X / 2 - o->tile_off[0]), cy = y - (o->Y / 2 - o->tile_off[1]), cz = z - (o->Z / 2 - o->tile_off[2]);
            if (cx * cx + cy * cy + cz * cz <= o->cfg.robot_r2_grids) o->ray_count[id] = -1;
        }
        const int c = o->ray_count[id];
        if (c > 0) o->inst_type[id] = GIE_VOX_OCCUPIED; else if (c < 0) o->inst_type[id] = GIE_VOX_FREE;
    }
    return 0;
}

/* ------------------------------------------------------------------ OGM: projective kernels */
static int robot_sphere(const gie_oracle *o, int x, int y, int z)
{
    if (!o->cfg.for_motion_planner) return 0;
    const int cx = x - (o->X / 2 - o->tile_off[0]), cy = y - (o->Y / 2 - o->tile_off[1]), cz = z - (o->Z / 2 - o->tile_off[2]);  /* _half_shift, local_batch.h:49 */
    return cx * cx + cy * cy + cz * cz <= o->cfg.robot_r2_grids;
}
static int pos_mod(int i, int n) { return (i % n + n) % n; }  /* vlp16_helper.h:11-15 */

/* VLP_FAST::setLocalOccupancy (vlp16_fast.cu:8-87) with VLP_HELPER::G2L (vlp16_helper.h:35-65).
 * getDist2Line (vlp16_helper.h:18-33) measures the distance from the point to the ray through
 * the point's own (phi, theta): identically 0 up to rounding (< 1e-4 m at 100 m range), so its
 * ">= voxel_width ⇒ unobservable" gate cannot fire for any voxel_width ≥ 1 mm and is omitted.
 * Quirk App.B #8: in the band real-0.3 <= ideal < real-0.1 the reference writes nothing. */
int go_ogm_multiscan(gie_oracle *o, const float *ranges, const gie_multiscan_param *p)
{
    const float w = o->cfg.voxel_width;
    o->pntcld_mode = 0;
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        if (robot_sphere(o, x, y, z)) { o->inst_type[id] = GIE_VOX_FREE; continue; }
        const float gx = (float)(x + o->pvt[0]) * w, gy = (float)(y + o->pvt[1]) * w, gz = (float)(z + o->pvt[2]) * w;
        const float gp[3] = { gx, gy, gz };
        float lp[3];
        om_transform(o->G2L, gp, lp);
        const float lx = lp[0], ly = lp[1], lz = lp[2];
        const float theta = om_atan2(ly, lx);
        int theta_idx = (int)floorf((theta - p->theta_min) / p->theta_inc + 0.5f);
        theta_idx = pos_mod(theta_idx, p->scan_num);
        const float range_hor = sqrtf(ly * ly + lx * lx);
        const float phi = om_atan2(lz, range_hor);
        const int phi_idx = (int)floorf((phi - p->phi_min) / p->phi_inc + 0.5f);
        if (phi_idx < 0 || phi_idx >= p->ring_num) continue;          /* depth = -1 */
        const float ideal = sqrtf(lx * lx + ly * ly);
        if (ideal < 0 || theta_idx < 0 || theta_idx >= p->scan_num) continue;
        const float real = ranges[phi_idx * p->scan_num + theta_idx];
        if (isnan(real) || real <= 0.3f) continue;
        if (ideal < real - 0.1f) {
            if (ideal < real - 0.3f) o->inst_type[id] = GIE_VOX_FREE;
        } else if ((double)ideal > (double)real + 0.1) {             /* "+ 0.1" is a double literal, vlp16_fast.cu:73 */
            /* not observable */
        } else if (gz >= o->cfg.ogm_min_h && gz <= o->cfg.ogm_max_h) {
            o->inst_type[id] = GIE_VOX_OCCUPIED;
        }
    }
    return 0;
}

/* A scan that arrives classified (include/gie.h gie_ogm_labels): what the projective kernels'
 * store phase leaves behind (vlp16_fast.cu:76-86) for a given label plane.  The plane replaces
 * the scan labels; the robot sphere is FREE as in every setLocalOccupancy. */
int go_ogm_labels(gie_oracle *o, const int8_t *labels)
{
    o->pntcld_mode = 0;
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        if (robot_sphere(o, x, y, z)) { o->inst_type[id] = GIE_VOX_FREE; continue; }
        const int8_t l = labels[id];
        o->inst_type[id] = (l == GIE_VOX_FREE || l == GIE_VOX_OCCUPIED) ? l : GIE_VOX_UNKNOWN;
    }
    return 0;
}

/* REALSENSE_FAST::setLocalOccupancy (realsense_fast.cu:9-94) with CAM_HELPER::G2L
 * (camera_helper.h:11-23). */
int go_ogm_depth(gie_oracle *o, const float *depth, const gie_cam_param *p)
{
    const float w = o->cfg.voxel_width;
    o->pntcld_mode = 0;
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        if (robot_sphere(o, x, y, z)) { o->inst_type[id] = GIE_VOX_FREE; continue; }
        const float gx = (float)(x + o->pvt[0]) * w, gy = (float)(y + o->pvt[1]) * w, gz = (float)(z + o->pvt[2]) * w;
        const float gp[3] = { gx, gy, gz };
        float lp[3];
        om_transform(o->G2L, gp, lp);
        const float lx = lp[0], ly = lp[1], lz = lp[2];
        const float ideal = lx;
        if (ideal <= 0.3f || ideal > 6.0f) continue;
        const float fpx = floorf(-ly * p->fx / ideal + p->cx + 0.5f);
        const float fpy = floorf(-lz * p->fy / ideal + p->cy + 0.5f);
        if (!(fpx >= 0.0f && fpx < (float)p->cols && fpy >= 0.0f && fpy < (float)p->rows)) continue;
        const int px = (int)fpx, py = (int)fpy;
        float real = depth[p->cols * py + px];
        if (real <= 0.21f) continue;
        if (isnan(real)) { if (p->valid_nan) real = 1000.f; else continue; }   /* SENS_FAR_DIST, cuda_macro.h:16 */
        if (ideal < real - w) o->inst_type[id] = GIE_VOX_FREE;
        else if (ideal > real + w) { /* not observable */ }
        else if (gz >= o->cfg.ogm_min_h && gz <= o->cfg.ogm_max_h) o->inst_type[id] = GIE_VOX_OCCUPIED;
    }
    return 0;
}

/* HOKUYO_FAST::setLocalOccupancy (hokuyo_fast.cu:9-81) with SCAN_HELPER::G2L (hokuyo_helper.h:17-33). */
int go_ogm_scan2d(gie_oracle *o, const float *ranges, const gie_scan_param *p)
{
    const float w = o->cfg.voxel_width;
    o->pntcld_mode = 0;
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        if (robot_sphere(o, x, y, z)) { o->inst_type[id] = GIE_VOX_FREE; continue; }
        const float gx = (float)(x + o->pvt[0]) * w, gy = (float)(y + o->pvt[1]) * w, gz = (float)(z + o->pvt[2]) * w;
        const float gp[3] = { gx, gy, gz };
        float lp[3];
        om_transform(o->G2L, gp, lp);
        const float lx = lp[0], ly = lp[1], lz = lp[2];
        const float theta = om_atan2(ly, lx);
        int theta_idx = (int)floorf((theta - p->theta_min) / p->theta_inc + 0.5f);
        theta_idx = pos_mod(theta_idx, p->scan_num);
        if (!(fabsf(lz) < w)) continue;                                 /* depth = -1 */
        const float ideal = sqrtf(lx * lx + ly * ly);
        const float real = ranges[theta_idx];
        if (isnan(real) || real <= 0.3f) continue;
        if (ideal < real - 0.3f) o->inst_type[id] = GIE_VOX_FREE;
        else if ((double)ideal > (double)real + 0.3) { /* not observable; double literal, hokuyo_fast.cu:67 */ }
        else if (gz >= o->cfg.ogm_min_h && gz <= o->cfg.ogm_max_h) o->inst_type[id] = GIE_VOX_OCCUPIED;
    }
    return 0;
}

int go_set_ext_boxes(gie_oracle *o, const float *ll, const float *ur, const uint8_t *act, int n)
{
    free(o->box_ll); free(o->box_ur); free(o->box_act);
    o->box_ll = o->box_ur = NULL; o->box_act = NULL; o->nbox = n;
    if (n > 0) {
        o->box_ll = (float *)malloc(sizeof(float) * 3 * (size_t)n); memcpy(o->box_ll, ll, sizeof(float) * 3 * (size_t)n);
        o->box_ur = (float *)malloc(sizeof(float) * 3 * (size_t)n); memcpy(o->box_ur, ur, sizeof(float) * 3 * (size_t)n);
        o->box_act = (uint8_t *)malloc((size_t)n); memcpy(o->box_act, act, (size_t)n);
    }
    return 0;
}

/* ------------------------------------------------------------------ fuse */
static int inside_aabb(const float p[3], const float *ll, const float *ur)   /* voxmap_utils.cuh:202-207 */
{ return p[0] >= ll[0] && p[1] >= ll[1] && p[2] >= ll[2] && p[0] <= ur[0] && p[1] <= ur[1] && p[2] <= ur[2]; }

/* set_hashvoxel_occ_val, voxmap_utils.cuh:181-200 */
static void set_occ(ovox *v, float val, float a, int thresh)
{
    if (v->vox_type != GIE_VOX_UNKNOWN) val = a * val + (1.0f - a) * (float)v->occ_val;
    else val = a * val + (1.0f - a) * 0.0f;
    if (val > 254.0f) val = 254.0f;
    if (val < 1.0f) val = 1.0f;
    v->occ_val = (uint8_t)val;
    v->vox_type = (v->occ_val > thresh) ? GIE_VOX_OCCUPIED : GIE_VOX_FREE;
}

/* GlbHashMap::updateHashOGM (glb_hash_map.cu:115-143): allocHashTB (:58-113) — a block is
 * allocated for every voxel the scan observed — then updateHashOGMWithPntCld
 * (unify_helper.cuh:35-116) or updateHashOGMWithSensor (:118-197). */
/* Block-pool lifecycle, gie_config.retain_radius_blocks = R > 0 (the reference itself never erases a block — its pool only
 * shrinks and allocate_n throws when it is empty, blockalloc.h:50-67; the free list it declares, :69-118, is what is fed
 * here): before anything is allocated, every block whose block coordinate lies more than R blocks (Chebyshev) outside the
 * block box of the local volume +-1 voxel is erased; its voxels read as defaults from then on. */
static void evict_far_blocks(gie_oracle *o)
{
    const int R = o->cfg.retain_radius_blocks;
    if (R <= 0) return;
    const int sz[3] = { o->X, o->Y, o->Z };
    int lo[3], hi[3];
    for (int i = 0; i < 3; i++) { lo[i] = fdiv8(o->pvt[i] - 1) - R; hi[i] = fdiv8(o->pvt[i] + sz[i]) + R; }
    int kept = 0;
    for (int s = 0; s < o->nblocks; s++) {
        oblock *b = o->blocks[s];
        int far = 0;
        for (int i = 0; i < 3; i++) far |= (b->key[i] < lo[i]) | (b->key[i] > hi[i]);
        if (far) { free(b); continue; }
        if (kept != s) for (int i = 0; i < VBSZ; i++) b->v[i].blk = kept;
        o->blocks[kept++] = b;
    }
    if (kept == o->nblocks) return;
    o->nblocks = kept;
    for (int i = 0; i < o->hcap; i++) o->htab[i] = -1;
    for (int i = 0; i < o->nblocks; i++) htab_insert(o, i);
}

int go_fuse(gie_oracle *o)
{
    const float w = o->cfg.voxel_width;
    o->st.blocks_new = 0;
    evict_far_blocks(o);
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++)
        if (o->inst_type[lid(o, x, y, z)] != GIE_VOX_UNKNOWN)
            blk_get_or_alloc(o, fdiv8(x + o->pvt[0]), fdiv8(y + o->pvt[1]), fdiv8(z + o->pvt[2]));
    o->st.blocks_total = o->nblocks;
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        const int count = o->ray_count[id];
        const int8_t nt = o->inst_type[id];
        o->ray_count[id] = 0;
        o->inst_type[id] = GIE_VOX_UNKNOWN;
        const int gx = x + o->pvt[0], gy = y + o->pvt[1], gz = z + o->pvt[2];
        ovox *v = vox_find(o, gx, gy, gz);
        if (!v) { o->glb_type[id] = GIE_VOX_UNKNOWN; continue; }
        const float gp[3] = { (float)gx * w, (float)gy * w, (float)gz * w };
        int occ_flag = 0;
        const int8_t ty_before = v->vox_type;
        if (o->nbox > 0 && o->box_act[0] && !inside_aabb(gp, o->box_ll, o->box_ur)) occ_flag = 1;
        else for (int i = 1; i < o->nbox; i++)
            if (o->box_act[i] && inside_aabb(gp, o->box_ll + 3 * i, o->box_ur + 3 * i)) { occ_flag = 1; break; }
        if (o->pntcld_mode) {
            if (count > 0 || occ_flag) set_occ(v, 250.f, 1.f, o->cfg.occupancy_threshold);
            else if (count < 0) {
                float pb = (float)(-count) / 10.f; if (pb > 1.f) pb = 1.f;
                set_occ(v, 0.f, pb, o->cfg.occupancy_threshold);
            }
        } else {
            if (nt == GIE_VOX_OCCUPIED || occ_flag) set_occ(v, 250.f, 0.8f, o->cfg.occupancy_threshold);
            else if (nt == GIE_VOX_FREE) set_occ(v, 0.f, 0.5f, o->cfg.occupancy_threshold);
        }
        if (v->vox_type != ty_before) touch(o, v);
        o->glb_type[id] = v->vox_type;
    }
    return 0;
}

/* ------------------------------------------------------------------ batch EDT */
/* EDT_OCC::batchEDTUpdate (local_edt.cu:7-28): EDTphase1 (local_edt_core.h:14-82, two sweeps
 * along y), EDTphase2 (:84-135, Meijster lower envelope along x with LocMap::f/sep,
 * local_batch.h:494-508, truncating division), EDTphase3 (:137-193, along z with f_z/sep_z,
 * :510-520).  The six cuTT transposes only re-lay the arrays (cutt.h:47-58) and vanish here.
 * Closest-obstacle components the reference leaves to stale memory (App. B #1) can only
 * surface when the volume holds no obstacle; they are the "invalid" marker here. */
int go_batch_edt(gie_oracle *o)
{
    const int X = o->X, Y = o->Y, Z = o->Z, MW = o->max_width, INV = o->inv_coc;
    int mx = X > Z ? X : Z;
    int *s = (int *)malloc(sizeof(int) * (size_t)mx), *t = (int *)malloc(sizeof(int) * (size_t)mx);
    /* phase 1 */
    for (int z = 0; z < Z; z++) for (int x = 0; x < X; x++) {
        int id = lid(o, x, 0, z);
        if (o->glb_type[id] == GIE_VOX_OCCUPIED) { o->g1[id] = 0; o->cy1[id] = 0; }
        else { o->g1[id] = MW; o->cy1[id] = INV; }
        for (int y = 1; y < Y; y++) {
            id = lid(o, x, y, z);
            const int pid = lid(o, x, y - 1, z);
            if (o->glb_type[id] == GIE_VOX_OCCUPIED) { o->g1[id] = 0; o->cy1[id] = y; }
            else if (o->cy1[pid] < MW) { o->g1[id] = 1 + o->g1[pid]; o->cy1[id] = o->cy1[pid]; }
            else { o->g1[id] = MW; o->cy1[id] = INV; }
        }
        for (int y = Y - 2; y >= 0; y--) {
            id = lid(o, x, y, z);
            const int nid = lid(o, x, y + 1, z);
            if (o->g1[nid] < o->g1[id]) {
                if (o->cy1[nid] < MW) { o->g1[id] = 1 + o->g1[nid]; o->cy1[id] = o->cy1[nid]; }
                else o->g1[id] = MW;
            }
        }
    }
    /* phase 2 */
    for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) {
#define A2(i) (o->g1[lid(o, (i), y, z)])
#define F2(u, i) (((u) - (i)) * ((u) - (i)) + A2(i) * A2(i))
        int q = 0; s[0] = 0; t[0] = 0;
        for (int u = 1; u < X; u++) {
            while (q >= 0 && F2(t[q], s[q]) > F2(t[q], u)) q--;
            if (q < 0) { q = 0; s[0] = u; }
            else {
                const int i = s[q];
                const int w = 1 + (u * u - i * i + A2(u) * A2(u) - A2(i) * A2(i)) / (2 * (u - i));
                if (w < X) { q++; s[q] = u; t[q] = w; }
            }
        }
        for (int u = X - 1; u >= 0; u--) {
            const int id = lid(o, u, y, z), sid = lid(o, s[q], y, z);
            o->d2[id] = F2(u, s[q]);
            o->cx2[id] = s[q];
            o->cy2[id] = (o->cy1[sid] < MW) ? o->cy1[sid] : INV;
            if (u == t[q]) q--;
        }
#undef A2
#undef F2
    }
    /* phase 3 (skipped when Z == 1, local_edt.cu:21) */
    for (int y = 0; y < Y; y++) for (int x = 0; x < X; x++) {
        if (Z == 1) {
            const int id = lid(o, x, y, 0);
            o->aux[id] = o->d2[id];
            o->bcoc[3 * id] = o->cx2[id]; o->bcoc[3 * id + 1] = o->cy2[id]; o->bcoc[3 * id + 2] = 0;
            continue;
        }
#define A3(i) (o->d2[lid(o, x, y, (i))])
#define F3(u, i) (((u) - (i)) * ((u) - (i)) + A3(i))
        int q = 0; s[0] = 0; t[0] = 0;
        for (int u = 1; u < Z; u++) {
            while (q >= 0 && F3(t[q], s[q]) > F3(t[q], u)) q--;
            if (q < 0) { q = 0; s[0] = u; }
            else {
                const int i = s[q];
                const int w = 1 + (u * u - i * i + A3(u) - A3(i)) / (2 * (u - i));
                if (w < Z) { q++; s[q] = u; t[q] = w; }
            }
        }
        for (int u = Z - 1; u >= 0; u--) {
            const int id = lid(o, x, y, u), sid = lid(o, x, y, s[q]);
            o->aux[id] = F3(u, s[q]);
            o->bcoc[3 * id] = o->cx2[sid];           /* always a valid x index (local_edt_core.h:120-122,178-181) */
            o->bcoc[3 * id + 1] = o->cy2[sid];
            o->bcoc[3 * id + 2] = s[q];
            if (u == t[q]) q--;
        }
#undef A3
#undef F3
    }
    free(s); free(t);
    memcpy(o->bdist, o->aux, sizeof(int32_t) * (size_t)o->N);
    return 0;
}

/* ------------------------------------------------------------------ merge: Mark / frontiers */
/* MarkLimitedObserve, unify_helper.cuh:201-273 */
static void mark_limited_observe(gie_oracle *o)
{
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        if (o->glb_type[id] == GIE_VOX_UNKNOWN) continue;
        int cn[3] = { o->bcoc[3 * id], o->bcoc[3 * id + 1], o->bcoc[3 * id + 2] };
        const int dn = o->aux[id];
        if (cn[0] > o->max_width || cn[1] > o->max_width || cn[2] > o->max_width || cn[0] < 0 || cn[1] < 0 || cn[2] < 0) {
            o->pair_dist[id] = o->empty_value;    /* invalid_coc_buf (voxmap_utils.cuh:174-179): see nothing */
            o->pair_par[id] = PAR_NONE;
            o->aux[id] = o->empty_value;
        }
        ovox *v = vox_find(o, x + o->pvt[0], y + o->pvt[1], z + o->pvt[2]);
        if (!v) continue; /* the reference asserts: an observed voxel always has its block */
        const int dold = v->dist_sq;
        const int ol[3] = { v->coc[0] - o->pvt[0], v->coc[1] - o->pvt[1], v->coc[2] - o->pvt[2] };
        /* with tiling: inside the union of all tiles (include/gie.h gie_set_tile) */
        const int old_in = ol[0] >= o->whole_lo[0] && ol[0] < o->whole_hi[0] && ol[1] >= o->whole_lo[1] && ol[1] < o->whole_hi[1] && ol[2] >= o->whole_lo[2] && ol[2] < o->whole_hi[2];
        if (dn > dold && !old_in) { cn[0] = ol[0]; cn[1] = ol[1]; cn[2] = ol[2]; o->aux[id] = dold; } /* limited observation */
        /* loc2wave_range, done in 64-bit because an EMPTY_KEY/invalid coc is far away */
        const long long wx = (long long)cn[0] + o->pvt[0] - o->upvt[0];
        const long long wy = (long long)cn[1] + o->pvt[1] - o->upvt[1];
        const long long wz = (long long)cn[2] + o->pvt[2] - o->upvt[2];
        if (!(wx >= 0 && wx < o->wr[0] && wy >= 0 && wy < o->wr[1] && wz >= 0 && wz < o->wr[2])) {
            o->pair_dist[id] = o->empty_value;    /* parent id left as is (App. B #4) */
            o->aux[id] = o->empty_value;
        } else {
            o->pair_dist[id] = o->aux[id];
            o->pair_par[id] = pack_wr((int)wx, (int)wy, (int)wz);
        }
        o->g_bak[id] = o->aux[id];
        o->coc_bak[id] = o->pair_par[id];
    }
}

/* obtainFrontiers, unify_helper.cuh:275-446 */
static const int DIRS[6][3] = { { -1, 0, 0 }, { 1, 0, 0 }, { 0, -1, 0 }, { 0, 1, 0 }, { 0, 0, -1 }, { 0, 0, 1 } }; /* glb_hash_map.h:59-61 */

static void obtain_frontiers(gie_oracle *o, queue *fa, queue *fb, queue *fc)
{
    const int ct = o->map_ct;
    for (int i = 0; i < o->N; i++) o->wave_layer[i] = o->empty_value;
    /* The reference flips FREE→FNT in place while neighbours read _glb_type; they only test
     * UNKNOWN / OCCUPIED, which FNT does not alias, so the in-place edit is order-free. */
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        const int8_t ty = o->glb_type[id];
        if (ty == GIE_VOX_UNKNOWN) continue;
        int cw[3]; unpack_wr(o->coc_bak[id], &cw[0], &cw[1], &cw[2]);
        const int cl[3] = { cw[0] + o->upvt[0] - o->pvt[0], cw[1] + o->upvt[1] - o->pvt[1], cw[2] + o->upvt[2] - o->pvt[2] };
        const int cg[3] = { cl[0] + o->pvt[0], cl[1] + o->pvt[1], cl[2] + o->pvt[2] };
        const int cd = o->g_bak[id];
        if (!in_loc(o, cl[0], cl[1], cl[2])) continue;
        int cur_in_q = 0, has_unknown = 0;
        for (int k = 0; k < 6; k++) {
            const int nx = x + DIRS[k][0], ny = y + DIRS[k][1], nz = z + DIRS[k][2];
            if (in_loc(o, nx, ny, nz)) {
                const int nid = lid(o, nx, ny, nz);
                if (o->glb_type[nid] == GIE_VOX_UNKNOWN) { has_unknown = 1; continue; }
                int nw[3]; unpack_wr(o->coc_bak[nid], &nw[0], &nw[1], &nw[2]);
                const int nl[3] = { nw[0] + o->upvt[0] - o->pvt[0], nw[1] + o->upvt[1] - o->pvt[1], nw[2] + o->upvt[2] - o->pvt[2] };
                if (!in_loc(o, nl[0], nl[1], nl[2]) && in_wr(o, nw[0], nw[1], nw[2])) {
                    const int d = d2i(nl[0], nl[1], nl[2], x, y, z);
                    if (d < cd) {
                        o->pair_dist[id] = d; o->pair_par[id] = pack_wr(nw[0], nw[1], nw[2]);
                        if (!cur_in_q) { cur_in_q = 1; o->wave_layer[id] = 1; q_push(fc, x, y, z); }
                    }
                }
            } else {
                const int ng[3] = { nx + o->pvt[0], ny + o->pvt[1], nz + o->pvt[2] };
                ovox *nv = vox_find(o, ng[0], ng[1], ng[2]);
                if (!nv) { has_unknown = 1; continue; }
                if (nv->vox_type == GIE_VOX_UNKNOWN) { has_unknown = 1; continue; }
                const int nd = nv->dist_sq;
                if (invalid_dist_glb(o, nd)) continue;
                if (invalid_coc_glb(nv->coc)) continue;
                const int nw[3] = { nv->coc[0] - o->upvt[0], nv->coc[1] - o->upvt[1], nv->coc[2] - o->upvt[2] };
                const int nl[3] = { nv->coc[0] - o->pvt[0], nv->coc[1] - o->pvt[1], nv->coc[2] - o->pvt[2] };
                const int n_valid = in_wr(o, nw[0], nw[1], nw[2]);
                const int n_local = in_loc(o, nl[0], nl[1], nl[2]);
                /* tiling: the neighbour is either a ghost (another tile's voxel, refreshed this update: its obstacle counts
                 * wherever it lies outside this tile) or a remembered voxel outside the WHOLE volume — and what that one
                 * remembers about an obstacle inside the whole volume is another tile's business (it may have vanished since:
                 * only the owner knows); without tiling whole = local and this is the reference's test */
                const int n_hidden = in_whole(o, nx, ny, nz) ? !n_local : !in_whole(o, nl[0], nl[1], nl[2]);
                if (n_hidden && n_valid) {
                    const int d = d2i(nl[0], nl[1], nl[2], x, y, z);
                    if (d < cd) {
                        o->pair_dist[id] = d; o->pair_par[id] = pack_wr(nw[0], nw[1], nw[2]);
                        if (!cur_in_q) { cur_in_q = 1; o->wave_layer[id] = 1; q_push(fc, x, y, z); }
                    }
                }
                /* tiling (include/gie.h): another tile's voxel is only ever read as a ghost, never raised / lowered from here */
                if (o->cfg.fast_mode || in_whole(o, nx, ny, nz)) continue;
                const int c2n = d2i(nx, ny, nz, cl[0], cl[1], cl[2]);
                if (c2n < nd) {                                      /* lower out */
                    nv->wave_layer = 1; nv->update_ct = ct;
                    nv->pair_dist = c2n; nv->pair_par = pack_wr(cw[0], cw[1], cw[2]);
                    q_push(fb, ng[0], ng[1], ng[2]);
                } else if (c2n > nd && n_local) {                    /* raise out */
                    if (o->glb_type[lid(o, nl[0], nl[1], nl[2])] != GIE_VOX_OCCUPIED) {
                        nv->dist_sq = c2n; nv->coc[0] = cg[0]; nv->coc[1] = cg[1]; nv->coc[2] = cg[2];
                        touch(o, nv);
                        nv->wave_layer = -ct;
                        nv->pair_dist = c2n; nv->pair_par = pack_wr(cw[0], cw[1], cw[2]);
                        q_push(fa, ng[0], ng[1], ng[2]);
                    }
                }
            }
        }
        if (ty == GIE_VOX_FREE && has_unknown) o->glb_type[id] = GIE_VOX_FNT;
    }
}

/* ------------------------------------------------------------------ waves (canonical schedule) */
typedef struct { ovox *v; int g[3]; int dist; int64_t par; } prop_rec;

/* Wave A: raise_outside (wave_core.cuh:103-224) in the canonical CHECKERBOARD BLOCK-ROUND schedule: the 8x8x8
 * blocks are coloured by the parity of bx+by+bz, and rounds alternate between the colours, starting with colour 0.  In a
 * round every block of the round's colour that holds pending voxels — seeds, or voxels a neighbouring block proposed to raise
 * in the round before — runs level after level INSIDE itself to exhaustion.  Blocks that run in the same round are never
 * 6-adjacent, so what a block reads of its neighbours is stable while it runs, and the order of the blocks does not matter.
 * Inside a block every entry of a level reads the state at the START of the level; raise proposals are min-resolved per
 * target voxel — (dist, parent) — and applied by the unique winner when the level is over (same block) or when the round is
 * over (a voxel of a neighbouring block, which then is pending in the next round); an entry's own lowering is applied when
 * its level is over.  levels_a counts the rounds that had work, visits_a the voxels taken up. */
static int cmp_i3_block(const void *a, const void *b)
{
    const i3 *p = (const i3 *)a, *q = (const i3 *)b;
    const int kp[3] = { fdiv8(p->z), fdiv8(p->y), fdiv8(p->x) }, kq[3] = { fdiv8(q->z), fdiv8(q->y), fdiv8(q->x) };
    for (int i = 0; i < 3; i++) if (kp[i] != kq[i]) return kp[i] < kq[i] ? -1 : 1;
    return 0;
}
static int same_block(const i3 *p, int x, int y, int z) { return fdiv8(p->x) == fdiv8(x) && fdiv8(p->y) == fdiv8(y) && fdiv8(p->z) == fdiv8(z); }
static int block_colour(int x, int y, int z) { return (fdiv8(x) + fdiv8(y) + fdiv8(z)) & 1; }

/* test instrumentation: how often raise_outside's stale-pair corner (below) was taken since the library was loaded */
int go_debug_stale_pairs = 0;
int go_debug_stale_last[3] = { 0, 0, 0 };                      /* ... and the voxel it was taken for last */
static void wave_a(gie_oracle *o, queue *front, queue *fb)
{
    const int ct = o->map_ct;
    queue pend[2] = { { 0, 0, 0 }, { 0, 0, 0 } };
    for (int i = 0; i < front->n; i++) q_push(&pend[block_colour(front->d[i].x, front->d[i].y, front->d[i].z)], front->d[i].x, front->d[i].y, front->d[i].z);
    q_free(front);
    typedef struct { int lowered; int dist; int coc[3]; int pair_set; int pair_dist; int64_t pair_par; } low_rec;
    for (int h = 0; pend[0].n + pend[1].n > 0; h++) {
        queue cur = pend[h & 1];
        pend[h & 1].d = NULL; pend[h & 1].n = pend[h & 1].cap = 0;
        if (cur.n == 0) continue;
        o->st.levels_a++;
        qsort(cur.d, (size_t)cur.n, sizeof(i3), cmp_i3_block);
        queue xtouched = { 0, 0, 0 };
        for (int b0 = 0; b0 < cur.n;) {
            int e1 = b0;
            while (e1 < cur.n && same_block(&cur.d[b0], cur.d[e1].x, cur.d[e1].y, cur.d[e1].z)) e1++;
            const i3 blk = cur.d[b0];
            queue L = { 0, 0, 0 };
            for (int e = b0; e < e1; e++) q_push(&L, cur.d[e].x, cur.d[e].y, cur.d[e].z);
            b0 = e1;
            while (L.n > 0) {                                             /* one level inside the block */
                o->st.visits_a += L.n;
                low_rec *low = (low_rec *)calloc((size_t)L.n, sizeof(low_rec));
                queue touched = { 0, 0, 0 }, Ln = { 0, 0, 0 };
                /* phase 1: every entry reads the level-start state; writes only proposals */
                for (int e = 0; e < L.n; e++) {
                    const int g[3] = { L.d[e].x, L.d[e].y, L.d[e].z };
                    ovox *c = vox_find(o, g[0], g[1], g[2]);
                    if (!c) continue;
                    if (c->dist_sq > o->cfg.cutoff_grids_sq) continue;
                    const int lc[3] = { c->coc[0], c->coc[1], c->coc[2] };
                    const int lcw[3] = { lc[0] - o->upvt[0], lc[1] - o->upvt[1], lc[2] - o->upvt[2] };
                    int cd = c->dist_sq;
                    low_rec *lr = &low[e];
                    for (int k = 0; k < 6; k++) {
                        const int ng[3] = { g[0] + DIRS[k][0], g[1] + DIRS[k][1], g[2] + DIRS[k][2] };
                        if (in_loc(o, ng[0] - o->pvt[0], ng[1] - o->pvt[1], ng[2] - o->pvt[2])) continue;
                        if (in_whole(o, ng[0] - o->pvt[0], ng[1] - o->pvt[1], ng[2] - o->pvt[2])) continue;   /* tiling: not into another tile's territory */
                        ovox *nv = vox_find(o, ng[0], ng[1], ng[2]);
                        if (!nv) continue;
                        if (nv->vox_type == GIE_VOX_UNKNOWN || invalid_coc_glb(nv->coc) || invalid_dist_glb(o, nv->dist_sq)) continue;
                        if (nv->wave_layer == -ct) continue;
                        if (nv->coc[0] == lc[0] && nv->coc[1] == lc[1] && nv->coc[2] == lc[2]) continue;
                        const int nl[3] = { nv->coc[0] - o->pvt[0], nv->coc[1] - o->pvt[1], nv->coc[2] - o->pvt[2] };
                        if (in_loc(o, nl[0], nl[1], nl[2]) && o->aux[lid(o, nl[0], nl[1], nl[2])] != 0) {
                            const int d = d2i(lc[0], lc[1], lc[2], ng[0], ng[1], ng[2]);
                            const int64_t par = pack_wr(lcw[0], lcw[1], lcw[2]);
                            if (same_block(&blk, ng[0], ng[1], ng[2])) {
                                if (pair_less(d, par, nv->prop_dist, nv->prop_par)) { nv->prop_dist = d; nv->prop_par = par; }
                                q_push(&touched, ng[0], ng[1], ng[2]);
                            } else {
                                if (pair_less(d, par, nv->xprop_dist, nv->xprop_par)) { nv->xprop_dist = d; nv->xprop_par = par; }
                                q_push(&xtouched, ng[0], ng[1], ng[2]);
                            }
                        } else {
                            const int d = d2i(nv->coc[0], nv->coc[1], nv->coc[2], g[0], g[1], g[2]);
                            if (cd > d) {
                                cd = d;
                                lr->lowered = 1; lr->dist = d; lr->coc[0] = nv->coc[0]; lr->coc[1] = nv->coc[1]; lr->coc[2] = nv->coc[2];
                                const int nw[3] = { nv->coc[0] - o->upvt[0], nv->coc[1] - o->upvt[1], nv->coc[2] - o->upvt[2] };
                                /* wave_core.cuh:199-221: distance and obstacle are overwritten first; an obstacle outside the
                                 * wave range then `continue`s -- the pair of an EARLIER direction's lowering (inside the wave
                                 * range) stays, and the voxel stays in frontier B with it ("stale pair": wave B will commit
                                 * that pair over the nearer, un-encodable obstacle) */
                                if (!in_wr(o, nw[0], nw[1], nw[2])) { if (lr->pair_set) { go_debug_stale_pairs++; go_debug_stale_last[0] = g[0]; go_debug_stale_last[1] = g[1]; go_debug_stale_last[2] = g[2]; } continue; }
                                lr->pair_set = 1; lr->pair_dist = d; lr->pair_par = pack_wr(nw[0], nw[1], nw[2]);
                            }
                        }
                    }
                }
                /* phase 2: apply */
                for (int e = 0; e < L.n; e++) {
                    if (!low[e].lowered) continue;
                    ovox *c = vox_find(o, L.d[e].x, L.d[e].y, L.d[e].z);
                    c->dist_sq = low[e].dist; c->coc[0] = low[e].coc[0]; c->coc[1] = low[e].coc[1]; c->coc[2] = low[e].coc[2];
                    touch(o, c);
                    c->wave_layer = 1; c->update_ct = ct;
                    if (low[e].pair_set) {
                        c->pair_dist = low[e].pair_dist; c->pair_par = low[e].pair_par;
                        q_push(fb, L.d[e].x, L.d[e].y, L.d[e].z);
                    }
                }
                for (int i = 0; i < touched.n; i++) {
                    ovox *nv = vox_find(o, touched.d[i].x, touched.d[i].y, touched.d[i].z);
                    if (nv->prop_dist == 0x7fffffff) continue;               /* already applied */
                    int lw[3]; unpack_wr(nv->prop_par, &lw[0], &lw[1], &lw[2]);
                    nv->dist_sq = nv->prop_dist;
                    nv->coc[0] = lw[0] + o->upvt[0]; nv->coc[1] = lw[1] + o->upvt[1]; nv->coc[2] = lw[2] + o->upvt[2];
                    touch(o, nv);
                    nv->wave_layer = -ct; nv->update_ct = -ct;
                    nv->pair_dist = nv->prop_dist; nv->pair_par = nv->prop_par;
                    nv->prop_dist = 0x7fffffff; nv->prop_par = 0;
                    q_push(&Ln, touched.d[i].x, touched.d[i].y, touched.d[i].z);
                }
                free(low); q_free(&touched);
                q_free(&L); L = Ln;
            }
            q_free(&L);
        }
        /* end of the round: the raises proposed into the neighbouring blocks (none of which ran in this round) */
        for (int i = 0; i < xtouched.n; i++) {
            ovox *nv = vox_find(o, xtouched.d[i].x, xtouched.d[i].y, xtouched.d[i].z);
            if (nv->xprop_dist == 0x7fffffff) continue;
            int lw[3]; unpack_wr(nv->xprop_par, &lw[0], &lw[1], &lw[2]);
            nv->dist_sq = nv->xprop_dist;
            nv->coc[0] = lw[0] + o->upvt[0]; nv->coc[1] = lw[1] + o->upvt[1]; nv->coc[2] = lw[2] + o->upvt[2];
            touch(o, nv);
            nv->wave_layer = -ct; nv->update_ct = -ct;
            nv->pair_dist = nv->xprop_dist; nv->pair_par = nv->xprop_par;
            nv->xprop_dist = 0x7fffffff; nv->xprop_par = 0;
            q_push(&pend[(h & 1) ^ 1], xtouched.d[i].x, xtouched.d[i].y, xtouched.d[i].z);
        }
        q_free(&xtouched);
        q_free(&cur);
    }
}

/* first face of the volume a boundary voxel lies on → dense slot in a 2(XY+YZ+XZ) table;
 * only used to de-duplicate/resolve writers, any injective map works */
static void dedupe_global(gie_oracle *o, queue *q)
{   /* keep the first occurrence of every coordinate (set semantics) using prop_par as a mark */
    int m = 0;
    for (int i = 0; i < q->n; i++) {
        ovox *v = vox_find(o, q->d[i].x, q->d[i].y, q->d[i].z);
        if (!v) continue;
        if (v->prop_par == -7) continue;
        v->prop_par = -7; q->d[m++] = q->d[i];
    }
    q->n = m;
    for (int i = 0; i < q->n; i++) vox_find(o, q->d[i].x, q->d[i].y, q->d[i].z)->prop_par = 0;
}

/* Wave B: lower_outside (wave_core.cuh:229-350) in the canonical BLOCK-ROUND schedule (DESIGN.md "Canonical wave schedule", the
 * twin of wave C's tile rounds): the hashed voxels are taken 8x8x8 block by block.  In a round every block that holds pending
 * voxels — the seeds, or voxels that received a proposal from another block in the round before — runs a level-synchronous
 * BFS INSIDE itself to exhaustion (the reference's BFS_in_block idea, wave_core.cuh:395-469); what it proposes to voxels of OTHER blocks
 * is collected — minimum (dist, parent) per voxel — and applied at the end of the round, where a voxel that improved joins the next
 * round.  Blocks do not see each other inside a round, so their order is irrelevant.  What the wave proposes to voxels INSIDE the
 * volume is collected over the whole wave — minimum per voxel — and stored when the wave is over (the reference's plain,
 * unconditional store, :336-346: any of the proposers may be the last writer; the canonical one is the smallest).
 * levels_b counts rounds, visits_b the voxels taken up (expanded or cut off). */
static void wave_b(gie_oracle *o, queue *front, queue *fc)
{
    dedupe_global(o, front);
    queue cur = *front;
    front->d = NULL; front->n = front->cap = 0;
    queue inl = { 0, 0, 0 };                                  /* voxels inside the volume that received a proposal */
    typedef struct { int active; int coc[3]; int64_t par; } snap;
    while (cur.n > 0) {
        o->st.levels_b++;
        qsort(cur.d, (size_t)cur.n, sizeof(i3), cmp_i3_block);            /* group the pending voxels by block */
        queue xtouched = { 0, 0, 0 };
        for (int b0 = 0; b0 < cur.n;) {
            int e1 = b0;
            while (e1 < cur.n && same_block(&cur.d[b0], cur.d[e1].x, cur.d[e1].y, cur.d[e1].z)) e1++;
            const i3 blk = cur.d[b0];
            queue L = { 0, 0, 0 };
            for (int e = b0; e < e1; e++) q_push(&L, cur.d[e].x, cur.d[e].y, cur.d[e].z);
            b0 = e1;
            while (L.n > 0) {                                             /* one level inside the block */
                o->st.visits_b += L.n;
                snap *sn = (snap *)calloc((size_t)L.n, sizeof(snap));
                /* the cut-off looks at the distance stored BEFORE the pair is committed (:262-266); then commit */
                for (int e = 0; e < L.n; e++) {
                    ovox *c = vox_find(o, L.d[e].x, L.d[e].y, L.d[e].z);
                    if (!c) continue;
                    if (c->dist_sq > o->cfg.cutoff_grids_sq) continue;
                    c->wave_layer = BLACK;
                    int cw[3]; unpack_wr(c->pair_par, &cw[0], &cw[1], &cw[2]);
                    c->coc[0] = cw[0] + o->upvt[0]; c->coc[1] = cw[1] + o->upvt[1]; c->coc[2] = cw[2] + o->upvt[2];
                    c->dist_sq = c->pair_dist;
                    touch(o, c);
                    sn[e].active = 1; sn[e].par = c->pair_par;
                    sn[e].coc[0] = c->coc[0]; sn[e].coc[1] = c->coc[1]; sn[e].coc[2] = c->coc[2];
                }
                queue touched = { 0, 0, 0 }, Ln = { 0, 0, 0 };
                for (int e = 0; e < L.n; e++) {
                    if (!sn[e].active) continue;
                    const int g[3] = { L.d[e].x, L.d[e].y, L.d[e].z };
                    for (int k = 0; k < 6; k++) {
                        const int ng[3] = { g[0] + DIRS[k][0], g[1] + DIRS[k][1], g[2] + DIRS[k][2] };
                        const int nb[3] = { ng[0] - o->pvt[0], ng[1] - o->pvt[1], ng[2] - o->pvt[2] };
                        const int cand = d2i(sn[e].coc[0], sn[e].coc[1], sn[e].coc[2], ng[0], ng[1], ng[2]);
                        if (!in_loc(o, nb[0], nb[1], nb[2])) {
                            if (in_whole(o, nb[0], nb[1], nb[2])) continue;   /* tiling: not into another tile's territory */
                            ovox *nv = vox_find(o, ng[0], ng[1], ng[2]);
                            if (!nv) continue;
                            if (nv->vox_type == GIE_VOX_UNKNOWN) continue;
                            if (invalid_coc_glb(nv->coc)) continue;
                            if (cand >= o->empty_value) continue;
                            if (same_block(&blk, ng[0], ng[1], ng[2])) {
                                if (pair_less(cand, sn[e].par, nv->prop_dist, nv->prop_par)) { nv->prop_dist = cand; nv->prop_par = sn[e].par; }
                                q_push(&touched, ng[0], ng[1], ng[2]);
                            } else {
                                if (pair_less(cand, sn[e].par, nv->xprop_dist, nv->xprop_par)) { nv->xprop_dist = cand; nv->xprop_par = sn[e].par; }
                                q_push(&xtouched, ng[0], ng[1], ng[2]);
                            }
                        } else {
                            const int nid = lid(o, nb[0], nb[1], nb[2]);
                            {   /* tiling: an obstacle inside the whole volume but not in this tile is its owner's to vouch for (see obtain_frontiers) */
                                const int cl3[3] = { sn[e].coc[0] - o->pvt[0], sn[e].coc[1] - o->pvt[1], sn[e].coc[2] - o->pvt[2] };
                                if (in_whole(o, cl3[0], cl3[1], cl3[2]) && !in_loc(o, cl3[0], cl3[1], cl3[2])) continue;
                            }
                            if (o->aux[nid] > cand) {
                                if (o->lprop_dist[nid] == 0x7fffffff) q_push(&inl, nb[0], nb[1], nb[2]);
                                if (pair_less(cand, sn[e].par, o->lprop_dist[nid], o->lprop_par[nid])) { o->lprop_dist[nid] = cand; o->lprop_par[nid] = sn[e].par; }
                            }
                        }
                    }
                }
                /* strict improvement over the pair at the start of the level (id_atomicMin, wave_core.cuh:9-22) */
                for (int i = 0; i < touched.n; i++) {
                    ovox *nv = vox_find(o, touched.d[i].x, touched.d[i].y, touched.d[i].z);
                    if (nv->prop_dist == 0x7fffffff) continue;
                    if (nv->pair_dist > nv->prop_dist) {
                        nv->pair_dist = nv->prop_dist; nv->pair_par = nv->prop_par;
                        q_push(&Ln, touched.d[i].x, touched.d[i].y, touched.d[i].z);
                    }
                    nv->prop_dist = 0x7fffffff; nv->prop_par = 0;
                }
                q_free(&touched); free(sn);
                q_free(&L); L = Ln;
            }
            q_free(&L);
        }
        /* end of the round: what crossed a block border */
        queue next = { 0, 0, 0 };
        for (int i = 0; i < xtouched.n; i++) {
            ovox *nv = vox_find(o, xtouched.d[i].x, xtouched.d[i].y, xtouched.d[i].z);
            if (nv->xprop_dist == 0x7fffffff) continue;
            if (nv->pair_dist > nv->xprop_dist) {
                nv->pair_dist = nv->xprop_dist; nv->pair_par = nv->xprop_par;
                q_push(&next, xtouched.d[i].x, xtouched.d[i].y, xtouched.d[i].z);
            }
            nv->xprop_dist = 0x7fffffff; nv->xprop_par = 0;
        }
        q_free(&xtouched);
        q_free(&cur); cur = next;
    }
    q_free(&cur);
    /* the stores into the volume (the reference stores the pair unconditionally, :336-346) */
    for (int i = 0; i < inl.n; i++) {
        const int nid = lid(o, inl.d[i].x, inl.d[i].y, inl.d[i].z);
        o->pair_dist[nid] = o->lprop_dist[nid]; o->pair_par[nid] = o->lprop_par[nid];
        o->lprop_dist[nid] = 0x7fffffff; o->lprop_par[nid] = 0;
        if (o->wave_layer[nid] != 1) q_push(fc, inl.d[i].x, inl.d[i].y, inl.d[i].z);
    }
    q_free(&inl);
}

/* Wave C: lower_inside (wave_core.cuh:353-393). */
/* lower_inside in the canonical TILE-ROUND schedule (DESIGN.md "Canonical wave schedule"): the volume is cut into its 8x8x8
 * tiles; in a round every tile that holds pending voxels runs a level-synchronous BFS INSIDE itself to exhaustion (like the
 * reference's BFS_in_block, wave_core.cuh:395-469), and what it proposes to voxels of OTHER tiles is collected — minimum
 * (dist, parent) per voxel — and applied at the end of the round, where a voxel that improved joins the next round.  Tiles
 * do not see each other inside a round, so their order is irrelevant.  levels_c counts rounds, visits_c expansions. */
static int tile_of(const gie_oracle *o, int x, int y, int z)
{ return ((z >> 3) * ((o->Y + 7) >> 3) + (y >> 3)) * ((o->X + 7) >> 3) + (x >> 3); }
static int cmp_i3_tile(const void *a, const void *b, void *ctx)
{
    const gie_oracle *o = (const gie_oracle *)ctx;
    const i3 *p = (const i3 *)a, *q = (const i3 *)b;
    const int tp = tile_of(o, p->x, p->y, p->z), tq = tile_of(o, q->x, q->y, q->z);
    return tp < tq ? -1 : (tp > tq ? 1 : 0);
}
static void wave_c(gie_oracle *o, queue *front)
{
    /* set semantics for the seed list (lower_outside may push a voxel more than once) */
    {
        int m = 0;
        for (int i = 0; i < front->n; i++) {
            const int nid = lid(o, front->d[i].x, front->d[i].y, front->d[i].z);
            if (o->lprop_par[nid] == -7) continue;
            o->lprop_par[nid] = -7; front->d[m++] = front->d[i];
        }
        front->n = m;
        for (int i = 0; i < front->n; i++) o->lprop_par[lid(o, front->d[i].x, front->d[i].y, front->d[i].z)] = 0;
    }
    queue cur = *front;
    front->d = NULL; front->n = front->cap = 0;
    /* proposals across tile borders of the running round */
    int32_t *xd = (int32_t *)malloc(sizeof(int32_t) * (size_t)o->N);
    int64_t *xp = (int64_t *)calloc((size_t)o->N, sizeof(int64_t));
    for (int i = 0; i < o->N; i++) xd[i] = 0x7fffffff;
    while (cur.n > 0) {
        o->st.levels_c++;
        qsort_r(cur.d, (size_t)cur.n, sizeof(i3), cmp_i3_tile, o);       /* group the pending voxels by tile */
        queue xtouched = { 0, 0, 0 };
        for (int b = 0; b < cur.n;) {
            const int T = tile_of(o, cur.d[b].x, cur.d[b].y, cur.d[b].z);
            int e1 = b;
            while (e1 < cur.n && tile_of(o, cur.d[e1].x, cur.d[e1].y, cur.d[e1].z) == T) e1++;
            queue L = { 0, 0, 0 };
            for (int e = b; e < e1; e++) q_push(&L, cur.d[e].x, cur.d[e].y, cur.d[e].z);
            b = e1;
            while (L.n > 0) {                                             /* one level inside the tile */
                o->st.visits_c += L.n;
                int64_t *par = (int64_t *)malloc(sizeof(int64_t) * (size_t)L.n);
                for (int e = 0; e < L.n; e++) {
                    const int id = lid(o, L.d[e].x, L.d[e].y, L.d[e].z);
                    o->wave_layer[id] = BLACK;
                    par[e] = o->pair_par[id];
                }
                queue touched = { 0, 0, 0 }, Ln = { 0, 0, 0 };
                for (int e = 0; e < L.n; e++) {
                    int cw[3]; unpack_wr(par[e], &cw[0], &cw[1], &cw[2]);
                    const int cl[3] = { cw[0] + o->upvt[0] - o->pvt[0], cw[1] + o->upvt[1] - o->pvt[1], cw[2] + o->upvt[2] - o->pvt[2] };
                    for (int k = 0; k < 6; k++) {
                        const int nx = L.d[e].x + DIRS[k][0], ny = L.d[e].y + DIRS[k][1], nz = L.d[e].z + DIRS[k][2];
                        if (!in_loc(o, nx, ny, nz)) continue;
                        const int nid = lid(o, nx, ny, nz);
                        const int cand = d2i(cl[0], cl[1], cl[2], nx, ny, nz);
                        if (cand >= o->empty_value) continue;
                        if (tile_of(o, nx, ny, nz) == T) {
                            if (pair_less(cand, par[e], o->lprop_dist[nid], o->lprop_par[nid])) { o->lprop_dist[nid] = cand; o->lprop_par[nid] = par[e]; }
                            q_push(&touched, nx, ny, nz);
                        } else {
                            if (pair_less(cand, par[e], xd[nid], xp[nid])) { xd[nid] = cand; xp[nid] = par[e]; }
                            q_push(&xtouched, nx, ny, nz);
                        }
                    }
                }
                for (int i = 0; i < touched.n; i++) {
                    const int nid = lid(o, touched.d[i].x, touched.d[i].y, touched.d[i].z);
                    if (o->lprop_dist[nid] == 0x7fffffff) continue;
                    if (o->pair_dist[nid] > o->lprop_dist[nid]) {
                        o->pair_dist[nid] = o->lprop_dist[nid]; o->pair_par[nid] = o->lprop_par[nid];
                        o->wave_layer[nid] = GRAY0;
                        q_push(&Ln, touched.d[i].x, touched.d[i].y, touched.d[i].z);
                    }
                    o->lprop_dist[nid] = 0x7fffffff; o->lprop_par[nid] = 0;
                }
                q_free(&touched); free(par);
                q_free(&L); L = Ln;
            }
            q_free(&L);
        }
        /* end of the round: what crossed a tile border */
        queue next = { 0, 0, 0 };
        for (int i = 0; i < xtouched.n; i++) {
            const int nid = lid(o, xtouched.d[i].x, xtouched.d[i].y, xtouched.d[i].z);
            if (xd[nid] == 0x7fffffff) continue;
            if (o->pair_dist[nid] > xd[nid]) {
                o->pair_dist[nid] = xd[nid]; o->pair_par[nid] = xp[nid];
                o->wave_layer[nid] = GRAY1;
                q_push(&next, xtouched.d[i].x, xtouched.d[i].y, xtouched.d[i].z);
            }
            xd[nid] = 0x7fffffff; xp[nid] = 0;
        }
        q_free(&xtouched);
        q_free(&cur); cur = next;
    }
    q_free(&cur);
    free(xd); free(xp);
}

/* UpdateHashBatch, unify_helper.cuh:448-523 */
static void update_hash_batch(gie_oracle *o)
{
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        const int id = lid(o, x, y, z);
        const int8_t ty = o->glb_type[id];
        if (ty == GIE_VOX_UNKNOWN) continue;
        if (o->pair_dist[id] == o->empty_value) {
            if (o->pair_par[id] == PAR_NONE) o->edt[id] = (float)o->max_loc_dist_sq;   /* App. B #7 */
            continue;
        }
        ovox *v = vox_find(o, x + o->pvt[0], y + o->pvt[1], z + o->pvt[2]);
        if (!v) continue;
        int cw[3]; unpack_wr(o->pair_par[id], &cw[0], &cw[1], &cw[2]);
        const int nc[3] = { cw[0] + o->upvt[0], cw[1] + o->upvt[1], cw[2] + o->upvt[2] };
        if (v->dist_sq != o->pair_dist[id] || v->coc[0] != nc[0] || v->coc[1] != nc[1] || v->coc[2] != nc[2] ||
            (ty == GIE_VOX_FNT && v->vox_type != GIE_VOX_FNT)) touch(o, v);
        v->coc[0] = nc[0]; v->coc[1] = nc[1]; v->coc[2] = nc[2];
        v->dist_sq = o->pair_dist[id];
        o->edt[id] = sqrtf((float)o->pair_dist[id]);
        v->pair_dist = o->pair_dist[id]; v->pair_par = o->pair_par[id];
        if (ty == GIE_VOX_FNT) v->vox_type = GIE_VOX_FNT;
    }
}

/* union of all tiles in local coordinates (= the local volume without tiling) */
static int in_whole(const gie_oracle *o, int x, int y, int z)
{ return x >= o->whole_lo[0] && x < o->whole_hi[0] && y >= o->whole_lo[1] && y < o->whole_hi[1] && z >= o->whole_lo[2] && z < o->whole_hi[2]; }

static int merge_rest(gie_oracle *o);
/* GlbHashMap::mergeNewObsv, glb_hash_map.cu:146-207 */
int go_merge(gie_oracle *o)
{
    mark_limited_observe(o);
    return merge_rest(o);
}
/* the two halves of a tiled run's merge (include/gie.h): Mark + commit of the Mark-time pairs | the rest */
int go_merge_begin_tiled(gie_oracle *o) { mark_limited_observe(o); update_hash_batch(o); return 0; }
int go_merge_end(gie_oracle *o) { return merge_rest(o); }
static int merge_rest(gie_oracle *o)
{
    queue fa = { 0, 0, 0 }, fb = { 0, 0, 0 }, fc = { 0, 0, 0 };
    obtain_frontiers(o, &fa, &fb, &fc);
    o->st.seeds_a = fa.n; o->st.seeds_b = fb.n; o->st.seeds_c = fc.n;
    if (!o->cfg.fast_mode) {
        wave_a(o, &fa, &fb);
        o->st.front_b = fb.n;
        wave_b(o, &fb, &fc);
    }
    o->st.front_c = fc.n;
    wave_c(o, &fc);
    update_hash_batch(o);
    o->tot_vis[0] += o->st.visits_a; o->tot_vis[1] += o->st.visits_b; o->tot_vis[2] += o->st.visits_c;
    q_free(&fa); q_free(&fb); q_free(&fc);
    return 0;
}

int go_step(gie_oracle *o) { go_fuse(o); go_batch_edt(o); go_merge(o); return 0; }

/* ------------------------------------------------------------------ tiling: halo exchange + refinement
 * (no reference counterpart — the reference is single-GPU; semantics documented in include/gie.h) */
static void face_coord(const gie_oracle *o, int face, int i, int depth_off, int *x, int *y, int *z)
{
    const int axis = face >> 1, hi = face & 1;
    const int sz[3] = { o->X, o->Y, o->Z };
    const int along = hi ? sz[axis] - 1 + depth_off : -depth_off;
    if (axis == 0) { *x = along; *y = i % o->Y; *z = i / o->Y; }
    else if (axis == 1) { *x = i % o->X; *y = along; *z = i / o->X; }
    else { *x = i % o->X; *y = i / o->X; *z = along; }
}
int go_set_tile(gie_oracle *o, const int32_t off[3], const int32_t whole[3])
{ for (int i = 0; i < 3; i++) { o->next_off[i] = off[i]; o->next_whole[i] = whole[i]; } return 0; }
int go_halo_count(gie_oracle *o, int face)
{ const int axis = face >> 1; return axis == 0 ? o->Y * o->Z : (axis == 1 ? o->X * o->Z : o->X * o->Y); }

int go_halo_export(gie_oracle *o, int face, gie_halo_voxel *out)
{
    const int n = go_halo_count(o, face);
    for (int i = 0; i < n; i++) {
        int x, y, z;
        face_coord(o, face, i, 0, &x, &y, &z);
        ovox *v = vox_find(o, x + o->pvt[0], y + o->pvt[1], z + o->pvt[2]);
        memset(&out[i], 0, sizeof(out[i]));
        if (!v || v->vox_type == GIE_VOX_UNKNOWN) {
            out[i].vox_type = GIE_VOX_UNKNOWN; out[i].dist_sq = o->empty_value;
            out[i].coc[0] = out[i].coc[1] = out[i].coc[2] = GIE_EMPTY_VALUE;
        } else {
            out[i].vox_type = v->vox_type; out[i].dist_sq = v->dist_sq; out[i].occ_val = v->occ_val;
            out[i].coc[0] = v->coc[0]; out[i].coc[1] = v->coc[1]; out[i].coc[2] = v->coc[2];
        }
    }
    return 0;
}

int go_halo_import(gie_oracle *o, int face, const gie_halo_voxel *in)
{
    const int n = go_halo_count(o, face);
    for (int i = 0; i < n; i++) {
        if (in[i].vox_type == GIE_VOX_UNKNOWN) continue;
        int x, y, z;
        face_coord(o, face, i, 1, &x, &y, &z);
        const int gx = x + o->pvt[0], gy = y + o->pvt[1], gz = z + o->pvt[2];
        oblock *b = blk_get_or_alloc(o, fdiv8(gx), fdiv8(gy), fdiv8(gz));
        ovox *v = &b->v[vox_in_blk(gx, gy, gz)];
        v->vox_type = in[i].vox_type; v->occ_val = in[i].occ_val; v->dist_sq = in[i].dist_sq;
        v->coc[0] = in[i].coc[0]; v->coc[1] = in[i].coc[1]; v->coc[2] = in[i].coc[2];
        touch(o, v);
    }
    return 0;
}

int go_refine(gie_oracle *o, int32_t *seeded)
{
    queue fc = { 0, 0, 0 };
    for (int z = 0; z < o->Z; z++) for (int y = 0; y < o->Y; y++) for (int x = 0; x < o->X; x++) {
        if (!(x == 0 || y == 0 || z == 0 || x == o->X - 1 || y == o->Y - 1 || z == o->Z - 1)) continue;
        const int id = lid(o, x, y, z);
        if (o->glb_type[id] == GIE_VOX_UNKNOWN) continue;
        const int cd = o->pair_dist[id];
        int hit = 0, sd = 0; int64_t sp = 0;
        for (int k = 0; k < 6; k++) {
            const int nx = x + DIRS[k][0], ny = y + DIRS[k][1], nz = z + DIRS[k][2];
            if (in_loc(o, nx, ny, nz)) continue;
            ovox *nv = vox_find(o, nx + o->pvt[0], ny + o->pvt[1], nz + o->pvt[2]);
            if (!nv || nv->vox_type == GIE_VOX_UNKNOWN) continue;
            if (invalid_dist_glb(o, nv->dist_sq) || invalid_coc_glb(nv->coc)) continue;
            const int nw[3] = { nv->coc[0] - o->upvt[0], nv->coc[1] - o->upvt[1], nv->coc[2] - o->upvt[2] };
            const int nl[3] = { nv->coc[0] - o->pvt[0], nv->coc[1] - o->pvt[1], nv->coc[2] - o->pvt[2] };
            /* a ghost vouches for an obstacle anywhere outside this tile, a remembered voxel outside the whole volume only for
             * one outside the whole volume (obtain_frontiers) */
            const int hidden = in_whole(o, nx, ny, nz) ? !in_loc(o, nl[0], nl[1], nl[2]) : !in_whole(o, nl[0], nl[1], nl[2]);
            if (!hidden || !in_wr(o, nw[0], nw[1], nw[2])) continue;
            const int d = d2i(nl[0], nl[1], nl[2], x, y, z);
            if (d < cd) { sd = d; sp = pack_wr(nw[0], nw[1], nw[2]); hit = 1; }
        }
        if (hit) { o->pair_dist[id] = sd; o->pair_par[id] = sp; q_push(&fc, x, y, z); }
    }
    if (seeded) *seeded = fc.n;
    o->st.visits_c = 0; o->st.levels_c = 0;
    o->st.front_c = fc.n;
    wave_c(o, &fc);
    update_hash_batch(o);
    o->tot_vis[2] += o->st.visits_c;
    q_free(&fc);
    return 0;
}

/* ------------------------------------------------------------------ readers */
int go_read_local(gie_oracle *o, float *edt, int8_t *type, int32_t *dist_sq, int32_t *coc_xyz)
{
    if (edt) memcpy(edt, o->edt, sizeof(float) * (size_t)o->N);
    if (type) memcpy(type, o->glb_type, (size_t)o->N);
    if (dist_sq) memcpy(dist_sq, o->pair_dist, sizeof(int32_t) * (size_t)o->N);
    if (coc_xyz) for (int i = 0; i < o->N; i++) {
        if (o->pair_par[i] == PAR_NONE || o->pair_dist[i] >= o->empty_value) {
            coc_xyz[3 * i] = coc_xyz[3 * i + 1] = coc_xyz[3 * i + 2] = GIE_EMPTY_VALUE;
        } else {
            int w[3]; unpack_wr(o->pair_par[i], &w[0], &w[1], &w[2]);
            coc_xyz[3 * i] = w[0] + o->upvt[0]; coc_xyz[3 * i + 1] = w[1] + o->upvt[1]; coc_xyz[3 * i + 2] = w[2] + o->upvt[2];
        }
    }
    return 0;
}
int go_read_ogm(gie_oracle *o, int8_t *inst_type, int32_t *ray_count)
{
    if (inst_type) memcpy(inst_type, o->inst_type, (size_t)o->N);
    if (ray_count) memcpy(ray_count, o->ray_count, sizeof(int32_t) * (size_t)o->N);
    return 0;
}
int go_read_batch_edt(gie_oracle *o, int32_t *dist_sq, int32_t *coc)
{
    if (dist_sq) memcpy(dist_sq, o->bdist, sizeof(int32_t) * (size_t)o->N);
    if (coc) for (int i = 0; i < o->N; i++) {
        const int bad = o->bcoc[3 * i] >= o->max_width || o->bcoc[3 * i + 1] >= o->max_width || o->bcoc[3 * i + 2] >= o->max_width;
        for (int k = 0; k < 3; k++) coc[3 * i + k] = bad ? -1 : o->bcoc[3 * i + k];
    }
    return 0;
}
/* LocMap::convertCostMap (local_batch.h:382-391): d = edt, o = (bool)type -- SeenDist.o is a `bool` assigned from a
 * `char` (local_batch.h:19-24,389), so it is 1 for every known type and 0 for UNKNOWN --, s untouched (App. B #9) */
int go_read_costmap(gie_oracle *o, gie_seendist *payload, gie_costmap_hdr *hdr)
{
    if (payload) for (int i = 0; i < o->N; i++) { payload[i].d = o->edt[i]; payload[i].s = 0; payload[i].o = (uint8_t)(o->glb_type[i] != 0); payload[i].pad[0] = payload[i].pad[1] = 0; }
    if (hdr) {
        hdr->x_size = o->X; hdr->y_size = o->Y; hdr->z_size = o->Z;
        hdr->x_origin = o->msg_origin[0]; hdr->y_origin = o->msg_origin[1]; hdr->z_origin = o->msg_origin[2];
        hdr->width = o->cfg.voxel_width; hdr->type = 1; hdr->pad[0] = hdr->pad[1] = hdr->pad[2] = 0;
    }
    return 0;
}
int go_query_global(gie_oracle *o, const int32_t *xyz, int n, gie_voxel *out)
{
    for (int i = 0; i < n; i++) {
        ovox *v = vox_find(o, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        out[i].pad = 0;
        if (!v) { out[i].occ_val = 0; out[i].vox_type = 0; out[i].dist_sq = o->empty_value; out[i].coc[0] = out[i].coc[1] = out[i].coc[2] = EMPTY_KEY_C; }
        else { out[i].occ_val = v->occ_val; out[i].vox_type = v->vox_type; out[i].dist_sq = v->dist_sq; out[i].coc[0] = v->coc[0]; out[i].coc[1] = v->coc[1]; out[i].coc[2] = v->coc[2]; }
    }
    return 0;
}
int go_get_stats(gie_oracle *o, gie_frame_stats *s)
{
    *s = o->st; s->blocks_total = o->nblocks;
    s->total_visits_a = o->tot_vis[0]; s->total_visits_b = o->tot_vis[1]; s->total_visits_c = o->tot_vis[2];
    return 0;
}
/* GlbHashMap::streamPipeline + streamD2H (glb_hash_map.cu:209-247): the blocks flagged since the
 * last call, each as its key and its 512 voxels in the reference's in-block order
 * (get_voxID_in_VB, voxmap_utils.cuh:104-109: x*64 + y*8 + z).  Blocks come in slot order; a
 * NULL output pointer only counts. */
int go_stream_enable(gie_oracle *o, int on) { o->track = on != 0; return 0; }
int go_stream_changed(gie_oracle *o, int32_t *keys, gie_voxel *vox, int max_blocks, int32_t *n_changed)
{
    int n = 0, w = 0;
    for (int s = 0; s < o->nblocks; s++) {
        oblock *b = o->blocks[s];
        if (!b->dirty) continue;
        n++;
        if (!keys || !vox || w >= max_blocks) continue;
        keys[3 * w] = b->key[0]; keys[3 * w + 1] = b->key[1]; keys[3 * w + 2] = b->key[2];
        for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
            const ovox *v = &b->v[vox_in_blk(x, y, z)];
            gie_voxel *d = &vox[(size_t)w * VBSZ + (size_t)(x * 64 + y * 8 + z)];
            d->occ_val = v->occ_val; d->vox_type = v->vox_type; d->pad = 0; d->dist_sq = v->dist_sq;
            d->coc[0] = v->coc[0]; d->coc[1] = v->coc[1]; d->coc[2] = v->coc[2];
        }
        b->dirty = 0;
        w++;
    }
    if (n_changed) *n_changed = n;
    return 0;
}
int go_get_pivot(gie_oracle *o, int32_t p[3]) { p[0] = o->pvt[0]; p[1] = o->pvt[1]; p[2] = o->pvt[2]; return 0; }

/* ------------------------------------------------------------------ ground truth for the EDT stage */
/* Exact EDT by definition: for every voxel the minimum squared distance to any occupied voxel
 * (occ[i] != 0).  out = INT32_MAX where there is no obstacle.  O(N*M): small grids only. */
int go_brute_force_edt(const int8_t *occ, int X, int Y, int Z, int32_t *out)
{
    const int N = X * Y * Z;
    int *ox = (int *)malloc(sizeof(int) * (size_t)N), *oy = (int *)malloc(sizeof(int) * (size_t)N), *oz = (int *)malloc(sizeof(int) * (size_t)N);
    int m = 0;
    for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) for (int x = 0; x < X; x++)
        if (occ[(z * Y + y) * X + x]) { ox[m] = x; oy[m] = y; oz[m] = z; m++; }
    for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) for (int x = 0; x < X; x++) {
        int best = 0x7fffffff;
        for (int i = 0; i < m; i++) {
            const int dx = x - ox[i], dy = y - oy[i], dz = z - oz[i];
            const int d = dx * dx + dy * dy + dz * dz;
            if (d < best) best = d;
        }
        out[(z * Y + y) * X + x] = best;
    }
    free(ox); free(oy); free(oz);
    return m;
}

/* Batch EDT alone on a caller-supplied type grid (used to pin phase 1-3 against brute force
 * and as the timed CPU EDT baseline). */
int go_edt_only(gie_oracle *o, const int8_t *glb_type, int32_t *dist_sq, int32_t *coc)
{
    memcpy(o->glb_type, glb_type, (size_t)o->N);
    go_batch_edt(o);
    return go_read_batch_edt(o, dist_sq, coc);
}
