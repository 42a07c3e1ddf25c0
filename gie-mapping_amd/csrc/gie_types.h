/*
 * gie_types.h — device-side data layout of the MI355X map-update path.
 *
 * Local volume (dense, x fastest, N = X*Y*Z; counterpart of LocMap's arrays,
 * include/map_structure/local_batch.h:541-561) and the block-sparse global map (8x8x8 voxel
 * blocks, counterpart of GlbVoxel/VoxelBlock + vhashing, include/par_wave/voxmap_utils.cuh:29-48,
 * include/vox_hash/vhashing.h) laid out as structure-of-arrays planes so that every sweep
 * touches only the fields it needs.
 *
 * All functions that are plain per-voxel logic are GIE_HD so that the test-only host emulation
 * (tests/emu) can run them; nothing in the product library ever executes them on the CPU.
 */
#ifndef GIE_TYPES_H
#define GIE_TYPES_H

#include <stdint.h>
#include "../../include/gie.h"
#include "../../include/gie_math.h"

#define GIE_VB 8
#define GIE_VBSZ 512
/* address of a voxel of the global map: slot * 512 + in-block index.  64 bits: a pool beyond 2^31 / 512 = 4.19 M blocks (288 GB hold
 * about 15 M) is addressed like a small one */
typedef long long gie_vaddr;

/* ---- 64-bit (dist, parent) pair: [63:42] dist (22 bit) | [41] new-in-this-level | [40:0] parent
 * parent = closest obstacle in wave-range coordinates, x | y<<14 | z<<28 (x,y < 16384, z < 8192).
 * One native 64-bit atomicMin gives: lower dist wins; at equal dist a value that was there at
 * the start of the BFS level beats candidates of the level (the reference's strict '>' in
 * id_atomicMin, wave_core.cuh:9-22); among candidates of one level the smaller parent id wins
 * (canonical tie rule). */
#define GIE_PAIR_DIST_SHIFT 42
#define GIE_PAIR_NEW (1ull << 41)
#define GIE_PAIR_PAR_MASK ((1ull << 41) - 1ull)
#define GIE_PAR_NONE GIE_PAIR_PAR_MASK /* the reference's 0xffffffff "see nothing" id */
#define GIE_NOPROP 0xffffffffffffffffull

GIE_HD uint64_t gie_pair_make(int dist, uint64_t par) { return ((uint64_t)(uint32_t)dist << GIE_PAIR_DIST_SHIFT) | (par & GIE_PAIR_PAR_MASK); }
GIE_HD int gie_pair_dist(uint64_t p) { return (int)(p >> GIE_PAIR_DIST_SHIFT); }
GIE_HD uint64_t gie_pair_par(uint64_t p) { return p & GIE_PAIR_PAR_MASK; }
GIE_HD uint64_t gie_pack_wr(int x, int y, int z) { return (uint64_t)(uint32_t)x | ((uint64_t)(uint32_t)y << 14) | ((uint64_t)(uint32_t)z << 28); }
GIE_HD void gie_unpack_wr(uint64_t id, int *x, int *y, int *z)
{
    *x = (int)(id & 0x3fff); *y = (int)((id >> 14) & 0x3fff); *z = (int)((id >> 28) & 0x1fff);
    if ((id & GIE_PAIR_PAR_MASK) == GIE_PAR_NONE) *z = 0x3fff; /* keep NONE outside every wave range */
}

/* ---- global coordinate packed in 64 bits (21 bits per axis, offset 2^20): block keys, queue
 * entries and the stored closest obstacle coc_glb.  EMPTY_KEY (999999) is representable. */
#define GIE_CRD_OFF (1 << 20)
GIE_HD uint64_t gie_pack_crd(int x, int y, int z)
{ return (uint64_t)(uint32_t)(x + GIE_CRD_OFF) | ((uint64_t)(uint32_t)(y + GIE_CRD_OFF) << 21) | ((uint64_t)(uint32_t)(z + GIE_CRD_OFF) << 42); }
GIE_HD void gie_unpack_crd(uint64_t k, int *x, int *y, int *z)
{ *x = (int)(k & 0x1fffff) - GIE_CRD_OFF; *y = (int)((k >> 21) & 0x1fffff) - GIE_CRD_OFF; *z = (int)((k >> 42) & 0x1fffff) - GIE_CRD_OFF; }
#define GIE_KEY_EMPTY 0xffffffffffffffffull
#define GIE_KEY_TOMB 0xfffffffffffffffeull    /* hash cell of an erased block (gie_config.retain_radius_blocks): a probe walks past it, an insert may take it */

/* batch-EDT closest obstacle in local coordinates, 10 bits per axis (dims <= 1024) */
#define GIE_BCOC_NONE 0xffffffffu
GIE_HD uint32_t gie_pack_bcoc(int x, int y, int z) { return (uint32_t)x | ((uint32_t)y << 10) | ((uint32_t)z << 20); }
/* batch dist² of local voxel (x,y,z) from its packed closest obstacle — the reference's `_aux`
 * plane after batchEDTUpdate is never stored: it is a function of `_coc_idx_aux` */
GIE_HD int gie_bcoc_dist(uint32_t bc, int x, int y, int z, int none_value)
{
    if (bc == GIE_BCOC_NONE) return none_value;
    const int dx = x - (int)(bc & 1023u), dy = y - (int)((bc >> 10) & 1023u), dz = z - (int)(bc >> 20);
    return dx * dx + dy * dy + dz * dz;
}

typedef struct gie_ctx {
    /* ---- configuration (LocMap members, local_batch.h:523-568) */
    int X, Y, Z, N;
    float voxel_width;
    int occ_thresh;
    float min_h, max_h;
    int cutoff_sq;
    int fast_mode, for_motion_planner, robot_r2;
    int bar_fault;          /* test hook (gie_debug_fault_barrier): the wavefront kernel's grid barrier waits for a workgroup that does not exist, with a short spin limit */
    int max_width, max_loc_dist_sq;
    int wr[3];              /* _wave_range */
    int empty_value;        /* EMPTY_VALUE or the wide sentinel */
    int invalid_dist_min;   /* invalid_dist_glb threshold */
    /* ---- per frame */
    int map_ct;
    int pvt[3], upvt[3];
    int tile_off[3];        /* volume centre minus sensor voxel (tiling); 0 in the reference */
    int whole_lo[3], whole_hi[3]; /* union of all tiles in local coordinates; = [0,size) without tiling */
    float origin[3];
    gie_se3 L2G, G2L;
    int pntcld_mode;
    uint32_t stamp_base;    /* frame-unique base for the local/global de-dup stamps */
    /* ---- local volume planes */
    int32_t *ray_count;     /* _ray_count */
    int8_t *inst_type;      /* _inst_type */
    int8_t *glb_type;       /* _glb_type  */
    float *edt;             /* _edt_D     */
    uint16_t *cy1;          /* EDT pass Y: closest y in the column, 0xffff none */
    uint32_t *cxy2;         /* EDT pass X: cx | cy<<16, 0xffffffff none */
    uint32_t *bcoc;         /* _coc_idx_aux: batch closest obstacle, local, packed */
    uint64_t *pair;         /* _dist_id_pair (persists across frames by local index) */
    uint32_t *wl;           /* _loc_wave_layer as frame-stamped marks */
    uint8_t *tflag;         /* per local 8x8x8 tile: some voxel's Mark-time closest obstacle lies outside the volume */
    int tfd[3];             /* tile grid dims */
    uint8_t *tknown, *tunk; /* per tile: holds a known / an unknown voxel after this frame's fuse */
    uint8_t *tknown_prev;   /* tknown of the previous frame = "_glb_type of the tile is not all UNKNOWN yet" */
    uint8_t *tray;          /* per tile: a ray touched it this scan (ray-casting OGM) */
    uint8_t *tact;          /* per tile: fuse has to look at it (overlaps an existing block, or held a known voxel last frame) */
    uint8_t *tsum;          /* per tile: obtainFrontiers has something to look at */
    int32_t *tbmax;            /* per tile: the largest batch-EDT distance of this update as far as pass Z's streaming form has seen it (values above 80: "or more"; k_edt_z_stream, read by k_markc for the lazy tiles' bounds) */
    int32_t *tmax, *tmax_prev; /* per tile: 1 + the largest distance this / the previous (fused) map update committed in it; 0x7fffffff: a voxel of
                                * the tile was not committed; 0: the tile was not looked at */
    uint8_t *ucol;          /* per z-column of eight voxels (index ((z >> 3) * Y + y) * X + x), bit z & 7: the local index has turned from unknown to known and Mark has not written its pair since (gie_ops.h "`_edt_D` is derived") */
    int oldskip;            /* this update's batch EDT flags the tiles whose stored records Mark need not read (gie_tile_oldskip) */
    uint8_t *tskip;         /* per tile: Mark need not read the stored global records of this tile (gie_tile_oldskip) — and, with coc_defer, does
                             * not WRITE them either: the pair plane is the record of such a tile's voxels until they are caught up or leave */
    uint8_t *tskip_prev;    /* tskip of the gie_fuse before (the two alternate), for the catch-up of the deferred records */
    int ts_pvt[3];          /* the pivot tskip's tiles refer to (= the pose of the last gie_fuse; c.pvt moves with gie_set_pose) */
    const int8_t *scan_labels; /* non-null: the scan is a label plane the caller (or the library's staging buffer) still holds — gie_ogm_labels_dev
                             * has only flagged its blocks — and gie_fuse reads the labels from there: `_inst_type` is neither written nor reset */
    int wr_inside;          /* every voxel of the local volume lies inside the wave range (always, unless a tile offset pushes the volume out of it) */
    int skip2_ok;           /* tskip_prev describes the tiles of the update right before this one, at the pose prev_shift refers to: a tile may be flagged 2 */
    int catchup_fast;       /* gie_tile_oldskip also says which tiles of this update need their deferred records stored (tskip_prev is the update before's, at prev_shift) */
    int coc_defer;          /* Mark + commit leaves the stored obstacle of skip tiles' voxels unwritten this update (gie_ops.h "deferred records") */
    /* LAZY PAIRS (round 6, gie_ops.h "lazy pairs"): in a tile the sweep's short way handles (tskip 2, records deferred, volume inside the wave
     * range) the pair of a voxel is a function of its batch obstacle, and the sweep does not store it: the tile is flagged, readers derive */
    uint8_t *tlazy;         /* per tile: the pairs of the tile's voxels are NOT in the pair plane: they are gie_pair_of_bcoc(bcoc_lazy[id]) at pivots pp_pvt / pp_upvt.
                             * ONE plane, never swapped or cleared with the frame: it says what the pair plane holds NOW (set by the sweep, taken away by whoever
                             * writes the tile's pairs) */
    const uint32_t *bcoc_lazy; /* the batch-obstacle plane the flags refer to (the one of the last merge; `bcoc` alternates between two planes) */
    int lazy_ok;            /* this update's sweep may leave pairs out: the volume lies inside the wave range and is not one tile of several */
    int qdefer;             /* readers of single global voxels (gie_query_global*): a voxel of a tskip tile has its record in the pair plane ... */
    int pp_pvt[3], pp_upvt[3]; /* ... which was written at this pivot / wave-range pivot */
    int prev_valid;         /* tmax_prev describes the map update right before this one */
    int prev_shift[3];      /* previous local coordinate = local coordinate + prev_shift */
    uint8_t *zocc;          /* per z-plane: holds an OCCUPIED voxel after this frame's fuse (EDT passes skip empty planes) */
    uint64_t *zneed;        /* per (x,y) tile column: bit tz set = somebody reads the batch EDT of tile (tx,ty,tz) */
    uint32_t *zredo;        /* per workgroup tile of pass Z's column kernel (16 columns x one y): the streaming form (k_edt_z_stream) could not
                             * finish a column of it — the column kernel does the tile (zero = the frame clear) */
    uint16_t *zlist;        /* the planes with obstacles, ascending */
    int32_t *zcount;        /* how many */
    int32_t *tl_known;      /* tiles that hold a known voxel (built by k_edt_prep); count in cnt[GIE_CNT_TL_KNOWN] */
    int32_t *tl_swept;      /* ... and, of those, the tiles not flagged 2 in tskip — the ones the fused sweep walks while the others are lazy (k_markc); count in cnt[GIE_CNT_TL_SWEPT] */
    int32_t *tl_front;      /* tiles obtainFrontiers has to look at (tsum); count in cnt[GIE_CNT_TL_FRONT] */
    int force_lists;        /* -1: every kernel chooses lists or a volume sweep from its list's length; 0 / 1: forced (tests) */
    uint64_t *lprop;        /* per boundary-face voxel: wave-B proposal for inside voxels */
    uint64_t *cand[2];      /* wave C candidate planes (BFS level parity), all-ones = none */
    /* ---- block table of the frame: slot of every block overlapping the volume +-1 voxel */
    int tb0[3];             /* block coordinate of table cell 0 */
    int tdim[3];
    int32_t *blk_tab;       /* slot or -1 */
    const int32_t *tab_prev; /* the block table of the fuse before (or this one again, for a second allocation pass of the same update), or null: a
                              * cell it holds a slot for needs no hash lookup (blocks inside the volume's box are never erased or moved) */
    int tab_prev_d[3];      /* cell (bx, by, bz) of this table = cell (bx, by, bz) + tab_prev_d of that one */
    uint8_t *blk_need;      /* observed-this-scan flag per table cell */
    int32_t *blk_new;       /* scratch: new-block flag / rank */
    /* ---- global map: hash + SoA block pool */
    uint64_t *hkeys;        /* open addressing, GIE_KEY_EMPTY = free */
    int32_t *hvals;
    uint32_t hmask;
    int max_blocks;
    int32_t *pool_count;    /* device scalars: [0] slots handed out by the bump allocator, [1] entries on the free list */
    int32_t *free_list;     /* slots of erased blocks (BlockAllocBase's links, blockalloc.h:69-118, as a stack) */
    int retain;             /* gie_config.retain_radius_blocks (0: blocks are never erased) */
    int vb_lo[3], vb_hi[3]; /* block box of the local volume +-1 voxel */
    uint64_t *g_key;        /* block key per slot */
    int32_t *g_nbr;         /* per slot: the slots of the six face neighbours (-x +x -y +y -z +z; 8 words per block), written when a block is
                             * initialised, for itself and into its neighbours' rows; an entry may outlive its block (erasure): the reader
                             * holds the named slot's key against the key it expects (waves A / B, fetched with the halo) */
    uint8_t *g_occ;         /* planes, 512 per slot, in-block index x | y<<3 | z<<6 */
    int8_t *g_type;
    uint64_t *g_coc;        /* packed global coord */
    uint64_t *g_pair;
    uint64_t *g_prop;       /* proposals to hashed voxels: wave A's raises; wave B's block rounds (plane of odd rounds + the seeds) */
    uint64_t *g_prop2;      /*   ... and the plane of wave B's even rounds */
    int32_t *wb_list[2];    /* wave B: the active blocks (slots) of a round, by round parity */
    int32_t *wb_flag[2];    /*         ... and their membership flags, one word per slot */
    int32_t *lvlb_next, *lvlb_vis; /* wave B: active blocks / voxels taken up per round (GIE_MAX_LEVELS words each) */
    int32_t *lvla_next, *lvla_vis; /* wave A: likewise (its rounds alternate between the two colours of the blocks) */
    int32_t *g_wl;          /* wave_layer (-map_ct raise stamp / level stamps) */
    int track;              /* changed-block flags on (gie_stream_enable) */
    int fused;              /* Mark and commit run as one sweep, wave C commits what it merges (gie_ops.h "Mark + commit") */
    int32_t *g_dirty;       /* per slot: a voxel's type / distance / closest obstacle changed since the last stream */
    /* ---- ext boxes */
    int nbox;
    const float *box_ll, *box_ur;
    const uint8_t *box_act;
    /* ---- frontier queues + counters */
    uint64_t *qa, *qb;      /* the seeds of waves A / B: packed global coordinates ... */
    gie_vaddr *qa_a, *qb_a; /* ... and addresses (slot * 512 + in-block index) */
    int32_t *qc[2];
    int qcap_ab, qcap_c;
    int32_t *cnt;           /* device counters, see GIE_CNT_* */
    int32_t *lvl_next, *lvl_vis; /* wave C: active tiles / visits per round (GIE_MAX_LEVELS words each) */
    int32_t *wc_list[2];         /* wave C: the active tiles of a round (parity of the round) */
    int32_t *wc_flag[2];         /*         ... and their membership flags, one word per tile */
    const int32_t *gate;         /* gie_round_gate: null, or a device word — the kernels of a halo exchange round (export, ghost blocks, import,
                                  * refinement) return at once while it holds 0 ("no tile changed in the round before") */
} gie_ctx;
/* a gated launch that has nothing to do (every kernel of an exchange round asks first; uniform over the grid) */
#define GIE_GATE_CLOSED(c) ((c).gate != nullptr && *(const volatile int32_t *)(c).gate == 0)

enum {
    GIE_CNT_A = 0, GIE_CNT_B, GIE_CNT_C,        /* seed counts from obtainFrontiers */
    GIE_CNT_TL_SWEPT,                           /* entries in tl_swept */
    GIE_CNT_FREE4, GIE_CNT_FREE5,               /* (unused since the waves run in block / tile rounds) */
    GIE_CNT_LV0, GIE_CNT_LV1, GIE_CNT_LV2,      /* wave C: entries expanded in a level (rotating) */
    GIE_CNT_ERR,                                /* sticky error flags */
    GIE_CNT_NEWBLK,                             /* blocks allocated this frame */
    GIE_CNT_VIS_A, GIE_CNT_VIS_B, GIE_CNT_VIS_C,
    GIE_CNT_LVL_A, GIE_CNT_LVL_B, GIE_CNT_LVL_C,
    GIE_CNT_FRONT_B, GIE_CNT_FRONT_C,
    GIE_CNT_SEED_A, GIE_CNT_SEED_B, GIE_CNT_SEED_C,
    GIE_CNT_ZSTREAM,                            /* pass Z: the streaming form has done the volume, the column kernel only repairs the tiles flagged in zredo */
    GIE_CNT_ZFAIL,                              /* pass Z, streaming form: slabs it gave up (the column kernel has nothing to repair when 0) */
    GIE_CNT_STATE1, GIE_CNT_STATE2,             /* lengths of the fuse-time tile lists: deferred records to store (be_tile_oldskip, be_coc_catchup), lazy tiles to write out (be_pair_materialise) */
    GIE_CNT_FRAME_END = 28,                     /* [0, FRAME_END) minus ERR are zeroed every frame */
    GIE_CNT_TOT_A = 28, GIE_CNT_TOT_B = 30, GIE_CNT_TOT_C = 32, /* 64-bit running totals (2 words each) */
    GIE_CNT_BAR_B = 34,                         /* grid-barrier word of wave C's launch (first word of the second cleared range) */
    GIE_CNT_BAR_C = 35,                         /* grid-barrier word of the launch of waves A / B */
    GIE_CNT_NEWLIST = 36,                       /* entries in the list of blocks to initialise (blk_new) */
    GIE_CNT_TL_KNOWN = 37, GIE_CNT_TL_FRONT = 38, /* entries in the tile lists tl_known / tl_front */
    GIE_CNT_BAR_AB2 = 39,                       /* (unused) */
    GIE_CNT_TL_FUSE = 40,                       /* entries in the fuse tile list (shares the tl_front buffer: consumed before Mark) */
    GIE_CNT_BARFAIL = 42,                       /* a grid barrier of THIS map update timed out (the sticky GIE_ERRF_BARRIER is the host's copy) */
    GIE_CNT_TSKIP = 41,                         /* tiles whose stored records Mark does not read (counted by the test-only emulation) */
    GIE_CNT_INL = 43,                           /* wave B: voxels inside the volume that received a proposal (listed in qc[1]) */
    GIE_CNT_AUX_END = 44,                       /* [BAR_B, AUX_END) is zeroed every frame too */
    GIE_CNT_LAZY_EXACT = 44,                    /* the last fused sweep took its lazy tiles' bounds from pass Z (exact), not from samples (gie_tile_oldskip's hysteresis); kept across updates */
    GIE_CNT_NUM = 48
};
#define GIE_MAX_LEVELS 4096
#define GIE_ERRF_POOL 1
#define GIE_ERRF_QUEUE 2
#define GIE_ERRF_HASH 4
#define GIE_ERRF_BARRIER 8

/* regions zeroed by one launch at the start of a map update */
#define GIE_CLEAR_MAX 16
typedef struct gie_clear_list { void *p[GIE_CLEAR_MAX]; uint32_t bytes[GIE_CLEAR_MAX]; int n; } gie_clear_list;
/* workgroups a region of `bytes` gets in the clear launches: one per 16 KB (a region of a few MB cleared by a fixed 32 workgroups was
 * 20 us of every map update), at least one, at most 256 */
#define GIE_CLEAR_WG_BYTES 16384u
GIE_HD int gie_clear_wgs(uint32_t bytes) { const uint32_t n = (bytes + GIE_CLEAR_WG_BYTES - 1u) / GIE_CLEAR_WG_BYTES; return n < 1u ? 1 : (n > 256u ? 256 : (int)n); }
GIE_HD int gie_clear_total_wgs(const gie_clear_list &l) { int t = 0; for (int i = 0; i < l.n; i++) t += gie_clear_wgs(l.bytes[i]); return t; }

/* stamps in ctx.wl (local) */
#define GIE_WL_SEED(c) ((c).stamp_base + 1u)
#define GIE_WL_PUSHED(c) ((c).stamp_base + 2u)
#define GIE_WL_LEVEL(c, lvl) ((c).stamp_base + 8u + (uint32_t)(lvl))

GIE_HD int gie_in_loc(const gie_ctx &c, int x, int y, int z) { return x >= 0 && x < c.X && y >= 0 && y < c.Y && z >= 0 && z < c.Z; }
GIE_HD int gie_in_whole(const gie_ctx &c, int x, int y, int z)
{ return x >= c.whole_lo[0] && x < c.whole_hi[0] && y >= c.whole_lo[1] && y < c.whole_hi[1] && z >= c.whole_lo[2] && z < c.whole_hi[2]; }
GIE_HD int gie_in_wr(const gie_ctx &c, int x, int y, int z) { return x >= 0 && x < c.wr[0] && y >= 0 && y < c.wr[1] && z >= 0 && z < c.wr[2]; }
GIE_HD int gie_lid(const gie_ctx &c, int x, int y, int z) { return (z * c.Y + y) * c.X + x; }
/* few listed tiles: walk the list; many (more than 1/8 of the volume): sweep the volume, whose
 * 64-voxel rows coalesce better.  Decided on the device from the list's length, so it is right
 * for THIS map update however far the host has run ahead. */
GIE_HD int gie_use_lists(const gie_ctx &c, int listed)
{
    if (c.force_lists >= 0) return c.force_lists;
    return (long long)listed * 8 <= (long long)c.tfd[0] * c.tfd[1] * c.tfd[2];
}
/* pass Z of the batch EDT: the list form (a workgroup per known tile, planes taken outwards from the tile) however many tiles are
 * known, if the volume is at most eight tiles high — a column of 40 voxels is too short for the column kernel's LDS tiles
 * to pay (BASELINE config 4, 320 x 320 x 40: GIE_Z_SHORT_LISTS) */
#ifndef GIE_Z_SHORT_LISTS
#define GIE_Z_SHORT_LISTS 1
#endif
#ifndef GIE_Z_LIST_DIV
#define GIE_Z_LIST_DIV 8          /* the list form while at most 1 / GIE_Z_LIST_DIV of the tiles are known */
#endif
GIE_HD int gie_z_use_lists(const gie_ctx &c, int listed)
{
    if (c.force_lists >= 0) return c.force_lists;
    if (GIE_Z_SHORT_LISTS && c.tfd[2] <= 8) return 1;
    return (long long)listed * GIE_Z_LIST_DIV <= (long long)c.tfd[0] * c.tfd[1] * c.tfd[2];
}
GIE_HD int gie_tile_index(const gie_ctx &c, int x, int y, int z) { return ((z >> 3) * c.tfd[1] + (y >> 3)) * c.tfd[0] + (x >> 3); }
GIE_HD int gie_vox_in_blk(int gx, int gy, int gz) { return ((gz & 7) << 6) | ((gy & 7) << 3) | (gx & 7); }
GIE_HD int gie_d2(int ax, int ay, int az, int bx, int by, int bz)
{
    const long long dx = ax - bx, dy = ay - by, dz = az - bz;
    const long long r = dx * dx + dy * dy + dz * dz;
    return r > 0x7fffffffLL ? 0x7fffffff : (int)r;
}
GIE_HD int gie_invalid_dist(const gie_ctx &c, int d) { return d < 0 || d >= c.invalid_dist_min; }
GIE_HD int gie_invalid_coc(int x, int y, int z) { return x > 900000 || y > 900000 || z > 900000; }

/* table cell of a global voxel (must lie within the volume +-1 voxel) */
GIE_HD int gie_tab_index(const gie_ctx &c, int gx, int gy, int gz)
{
    const int bx = (gx >> 3) - c.tb0[0], by = (gy >> 3) - c.tb0[1], bz = (gz >> 3) - c.tb0[2];
    return (bz * c.tdim[1] + by) * c.tdim[0] + bx;
}

/* Hash of a block coordinate for the open-addressing table.  NOT the reference's BlockHasher (voxmap_utils.cuh:69-81:
 * (x·73856093) ^ (y·19349669) ^ (z·83492791)): on the dense grids of block coordinates a map consists of, that function
 * clusters under linear probing — simulated for the blocks of 13 C5 map updates at load 0.17: 2.7 probes per hit on
 * average, 16 at the 99th percentile, 52 at worst — and a wave of the BFS kernels waits for its slowest lane's chain of
 * dependent probes (a phase of waves A / B took 35-45 us whatever its size).  A 64-bit finaliser over the injective
 * packed key: 1.1 / 3 / 11.  The table is this library's own structure; nothing outside sees the hash. */
GIE_HD uint32_t gie_hash_key(int bx, int by, int bz)
{
    uint64_t h = gie_pack_crd(bx, by, bz);
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return (uint32_t)h;
}

/* read-only lookup: HashTableBase::get_alloc_blk_id (vhashing.h:125-134) */
GIE_HD int gie_hash_find(const gie_ctx &c, int bx, int by, int bz)
{
    const uint64_t key = gie_pack_crd(bx, by, bz);
    uint32_t h = gie_hash_key(bx, by, bz) & c.hmask;
    for (uint32_t probes = 0; probes <= c.hmask; probes++) {     /* bounded like gie_key_insert: a table without an EMPTY cell left must not hang the device */
        const uint64_t k = c.hkeys[h];
        const int v = c.hvals[h];                 /* fetched WITH the key (same index): a hit costs one round trip, not two */
        if (k == key) return v;
        if (k == GIE_KEY_EMPTY) return -1;
        h = (h + 1) & c.hmask;
    }
    return -1;
}

/* first face of the volume a boundary voxel lies on → slot in the 2(XY+YZ+XZ) proposal table */
GIE_HD int gie_bdr_index(const gie_ctx &c, int x, int y, int z)
{
    const int YZ = c.Y * c.Z, XZ = c.X * c.Z, XY = c.X * c.Y;
    if (x == 0) return y + c.Y * z;
    if (x == c.X - 1) return YZ + y + c.Y * z;
    if (y == 0) return 2 * YZ + x + c.X * z;
    if (y == c.Y - 1) return 2 * YZ + XZ + x + c.X * z;
    if (z == 0) return 2 * YZ + 2 * XZ + x + c.X * y;
    return 2 * YZ + 2 * XZ + XY + x + c.X * y;
}

/* Test / measurement switches.  The PRODUCTION library (built without -DGIE_TEST_HOOKS: libgie_hip.so) reads nothing from the
 * environment: every switch is its default, a compile-time constant, and the gie_debug_* hooks do not exist — its results depend
 * on its arguments only.  The test build of the same sources (libgie_hip_test.so, tests/hooks_py.py; the CPU emulation of tests/emu
 * is always one) reads GIE_<NAME> once per process.  What a caller may legitimately tune is in gie_config (wave_workgroups,
 * place_tries). */
#if defined(GIE_TEST_HOOKS)
#include <stdlib.h>
static inline int gie_switch_env(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#define GIE_SWITCH(name, dflt) gie_switch_env(name, dflt)
#else
#define GIE_SWITCH(name, dflt) (dflt)
#endif

#endif /* GIE_TYPES_H */
