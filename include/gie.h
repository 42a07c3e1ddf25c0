/*
 * gie.h — C-ABI of the MI355X-native incremental-EDT map-update path.
 *
 * This is the drop-in boundary for the per-frame hot path of JINXER000/GIE-mapping
 * (VOLMAPNODE::publishMap, src/volumetric_mapper.cpp:138-224).  The reference has no FFI
 * layer; its boundary is the set of C++ objects the ROS node calls.  Every entry point below
 * names the reference symbol it replaces (file:line relative to the reference tree).
 *
 * Conventions: opaque handle, int status codes (0 = GIE_OK), gie_last_error() for the text,
 * no exceptions cross the ABI, one handle is not thread-safe (the reference is single-threaded,
 * src/main.cpp:7), buffers are caller-owned HOST pointers unless the name ends in _dev.
 * All dense local-volume arrays are x-fastest: idx = z*X*Y + y*X + x (local_batch.h:394-407).
 */
#ifndef GIE_H
#define GIE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIE_OK 0
#define GIE_ERR_INVALID 1   /* bad argument / bad call order            */
#define GIE_ERR_DEVICE 2    /* HIP runtime error                        */
#define GIE_ERR_CAPACITY 3  /* block pool / hash / frontier queue full  */
#define GIE_ERR_TIMEOUT 4   /* the wavefront kernel's grid barrier timed out: its workgroups were kept off the device for
                               seconds (another process's kernels holding the compute units).  The map update that hit it is
                               incomplete (waves cut short, distances may be over-estimates until the region is observed
                               again); the condition is reported once and cleared — the next update runs normally.  What the
                               incomplete update leaves in the map: Mark's own distances are committed (Mark and commit are one
                               sweep unless the changed-block flags are on — then nothing of that update is committed), what the
                               waves would have lowered or raised behind them is not.  One process per device is the supported
                               configuration. */

/* voxel types, local_batch.h:7-10 */
#define GIE_VOX_UNKNOWN 0
#define GIE_VOX_FREE 1
#define GIE_VOX_OCCUPIED 2
#define GIE_VOX_FNT 3

/* sentinels, voxmap_utils.cuh:8-9 */
#define GIE_EMPTY_VALUE 999999

typedef struct gie_mapper gie_mapper;

/* LocMap ctor arguments (local_batch.h:35-60) + the Parameters fields that reach the GPU code
 * (parameters.h:69-132) + GlbHashMap capacity (glb_hash_map.cu:9-47). */
typedef struct gie_config {
    float voxel_width;            /* m, Parameters::voxel_width                               */
    int32_t local_size[3];        /* voxels (X,Y,Z) = local_size_{x,y,z}/voxel_width           */
    int32_t occupancy_threshold;  /* 0..255, default 180                                       */
    float ogm_min_h, ogm_max_h;   /* height gate for occupied hits (ogm/min_height,max_height) */
    int32_t cutoff_grids_sq;      /* ceil(cutoff_dist/voxel_width)^2, parameters.h:95,134-138  */
    int32_t fast_mode;            /* wave/fast_mode (code default true, parameters.h:93)       */
    int32_t for_motion_planner;   /* robot sphere forced FREE in the OGM kernels               */
    int32_t robot_r2_grids;       /* ceil(robot_r/voxel_width)^2                               */
    int32_t max_blocks;           /* block pool capacity (hash/block_max); 0 = size from volume.  19.5 KB of device memory per block; voxel addresses are 64-bit, so the device's memory is the limit */
    int32_t device_id;            /* HIP device ordinal                                         */
    int32_t retain_radius_blocks; /* block-pool lifecycle.  0 = the reference's rule: blocks are never freed and the pool only
                                     shrinks (BlockAllocBase::allocate_n throws when it is empty, blockalloc.h:50-67; its free
                                     list, blockalloc.h:69-118, is never fed).  R > 0: at the start of every gie_fuse the
                                     blocks whose block coordinate lies more than R blocks (Chebyshev) outside the block box
                                     of the local volume +-1 voxel are erased — their voxels revert to the defaults
                                     (UNKNOWN, EMPTY_VALUE, EMPTY_KEY) — and go back to the pool's free list, so a robot
                                     that drives on forever runs on a fixed pool.  Erased blocks are not streamed. */
    int32_t wave_workgroups;      /* workgroups of the persistent wavefront kernel (waves A / B / C; its grid barrier needs all of them resident
                                     at once).  0 = half the device's compute units, so that two mappers or processes with default settings
                                     share a device; a process that owns its device may ask for more (160 of 256 measured best on an MI355X),
                                     N processes on one device for at most (compute units) / N each. */
    int32_t place_tries;          /* 0 / 1 = off.  T > 1: gie_create times the Mark + commit sweep's memory pattern on the mapper's four large
                                     planes and re-allocates each up to T - 1 times, keeping the faster physical placement (the same kernel runs
                                     0.76 or 0.84 ms from one allocation to the next on an MI355X).  Costs tens of milliseconds and, transiently,
                                     a second copy of one plane; only for volumes of 16 M voxels and more.  Off by default: create is then
                                     deterministic and holds no more memory than the mapper keeps. */
    int32_t reserved[3];
} gie_config;

/* MulScanParam, include/cuda_toolkit/occupancy/vlp16/multiscan_param.h:4-27 */
typedef struct gie_multiscan_param {
    int32_t scan_num, ring_num;
    float max_r, theta_inc, theta_min, phi_inc, phi_min;
} gie_multiscan_param;

/* CamParam, include/cuda_toolkit/occupancy/realsense/camera_param.h:4-27 */
typedef struct gie_cam_param {
    int32_t rows, cols;
    float cx, cy, fx, fy;
    int32_t valid_nan;
} gie_cam_param;

/* ScanParam, include/cuda_toolkit/occupancy/hokuyo/scan_param.h:4-19 */
typedef struct gie_scan_param {
    int32_t scan_num;
    float max_r, theta_inc, theta_min;
} gie_scan_param;

/* One global voxel as planners see it (GlbVoxel, voxmap_utils.cuh:29-44). */
typedef struct gie_voxel {
    uint8_t occ_val;
    int8_t vox_type;
    int16_t pad;
    int32_t dist_sq;
    int32_t coc[3];
} gie_voxel;

/* CostMap.msg header fields (msg/CostMap.msg:1-15, volumetric_mapper.cpp:375-389). */
typedef struct gie_costmap_hdr {
    int32_t x_size, y_size, z_size;
    float x_origin, y_origin, z_origin;
    float width;
    uint8_t type; /* TYPE_EDT = 1 */
    uint8_t pad[3];
} gie_costmap_hdr;

/* SeenDist payload element, local_batch.h:19-24 (8 bytes: float d; bool s; bool o; 2 bytes of tail padding).
 * d = _edt_D in VOXEL units; o = `bool` converted from the voxel's type (local_batch.h:389): 1 for FREE /
 * OCCUPIED / FNT, 0 for UNKNOWN; s is never written by the reference (here: 0, padding 0). */
typedef struct gie_seendist {
    float d;
    uint8_t s;
    uint8_t o;
    uint8_t pad[2];
} gie_seendist;

/* Per-frame counters (the reference prints/keeps: glb_hash_map.cu:86,170-172; wave_helper.h). */
typedef struct gie_frame_stats {
    int32_t frame;            /* map_ct (_time)                                  */
    int32_t blocks_total;     /* blocks in the pool                              */
    int32_t blocks_new;       /* "New blocks allocated"                          */
    int32_t seeds_a, seeds_b, seeds_c; /* after obtainFrontiers                 */
    int32_t front_b, front_c; /* |B| after wave A, |C| after wave B              */
    int32_t visits_a, visits_b, visits_c; /* frontier entries expanded          */
    int32_t levels_a, levels_b, levels_c;
    float us_ogm, us_fuse, us_edt, us_merge; /* device time of the last step's stages; filled while gie_profile_enable is on, else 0 */
    int64_t total_visits_a, total_visits_b, total_visits_c; /* since gie_create */
    int32_t known_tiles;      /* 8x8x8 tiles of the local volume that hold a known voxel (what Mark / commit sweep) */
    int32_t frontier_tiles;   /* tiles obtainFrontiers looks at voxel by voxel (the voxels on the six faces come on top) */
} gie_frame_stats;

const char *gie_last_error(void);

/* LocMap::LocMap + create_gpu_map (local_batch.h:35-89), GlbHashMap::GlbHashMap
 * (glb_hash_map.cu:9-47), setupRotationPlan (volumetric_mapper.cpp:344-373), warmupCuda. */
gie_mapper *gie_create(const gie_config *cfg);
void gie_destroy(gie_mapper *h);

/* odom2trans/trans2proj (projection.h:14-33) + LocMap::calculate_pivot_origin /
 * calculate_update_pivot (local_batch.h:129-166).  Also advances the map tick (_time++). */
int gie_set_pose(gie_mapper *h, const float pos[3], const float quat_wxyz[4]);

/* PntcldMapMaker::updateLocalOGM (pntcld_map_maker.cpp:63-73) →
 * PNTCLD_RAYCAST::localOGMKernels (pntcld_raycast.cu:105-117). xyz: n sensor-frame points.
 * n == 0 is a valid (empty) scan.  A point whose global-frame coordinates are not finite or lie
 * beyond +-1e6 m is ignored (gie_math.h gie_point_ok; undefined in the reference). */
int gie_ogm_pointcloud(gie_mapper *h, const float *xyz, int n);
int gie_ogm_pointcloud_dev(gie_mapper *h, const float *d_xyz, int n);
/* Vlp16MapMaker::updateLocalOGM (vlp16_map_maker.cpp:51-71) → VLP_FAST::localOGMKernels
 * (vlp16_fast.cu:89-97). ranges: ring-major [ring_num][scan_num] horizontal ranges. */
int gie_ogm_multiscan(gie_mapper *h, const float *ranges, const gie_multiscan_param *p);
int gie_ogm_multiscan_dev(gie_mapper *h, const float *d_ranges, const gie_multiscan_param *p);
/* RealsenseMapMaker::updateLocalOGM (realsense_map_maker.cpp:46-52) →
 * REALSENSE_FAST::localOGMKernels (realsense_fast.cu:97-104). depth: row-major rows x cols. */
int gie_ogm_depth(gie_mapper *h, const float *depth, const gie_cam_param *p);
int gie_ogm_depth_dev(gie_mapper *h, const float *d_depth, const gie_cam_param *p);
/* HokuyoMapMaker::updateLocalOGM (hokuyo_map_maker.cpp:44-50) → HOKUYO_FAST::localOGMKernels. */
int gie_ogm_scan2d(gie_mapper *h, const float *ranges, const gie_scan_param *p);
/* A scan that arrives already classified: labels[idx] for every voxel of the local volume
 * (x fastest), GIE_VOX_UNKNOWN = not observed, GIE_VOX_FREE, GIE_VOX_OCCUPIED (anything else is
 * ignored).  This is the state the reference's projective kernels leave behind — `_inst_type` plus
 * the block key of every observed voxel (setLocalOccupancy: vlp16_fast.cu:76-86,
 * realsense_fast.cu:80-93, hokuyo_fast.cu:68-80) — without a sensor model in front of it; fused
 * like any projective scan (updateHashOGMWithSensor).  It is how the sensor-less synthetic world
 * of BASELINE config 5 (occupancy from a hash of the voxel, full observation) is
 * fed.  The robot sphere of for_motion_planner is forced FREE as in every OGM kernel. */
int gie_ogm_labels(gie_mapper *h, const int8_t *labels);
/* _dev: a device-resident plane, copied into the mapper's own scan plane by this call's kernel on the mapper's stream
 * (gie_get_stream): d_labels may be reused once that kernel has run, as with every other gie_ogm_*_dev entry point. */
int gie_ogm_labels_dev(gie_mapper *h, const int8_t *d_labels);
/* _dev_borrow (round 6; rounds 5's gie_ogm_labels_dev did this unasked): the plane MAY be read IN PLACE by gie_fuse instead (no copy
 * into `_inst_type`, no reset of it: 2 bytes per voxel less) — when the volume's X is a multiple of 16, d_labels is 16-byte aligned and
 * for_motion_planner is off; *borrowed (may be NULL) says whether it is.  A borrowed plane must stay unchanged until the gie_fuse /
 * gie_step of this map update has been executed on the mapper's stream — or until another gie_ogm_* / gie_read_ogm call, which
 * copies it first. */
int gie_ogm_labels_dev_borrow(gie_mapper *h, const int8_t *d_labels, int *borrowed);

/* Ext_Obs_Wrapper boxes as consumed by the fuse kernels (pre_map.cu:80-101,
 * unify_helper.cuh:68-86). ll/ur: n x 3 floats (metres); active: n flags. Box 0 is the inverted
 * "fence". n = 0 clears. */
int gie_set_ext_boxes(gie_mapper *h, const float *ll, const float *ur, const uint8_t *active, int n);

/* GlbHashMap::updateHashOGM (glb_hash_map.cu:115-143): allocHashTB + updateHashOGMWith*.  Needs a scan: GIE_ERR_INVALID when no
 * gie_ogm_* call has been made since the last gie_fuse (an update that only changes the external boxes feeds an empty scan:
 * gie_ogm_pointcloud(h, NULL, 0)). */
int gie_fuse(gie_mapper *h);
/* EDT_OCC::batchEDTUpdate (local_edt.cu:7-28). */
int gie_batch_edt(gie_mapper *h);
/* GlbHashMap::mergeNewObsv (glb_hash_map.cu:146-207).  One merge per map update: GIE_ERR_INVALID without a gie_fuse since the
 * last merge (the seed counters and barrier words of the wavefront kernel are cleared with the frame). */
int gie_merge(gie_mapper *h);
/* The two halves of gie_merge for a TILED run (gie_set_tile): gie_merge_begin_tiled = MarkLimitedObserve + commit of the
 * Mark-time pairs, so that the face layers exported next are THIS map update's state; gie_merge_end = obtainFrontiers +
 * waves + commit, run after the neighbours' layers have been imported.  gie_merge = begin + end. */
int gie_merge_begin(gie_mapper *h);
int gie_merge_begin_tiled(gie_mapper *h);
int gie_merge_end(gie_mapper *h);
/* fuse + batch_edt + merge, asynchronous on the mapper's stream. */
int gie_step(gie_mapper *h);
/* GPU_DEV_SYNC (cuda_macro.h:38); also surfaces device-side capacity errors. */
int gie_sync(gie_mapper *h);

/* LocMap::copy_edt_2_host / copy_ogm_2_host (local_batch.h:370-378) + the pair contents.
 * Any pointer may be NULL. edt: N floats (_edt_D, voxel units); type: N (_glb_type);
 * dist_sq: N (pair distance); coc_xyz: 3N global coords of the closest obstacle
 * (GIE_EMPTY_VALUE x3 when there is none). */
int gie_read_local(gie_mapper *h, float *edt, int8_t *type, int32_t *dist_sq, int32_t *coc_xyz);
/* Intermediate state for parity tests: OGM scan labels / hit-miss counters (before fuse).  After a
 * ray-cast scan the FREE / OCCUPIED labels of the counted cells (getAllocKeys) are only written
 * when this call asks for them: fuse works from the counters. */
int gie_read_ogm(gie_mapper *h, int8_t *inst_type, int32_t *ray_count);
/* Batch EDT result before the merge: dist² (_aux) and local closest-obstacle coords
 * (_coc_idx_aux unpacked; -1 x3 when the volume holds no obstacle).  The map update itself only
 * produces this plane where it is read (voxels of tiles that hold a known voxel); this call
 * completes it over the whole volume first. */
int gie_read_batch_edt(gie_mapper *h, int32_t *dist_sq, int32_t *coc_xyz_local);
/* LocMap::convertCostMap (local_batch.h:382-391) + setupEDTmsg4Motion
 * (volumetric_mapper.cpp:375-389). payload: N gie_seendist. */
int gie_read_costmap(gie_mapper *h, gie_seendist *payload, gie_costmap_hdr *hdr);
/* The same payload for consumers that cannot afford the stall (at 512^3 it is 1.07 GB per frame; the reference's blocking copy into
 * pageable memory, local_batch.h:370-391, was written for volumes 100x smaller):
 *  gie_read_costmap_dev   into a DEVICE buffer of the caller's (a GPU planner), asynchronous on the mapper's stream;
 *  gie_costmap_publish    conversion on the mapper's stream + ONE asynchronous copy into pinned host memory the library owns, on a
 *                         copy stream of its own: returns at once, and the next map update does not queue up behind the copy;
 *  gie_costmap_acquire    waits for the last publish and hands out its payload (N gie_seendist in pinned memory; valid until the
 *                         publish after the next one — two buffers alternate). */
int gie_read_costmap_dev(gie_mapper *h, gie_seendist *d_payload, gie_costmap_hdr *hdr);
int gie_costmap_publish(gie_mapper *h, gie_costmap_hdr *hdr);
int gie_costmap_acquire(gie_mapper *h, const gie_seendist **payload);
/* Hash lookup + retrive_vox_D (voxmap_utils.cuh:94-132) for n global coords. Voxels of
 * unallocated blocks come back as a default GlbVoxel (UNKNOWN, EMPTY_VALUE, EMPTY_KEY). */
int gie_query_global(gie_mapper *h, const int32_t *xyz, int n, gie_voxel *out);
/* Device-side access to the global map for GPU planners — the integration the reference recommends (README.md:163-165:
 * get_VB_key / get_voxID_in_VB / hash_table_D lookups inside the planner's own kernels, voxmap_utils.cuh:94-132): d_xyz (n x 3)
 * and d_out (n) are DEVICE buffers, the lookup kernel is enqueued on the mapper's stream (gie_get_stream), nothing is copied and
 * the host does not wait.  Same records as gie_query_global. */
int gie_query_global_dev(gie_mapper *h, const int32_t *d_xyz, int n, gie_voxel *d_out);
int gie_get_stats(gie_mapper *h, gie_frame_stats *out);

/* ---- changed-block streaming: the CPU mirror the reference keeps for RViz and CPU planners.
 * GlbHashMap::streamPipeline / streamD2H / getUpdatedAddr (glb_hash_map.cu:209-247,
 * unify_helper.cuh:11-32), fed by the stream_VB_keys_D appends of the fuse / wave / commit kernels
 * (unify_helper.cuh:103-110,184-191,510-520; wave_core.cuh:129-134), switched by
 * display_glb_edt / display_glb_ogm (volumetric_mapper.cpp:182,196-198).
 * gie_stream_enable(1) makes the update kernels flag every block in which a voxel's type,
 * distance or closest obstacle changes (off by default: README.md:154).  gie_stream_changed
 * hands over the flagged blocks — key (block coordinate, 3 ints) + GIE_BLOCK_VOXELS voxels in the
 * reference's in-block order get_voxID_in_VB = (x&7)*64 + (y&7)*8 + (z&7)
 * (voxmap_utils.cuh:104-109) — and clears their flags.  *n_changed = flagged blocks before the
 * call; at most max_blocks are delivered (the rest stay flagged), in unspecified order; NULL
 * keys/blocks only counts.  One gather kernel + one batched copy per chunk instead of the
 * reference's 20 KB memcpy per block. */
#define GIE_BLOCK_VOXELS 512
/* (takes effect with the next gie_fuse: a map update runs in one order of kernels — the fused Mark + commit sweep, or the reference's
 * Mark ... commit with the flags — from its fuse to its merge) */
int gie_stream_enable(gie_mapper *h, int on);
int gie_stream_changed(gie_mapper *h, int32_t *keys, gie_voxel *blocks, int max_blocks, int32_t *n_changed);

/* ---- spatial tiling across GPUs (no counterpart in the reference, which is single-GPU; SURVEY
 * §8e).  A large volume is cut into tiles, one mapper per tile/GPU.  Per map update every tile runs
 * fuse, batch EDT and gie_merge_begin_tiled on its own volume, exports the one-voxel layer on each of
 * its faces, imports its neighbours' layers as "ghost" voxels just outside its own volume — the role
 * old out-of-volume voxels play in the reference, but refreshed every update — and finishes with
 * gie_merge_end.  Then rounds of export / exchange / import / gie_refine (obtainFrontiers' C seeds
 * restricted to the faces → wave C → commit) repeat until no tile changes.  Voxels of ANOTHER tile
 * are only ever read as ghosts: waves A / B (raise / lower outside) run outside the whole volume.
 * face = 2*axis + side (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z); a layer is indexed b*A + a with (a,b) the
 * two remaining axes in x<y<z order. */
typedef struct gie_halo_voxel {
    int32_t dist_sq;
    int32_t coc[3];
    int8_t vox_type;
    uint8_t occ_val;   /* travels too: a ghost voxel that enters the tile next frame keeps its occupancy state */
    int8_t pad[2];
} gie_halo_voxel;
/* Tile placement: the local volume is centred `off` voxels away from the sensor position
 * (_pvt = round(pos/w) - size/2 + off); every tile of one robot shares the sensor pose and the
 * wave-range pivot.  `whole` = size of the union of all tiles (centred on the sensor like an
 * ordinary local volume): MarkLimitedObserve keeps an old distance only when its closest
 * obstacle lies outside the WHOLE volume — one inside another tile is re-derived from that
 * tile's current state through the halo exchange instead of being trusted.  Takes effect at the
 * next gie_set_pose; default off = 0, whole = local_size = the reference. */
int gie_set_tile(gie_mapper *h, const int32_t off[3], const int32_t whole[3]);
int gie_halo_count(gie_mapper *h, int face);
int gie_halo_export(gie_mapper *h, int face, gie_halo_voxel *out);
int gie_halo_import(gie_mapper *h, int face, const gie_halo_voxel *in);
/* same with DEVICE buffers (RCCL send/recv tensors), asynchronous on the mapper's stream */
int gie_halo_export_dev(gie_mapper *h, int face, gie_halo_voxel *d_out);
int gie_halo_import_dev(gie_mapper *h, int face, const gie_halo_voxel *d_in);
/* SPARSE face layers: only the KNOWN voxels of a layer, as (index in the layer, record) in no particular order.  An unknown ghost
 * changes nothing on import, so the two forms are interchangeable and may be mixed from round to round; a layer of a sparsely
 * observed face (a lidar's) is a few per cent of the dense one, a fully observed face (BASELINE config 5) gains nothing.  `out`
 * holds gie_halo_count(face) entries at most; the count travels with the payload (the _dev forms keep it on the device: the
 * import launches over the whole layer and looks at the first *d_count entries, so nothing waits for the host). */
typedef struct gie_halo_entry { int32_t index; gie_halo_voxel v; } gie_halo_entry;
int gie_halo_export_sparse(gie_mapper *h, int face, gie_halo_entry *out, int32_t *count);
int gie_halo_import_sparse(gie_mapper *h, int face, const gie_halo_entry *in, int32_t count);
int gie_halo_export_sparse_dev(gie_mapper *h, int face, gie_halo_entry *d_out, int32_t *d_count);
int gie_halo_import_sparse_dev(gie_mapper *h, int face, const gie_halo_entry *d_in, const int32_t *d_count);
/* returns the number of voxels seeded from ghost neighbours in *seeded (0 = nothing changed) */
/* Several faces in one call (NULL entry = face not exchanged): one launch per step for all of
 * them and one block allocation for all ghost layers. */
int gie_halo_export_all_dev(gie_mapper *h, gie_halo_voxel *const d_out[6]);
int gie_halo_import_all_dev(gie_mapper *h, const gie_halo_voxel *const d_in[6]);
int gie_refine(gie_mapper *h, int32_t *seeded);   /* seeded == NULL: enqueue only (no synchronisation) */
/* Exchange rounds "until no GPU changed"  WITHOUT the host in the loop.  A caller enqueues a fixed upper bound of
 * rounds per map update on the mapper's stream — export, transfer, import, gie_refine_dev — and all-reduces (max) the word
 * gie_refine_dev leaves over the ranks, on the same stream (RCCL), into a device word it hands to gie_round_gate before the
 * next round: while that word is 0 every kernel of a round (export, ghost-block allocation, import, refinement) returns at once.
 * gie_refine_dev = gie_refine(h, NULL) + *d_changed = voxels seeded from ghost neighbours by this round (0 when the gate kept
 * it from running).  gie_round_gate(h, NULL) opens the gate (the first round of an update always runs).  gie_round_end closes a
 * map update's rounds: opens the gate and counts the update as unconverged when *d_go (the all-reduced word of the LAST round;
 * NULL = not known) is still non-zero.  gie_round_stats synchronises: out = { rounds enqueued, rounds that ran, map updates,
 * map updates left unconverged } since gie_create.  A gate lasts until gie_round_end or the next gie_set_pose, whichever comes
 * first (a caller that leaves a round half way does not gate the next update).  No counterpart in the reference (single GPU). */
int gie_round_gate(gie_mapper *h, const int32_t *d_go);
int gie_refine_dev(gie_mapper *h, int32_t *d_changed);
int gie_round_end(gie_mapper *h, const int32_t *d_go);
int gie_round_stats(gie_mapper *h, int64_t out[4]);
/* The HIP stream all work of this mapper is enqueued on (a hipStream_t), so that a caller can order
 * its own device work — e.g. the RCCL transfers of the halo layers — with it instead of
 * synchronising the host.  NULL for implementations without streams. */
int gie_get_stream(gie_mapper *h, void **stream);

/* Per-kernel device time (the reference only has the two std::chrono spans of
 * volumetric_mapper.cpp:153,187-203).  When enabled, every kernel launch of the frame carries
 * start / stop HIP events on the mapper's stream (on its own dispatch packet: the time is the
 * kernel's execution, `launches` counts the brackets a kernel group was timed in);
 * gie_profile_read synchronises, returns the accumulated totals since the last read and resets
 * them.  Profiling is not free: a timed dispatch makes the next one wait for its completion
 * signal (about +0.1 ms per map update on an MI355X) — leave it off when measuring throughput. */
typedef struct gie_kernel_time {
    char name[24];
    float total_ms;
    int32_t launches;
} gie_kernel_time;
int gie_profile_enable(gie_mapper *h, int on);
int gie_profile_read(gie_mapper *h, gie_kernel_time *out, int max_entries);
/* local pivot _pvt (global coord of local voxel 0,0,0) of the current frame. */
int gie_get_pivot(gie_mapper *h, int32_t pvt[3]);

#ifdef __cplusplus
}
#endif
#endif /* GIE_H */
