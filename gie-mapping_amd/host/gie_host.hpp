/*
 * gie_host.hpp — ROS-free C++ host layer over the C-ABI (include/gie.h).
 *
 * Counterpart of what the reference keeps on the host around the GPU path:
 *   Parameters            include/parameters.h:16-139 (defaults, flt2GridsSq) + the cfg yaml keys
 *   Vlp16Adapter          Vlp16MapMaker::convertPyntCld, src/vlp16_map_maker.cpp:73-147
 *   PointCloudAdapter     PntcldMapMaker::pntcld_process, src/pntcld_map_maker.cpp:49-61
 *   ExtObstacles          Ext_Obs_Wrapper, src/kernel/pre_map/pre_map.cu:4-101, and the DBSCAN
 *                         clustering of VOLMAPNODE::clustring, src/volumetric_mapper.cpp:391-491
 *   CsvLog                csvfile, include/simple_logger.h:18-85
 *   GroundTruthCheck      Gnd_truth_checker::cmp_dist, include/gt_checker.h:30-80
 *   BlockMirror           hash_table_H_std / VB_values_H / VB_keys_H fed by streamPipeline,
 *                         src/kernel/par_wave/glb_hash_map.cu:209-247, README.md:163-170
 *   VolumetricMapper      VOLMAPNODE ctor + publishMap, src/volumetric_mapper.cpp:6-224, with
 *                         CostMap = msg/CostMap.msg
 * The ROS transport (subscriptions, message_filters, tf broadcast, RViz clouds) is not part of it.
 * Written from the behaviour described in those files; no reference code is reused (PCL's
 * KD-tree / MomentOfInertia are replaced by plain loops).
 */
#ifndef GIE_HOST_HPP
#define GIE_HOST_HPP

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gie.h"

namespace gie_host {

struct Vec3 { float x, y, z; };

/* ---------------------------------------------------------------- parameters */
struct Parameters {
    bool for_motion_planner = false;
    float robot_r = 0.4f;
    bool display_loc_edt = false, display_loc_ogm = false, display_glb_edt = true, display_glb_ogm = true;
    bool profile_loc_rms = false, profile_glb_rms = false;
    std::string log_name = "GIE_log.csv";
    std::string data_case = "ugv_corridor";
    int vis_interval = 1;
    int occupancy_threshold = 180;
    float vis_height = 1.0f, ugv_height = -1.0f;
    float voxel_width = 0.2f;
    float local_size_x = 10, local_size_y = 10, local_size_z = 3;
    float ogm_min_h = 0.2f, ogm_max_h = 10.0f;
    bool fast_mode = true;               /* code default, parameters.h:93 */
    float cutoff_dist = 6.0f;
    int max_bucket = 10000, max_block = 19997;
    int wave_workgroups = 0, place_tries = 0;   /* gie_config fields of the same name (no counterpart in the reference): "gpu/wave_workgroups", "gpu/place_tries" */
    bool is_ext_obsv_3D = false;
    std::vector<Vec3> obsbbx_ll{ { -3.6f, -3.2f, 0.2f } }, obsbbx_ur{ { 4.4f, 3.4f, 2.6f } };   /* the hard-coded fence */

    int flt2GridsSq(float rad) const { const int g = (int)std::ceil(rad / voxel_width); return g * g; }
    int cutoff_grids_sq() const { return flt2GridsSq(cutoff_dist); }
    int robot_r2_grids() const { return flt2GridsSq(robot_r); }

    /* cfg yaml: flat "key: value" lines with one level of nesting (ogm:, wave:, hash:) */
    void load_yaml(const std::string &path)
    {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open " + path);
        std::string line, section;
        while (std::getline(f, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            if (line.find_first_not_of(" \t\r") == std::string::npos) continue;
            const bool nested = line[0] == ' ' || line[0] == '\t';
            const size_t colon = line.find(':');
            if (colon == std::string::npos) continue;
            auto trim = [](std::string s) { const size_t a = s.find_first_not_of(" \t\r\""), b = s.find_last_not_of(" \t\r\""); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
            const std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
            if (val.empty()) { section = key; continue; }
            if (!nested) section.clear();
            set(section.empty() ? key : section + "/" + key, val);
        }
    }
    void set(const std::string &k, const std::string &v)
    {
        auto b = [&] { return v == "true" || v == "True" || v == "1"; };
        auto fl = [&] { return std::stof(v); };
        auto in = [&] { return std::stoi(v); };
        if (k == "for_motion_planner") for_motion_planner = b(); else if (k == "robot_r") robot_r = fl();
        else if (k == "display_loc_edt") display_loc_edt = b(); else if (k == "display_loc_ogm") display_loc_ogm = b();
        else if (k == "display_glb_edt") display_glb_edt = b(); else if (k == "display_glb_ogm") display_glb_ogm = b();
        else if (k == "profile_loc_rms") profile_loc_rms = b(); else if (k == "profile_glb_rms") profile_glb_rms = b();
        else if (k == "log_name") log_name = v; else if (k == "data_case") data_case = v;
        else if (k == "vis_interval") vis_interval = in(); else if (k == "occupancy_threshold") occupancy_threshold = in();
        else if (k == "vis_height") vis_height = fl(); else if (k == "ugv_height") ugv_height = fl();
        else if (k == "voxel_width") voxel_width = fl();
        else if (k == "local_size_x") local_size_x = fl(); else if (k == "local_size_y") local_size_y = fl(); else if (k == "local_size_z") local_size_z = fl();
        else if (k == "ogm/min_height") ogm_min_h = fl(); else if (k == "ogm/max_height") ogm_max_h = fl();
        else if (k == "wave/fast_mode") fast_mode = b(); else if (k == "wave/cutoff_dist") cutoff_dist = fl();
        else if (k == "hash/bucket_max") max_bucket = in(); else if (k == "hash/block_max") max_block = in();
        else if (k == "gpu/wave_workgroups") wave_workgroups = in(); else if (k == "gpu/place_tries") place_tries = in();
        else if (k == "is_ext_obsv_3D") is_ext_obsv_3D = b();
    }
    gie_config to_config(int device_id = 0) const
    {
        gie_config c;
        std::memset(&c, 0, sizeof(c));
        c.voxel_width = voxel_width;
        c.local_size[0] = (int)(local_size_x / voxel_width);      /* float → int truncation, volumetric_mapper.cpp:70 */
        c.local_size[1] = (int)(local_size_y / voxel_width);
        c.local_size[2] = (int)(local_size_z / voxel_width);
        c.occupancy_threshold = occupancy_threshold;
        c.ogm_min_h = ogm_min_h; c.ogm_max_h = ogm_max_h;
        c.cutoff_grids_sq = cutoff_grids_sq();
        c.fast_mode = fast_mode; c.for_motion_planner = for_motion_planner; c.robot_r2_grids = robot_r2_grids();
        c.max_blocks = 0;            /* hash/block_max of the shipped yaml files (≈12-22 k) is far too small for big volumes */
        c.device_id = device_id;
        c.wave_workgroups = wave_workgroups; c.place_tries = place_tries;
        return c;
    }
};

/* ---------------------------------------------------------------- sensor adapters */
struct PointXYZIR { float x, y, z, intensity; uint16_t ring; };

class Vlp16Adapter {
public:
    explicit Vlp16Adapter(const gie_multiscan_param &p = { 440, 16, 10.f, (float)(2.0 * M_PI / 440), (float)-M_PI, (float)(2.0 / 180.0 * M_PI), (float)(-15.0 / 180.0 * M_PI) },
                          bool use_rs_lidar = false)
        : p_(p), rs_(use_rs_lidar), ranges_((size_t)p.scan_num * p.ring_num) {}
    /* ring binning of one cloud: ranges[ring][bin] = horizontal range, +inf where nothing fell */
    const float *convert(const PointXYZIR *pts, size_t n)
    {
        std::fill(ranges_.begin(), ranges_.end(), std::numeric_limits<float>::infinity());
        const float res = std::fabs(p_.theta_inc);
        for (size_t i = 0; i < n; i++) {
            uint16_t r = pts[i].ring;
            if (rs_ && r >= 8) r = (uint16_t)(23 - r);            /* USE_RS_LIDAR remap */
            if (r >= p_.ring_num) continue;
            const int bin = (int)((atan2f(pts[i].y, pts[i].x) + (float)M_PI) / res);
            if (bin >= 0 && bin < p_.scan_num) ranges_[(size_t)r * p_.scan_num + bin] = sqrtf(pts[i].x * pts[i].x + pts[i].y * pts[i].y);
        }
        return ranges_.data();
    }
    const gie_multiscan_param &param() const { return p_; }
private:
    gie_multiscan_param p_;
    bool rs_;
    std::vector<float> ranges_;
};

class PointCloudAdapter {
public:
    explicit PointCloudAdapter(size_t cld_sz) : cap_(cld_sz), xyz_(cld_sz * 3) {}
    /* keeps at most cld_sz points, in order */
    size_t process(const float *xyz, size_t n)
    {
        const size_t m = std::min(n, cap_);
        std::memcpy(xyz_.data(), xyz, m * 3 * sizeof(float));
        return valid_ = m;
    }
    const float *data() const { return xyz_.data(); }
    size_t valid() const { return valid_; }
private:
    size_t cap_, valid_ = 0;
    std::vector<float> xyz_;
};

/* ---------------------------------------------------------------- external obstacles */
class ExtObstacles {
public:
    void assign_premap(const std::vector<Vec3> &ll, const std::vector<Vec3> &ur) { ll_ = ll; ur_ = ur; }
    void append(const Vec3 &ll, const Vec3 &ur) { ll_.push_back(ll); ur_.push_back(ur); }
    static bool intersects(const Vec3 &a_ll, const Vec3 &a_ur, const Vec3 &b_ll, const Vec3 &b_ur)
    {
        return a_ll.x <= b_ur.x && a_ur.x >= b_ll.x && a_ll.y <= b_ur.y && a_ur.y >= b_ll.y && a_ll.z <= b_ur.z && a_ur.z >= b_ll.z;
    }
    /* box 0 (the fence) is never active; the others are active when they touch the local volume */
    void activate(const Vec3 &loc_ll, const Vec3 &loc_ur)
    {
        act_.assign(ll_.size(), 0);
        for (size_t i = 1; i < ll_.size(); i++) act_[i] = intersects(loc_ll, loc_ur, ll_[i], ur_[i]) ? 1 : 0;
    }
    int upload(gie_mapper *m) const
    {
        return gie_set_ext_boxes(m, ll_.empty() ? nullptr : &ll_[0].x, ur_.empty() ? nullptr : &ur_[0].x, act_.empty() ? nullptr : act_.data(), (int)ll_.size());
    }
    size_t size() const { return ll_.size(); }
    const Vec3 &ll(size_t i) const { return ll_[i]; }
    const Vec3 &ur(size_t i) const { return ur_[i]; }
    const std::vector<uint8_t> &active() const { return act_; }

    /* Density clustering of the external cloud (VOLMAPNODE::clustring): neighbourhood radius
     * 0.3 m; a point seeds a group with everything in its neighbourhood, and the group keeps
     * growing through members that have >= 3 neighbours (themselves included); groups of >= 4
     * points become one axis-aligned box each.  z extent is fixed to [0.2, 2.6] unless is_3d.
     * Neighbour search: uniform buckets of one radius, so a query looks at 27 buckets. */
    void cluster_cloud(const float *xyz, size_t n, bool is_3d, const std::vector<Vec3> &pre_ll, const std::vector<Vec3> &pre_ur)
    {
        assign_premap(pre_ll, pre_ur);
        if (n == 0) return;
        const float rad = 0.3f, rad2 = rad * rad;
        const size_t min_core = 3, min_group = 4;
        struct Cell { int64_t key; int idx; };
        auto cell_of = [&](const float *p, int d) { return (int64_t)std::floor(p[d] / rad); };
        auto key_of = [](int64_t cx, int64_t cy, int64_t cz) { return ((cx & 0x1fffff) << 42) | ((cy & 0x1fffff) << 21) | (cz & 0x1fffff); };
        std::vector<Cell> cells(n);
        for (size_t i = 0; i < n; i++) cells[i] = { key_of(cell_of(xyz + 3 * i, 0), cell_of(xyz + 3 * i, 1), cell_of(xyz + 3 * i, 2)), (int)i };
        std::sort(cells.begin(), cells.end(), [](const Cell &a, const Cell &b) { return a.key != b.key ? a.key < b.key : a.idx < b.idx; });
        auto around = [&](int i, std::vector<int> &out) {
            out.clear();
            const float *p = xyz + 3 * (size_t)i;
            const int64_t cx = cell_of(p, 0), cy = cell_of(p, 1), cz = cell_of(p, 2);
            for (int64_t dx = -1; dx <= 1; dx++) for (int64_t dy = -1; dy <= 1; dy++) for (int64_t dz = -1; dz <= 1; dz++) {
                const int64_t k = key_of(cx + dx, cy + dy, cz + dz);
                auto it = std::lower_bound(cells.begin(), cells.end(), k, [](const Cell &c, int64_t key) { return c.key < key; });
                for (; it != cells.end() && it->key == k; ++it) {
                    const float *q = xyz + 3 * (size_t)it->idx;
                    const float ex = p[0] - q[0], ey = p[1] - q[1], ez = p[2] - q[2];
                    if (ex * ex + ey * ey + ez * ez <= rad2) out.push_back(it->idx);
                }
            }
        };
        enum : uint8_t { FRESH, PENDING, SETTLED };
        std::vector<uint8_t> mark(n, FRESH);
        std::vector<int> near, group;
        for (size_t s = 0; s < n; s++) {
            if (mark[s] == SETTLED) continue;
            group.assign(1, (int)s);
            mark[s] = SETTLED;
            around((int)s, near);
            for (int j : near) if (j != (int)s) { group.push_back(j); mark[j] = PENDING; }     /* seeds take their whole neighbourhood */
            for (size_t head = 1; head < group.size(); head++) {
                const int p = group[head];
                if (mark[p] == SETTLED) continue;
                around(p, near);
                if (near.size() >= min_core)
                    for (int j : near) if (mark[j] == FRESH) { group.push_back(j); mark[j] = PENDING; }
                mark[p] = SETTLED;
            }
            if (group.size() < min_group) continue;
            Vec3 lo{ xyz[3 * group[0]], xyz[3 * group[0] + 1], xyz[3 * group[0] + 2] }, hi = lo;
            for (int p : group) {
                const float *q = xyz + 3 * (size_t)p;
                lo.x = std::min(lo.x, q[0]); lo.y = std::min(lo.y, q[1]); lo.z = std::min(lo.z, q[2]);
                hi.x = std::max(hi.x, q[0]); hi.y = std::max(hi.y, q[1]); hi.z = std::max(hi.z, q[2]);
            }
            if (!is_3d) { lo.z = 0.2f; hi.z = 2.6f; }
            append(lo, hi);
        }
    }
private:
    std::vector<Vec3> ll_, ur_;
    std::vector<uint8_t> act_;
};

/* ---------------------------------------------------------------- CSV log */
class CsvLog {
public:
    explicit CsvLog(const std::string &path, const std::string &sep = ",") : f_(path), sep_(sep) {}
    CsvLog &operator<<(const char *s) { f_ << '"' << s << '"' << sep_; return *this; }
    CsvLog &operator<<(const std::string &s) { f_ << '"' << s << '"' << sep_; return *this; }
    template <class T> CsvLog &operator<<(const T &v) { f_ << v << sep_; return *this; }
    void endrow() { f_ << std::endl; }
    bool ok() const { return (bool)f_; }
private:
    std::ofstream f_;
    std::string sep_;
};

/* ---------------------------------------------------------------- accuracy check */
struct RmsResult { double rms, max_err; size_t less, more, n; };
/* EDT value of every known voxel against the true distance to the nearest OCCUPIED voxel of the
 * same local volume (what the KD-tree query of the reference returns), both in metres */
inline RmsResult ground_truth_check(const float *edt, const int8_t *type, int X, int Y, int Z, float voxel_width)
{
    std::vector<int> ox, oy, oz;
    for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) for (int x = 0; x < X; x++)
        if (type[((size_t)z * Y + y) * X + x] == GIE_VOX_OCCUPIED) { ox.push_back(x); oy.push_back(y); oz.push_back(z); }
    RmsResult r{ 0, 0, 0, 0, 0 };
    if (ox.empty()) { r.rms = -1; return r; }
    double s2 = 0;
    for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) for (int x = 0; x < X; x++) {
        const size_t id = ((size_t)z * Y + y) * X + x;
        if (type[id] == GIE_VOX_UNKNOWN) continue;
        long best = std::numeric_limits<long>::max();
        for (size_t i = 0; i < ox.size(); i++) {
            const long dx = x - ox[i], dy = y - oy[i], dz = z - oz[i];
            best = std::min(best, dx * dx + dy * dy + dz * dz);
        }
        const double err = std::sqrt((double)best) * voxel_width - (double)edt[id] * voxel_width;
        if (err > 0.001) r.less++; else if (err < -0.001) r.more++;
        s2 += err * err; r.max_err = std::max(r.max_err, std::fabs(err)); r.n++;
    }
    r.rms = r.n ? std::sqrt(s2 / (double)r.n) : -1;
    return r;
}

/* ---------------------------------------------------------------- CPU mirror of the global map */
/* What CPU planners and the RViz publishers read: block key → index into a growing array of
 * blocks; voxels inside a block in get_voxID_in_VB order.  update() pulls the blocks the device
 * flagged since the last call. */
class BlockMirror {
public:
    struct Key { int32_t x, y, z; bool operator==(const Key &o) const { return x == o.x && y == o.y && z == o.z; } };
    struct KeyHash { size_t operator()(const Key &k) const { return ((size_t)(uint32_t)k.x * 73856093u) ^ ((size_t)(uint32_t)k.y * 19349669u) ^ ((size_t)(uint32_t)k.z * 83492791u); } };
    static int vox_id(int gx, int gy, int gz) { return (gx & 7) * 64 + (gy & 7) * 8 + (gz & 7); }
    static Key key_of(int gx, int gy, int gz) { return { gx >> 3, gy >> 3, gz >> 3 }; }

    /* returns the number of blocks refreshed */
    int update(gie_mapper *m)
    {
        int32_t n = 0;
        if (gie_stream_changed(m, nullptr, nullptr, 0, &n) != GIE_OK) throw std::runtime_error(std::string("gie_stream_changed: ") + gie_last_error());
        if (n == 0) return 0;
        keys_.resize(3 * (size_t)n); vox_.resize((size_t)n * GIE_BLOCK_VOXELS);
        int32_t before = 0;
        if (gie_stream_changed(m, keys_.data(), vox_.data(), n, &before) != GIE_OK) throw std::runtime_error(std::string("gie_stream_changed: ") + gie_last_error());
        for (int i = 0; i < n; i++) {
            const Key k{ keys_[3 * i], keys_[3 * i + 1], keys_[3 * i + 2] };
            auto it = index.find(k);
            size_t slot;
            if (it == index.end()) { slot = block_keys.size(); index.emplace(k, slot); block_keys.push_back(k); blocks.resize((slot + 1) * GIE_BLOCK_VOXELS); }
            else slot = it->second;
            std::memcpy(&blocks[slot * GIE_BLOCK_VOXELS], &vox_[(size_t)i * GIE_BLOCK_VOXELS], GIE_BLOCK_VOXELS * sizeof(gie_voxel));
        }
        return n;
    }
    /* nullptr when the block was never streamed */
    const gie_voxel *find(int gx, int gy, int gz) const
    {
        auto it = index.find(key_of(gx, gy, gz));
        return it == index.end() ? nullptr : &blocks[it->second * GIE_BLOCK_VOXELS + vox_id(gx, gy, gz)];
    }
    std::unordered_map<Key, size_t, KeyHash> index;     /* hash_table_H_std */
    std::vector<Key> block_keys;                        /* VB_keys_H */
    std::vector<gie_voxel> blocks;                      /* VB_values_H */
private:
    std::vector<int32_t> keys_;
    std::vector<gie_voxel> vox_;
};

/* ---------------------------------------------------------------- the node's per-frame logic */
struct CostMap {                        /* msg/CostMap.msg */
    int32_t x_size = 0, y_size = 0, z_size = 0;
    float x_origin = 0, y_origin = 0, z_origin = 0, width = 0;
    uint8_t type = 1;                   /* TYPE_EDT */
    std::vector<gie_seendist> payload8;
};

struct Pose { float pos[3]; float quat_wxyz[4]; };

class VolumetricMapper {
public:
    explicit VolumetricMapper(const Parameters &p, int device_id = 0) : param(p), cfg_(p.to_config(device_id))
    {
        m_ = gie_create(&cfg_);
        if (!m_) throw std::runtime_error(std::string("gie_create: ") + gie_last_error());
        ext.assign_premap(p.obsbbx_ll, p.obsbbx_ur);
        streaming_ = p.display_glb_edt || p.display_glb_ogm;       /* volumetric_mapper.cpp:182,196-198 */
        if (streaming_) chk(gie_stream_enable(m_, 1));
        if (p.for_motion_planner) {
            cost_map.x_size = cfg_.local_size[0]; cost_map.y_size = cfg_.local_size[1]; cost_map.z_size = cfg_.local_size[2];
            cost_map.payload8.resize((size_t)cfg_.local_size[0] * cfg_.local_size[1] * cfg_.local_size[2]);
        }
    }
    ~VolumetricMapper() { if (m_) gie_destroy(m_); }
    VolumetricMapper(const VolumetricMapper &) = delete;
    VolumetricMapper &operator=(const VolumetricMapper &) = delete;

    enum Sensor { DEPTH, SCAN2D, MULTISCAN, POINTCLOUD };
    struct Frame {
        Sensor kind;
        const float *data; int n;       /* depth: rows*cols; scan2d: ranges; multiscan: ring-major image; pointcloud: n points xyz */
        gie_cam_param cam; gie_scan_param scan; gie_multiscan_param mscan;
    };

    /* VOLMAPNODE::publishMap (src/volumetric_mapper.cpp:138-224) without the ROS plumbing */
    void publishMap(Pose pose, const Frame &f)
    {
        using clk = std::chrono::steady_clock;
        if (param.ugv_height > 0) pose.pos[2] = param.ugv_height;
        auto t0 = clk::now();
        chk(gie_set_pose(m_, pose.pos, pose.quat_wxyz));
        switch (f.kind) {
        case DEPTH: chk(gie_ogm_depth(m_, f.data, &f.cam)); break;
        case SCAN2D: chk(gie_ogm_scan2d(m_, f.data, &f.scan)); break;
        case MULTISCAN: chk(gie_ogm_multiscan(m_, f.data, &f.mscan)); break;
        case POINTCLOUD: chk(gie_ogm_pointcloud(m_, f.data, f.n)); break;
        }
        /* update_ext_map (:498-508): boxes that touch the local volume become active */
        int32_t pvt[3];
        chk(gie_get_pivot(m_, pvt));                           /* _msg_origin = coord2pos(pivot) */
        const Vec3 ll{ (float)pvt[0] * param.voxel_width, (float)pvt[1] * param.voxel_width, (float)pvt[2] * param.voxel_width }, ur{ ll.x + param.local_size_x, ll.y + param.local_size_y, ll.z + param.local_size_z };
        ext.activate(ll, ur);
        chk(ext.upload(m_));
        chk(gie_fuse(m_));
        chk(gie_sync(m_));                                     /* GPU_DEV_SYNC, "only for profiling" (:186) */
        auto t1 = clk::now();
        chk(gie_batch_edt(m_));
        chk(gie_merge(m_));
        if (streaming_) streamed_blocks = mirror.update(m_);     /* streamPipeline is inside the reference's EDT timing too (:196-202) */
        chk(gie_sync(m_));
        auto t2 = clk::now();
        ogm_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        edt_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
        if (param.for_motion_planner) {
            gie_costmap_hdr h;
            chk(gie_read_costmap(m_, cost_map.payload8.data(), &h));
            cost_map.x_origin = h.x_origin; cost_map.y_origin = h.y_origin; cost_map.z_origin = h.z_origin;
            cost_map.width = h.width; cost_map.type = h.type;
        }
        frame++;
    }

    gie_mapper *handle() { return m_; }
    const gie_config &config() const { return cfg_; }

    Parameters param;
    ExtObstacles ext;
    BlockMirror mirror;
    int streamed_blocks = 0;
    CostMap cost_map;
    double ogm_ms = 0, edt_ms = 0;
    int frame = 0;
private:
    static void chk(int rc) { if (rc != GIE_OK) throw std::runtime_error(std::string("gie: ") + gie_last_error()); }
    gie_config cfg_;
    gie_mapper *m_ = nullptr;
    bool streaming_ = false;
};

} /* namespace gie_host */
#endif /* GIE_HOST_HPP */
