/*
 * gie_driver — replays a recorded frame file through VolumetricMapper (gie_host.hpp): the
 * stand-alone counterpart of running the reference node on a rosbag (the launch files).
 *
 *   gie_driver --frames in.gief [--yaml cfg.yaml] [--set key=value ...] [--out prefix]
 *              [--log run.csv] [--rms] [--device N]
 *
 * Frame file: "GIEF" u32 version(1) u32 count, then per record
 *   i32 kind, f32 pos[3], f32 quat_wxyz[4], i32 n_floats, i32 ip[4], f32 fp[6], f32 data[n_floats]
 *   kind 0 depth      ip = rows, cols, valid_nan        fp = cx, cy, fx, fy
 *        1 scan2d     ip = scan_num                     fp = max_r, theta_inc, theta_min
 *        2 multiscan  ip = scan_num, ring_num           fp = max_r, theta_inc, theta_min, phi_inc, phi_min
 *        3 pointcloud n_floats = 3 n (sensor frame xyz)
 *        4 ring cloud n_floats = 5 n (x, y, z, intensity, ring) → Vlp16Adapter → multiscan;
 *                     ip = use_rs_lidar
 *        5 external obstacle cloud, n_floats = 3 n (world frame) → ExtObstacles::cluster_cloud;
 *                     not a map update
 * Outputs (after the last frame): prefix.edt.f32 prefix.type.i8 prefix.dist.i32 prefix.coc.i32
 * prefix.costmap.bin (when for_motion_planner) prefix.boxes.f32 (ll,ur,active per box)
 * prefix.mirror.bin (when display_glb_edt or display_glb_ogm: i32 n, n x (i32 key[3]), then
 * n x 512 gie_voxel — the CPU mirror built from the changed-block stream).
 * The CSV log starts with the reference's three columns (volumetric_mapper.cpp:121-122,189,202)
 * and adds the per-frame counters.
 */
#include "gie_host.hpp"

using namespace gie_host;

static bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }
template <class T> static void dump(const std::string &path, const std::vector<T> &v)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}

int main(int argc, char **argv)
{
    std::string frames, yaml, out, log;
    std::vector<std::string> sets;
    bool rms = false;
    int device = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--frames") frames = next(); else if (a == "--yaml") yaml = next(); else if (a == "--out") out = next();
        else if (a == "--log") log = next(); else if (a == "--set") sets.push_back(next()); else if (a == "--rms") rms = true;
        else if (a == "--device") device = std::stoi(next());
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (frames.empty()) { fprintf(stderr, "usage: gie_driver --frames in.gief [--yaml cfg] [--set k=v] [--out prefix] [--log csv] [--rms]\n"); return 2; }
    try {
        Parameters p;
        if (!yaml.empty()) p.load_yaml(yaml);
        for (const std::string &s : sets) { const size_t e = s.find('='); if (e == std::string::npos) throw std::runtime_error("--set wants key=value"); p.set(s.substr(0, e), s.substr(e + 1)); }
        VolumetricMapper node(p, device);
        const gie_config &c = node.config();
        const size_t N = (size_t)c.local_size[0] * c.local_size[1] * c.local_size[2];

        FILE *f = fopen(frames.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + frames);
        char magic[4]; uint32_t ver = 0, count = 0;
        if (!rd(f, magic, 4) || memcmp(magic, "GIEF", 4) || !rd(f, &ver, 4) || ver != 1 || !rd(f, &count, 4)) throw std::runtime_error("bad frame file header");

        CsvLog *csv = log.empty() ? nullptr : new CsvLog(log);
        if (csv) { *csv << "Occupancy time" << "EDT time" << "RMSE" << "New blocks" << "Visits A" << "Visits B" << "Visits C" << "Streamed blocks"; csv->endrow(); }
        std::vector<float> data;
        Vlp16Adapter *vlp = nullptr;
        for (uint32_t k = 0; k < count; k++) {
            int32_t kind, n, ip[4]; Pose pose; float fp[6];
            if (!rd(f, &kind, 4) || !rd(f, pose.pos, 12) || !rd(f, pose.quat_wxyz, 16) || !rd(f, &n, 4) || !rd(f, ip, 16) || !rd(f, fp, 24) || n < 0) throw std::runtime_error("truncated record header");
            data.resize((size_t)n);
            if (n && !rd(f, data.data(), (size_t)n * 4)) throw std::runtime_error("truncated record data");
            VolumetricMapper::Frame fr;
            std::memset(&fr, 0, sizeof(fr));
            fr.data = data.data(); fr.n = n;
            std::vector<PointXYZIR> pts;
            switch (kind) {
            case 0: fr.kind = VolumetricMapper::DEPTH; fr.cam = { ip[0], ip[1], fp[0], fp[1], fp[2], fp[3], ip[2] }; break;
            case 1: fr.kind = VolumetricMapper::SCAN2D; fr.scan = { ip[0], fp[0], fp[1], fp[2] }; break;
            case 2: fr.kind = VolumetricMapper::MULTISCAN; fr.mscan = { ip[0], ip[1], fp[0], fp[1], fp[2], fp[3], fp[4] }; break;
            case 3: fr.kind = VolumetricMapper::POINTCLOUD; fr.n = n / 3; break;
            case 4: {
                if (!vlp) vlp = new Vlp16Adapter(gie_multiscan_param{ 440, 16, 10.f, (float)(2.0 * M_PI / 440), (float)-M_PI, (float)(2.0 / 180.0 * M_PI), (float)(-15.0 / 180.0 * M_PI) }, ip[0] != 0);
                pts.resize((size_t)n / 5);
                for (size_t i = 0; i < pts.size(); i++) pts[i] = { data[5 * i], data[5 * i + 1], data[5 * i + 2], data[5 * i + 3], (uint16_t)data[5 * i + 4] };
                fr.kind = VolumetricMapper::MULTISCAN; fr.mscan = vlp->param();
                fr.data = vlp->convert(pts.data(), pts.size()); fr.n = fr.mscan.scan_num * fr.mscan.ring_num;
                break;
            }
            case 5: node.ext.cluster_cloud(data.data(), (size_t)n / 3, p.is_ext_obsv_3D, p.obsbbx_ll, p.obsbbx_ur); continue;
            default: throw std::runtime_error("unknown record kind");
            }
            node.publishMap(pose, fr);
            gie_frame_stats st;
            if (gie_get_stats(node.handle(), &st) != GIE_OK) throw std::runtime_error(gie_last_error());
            double frame_rms = -1;
            if (csv && p.profile_loc_rms) {                    /* Gnd_truth_checker on the local volume; brute force, small volumes only */
                std::vector<float> e(N); std::vector<int8_t> t(N);
                if (gie_read_local(node.handle(), e.data(), t.data(), nullptr, nullptr) != GIE_OK) throw std::runtime_error(gie_last_error());
                frame_rms = ground_truth_check(e.data(), t.data(), c.local_size[0], c.local_size[1], c.local_size[2], c.voxel_width).rms;
            }
            if (csv) { *csv << (float)node.ogm_ms << (float)node.edt_ms << frame_rms << st.blocks_new << st.visits_a << st.visits_b << st.visits_c << node.streamed_blocks; csv->endrow(); }
        }
        fclose(f);
        delete csv; delete vlp;

        std::vector<float> edt(N); std::vector<int8_t> type(N); std::vector<int32_t> dist(N), coc(3 * N);
        if (gie_read_local(node.handle(), edt.data(), type.data(), dist.data(), coc.data()) != GIE_OK) throw std::runtime_error(gie_last_error());
        if (!out.empty()) {
            dump(out + ".edt.f32", edt); dump(out + ".type.i8", type); dump(out + ".dist.i32", dist); dump(out + ".coc.i32", coc);
            if (p.for_motion_planner) {
                std::vector<uint8_t> cm(sizeof(gie_costmap_hdr) - 4 + node.cost_map.payload8.size() * sizeof(gie_seendist));
                /* x/y/z size, origin, width (28 bytes), then the payload */
                const CostMap &m = node.cost_map;
                const int32_t sz[3] = { m.x_size, m.y_size, m.z_size }; const float og[4] = { m.x_origin, m.y_origin, m.z_origin, m.width };
                memcpy(cm.data(), sz, 12); memcpy(cm.data() + 12, og, 16);
                memcpy(cm.data() + 28, m.payload8.data(), m.payload8.size() * sizeof(gie_seendist));
                dump(out + ".costmap.bin", cm);
            }
            if (p.display_glb_edt || p.display_glb_ogm) {
                const BlockMirror &mr = node.mirror;
                const int32_t nb = (int32_t)mr.block_keys.size();
                std::vector<uint8_t> buf(4 + (size_t)nb * 12 + mr.blocks.size() * sizeof(gie_voxel));
                memcpy(buf.data(), &nb, 4);
                if (nb) { memcpy(buf.data() + 4, mr.block_keys.data(), (size_t)nb * 12); memcpy(buf.data() + 4 + (size_t)nb * 12, mr.blocks.data(), mr.blocks.size() * sizeof(gie_voxel)); }
                dump(out + ".mirror.bin", buf);
            }
            std::vector<float> boxes;
            /* boxes as uploaded for the last frame: 7 floats each */
            /* (the wrapper keeps them private; export through its accessors) */
            for (size_t i = 0; i < node.ext.size(); i++) {
                const Vec3 &l = node.ext.ll(i), &u = node.ext.ur(i);
                boxes.insert(boxes.end(), { l.x, l.y, l.z, u.x, u.y, u.z, i < node.ext.active().size() ? (float)node.ext.active()[i] : 0.f });
            }
            dump(out + ".boxes.f32", boxes);
        }
        if (rms) {
            const RmsResult r = ground_truth_check(edt.data(), type.data(), c.local_size[0], c.local_size[1], c.local_size[2], c.voxel_width);
            printf("rms %.6f max %.6f less %zu more %zu n %zu\n", r.rms, r.max_err, r.less, r.more, r.n);
        }
        printf("frames %d ogm_ms %.3f edt_ms %.3f\n", node.frame, node.ogm_ms, node.edt_ms);
    } catch (const std::exception &e) {
        fprintf(stderr, "gie_driver: %s\n", e.what());
        return 1;
    }
    return 0;
}
