/*
 * gie_tiled_driver — one RANK of a tiled replay: the same frame file as gie_driver, every rank feeds every frame to the
 * tile it owns and the ranks exchange their face layers through gie_tiled.hpp (no PyTorch anywhere).
 *
 *   gie_tiled_driver --frames in.gief --rank R --world W --port P [--transport socket|rccl] [--rounds N]
 *                    [--yaml cfg.yaml] [--set key=value ...] [--out prefix] [--device D]
 *
 * The parameters' local_size_* is the size of ONE tile; the whole volume is the 2^k arrangement of gie/tiling.py.
 * --transport socket : TCP on 127.0.0.1 (ports P .. P+W-1) with host staging — what the CPU tests run.
 * --transport rccl   : (built with -DGIE_WITH_RCCL) RCCL send / receive of the device buffers on the mapper's stream; the
 *                      sockets only carry the ncclUniqueId.  --device defaults to the rank.
 * --rounds N > 0     : N refinement rounds per update without a convergence test (stream ordered with rccl).
 * --rounds N < 0     : at most -N rounds enqueued per update, gated on the device by the all-reduced "changed" word (rccl only;
 *                      what bench.py --gpus N runs by default).
 * Outputs after the last frame: prefix.rR.{edt.f32,type.i8,dist.i32,coc.i32}; stdout: "rank R frames F rounds K".
 */
#include "gie_tiled.hpp"

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
    std::string frames, yaml, out, transport = "socket";
    std::vector<std::string> sets;
    int rank = -1, world = 0, port = 0, rounds = 0, device = -1;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--frames") frames = next(); else if (a == "--yaml") yaml = next(); else if (a == "--out") out = next();
        else if (a == "--set") sets.push_back(next()); else if (a == "--rank") rank = std::stoi(next()); else if (a == "--world") world = std::stoi(next());
        else if (a == "--port") port = std::stoi(next()); else if (a == "--transport") transport = next(); else if (a == "--rounds") rounds = std::stoi(next());
        else if (a == "--device") device = std::stoi(next());
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (frames.empty() || rank < 0 || world < 1 || rank >= world || port <= 0) {
        fprintf(stderr, "usage: gie_tiled_driver --frames in.gief --rank R --world W --port P [--transport socket|rccl] [--rounds N] [--yaml cfg] [--set k=v] [--out prefix]\n");
        return 2;
    }
    try {
        Parameters p;
        if (!yaml.empty()) p.load_yaml(yaml);
        for (const std::string &s : sets) { const size_t e = s.find('='); if (e == std::string::npos) throw std::runtime_error("--set wants key=value"); p.set(s.substr(0, e), s.substr(e + 1)); }
        const gie_config c0 = p.to_config(0);
        const int tile[3] = { c0.local_size[0], c0.local_size[1], c0.local_size[2] };
        TileLayout layout(world, rank, tile);
        SocketTransport sock(world, rank, port);
        HaloTransport *tr = &sock;
#if defined(GIE_WITH_RCCL)
        std::unique_ptr<RcclTransport> rccl;
        if (transport == "rccl") { rccl.reset(new RcclTransport(world, rank, device >= 0 ? device : rank, sock)); tr = rccl.get(); }
#else
        if (transport == "rccl") throw std::runtime_error("built without RCCL (compile with -DGIE_WITH_RCCL and link rccl + amdhip64)");
#endif
        {   /* Ranks that share a device (the socket transport puts every rank on device 0 unless told otherwise, and is what a single-device check uses): the persistent
             * grids of their wavefront kernels have to be resident side by side, or the grid barriers of two half-resident grids
             * wait for each other until they time out (GIE_ERR_TIMEOUT): gie_config.wave_workgroups; a value the
             * parameter file has set ("gpu/wave_workgroups") stands. */
            const int sharing = transport == "rccl" ? 1 : world;      /* (host-staged sockets: ranks of one host, as a rule of one device — also with an explicit --device) */
            if (sharing > 1 && p.wave_workgroups == 0) p.wave_workgroups = std::max(8, 192 / sharing);
        }
        TiledMapper node(p, layout, *tr, transport == "rccl" ? (device >= 0 ? device : rank) : (device >= 0 ? device : 0), rounds);
        const size_t N = (size_t)tile[0] * tile[1] * tile[2];

        FILE *f = fopen(frames.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + frames);
        char magic[4]; uint32_t ver = 0, count = 0;
        if (!rd(f, magic, 4) || memcmp(magic, "GIEF", 4) || !rd(f, &ver, 4) || ver != 1 || !rd(f, &count, 4)) throw std::runtime_error("bad frame file header");
        std::vector<float> data;
        int total_rounds = 0;
        for (uint32_t k = 0; k < count; k++) {
            int32_t kind, n, ip[4]; Pose pose; float fp[6];
            if (!rd(f, &kind, 4) || !rd(f, pose.pos, 12) || !rd(f, pose.quat_wxyz, 16) || !rd(f, &n, 4) || !rd(f, ip, 16) || !rd(f, fp, 24) || n < 0) throw std::runtime_error("truncated record header");
            data.resize((size_t)n);
            if (n && !rd(f, data.data(), (size_t)n * 4)) throw std::runtime_error("truncated record data");
            VolumetricMapper::Frame fr;
            std::memset(&fr, 0, sizeof(fr));
            fr.data = data.data(); fr.n = n;
            switch (kind) {
            case 0: fr.kind = VolumetricMapper::DEPTH; fr.cam = { ip[0], ip[1], fp[0], fp[1], fp[2], fp[3], ip[2] }; break;
            case 1: fr.kind = VolumetricMapper::SCAN2D; fr.scan = { ip[0], fp[0], fp[1], fp[2] }; break;
            case 2: fr.kind = VolumetricMapper::MULTISCAN; fr.mscan = { ip[0], ip[1], fp[0], fp[1], fp[2], fp[3], fp[4] }; break;
            case 3: fr.kind = VolumetricMapper::POINTCLOUD; fr.n = n / 3; break;
            default: throw std::runtime_error("record kind not supported by the tiled driver");
            }
            node.publishMap(pose, fr);
            total_rounds += node.rounds;
        }
        fclose(f);
        if (!out.empty()) {
            std::vector<float> edt(N); std::vector<int8_t> type(N); std::vector<int32_t> dist(N), coc(3 * N);
            if (gie_read_local(node.handle(), edt.data(), type.data(), dist.data(), coc.data()) != GIE_OK) throw std::runtime_error(gie_last_error());
            const std::string pre = out + ".r" + std::to_string(rank);
            dump(pre + ".edt.f32", edt); dump(pre + ".type.i8", type); dump(pre + ".dist.i32", dist); dump(pre + ".coc.i32", coc);
        }
        printf("rank %d frames %d rounds %d\n", rank, node.frame, total_rounds);
    } catch (const std::exception &e) {
        fprintf(stderr, "gie_tiled_driver[%d]: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}
