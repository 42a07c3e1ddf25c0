/*
 * gie_tiled.hpp — a tiled mapper for the C++ host layer: one rank (process, GPU) per tile of a larger local
 * volume, face layers exchanged between the ranks WITHOUT PyTorch (the reference is single-GPU,
 * so this has no counterpart there — its per-rank sequence is VOLMAPNODE::publishMap, volumetric_mapper.cpp:138-224,
 * with the exchange rounds of include/gie.h "spatial tiling" in the middle).
 *
 *   HaloTransport      what a rank needs from the fabric: exchange the face layers with its face neighbours,
 *                      and sum one integer over all ranks
 *   RcclTransport      (GIE_WITH_RCCL) ncclSend / ncclRecv of the DEVICE face buffers, grouped, enqueued on the
 *                      mapper's own HIP stream (gie_get_stream): export kernel -> RCCL -> import kernel are
 *                      stream ordered, the host does not wait; ncclAllReduce for the convergence test
 *   SocketTransport    TCP over 127.0.0.1 with host staging (gie_halo_export / gie_halo_import): the stand-in
 *                      the CPU tests run two ranks with, and the bootstrap channel for the ncclUniqueId
 *   TiledMapper        set_tile + publishMap: pose -> OGM -> fuse -> batch EDT -> gie_merge_begin_tiled ->
 *                      [export, exchange, import] -> gie_merge_end -> rounds of [export, exchange, import,
 *                      gie_refine] until no rank seeded anything (or a fixed number of stream-ordered rounds)
 *
 * Tile arithmetic mirrors gie/tiling.py (tile_grid, tile_offset_voxels, neighbours) so that the Python and
 * the C++ drivers of one run agree on who owns what.
 */
#ifndef GIE_TILED_HPP
#define GIE_TILED_HPP

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <thread>

#include "gie_host.hpp"

#if defined(GIE_WITH_RCCL)
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif

namespace gie_host {

/* ---------------------------------------------------------------- who owns what */
struct TileLayout {
    int world = 1, rank = 0;
    int grid[3] = { 1, 1, 1 };
    int tile[3] = { 0, 0, 0 };

    /* tiles per axis for a power-of-two world: 1 (1,1,1), 2 (2,1,1), 4 (2,2,1), 8 (2,2,2), ... */
    static void grid_of(int world, int g[3])
    {
        if (world < 1 || (world & (world - 1))) throw std::runtime_error("world size must be a power of two");
        g[0] = g[1] = g[2] = 1;
        for (int a = 0; g[0] * g[1] * g[2] < world; a = (a + 1) % 3) g[a] *= 2;
    }
    TileLayout(int world_, int rank_, const int tile_[3]) : world(world_), rank(rank_)
    {
        grid_of(world, grid);
        for (int i = 0; i < 3; i++) { tile[i] = tile_[i]; if (tile[i] % 8) throw std::runtime_error("tiles must be aligned to the 8-voxel blocks"); }
    }
    void index(int r, int idx[3]) const { idx[0] = r % grid[0]; idx[1] = (r / grid[0]) % grid[1]; idx[2] = r / (grid[0] * grid[1]); }
    /* offset (voxels) of this tile's centre from the centre of the whole volume: gie_set_tile's `off` */
    void offset(int32_t off[3]) const
    {
        int idx[3]; index(rank, idx);
        for (int i = 0; i < 3; i++) off[i] = idx[i] * tile[i] + tile[i] / 2 - (grid[i] * tile[i]) / 2;
    }
    void whole(int32_t w[3]) const { for (int i = 0; i < 3; i++) w[i] = grid[i] * tile[i]; }
    /* face (2 axis + side) -> neighbour rank, for the tiles that exist around this one */
    std::map<int, int> neighbours() const
    {
        std::map<int, int> out;
        int idx[3]; index(rank, idx);
        for (int axis = 0; axis < 3; axis++) for (int side = 0; side < 2; side++) {
            int j[3] = { idx[0], idx[1], idx[2] };
            j[axis] += side ? 1 : -1;
            if (j[axis] >= 0 && j[axis] < grid[axis]) out[2 * axis + side] = j[0] + grid[0] * (j[1] + grid[1] * j[2]);
        }
        return out;
    }
};

/* ---------------------------------------------------------------- transports */
struct HaloTransport {
    virtual ~HaloTransport() {}
    /* true: the face layers live in DEVICE buffers and the transfers are ordered on the mapper's stream */
    virtual bool device_resident() const = 0;
    /* send[f] goes to the neighbour across face f, which imports it as its face f ^ 1; recv[f] receives that neighbour's layer */
    virtual void exchange(const std::map<int, int> &nbs, const std::map<int, void *> &send, const std::map<int, void *> &recv,
                          const std::map<int, size_t> &bytes, void *stream) = 0;
    virtual long long allreduce_sum(long long v, void *stream) = 0;
    /* max over the ranks of a DEVICE word, in place, ordered on `stream` (gated rounds; device-resident transports only) */
    virtual void allreduce_max_dev(int32_t *, void *) { throw std::runtime_error("this transport has no device-side all-reduce (gated rounds need a device-resident transport)"); }
};

/* TCP between the ranks of one host: rank r listens on base_port + r; a full mesh of connections, made once.
 * Messages are {int32 tag, int64 bytes, payload}; tags keep the faces of one round apart. */
class SocketTransport : public HaloTransport {
public:
    SocketTransport(int world, int rank, int base_port) : world_(world), rank_(rank), peers_((size_t)world, -1)
    {
        const int lfd = socket(AF_INET, SOCK_STREAM, 0);
        int one = 1; setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{}; a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = htons((uint16_t)(base_port + rank));
        if (bind(lfd, (sockaddr *)&a, sizeof(a)) != 0 || listen(lfd, world) != 0) { close(lfd); throw std::runtime_error("SocketTransport: cannot listen"); }
        /* lower ranks connect to higher ones; every connection starts with the connector's rank */
        for (int p = rank + 1; p < world; p++) {
            int fd = -1;
            for (int tries = 0; tries < 600; tries++) {
                fd = socket(AF_INET, SOCK_STREAM, 0);
                sockaddr_in b{}; b.sin_family = AF_INET; b.sin_addr.s_addr = htonl(INADDR_LOOPBACK); b.sin_port = htons((uint16_t)(base_port + p));
                if (connect(fd, (sockaddr *)&b, sizeof(b)) == 0) break;
                close(fd); fd = -1; usleep(50000);
            }
            if (fd < 0) { close(lfd); throw std::runtime_error("SocketTransport: cannot reach rank " + std::to_string(p)); }
            setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            const int32_t me = rank; put(fd, &me, 4);
            peers_[(size_t)p] = fd;
        }
        for (int k = 0; k < rank; k++) {
            const int fd = accept(lfd, nullptr, nullptr);
            if (fd < 0) { close(lfd); throw std::runtime_error("SocketTransport: accept failed"); }
            setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            int32_t who = -1; get(fd, &who, 4);
            if (who < 0 || who >= rank || peers_[(size_t)who] >= 0) { close(lfd); throw std::runtime_error("SocketTransport: bad hello"); }
            peers_[(size_t)who] = fd;
        }
        close(lfd);
    }
    ~SocketTransport() override { for (int fd : peers_) if (fd >= 0) close(fd); }
    bool device_resident() const override { return false; }
    void exchange(const std::map<int, int> &nbs, const std::map<int, void *> &send, const std::map<int, void *> &recv,
                  const std::map<int, size_t> &bytes, void *) override
    {
        /* sends from a helper thread so that two ranks sending to each other cannot both block on full socket buffers */
        std::thread tx([&]() { for (const auto &fn : nbs) send_msg(fn.second, fn.first ^ 1, send.at(fn.first), bytes.at(fn.first)); });
        for (const auto &fn : nbs) recv_msg(fn.second, fn.first, recv.at(fn.first), bytes.at(fn.first));
        tx.join();
    }
    long long allreduce_sum(long long v, void *) override
    {
        long long total = v;
        if (rank_ == 0) {
            for (int p = 1; p < world_; p++) { long long x = 0; recv_msg(p, 100, &x, 8); total += x; }
            for (int p = 1; p < world_; p++) send_msg(p, 101, &total, 8);
        } else { send_msg(0, 100, &v, 8); recv_msg(0, 101, &total, 8); }
        return total;
    }
    /* rank 0's bytes to everybody (the ncclUniqueId) */
    void broadcast(void *p, size_t n)
    {
        if (rank_ == 0) for (int q = 1; q < world_; q++) send_msg(q, 102, p, n);
        else recv_msg(0, 102, p, n);
    }
private:
    static void put(int fd, const void *p, size_t n)
    { const char *c = (const char *)p; while (n) { const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) throw std::runtime_error("SocketTransport: send failed"); c += k; n -= (size_t)k; } }
    static void get(int fd, void *p, size_t n)
    { char *c = (char *)p; while (n) { const ssize_t k = ::recv(fd, c, n, 0); if (k <= 0) throw std::runtime_error("SocketTransport: peer closed"); c += k; n -= (size_t)k; } }
    void send_msg(int peer, int32_t tag, const void *p, size_t n) { const int64_t len = (int64_t)n; put(peers_[(size_t)peer], &tag, 4); put(peers_[(size_t)peer], &len, 8); put(peers_[(size_t)peer], p, n); }
    void recv_msg(int peer, int32_t tag, void *p, size_t n)
    {
        int32_t t = -1; int64_t len = -1;
        get(peers_[(size_t)peer], &t, 4); get(peers_[(size_t)peer], &len, 8);
        if (t != tag || len != (int64_t)n) throw std::runtime_error("SocketTransport: unexpected message (tag " + std::to_string(t) + ", want " + std::to_string(tag) + ")");
        get(peers_[(size_t)peer], p, n);
    }
    int world_, rank_;
    std::vector<int> peers_;
};

#if defined(GIE_WITH_RCCL)
/* RCCL over xGMI: the face layers never leave HBM; send / receive of all faces of a round are ONE group on the mapper's
 * stream.  xGMI is point-to-point (7 links per GPU): the <= 3 face neighbours of a 2x2x2 tile sit on distinct links, a
 * 512^2 face layer of 20 B voxels is 5.2 MB — the exchange is latency-, not bandwidth-bound. */
class RcclTransport : public HaloTransport {
public:
    /* `boot` carries the ncclUniqueId from rank 0 to the others */
    RcclTransport(int world, int rank, int device, SocketTransport &boot) : world_(world)
    {
        if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("RcclTransport: hipSetDevice failed");
        ncclUniqueId id;
        if (rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) throw std::runtime_error("ncclGetUniqueId failed");
        boot.broadcast(&id, sizeof(id));
        if (ncclCommInitRank(&comm_, world, id, rank) != ncclSuccess) throw std::runtime_error("ncclCommInitRank failed");
        if (hipMalloc(&d_sum_, 8) != hipSuccess) throw std::runtime_error("RcclTransport: hipMalloc failed");
    }
    ~RcclTransport() override { if (d_sum_) (void)hipFree(d_sum_); if (comm_) ncclCommDestroy(comm_); }
    bool device_resident() const override { return true; }
    void exchange(const std::map<int, int> &nbs, const std::map<int, void *> &send, const std::map<int, void *> &recv,
                  const std::map<int, size_t> &bytes, void *stream) override
    {
        ck(ncclGroupStart());
        for (const auto &fn : nbs) {
            ck(ncclSend(send.at(fn.first), bytes.at(fn.first), ncclUint8, fn.second, comm_, (hipStream_t)stream));
            ck(ncclRecv(recv.at(fn.first), bytes.at(fn.first), ncclUint8, fn.second, comm_, (hipStream_t)stream));
        }
        ck(ncclGroupEnd());
    }
    /* the "some tile changed" word of a gated round: max over the ranks, in place, on the stream — nobody waits */
    void allreduce_max_dev(int32_t *d_word, void *stream) override
    {
        ck(ncclAllReduce(d_word, d_word, 1, ncclInt32, ncclMax, comm_, (hipStream_t)stream));
    }
    long long allreduce_sum(long long v, void *stream) override
    {
        if (world_ == 1) return v;
        if (hipMemcpyAsync(d_sum_, &v, 8, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) throw std::runtime_error("RcclTransport: copy failed");
        ck(ncclAllReduce(d_sum_, d_sum_, 1, ncclInt64, ncclSum, comm_, (hipStream_t)stream));
        long long out = 0;
        if (hipMemcpyAsync(&out, d_sum_, 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
            throw std::runtime_error("RcclTransport: copy back failed");
        return out;
    }
private:
    static void ck(ncclResult_t r) { if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL: ") + ncclGetErrorString(r)); }
    int world_;
    ncclComm_t comm_ = nullptr;
    void *d_sum_ = nullptr;
};
#endif

/* ---------------------------------------------------------------- the tiled node */
class TiledMapper {
public:
    /* p.local_size_* is the TILE's size; fixed_rounds > 0: that many refinement rounds per update, enqueued without a
     * convergence test (device-resident transports: the host never waits); 0: rounds until no rank seeded anything, the host
     * reading the all-reduced seed count after every round; < 0: at most -fixed_rounds rounds enqueued per update and GATED ON THE
     * DEVICE by the all-reduced "changed" word of the round before (gie_round_gate / gie_refine_dev / gie_round_end: rounds until no
     * tile changed without the host in the loop — device-resident transports only; round_stats() says what ran) */
    TiledMapper(const Parameters &p, const TileLayout &layout, HaloTransport &tr, int device_id = 0, int fixed_rounds = 0)
        : param(p), layout_(layout), tr_(tr), fixed_rounds_(fixed_rounds), cfg_(p.to_config(device_id))
    {
        for (int i = 0; i < 3; i++) if (cfg_.local_size[i] != layout.tile[i]) throw std::runtime_error("TiledMapper: the parameters' local size is not the layout's tile size");
        m_ = gie_create(&cfg_);
        if (!m_) throw std::runtime_error(std::string("gie_create: ") + gie_last_error());
        int32_t off[3], whole[3];
        layout.offset(off); layout.whole(whole);
        chk(gie_set_tile(m_, off, whole));
        nbs_ = layout.neighbours();
        chk(gie_get_stream(m_, &stream_));
        for (const auto &fn : nbs_) {
            const size_t n = (size_t)gie_halo_count(m_, fn.first) * sizeof(gie_halo_voxel);
            bytes_[fn.first] = n;
#if defined(GIE_WITH_RCCL)
            if (tr_.device_resident()) {
                void *s = nullptr, *r = nullptr;
                if (hipMalloc(&s, n) != hipSuccess || hipMalloc(&r, n) != hipSuccess) throw std::runtime_error("TiledMapper: face buffer allocation failed");
                send_[fn.first] = s; recv_[fn.first] = r;
                continue;
            }
#endif
            host_send_[fn.first].resize(n); host_recv_[fn.first].resize(n);
            send_[fn.first] = host_send_[fn.first].data(); recv_[fn.first] = host_recv_[fn.first].data();
        }
        if (fixed_rounds_ < 0) {
            if (!tr_.device_resident()) throw std::runtime_error("TiledMapper: gated rounds need a device-resident transport");
#if defined(GIE_WITH_RCCL)
            if (hipMalloc((void **)&d_words_, 2 * sizeof(int32_t)) != hipSuccess) throw std::runtime_error("TiledMapper: hipMalloc failed");
#endif
        }
    }
    ~TiledMapper()
    {
        if (m_) gie_destroy(m_);
#if defined(GIE_WITH_RCCL)
        if (tr_.device_resident()) for (auto &kv : send_) { (void)hipFree(kv.second); (void)hipFree(recv_[kv.first]); }
        if (d_words_) (void)hipFree(d_words_);
#endif
    }
    TiledMapper(const TiledMapper &) = delete;
    TiledMapper &operator=(const TiledMapper &) = delete;

    /* one map update of this rank's tile; every rank calls it with the SAME pose and sensor frame */
    void publishMap(Pose pose, const VolumetricMapper::Frame &f)
    {
        if (param.ugv_height > 0) pose.pos[2] = param.ugv_height;
        chk(gie_set_pose(m_, pose.pos, pose.quat_wxyz));
        switch (f.kind) {
        case VolumetricMapper::DEPTH: chk(gie_ogm_depth(m_, f.data, &f.cam)); break;
        case VolumetricMapper::SCAN2D: chk(gie_ogm_scan2d(m_, f.data, &f.scan)); break;
        case VolumetricMapper::MULTISCAN: chk(gie_ogm_multiscan(m_, f.data, &f.mscan)); break;
        case VolumetricMapper::POINTCLOUD: chk(gie_ogm_pointcloud(m_, f.data, f.n)); break;
        }
        chk(gie_fuse(m_));
        chk(gie_batch_edt(m_));
        chk(gie_merge_begin_tiled(m_));
        round(true);                                       /* this update's face layers -> ghosts, then the rest of the merge */
        rounds = 0;
        if (fixed_rounds_ < 0) { gated_rounds(-fixed_rounds_); chk(gie_sync(m_)); frame++; return; }
        for (;;) {
            if (fixed_rounds_ > 0 && rounds >= fixed_rounds_) break;
            const long long seeded = round(false);
            rounds++;
            if (fixed_rounds_ <= 0 && seeded == 0) break;
            if (rounds >= 64) break;
        }
        chk(gie_sync(m_));
        frame++;
    }
    gie_mapper *handle() { return m_; }
    const gie_config &config() const { return cfg_; }
    /* gated mode: { rounds enqueued, rounds that ran, map updates, map updates left unconverged } since construction */
    void round_stats(int64_t out[4]) { chk(gie_round_stats(m_, out)); }
    Parameters param;
    int rounds = 0, frame = 0;
private:
    /* export -> exchange -> import, then gie_merge_end (first) or gie_refine; returns the seeds of all ranks (0 when not asked) */
    long long round(bool first)
    {
        if (tr_.device_resident()) {
            gie_halo_voxel *out[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
            const gie_halo_voxel *in[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
            for (const auto &fn : nbs_) { out[fn.first] = (gie_halo_voxel *)send_[fn.first]; in[fn.first] = (const gie_halo_voxel *)recv_[fn.first]; }
            chk(gie_halo_export_all_dev(m_, out));
            tr_.exchange(nbs_, send_, recv_, bytes_, stream_);
            chk(gie_halo_import_all_dev(m_, in));
        } else {
            for (const auto &fn : nbs_) chk(gie_halo_export(m_, fn.first, (gie_halo_voxel *)send_[fn.first]));
            tr_.exchange(nbs_, send_, recv_, bytes_, stream_);
            for (const auto &fn : nbs_) chk(gie_halo_import(m_, fn.first, (const gie_halo_voxel *)recv_[fn.first]));
        }
        if (first) { chk(gie_merge_end(m_)); return 0; }
        if (fixed_rounds_ > 0) { chk(gie_refine(m_, nullptr)); return 0; }
        int32_t seeded = 0;
        chk(gie_refine(m_, &seeded));
        return tr_.allreduce_sum(seeded, stream_);
    }
    /* SURVEY 8(e) "until no GPU changed" with nobody waiting for the host: `bound` rounds enqueued, each but the first gated by the
     * all-reduced (max) "changed" word of the round before — the sequence of gie/tiling.py exchange_converged_device */
    void gated_rounds(int bound)
    {
#if defined(GIE_WITH_RCCL)
        int32_t *d_changed = d_words_, *d_go = d_words_ + 1;
        gie_halo_voxel *out[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
        const gie_halo_voxel *in[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
        for (const auto &fn : nbs_) { out[fn.first] = (gie_halo_voxel *)send_[fn.first]; in[fn.first] = (const gie_halo_voxel *)recv_[fn.first]; }
        for (int k = 1; k <= bound; k++) {
            chk(gie_round_gate(m_, k > 1 ? d_go : nullptr));
            chk(gie_halo_export_all_dev(m_, out));
            tr_.exchange(nbs_, send_, recv_, bytes_, stream_);
            chk(gie_halo_import_all_dev(m_, in));
            chk(gie_refine_dev(m_, d_changed));
            if (hipMemcpyAsync(d_go, d_changed, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream_) != hipSuccess) throw std::runtime_error("TiledMapper: copy failed");
            tr_.allreduce_max_dev(d_go, stream_);
        }
        chk(gie_round_end(m_, d_go));
        rounds = bound;
#else
        (void)bound; throw std::runtime_error("built without RCCL");
#endif
    }
    static void chk(int rc) { if (rc != GIE_OK) throw std::runtime_error(std::string("gie: ") + gie_last_error()); }
    int32_t *d_words_ = nullptr;
    TileLayout layout_;
    HaloTransport &tr_;
    int fixed_rounds_;
    gie_config cfg_;
    gie_mapper *m_ = nullptr;
    void *stream_ = nullptr;
    std::map<int, int> nbs_;
    std::map<int, size_t> bytes_;
    std::map<int, void *> send_, recv_;
    std::map<int, std::vector<uint8_t>> host_send_, host_recv_;
};

} /* namespace gie_host */
#endif /* GIE_TILED_HPP */
