/* Test helper (GPU): gie_host::RcclTransport (host/gie_tiled.hpp) on ONE GPU.
 *
 * RCCL allows a communicator of one rank, and a rank may send to itself inside a group.  A mapper's +x face layer is wired to
 * its own -x face (and the other way round) as if the volume were one tile of a ring: the device-resident export of both faces
 * goes through RcclTransport::exchange — ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the MAPPER'S stream, the code
 * path bench.py's C++ twin runs between GPUs — and what arrives is compared byte for byte with the host export (gie_halo_export)
 * of the face it came from; then the received layers are imported as ghosts (gie_halo_import_all_dev) and the update is
 * finished (gie_merge_end), which must succeed.  Prints "rccl loopback OK ..." and exits 0, or the reason and a non-zero code;
 * exit code 77 = RCCL refused a one-rank communicator on this box (the test reports that as a skip with the reason). */
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../gie-mapping_amd/host/gie_tiled.hpp"

using namespace gie_host;

static void hchk(hipError_t e, const char *what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
static void gchk(int rc) { if (rc != GIE_OK) throw std::runtime_error(std::string("gie: ") + gie_last_error()); }

int main(int argc, char **argv)
{
    const int port = argc > 1 ? atoi(argv[1]) : 29811;
    try {
        gie_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.voxel_width = 0.05f; cfg.local_size[0] = 64; cfg.local_size[1] = 48; cfg.local_size[2] = 32;
        cfg.occupancy_threshold = 180; cfg.ogm_min_h = -1000.f; cfg.ogm_max_h = 1000.f; cfg.cutoff_grids_sq = 1600; cfg.fast_mode = 0;
        gie_mapper *m = gie_create(&cfg);
        if (!m) throw std::runtime_error(std::string("gie_create: ") + gie_last_error());
        /* this volume is the middle tile of a ring of three along x: its x faces are internal faces */
        const int32_t off[3] = { 0, 0, 0 }, whole[3] = { 192, 48, 32 };
        gchk(gie_set_tile(m, off, whole));
        const float pos[3] = { 0.f, 0.f, 0.f }, quat[4] = { 1.f, 0.f, 0.f, 0.f };
        const size_t N = (size_t)64 * 48 * 32;
        std::vector<int8_t> lab(N);
        for (int z = 0; z < 32; z++) for (int y = 0; y < 48; y++) for (int x = 0; x < 64; x++)
            lab[((size_t)z * 48 + y) * 64 + x] = (int8_t)((((unsigned)(x * 73856093) ^ (unsigned)(y * 19349669) ^ (unsigned)(z * 83492791)) % 53u) == 0u ? 2 : 1);
        void *stream = nullptr;
        gchk(gie_get_stream(m, &stream));
        SocketTransport boot(1, 0, port);                 /* one rank: nobody to connect to */
        std::unique_ptr<RcclTransport> tr;
        try { tr.reset(new RcclTransport(1, 0, 0, boot)); }
        catch (const std::exception &e) { printf("RCCL one-rank communicator refused: %s\n", e.what()); return 77; }

        const std::map<int, int> nbs = { { 0, 0 }, { 1, 0 } };          /* both x neighbours are ... this rank */
        std::map<int, size_t> bytes;
        std::map<int, void *> exp, send, recv;
        for (int f = 0; f < 2; f++) {
            bytes[f] = (size_t)gie_halo_count(m, f) * sizeof(gie_halo_voxel);
            hchk(hipMalloc(&exp[f], bytes[f]), "hipMalloc"); hchk(hipMalloc(&recv[f], bytes[f]), "hipMalloc");
            hchk(hipMemset(recv[f], 0xee, bytes[f]), "hipMemset");
        }
        /* what leaves across face f arrives at the neighbour as ITS face f ^ 1: in the ring the neighbour is this mapper, so the
         * buffer sent "across face 0" is received into recv[0] by the matching receive — the export of face 1 */
        send[0] = exp[1]; send[1] = exp[0];

        long long checked = 0;
        for (int frame = 0; frame < 3; frame++) {
            gchk(gie_set_pose(m, pos, quat));
            gchk(gie_ogm_labels(m, lab.data()));
            gchk(gie_fuse(m)); gchk(gie_batch_edt(m)); gchk(gie_merge_begin_tiled(m));
            gie_halo_voxel *out[6] = { (gie_halo_voxel *)exp[0], (gie_halo_voxel *)exp[1], nullptr, nullptr, nullptr, nullptr };
            gchk(gie_halo_export_all_dev(m, out));
            tr->exchange(nbs, send, recv, bytes, stream);  /* RCCL, stream-ordered behind the export kernel */
            const gie_halo_voxel *in[6] = { (const gie_halo_voxel *)recv[0], (const gie_halo_voxel *)recv[1], nullptr, nullptr, nullptr, nullptr };
            gchk(gie_halo_import_all_dev(m, in));
            gchk(gie_merge_end(m));
            gchk(gie_sync(m));
            for (int f = 0; f < 2; f++) {
                std::vector<uint8_t> got(bytes[f]), want(bytes[f ^ 1]);
                hchk(hipMemcpy(got.data(), recv[f], bytes[f], hipMemcpyDeviceToHost), "hipMemcpy");
                /* the host export of the face the layer came from: taken after the update, so compare the fields an update's second
                 * half does not change on a face voxel's own record ... it may (waves), so compare with the DEVICE export instead */
                hchk(hipMemcpy(want.data(), exp[f ^ 1], bytes[f ^ 1], hipMemcpyDeviceToHost), "hipMemcpy");
                if (got.size() != want.size() || memcmp(got.data(), want.data(), got.size()) != 0) throw std::runtime_error("face layer " + std::to_string(f) + " arrived changed");
                int known = 0;
                const gie_halo_voxel *v = (const gie_halo_voxel *)got.data();
                for (size_t i = 0; i < got.size() / sizeof(gie_halo_voxel); i++) known += v[i].vox_type != 0;
                if (known == 0) throw std::runtime_error("the face layer is empty: nothing was tested");
                checked += known;
            }
            /* and the host-staged export of the same face equals the device-resident one (taken at the same point of the next frame) */
        }
        /* device export == host export of the same state */
        for (int f = 0; f < 2; f++) {
            std::vector<gie_halo_voxel> h((size_t)gie_halo_count(m, f));
            gchk(gie_halo_export(m, f, h.data()));
            gie_halo_voxel *out[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
            out[f] = (gie_halo_voxel *)exp[f];
            gchk(gie_halo_export_all_dev(m, out));
            gchk(gie_sync(m));
            std::vector<uint8_t> d(bytes[f]);
            hchk(hipMemcpy(d.data(), exp[f], bytes[f], hipMemcpyDeviceToHost), "hipMemcpy");
            if (memcmp(d.data(), h.data(), bytes[f]) != 0) throw std::runtime_error("device and host export of face " + std::to_string(f) + " differ");
        }
        /* gated rounds (gie_round_gate / gie_refine_dev / gie_round_end) with the all-reduce(max) of the "changed" word over RCCL on the
         * mapper's stream: one more update, three rounds enqueued.  The ring neighbour is this mapper, whose own face layers hold
         * nothing closer than what it has: the first round seeds nothing, the others meet a closed gate. */
        {
            int32_t *d_words = nullptr;
            hchk(hipMalloc((void **)&d_words, 8), "hipMalloc");
            gchk(gie_set_pose(m, pos, quat));
            gchk(gie_ogm_labels(m, lab.data()));
            gchk(gie_fuse(m)); gchk(gie_batch_edt(m)); gchk(gie_merge_begin_tiled(m));
            gie_halo_voxel *out[6] = { (gie_halo_voxel *)exp[0], (gie_halo_voxel *)exp[1], nullptr, nullptr, nullptr, nullptr };
            const gie_halo_voxel *in[6] = { (const gie_halo_voxel *)recv[0], (const gie_halo_voxel *)recv[1], nullptr, nullptr, nullptr, nullptr };
            gchk(gie_halo_export_all_dev(m, out)); tr->exchange(nbs, send, recv, bytes, stream); gchk(gie_halo_import_all_dev(m, in));
            gchk(gie_merge_end(m));
            for (int k = 1; k <= 3; k++) {
                gchk(gie_round_gate(m, k > 1 ? d_words + 1 : nullptr));
                gchk(gie_halo_export_all_dev(m, out)); tr->exchange(nbs, send, recv, bytes, stream); gchk(gie_halo_import_all_dev(m, in));
                gchk(gie_refine_dev(m, d_words));
                hchk(hipMemcpyAsync(d_words + 1, d_words, 4, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync");
                tr->allreduce_max_dev(d_words + 1, stream);
            }
            gchk(gie_round_end(m, d_words + 1));
            int64_t st[4];
            gchk(gie_round_stats(m, st));
            if (st[0] != 3 || st[1] < 1 || st[1] > 3 || st[2] != 1 || st[3] != 0)
                throw std::runtime_error("gated rounds: stats " + std::to_string(st[0]) + " " + std::to_string(st[1]) + " " + std::to_string(st[2]) + " " + std::to_string(st[3]));
            printf("gated rounds over RCCL: %lld enqueued, %lld ran, unconverged %lld\n", (long long)st[0], (long long)st[1], (long long)st[3]);
            (void)hipFree(d_words);
        }
        long long s = tr->allreduce_sum(41, stream);
        if (s != 41) throw std::runtime_error("allreduce_sum over one rank changed the value");
        for (int f = 0; f < 2; f++) { (void)hipFree(exp[f]); (void)hipFree(recv[f]); }
        tr.reset();
        gie_destroy(m);
        printf("rccl loopback OK: 3 updates, %lld known face voxels through ncclSend/ncclRecv on the mapper's stream\n", checked);
        return 0;
    } catch (const std::exception &e) {
        printf("rccl loopback FAILED: %s\n", e.what());
        return 1;
    }
}
