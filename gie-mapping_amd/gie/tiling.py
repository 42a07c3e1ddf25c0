"""Spatial sharding of a large local volume over the GPUs of one node.

A volume of `grid` voxels is cut into block-aligned tiles, one per rank; rank r owns the voxels
(and the global voxel blocks) of its tile.  After every map update the tiles exchange the
one-voxel layers on their shared faces (export -> transfer -> import as ghost voxels -> refine);
the variants below differ in where the tiles live (one process / one rank each) and in who
orders the steps (the host, or the mappers' own streams).
"""
import numpy as np


def tile_grid(world_size):
    """Tiles per axis (tx, ty, tz) for a world size that is a power of two: 1→(1,1,1), 2→(2,1,1),
    4→(2,2,1), 8→(2,2,2), …"""
    if world_size < 1 or world_size & (world_size - 1):
        raise ValueError("world_size must be a power of two")
    t = [1, 1, 1]
    a = 0
    while t[0] * t[1] * t[2] < world_size:
        t[a] *= 2
        a = (a + 1) % 3
    return tuple(t)


def tile_of_rank(rank, world_size, tile_size):
    """Origin (in voxels, relative to the large volume's corner) and size of rank's tile."""
    tx, ty, tz = tile_grid(world_size)
    ix, iy, iz = rank % tx, (rank // tx) % ty, rank // (tx * ty)
    size = tuple(int(s) for s in tile_size)
    if any(s % 8 for s in size):
        raise ValueError("tiles must be aligned to the 8-voxel blocks")
    return (ix * size[0], iy * size[1], iz * size[2]), size


def tile_offset_voxels(rank, world_size, tile_size):
    """Integer offset (voxels) of the tile centre from the centre of the whole volume — the
    argument of Mapper.set_tile_offset when all tiles share one sensor pose."""
    origin, size = tile_of_rank(rank, world_size, tile_size)
    t = tile_grid(world_size)
    return tuple(int(origin[i] + size[i] // 2 - (t[i] * size[i]) // 2) for i in range(3))


def tile_centre_offset(rank, world_size, tile_size, voxel_width):
    """Metric offset of the tile centre from the centre of the whole volume: the pose a rank
    feeds its mapper is the shared sensor pose shifted by this."""
    origin, size = tile_of_rank(rank, world_size, tile_size)
    t = tile_grid(world_size)
    whole = np.array([t[i] * size[i] for i in range(3)], dtype=np.float64)
    centre = np.array(origin, dtype=np.float64) + 0.5 * np.array(size, dtype=np.float64)
    return tuple(((centre - 0.5 * whole) * voxel_width).tolist())


def aggregate(dist, seconds, voxels_per_rank, steps):
    """Whole-job throughput [Mvoxels/s] from the slowest rank's time (max-reduce)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        world = dist.get_world_size()
    else:
        world = 1
    t_max = float(t.item())
    return world * voxels_per_rank * steps / t_max / 1e6, t_max


# face = 2*axis + side (include/gie.h); the neighbour across my face f imports my layer as its face f^1
def neighbours(rank, world_size):
    """{face: neighbour rank} for the tiles that exist around `rank`."""
    t = tile_grid(world_size)
    idx = [rank % t[0], (rank // t[0]) % t[1], rank // (t[0] * t[1])]
    out = {}
    for axis in range(3):
        for side in (0, 1):
            j = list(idx)
            j[axis] += 1 if side else -1
            if 0 <= j[axis] < t[axis]:
                out[2 * axis + side] = j[0] + t[0] * (j[1] + t[1] * j[2])
    return out


# Every exchange function below expects its mappers in the middle of a TILED map update — Mapper.update(..., tiled=True) or
# Mapper.step_begin_tiled(): fuse, batch EDT and the first half of the merge are done, the face layers hold this update's
# Mark-time state.  Round 0 exports / imports them and finishes the merge (gie_merge_end: obtainFrontiers + waves with fresh
# ghosts); the rounds after it are export / import / gie_refine.

def exchange_until_stable_local(mappers, grid, max_rounds=64, sparse=False, sent=None):
    """All tiles live in this process (tests, single-GPU checks): export every shared face, hand
    it to the neighbour, finish the merge (round 0) or refine, repeat until no tile seeded anything.
    Returns the refinement rounds run.  sparse=True: the layers travel in their sparse form (known voxels only,
    gie_halo_export_sparse / gie_halo_import_sparse); `sent` (a list) collects the bytes of every layer handed over."""
    world = grid[0] * grid[1] * grid[2]
    assert world == len(mappers)
    rounds = 0
    for k in range(max_rounds + 1):
        layers = {}
        for r, m in enumerate(mappers):
            for face, nb in neighbours(r, world).items():
                layers[(nb, face ^ 1)] = m.halo_export_sparse(face) if sparse else m.halo_export(face)
                if sent is not None:
                    sent.append(layers[(nb, face ^ 1)].nbytes)
        for (r, face), layer in layers.items():
            if sparse:
                mappers[r].halo_import_sparse(face, layer)
            else:
                mappers[r].halo_import(face, layer)
        if k == 0:
            for m in mappers:
                m.merge_end()
            continue
        seeded = sum(m.refine() for m in mappers)
        rounds += 1
        if seeded == 0:
            break
    return rounds


def _local_face_buffers(mappers, world, device, bufs):
    """One tensor per (exporting tile, face), allocated once and kept in `bufs`: only the mappers' own streams touch
    them, so their lifetime must not hang on torch's allocator (a tensor dropped after round k could be handed out
    again while a neighbour's import of round k is still reading it)."""
    import torch
    if "layers" not in bufs:
        bufs["layers"] = {(r, face): torch.empty(m.halo_count(face) * 20, dtype=torch.uint8, device=device)
                          for r, m in enumerate(mappers) for face in neighbours(r, world)}
    return bufs["layers"]


def exchange_until_stable_local_device(mappers, grid, device, max_rounds=64, bufs=None):
    """In-process variant of the device-resident exchange (all tiles on one GPU): the export
    kernel of one mapper writes the tensor the import kernel of its neighbour reads."""
    world = grid[0] * grid[1] * grid[2]
    bufs = {} if bufs is None else bufs
    lay = _local_face_buffers(mappers, world, device, bufs)
    rounds = 0
    for k in range(max_rounds + 1):
        for r, m in enumerate(mappers):
            for face in neighbours(r, world):
                m.halo_export_dev(face, lay[(r, face)].data_ptr())
        for m in mappers:
            m.sync()
        for r, m in enumerate(mappers):
            for face, nb in neighbours(r, world).items():
                m.halo_import_dev(face, lay[(nb, face ^ 1)].data_ptr())
        if k == 0:
            for m in mappers:
                m.merge_end()
                m.sync()                                   # the layers are free again
            continue
        seeded = sum(m.refine() for m in mappers)          # synchronises every mapper: the layers are free again
        rounds += 1
        if seeded == 0:
            break
    return rounds


def exchange_until_stable_device(mapper, dist, rank, world_size, device, bufs=None, max_rounds=64, group=None, sparse=False):
    """Device-resident form of exchange_until_stable for backend "nccl" (RCCL over xGMI): the
    face layers are written by the export kernel straight into the send tensors and read by the
    import kernel from the receive tensors; nothing crosses PCIe except the seed count.
    sparse=True: gie_halo_export_sparse_dev compacts the known voxels of a layer; the neighbours exchange the entry counts
    (one more small batch and one host read per round — this form waits for the host anyway), then only that many entries."""
    import torch
    nbs = neighbours(rank, world_size)
    if bufs is None:
        bufs = {}
    esz = 24 if sparse else 20
    for face in nbs:
        if face not in bufs:
            n = mapper.halo_count(face) * esz
            bufs[face] = (torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n, dtype=torch.uint8, device=device))
            if sparse:
                bufs[("count", face)] = (torch.zeros(1, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int32, device=device))
    rounds = 0
    for k in range(max_rounds + 1):
        ops = []
        if sparse:
            cops = []
            for face, nb in sorted(nbs.items()):
                cs, cr = bufs[("count", face)]
                mapper.halo_export_sparse_dev(face, bufs[face][0].data_ptr(), cs.data_ptr())
                cops.append(dist.P2POp(dist.isend, cs, nb, group=group))
                cops.append(dist.P2POp(dist.irecv, cr, nb, group=group))
            mapper.sync()
            if cops:
                for w in dist.batch_isend_irecv(cops):
                    w.wait()
                torch.cuda.synchronize(device)
            for face, nb in sorted(nbs.items()):
                ns, nr = int(bufs[("count", face)][0].item()), int(bufs[("count", face)][1].item())
                if ns:
                    ops.append(dist.P2POp(dist.isend, bufs[face][0][:ns * esz], nb, group=group))
                if nr:
                    ops.append(dist.P2POp(dist.irecv, bufs[face][1][:nr * esz], nb, group=group))
        else:
            for face, nb in sorted(nbs.items()):
                snd, rcv = bufs[face]
                mapper.halo_export_dev(face, snd.data_ptr())
                ops.append(dist.P2POp(dist.isend, snd, nb, group=group))
                ops.append(dist.P2POp(dist.irecv, rcv, nb, group=group))
            mapper.sync()                               # export kernels ran on the mapper's own stream
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            torch.cuda.synchronize(device)
        for face in sorted(nbs):
            if sparse:
                mapper.halo_import_sparse_dev(face, bufs[face][1].data_ptr(), bufs[("count", face)][1].data_ptr())
            else:
                mapper.halo_import_dev(face, bufs[face][1].data_ptr())
        if k == 0:
            mapper.merge_end()
            continue
        n = torch.tensor([mapper.refine()], dtype=torch.int64, device=device)
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
        rounds += 1
        if int(n.item()) == 0:
            break
    return rounds


def exchange_rounds_device(mapper, dist, rank, world_size, device, bufs, rounds=1, group=None):
    """A fixed number of exchange rounds, ordered ON THE MAPPER'S STREAM: export kernels, the RCCL
    send / receive of the face layers, ghost import and refinement are enqueued back to back and
    the host never waits (no seed count comes back, so there is no convergence test: information
    crosses one tile boundary per round, the rest follows with the next map update)."""
    import torch
    if "ops" not in bufs:                                 # everything that does not change from round to round, once
        nbs = neighbours(rank, world_size)
        for face in nbs:
            n = mapper.halo_count(face) * 20
            bufs[face] = (torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n, dtype=torch.uint8, device=device))
        bufs["stream"] = torch.cuda.ExternalStream(mapper.stream_handle(), device=device)
        bufs["out"] = {face: bufs[face][0].data_ptr() for face in nbs}
        bufs["in"] = {face: bufs[face][1].data_ptr() for face in nbs}
        ops = []
        for face, nb in sorted(nbs.items()):
            snd, rcv = bufs[face]
            ops.append(dist.P2POp(dist.isend, snd, nb, group=group))
            ops.append(dist.P2POp(dist.irecv, rcv, nb, group=group))
        bufs["ops"] = ops
    ops = bufs["ops"]
    with torch.cuda.stream(bufs["stream"]):
        for k in range(rounds + 1):
            mapper.halo_export_all_dev(bufs["out"])
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()                              # stream-level: the current (= the mapper's) stream waits for RCCL
            mapper.halo_import_all_dev(bufs["in"])
            if k == 0:
                mapper.merge_end()                        # round 0 finishes the merge with this update's ghosts
            else:
                mapper.refine_async()
    return rounds


def exchange_converged_device(mapper, dist, rank, world_size, device, bufs, max_rounds=4, group=None):
    """rounds "until no GPU changed", ordered ON THE MAPPER'S STREAM with nobody waiting for the host.  Round 0 finishes
    the merge with this update's ghosts.  Then at most `max_rounds` refinement rounds are enqueued back to back: export, RCCL send /
    receive of the face layers, import, gie_refine_dev — which leaves the number of voxels it seeded in a device word —, and a 4-byte
    all-reduce(max) of that word over the ranks on the same stream.  The result gates the NEXT round (gie_round_gate): once no tile
    changed, every kernel of the remaining rounds returns at once (the transfers still run: a collective cannot be skipped by one
    side).  gie_round_end counts an update whose last all-reduce was still non-zero as unconverged (Mapper.round_stats()): the
    bound was too small — tests hold the bench's bound against the tiled oracle.  Works on the CPU too (emulated mappers, gloo):
    `device` "cpu", the "device words" are then host memory.
    Returns nothing the host could know without waiting: the rounds that ran are in round_stats()."""
    import torch
    cuda = getattr(device, "type", str(device)) == "cuda"
    if "go" not in bufs:
        nbs = neighbours(rank, world_size)
        for face in nbs:
            n = mapper.halo_count(face) * 20
            bufs[face] = (torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n, dtype=torch.uint8, device=device))
        bufs["stream"] = torch.cuda.ExternalStream(mapper.stream_handle(), device=device) if cuda else None
        bufs["out"] = {face: bufs[face][0].data_ptr() for face in nbs}
        bufs["in"] = {face: bufs[face][1].data_ptr() for face in nbs}
        ops = []
        for face, nb in sorted(nbs.items()):
            snd, rcv = bufs[face]
            ops.append(dist.P2POp(dist.isend, snd, nb, group=group))
            ops.append(dist.P2POp(dist.irecv, rcv, nb, group=group))
        bufs["ops"] = ops
        bufs["changed"] = torch.zeros(1, dtype=torch.int32, device=device)
        bufs["go"] = torch.ones(1, dtype=torch.int32, device=device)
    ops, changed, go = bufs["ops"], bufs["changed"], bufs["go"]

    def transfer():
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()                                  # stream-level on RCCL: the current (= the mapper's) stream waits for the transfer

    def rounds():
        mapper.round_gate(None)
        mapper.halo_export_all_dev(bufs["out"])
        transfer()
        mapper.halo_import_all_dev(bufs["in"])
        mapper.merge_end()                                # round 0 finishes the merge with this update's ghosts
        for k in range(1, max_rounds + 1):
            mapper.round_gate(go.data_ptr() if k > 1 else None)
            mapper.halo_export_all_dev(bufs["out"])
            transfer()
            mapper.halo_import_all_dev(bufs["in"])
            mapper.refine_dev(changed.data_ptr())
            go.copy_(changed)
            dist.all_reduce(go, op=dist.ReduceOp.MAX, group=group)
        mapper.round_end(go.data_ptr())

    if cuda:
        with torch.cuda.stream(bufs["stream"]):
            rounds()
    else:
        rounds()


def exchange_converged_local_device(mappers, grid, device, max_rounds=4, bufs=None):
    """exchange_converged_device with all tiles in this process (one GPU, or emulated mappers with device "cpu"): the neighbour's
    stream waits for an event on the exporter's stream instead of an RCCL transfer, and the all-reduce(max) of the "changed" words
    is one reduction on a side stream that waits for every mapper's refinement and that every mapper's next round waits for."""
    import torch
    cuda = getattr(device, "type", str(device)) == "cuda"
    world = grid[0] * grid[1] * grid[2]
    own = bufs is None
    bufs = {} if bufs is None else bufs
    lay = _local_face_buffers(mappers, world, device, bufs)
    if "go" not in bufs:
        bufs["streams"] = [torch.cuda.ExternalStream(m.stream_handle(), device=device) for m in mappers] if cuda else [None] * world
        bufs["side"] = torch.cuda.Stream(device=device) if cuda else None
        bufs["imported"] = []
        bufs["changed"] = torch.zeros(world, dtype=torch.int32, device=device)
        bufs["go"] = torch.ones(1, dtype=torch.int32, device=device)
    streams, side, changed, go = bufs["streams"], bufs["side"], bufs["changed"], bufs["go"]
    esz = changed.element_size()

    def record(r):
        if not cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(streams[r])
        return ev

    def wait(r, evs):
        if cuda:
            for ev in evs:
                streams[r].wait_event(ev)

    go_ready = []
    for k in range(max_rounds + 1):
        evs = []
        for r, m in enumerate(mappers):
            wait(r, bufs["imported"] + go_ready)          # the layers of the round before have been read; this round's gate is final
            m.round_gate(go.data_ptr() if k > 1 else None)
            m.halo_export_all_dev({face: lay[(r, face)].data_ptr() for face in neighbours(r, world)})
            evs.append(record(r))
        done, refined = [], []
        for r, m in enumerate(mappers):
            wait(r, evs)
            m.halo_import_all_dev({face: lay[(nb, face ^ 1)].data_ptr() for face, nb in neighbours(r, world).items()})
            done.append(record(r))
            if k == 0:
                m.merge_end()
            else:
                m.refine_dev(changed.data_ptr() + r * esz)
                refined.append(record(r))
        bufs["imported"] = done
        if k > 0:                                         # the all-reduce(max): one reduction everybody's next round waits for
            if cuda:
                for ev in refined:
                    side.wait_event(ev)
                with torch.cuda.stream(side):
                    torch.amax(changed, dim=0, keepdim=True, out=go)
                    ev = torch.cuda.Event()
                    ev.record(side)
                go_ready = [ev]
            else:
                torch.amax(changed, dim=0, keepdim=True, out=go)
    for r, m in enumerate(mappers):
        wait(r, go_ready)
        m.round_end(go.data_ptr())
    if own:
        for m in mappers:
            m.sync()


def exchange_rounds_local_device(mappers, grid, device, rounds=1, bufs=None):
    """The same stream-ordered rounds with all tiles in this process (one GPU): the neighbour's
    stream waits for an event on the exporter's stream instead of an RCCL transfer.  The face layers
    are allocated once (`bufs`, kept by the caller across calls); an exporter rewrites its layers only
    after every import of the round before has finished (events recorded behind the imports)."""
    import torch
    world = grid[0] * grid[1] * grid[2]
    own = bufs is None                                    # nobody keeps the layers alive after the call: finish before returning
    bufs = {} if bufs is None else bufs
    lay = _local_face_buffers(mappers, world, device, bufs)
    if "streams" not in bufs:
        bufs["streams"] = [torch.cuda.ExternalStream(m.stream_handle(), device=device) for m in mappers]
        bufs["imported"] = []
    streams = bufs["streams"]
    for k in range(rounds + 1):
        evs = []
        for r, m in enumerate(mappers):
            for ev in bufs["imported"]:                   # the layers of the round before have been read
                streams[r].wait_event(ev)
            m.halo_export_all_dev({face: lay[(r, face)].data_ptr() for face in neighbours(r, world)})
            ev = torch.cuda.Event()
            ev.record(streams[r])
            evs.append(ev)
        done = []
        for r, m in enumerate(mappers):
            for ev in evs:
                streams[r].wait_event(ev)
            m.halo_import_all_dev({face: lay[(nb, face ^ 1)].data_ptr() for face, nb in neighbours(r, world).items()})
            ev = torch.cuda.Event()
            ev.record(streams[r])
            done.append(ev)
            if k == 0:
                m.merge_end()
            else:
                m.refine_async()
        bufs["imported"] = done
    if own:
        for m in mappers:
            m.sync()
    return rounds


def exchange_until_stable(mapper, dist, rank, world_size, device=None, max_rounds=64, group=None, sparse=False):
    """One tile per rank: face layers travel with torch.distributed point-to-point ops (RCCL over
    xGMI with backend "nccl", gloo on CPU); a 1-int all-reduce(sum) of the seed counts is the
    convergence test.  Returns the refinement rounds run.  sparse=True: only the known voxels of a layer travel
    (gie_halo_export_sparse); the neighbours tell each other the entry counts first, so a round is two batches."""
    import torch
    from .mapper import HALO_DTYPE, HALO_ENTRY_DTYPE
    nbs = neighbours(rank, world_size)
    rounds = 0
    for k in range(max_rounds + 1):
        sends, recvs, ops = {}, {}, []
        if sparse:
            lays = {face: mapper.halo_export_sparse(face) for face in sorted(nbs)}
            cs = {face: torch.tensor([lays[face].shape[0]], dtype=torch.int64) for face in lays}
            cr = {face: torch.zeros(1, dtype=torch.int64) for face in lays}
            if device is not None:
                cs = {f: t.to(device) for f, t in cs.items()}; cr = {f: t.to(device) for f, t in cr.items()}
            cops = []
            for face, nb in sorted(nbs.items()):
                cops.append(dist.P2POp(dist.isend, cs[face], nb, group=group))
                cops.append(dist.P2POp(dist.irecv, cr[face], nb, group=group))
            if cops:
                for w in dist.batch_isend_irecv(cops):
                    w.wait()
        for face, nb in sorted(nbs.items()):
            if sparse:
                t = torch.from_numpy(lays[face].view(np.uint8).copy())
                r = torch.empty(int(cr[face].item()) * HALO_ENTRY_DTYPE.itemsize, dtype=torch.uint8)
            else:
                lay = mapper.halo_export(face)
                t = torch.from_numpy(lay.view(np.uint8).copy())
                r = torch.empty_like(t)
            if device is not None:
                t, r = t.to(device), r.to(device)
            sends[face], recvs[face] = t, r
            if t.numel():
                ops.append(dist.P2POp(dist.isend, t, nb, group=group))
            if r.numel():
                ops.append(dist.P2POp(dist.irecv, r, nb, group=group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for face in sorted(nbs):
            if sparse:
                mapper.halo_import_sparse(face, recvs[face].cpu().numpy().view(HALO_ENTRY_DTYPE))
            else:
                mapper.halo_import(face, recvs[face].cpu().numpy().view(HALO_DTYPE))
        if k == 0:
            mapper.merge_end()
            continue
        n = torch.tensor([mapper.refine()], dtype=torch.int64, device=device if device is not None else "cpu")
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
        rounds += 1
        if int(n.item()) == 0:
            break
    return rounds


def init_transport(torch, dist, rank, world_size, device, want="nccl", preflight_timeout_s=120):
    """The process group(s) of a tiled run.  The control plane (barriers, timing reductions, the agreement below) is always a
    gloo group — it exists wherever torch.distributed does.  The data plane is RCCL (backend "nccl", over xGMI) when `want`
    says so AND a PRE-FLIGHT on every rank succeeds: group creation, one all-reduce and one ring send / receive of a device
    buffer, checked.  Every rank then learns over gloo whether ANY rank failed, so that all of them take the same path: RCCL, or
    the face layers staged through the host over gloo with the reason recorded.  Returns {"backend", "group", "note"}: `group`
    is what the exchange functions above take (None = the default gloo group)."""
    import datetime
    if not dist.is_initialized():
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
    info = {"backend": "gloo", "group": None, "note": None}
    if want != "nccl" or world_size < 2:
        return info
    err, g = None, None
    try:
        g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=preflight_timeout_s))
        t = torch.ones(1, device=device)
        dist.all_reduce(t, group=g)
        snd = torch.full((4096,), rank % 251, dtype=torch.uint8, device=device)
        rcv = torch.empty_like(snd)
        ops = [dist.P2POp(dist.isend, snd, (rank + 1) % world_size, group=g), dist.P2POp(dist.irecv, rcv, (rank - 1) % world_size, group=g)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize(device)
        if int(t.item()) != world_size or int(rcv[0].item()) != ((rank - 1) % world_size) % 251 or int(rcv[-1].item()) != int(rcv[0].item()):
            err = "pre-flight data mismatch"
    except Exception as e:            # noqa: BLE001 — whatever RCCL raises (no xGMI, IPC refused, a peer that never arrives) is a reason to fall back
        err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
    flag = torch.tensor([1 if err else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)          # over gloo: one answer for everybody
    if int(flag.item()) == 0:
        info.update(backend="nccl", group=g)
    else:
        info["note"] = "RCCL pre-flight failed (%s): face layers staged through the host over gloo" % (err or "on another rank")
    return info
