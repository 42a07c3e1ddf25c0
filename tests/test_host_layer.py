"""C++ host layer (gie-mapping_amd/host): parameters, sensor adapters, external-obstacle
clustering, CSV log, accuracy check and the publishMap sequence, replayed by gie_driver.

On CPU the driver is linked against the test-only emulation of the device logic (tests/emu); the
expected maps come from the oracle driven from Python with the same frames."""
import os
import struct
import subprocess

import numpy as np
import pytest

import gie
from gie import scenes
from emu_py import load as load_emu, EMU_DIR
from oracle_py import OracleMapper
from parity import Scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "gie-mapping_amd", "host")
KIND = {"depth": 0, "scan2d": 1, "multiscan": 2, "pointcloud": 3, "ringcloud": 4, "extcloud": 5}


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    load_emu()
    exe = str(tmp_path_factory.mktemp("host") / "gie_driver_emu")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(HOST, "gie_driver.cpp"), "-o", exe,
                           "-L" + EMU_DIR, "-lgie_emu", "-Wl,-rpath," + EMU_DIR])
    return exe


def write_frames(path, records):
    with open(path, "wb") as f:
        f.write(b"GIEF" + struct.pack("<II", 1, len(records)))
        for kind, pos, q, ip, fp, data in records:
            data = np.ascontiguousarray(data, np.float32).ravel()
            ip = list(ip) + [0] * (4 - len(ip))
            fp = list(fp) + [0.0] * (6 - len(fp))
            f.write(struct.pack("<i3f4fi4i6f", KIND[kind], *pos, *q, data.size, *ip, *fp))
            f.write(data.tobytes())


def yaml_text(sc, extra=""):
    return ("for_motion_planner: %s\nrobot_r: 0.4\nvoxel_width: %g   # metres\n\nlocal_size_x: %g\nlocal_size_y: %g\n"
            "local_size_z: %g\noccupancy_threshold: 180\nugv_height: -1\nogm:\n  min_height: %g\n  max_height: %g\n\n"
            "wave:\n  fast_mode: %s\n  cutoff_dist: %g\nhash:\n  bucket_max: 20000\n  block_max: 21997\nlog_name: \"x.csv\"\n%s"
            % ("true" if sc.for_motion_planner else "false", sc.voxel, sc.size[0] * sc.voxel + 1e-4, sc.size[1] * sc.voxel + 1e-4,
               sc.size[2] * sc.voxel + 1e-4, sc.min_h, sc.max_h, "true" if sc.fast_mode else "false", sc.cutoff_dist, extra))


def ring_bin(pts, rings):
    """numpy statement of the ring binning (vlp16_map_maker.cpp:73-147): last writer wins."""
    img = np.full((16, 440), np.inf, np.float32)
    res = np.float32(2.0 * np.pi / 440)
    # the C library's atan2f (what the C++ adapter calls): numpy's float32 arctan2 can differ in the last bit,
    # which moves points that sit on a bin edge
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype, libm.atan2f.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
    ang = np.array([libm.atan2f(float(y), float(x)) for x, y in zip(pts[:, 0], pts[:, 1])], np.float32)
    b = ((ang + np.float32(np.pi)) / res).astype(np.int32)
    r = np.sqrt(pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]).astype(np.float32)
    for i in range(len(pts)):
        if 0 <= b[i] < 440 and rings[i] < 16:
            img[rings[i], b[i]] = r[i]
    return img


def load_out(prefix, size):
    shp = (size[2], size[1], size[0])
    return dict(edt=np.fromfile(prefix + ".edt.f32", np.float32).reshape(shp), type=np.fromfile(prefix + ".type.i8", np.int8).reshape(shp),
                dist_sq=np.fromfile(prefix + ".dist.i32", np.int32).reshape(shp), coc=np.fromfile(prefix + ".coc.i32", np.int32).reshape(shp + (3,)))


def test_driver_replay_matches_oracle(driver, tmp_path):
    _replay_mixed(driver, tmp_path, (48, 40, 16))


@pytest.mark.gpu
def test_driver_on_gpu_matches_oracle(tmp_path):
    """The same replay through the real binary (linked against libgie_hip.so) on the MI355X."""
    import __graft_entry__ as ge
    _replay_mixed(ge.build_host(), tmp_path, (96, 80, 32))


def _replay_mixed(driver, tmp_path, size):
    sc = Scenario("host_mixed", size, voxel=0.1, sensor="mixed", frames=5, cutoff_dist=1.5, for_motion_planner=True)
    records, frames = [], list(sc.frames_iter())
    world = scenes.BoxWorld(sc.seed, extent=sc.extent, n_boxes=sc.n_boxes, toggle_frac=sc.toggle)
    for pos, q, kind, data, kw in frames:
        if kind == "depth":
            records.append(("depth", pos, q, (data.shape[0], data.shape[1], 1), (kw["cx"], kw["cy"], kw["fx"], kw["fy"]), data))
        elif kind == "pointcloud":
            records.append(("pointcloud", pos, q, (), (), data))
        else:
            records.append(("multiscan", pos, q, (data.shape[1], data.shape[0]), (100.0, kw["theta_inc"], kw["theta_min"], kw["phi_inc"], kw["phi_min"]), data))
    # one more frame as a raw ring cloud through the Vlp16 adapter
    rpos, rq = scenes.pose(len(frames), sc.voxel)
    pts, rng = scenes.lidar_frame(world, len(frames), rpos, rq, az=900, max_range=30.0)
    rings = np.repeat(np.arange(16), 900)[np.isfinite(rng).ravel()]
    assert len(rings) == len(pts)
    cloud = np.concatenate([pts, np.zeros((len(pts), 1), np.float32), rings[:, None].astype(np.float32)], 1)
    records.append(("ringcloud", rpos, rq, (0,), (), cloud))
    fpath, ypath, out, log = (str(tmp_path / n) for n in ("in.gief", "cfg.yaml", "out", "run.csv"))
    write_frames(fpath, records)
    open(ypath, "w").write(yaml_text(sc))
    res = subprocess.run([driver, "--frames", fpath, "--yaml", ypath, "--out", out, "--log", log], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got = load_out(out, sc.size)

    o = OracleMapper(sc.config())
    for pos, q, kind, data, kw in frames:
        o.update(pos, q, kind, data, **({**kw, "max_r": 100.0} if kind == "multiscan" else kw))
    o.set_pose(rpos, rq)
    o.ogm_multiscan(ring_bin(pts, rings), theta_inc=np.float32(2 * np.pi / 440), theta_min=np.float32(-np.pi),
                    phi_inc=np.float32(2.0 / 180 * np.pi), phi_min=np.float32(-15.0 / 180 * np.pi), max_r=10.0)
    o.set_ext_boxes(np.array([[-3.6, -3.2, 0.2]], np.float32), np.array([[4.4, 3.4, 2.6]], np.float32), np.zeros(1, np.uint8))
    o.fuse(); o.batch_edt(); o.merge()
    want = o.read_local()
    for key in ("type", "dist_sq", "coc"):
        assert np.array_equal(got[key], want[key]), key
    assert np.allclose(got["edt"], want["edt"], rtol=1e-6, atol=0)
    # CostMap payload of the last frame
    payload, hdr = o.read_costmap()
    raw = np.fromfile(out + ".costmap.bin", np.uint8)
    assert tuple(np.frombuffer(raw[:12].tobytes(), np.int32)) == tuple(sc.size)
    og = np.frombuffer(raw[12:28].tobytes(), np.float32)
    assert np.allclose(og, [hdr.x_origin, hdr.y_origin, hdr.z_origin, hdr.width])
    assert raw[28:].tobytes() == np.ascontiguousarray(payload).tobytes()
    # log: header + one row per map frame, quoted strings, comma separated
    rows = open(log).read().strip().split("\n")
    assert rows[0].startswith('"Occupancy time","EDT time","RMSE",')
    assert len(rows) == 1 + len(records)
    assert all(len(r.rstrip(",").split(",")) == 8 for r in rows)
    # the CPU mirror built from the changed-block stream (display_glb_* default to true) equals the global map
    raw = np.fromfile(out + ".mirror.bin", np.uint8)
    nb = int(np.frombuffer(raw[:4].tobytes(), np.int32)[0])
    keys = np.frombuffer(raw[4:4 + 12 * nb].tobytes(), np.int32).reshape(nb, 3)
    blocks = np.frombuffer(raw[4 + 12 * nb:].tobytes(), gie.mapper.VOXEL_DTYPE).reshape(nb, 512)
    assert nb > 0 and len({tuple(k) for k in keys.tolist()}) == nb
    j = np.arange(512)
    for k, blk in list(zip(keys, blocks))[::7]:
        xyz = np.stack([k[0] * 8 + (j >> 6), k[1] * 8 + ((j >> 3) & 7), k[2] * 8 + (j & 7)], 1).astype(np.int32)
        g = o.query_global(xyz)
        for f in ("vox_type", "dist_sq", "coc"):
            assert np.array_equal(blk[f], g[f]), (tuple(k), f)
    o.close()


def dbscan_boxes(pts, is_3d):
    """Plain restatement of the clustering rule for the test (seed takes its neighbourhood; growth
    through points with >= 3 neighbours; groups >= 4)."""
    n = len(pts)
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    nb = [np.nonzero(d2[i] <= np.float32(0.3) ** 2)[0] for i in range(n)]
    state = np.zeros(n, int)
    boxes = []
    for s in range(n):
        if state[s] == 2:
            continue
        grp = [s]
        state[s] = 2
        for j in nb[s]:
            if j != s:
                grp.append(j); state[j] = 1
        h = 1
        while h < len(grp):
            p = grp[h]; h += 1
            if state[p] == 2:
                continue
            if len(nb[p]) >= 3:
                for j in nb[p]:
                    if state[j] == 0:
                        grp.append(j); state[j] = 1
            state[p] = 2
        if len(grp) >= 4:
            g = pts[grp]
            lo, hi = g.min(0), g.max(0)
            if not is_3d:
                lo[2], hi[2] = 0.2, 2.6
            boxes.append((lo, hi))
    return boxes


@pytest.mark.parametrize("is_3d", [False, True])
def test_external_obstacles_cluster_and_fuse(driver, tmp_path, is_3d):
    sc = Scenario("host_ext", (40, 40, 24), voxel=0.1, sensor="depth", frames=2, cutoff_dist=1.5)
    rng = np.random.default_rng(5)
    blobs = [rng.normal(c, 0.08, size=(40, 3)) for c in ([1.0, 0.5, 1.0], [-0.8, -0.6, 0.8], [30.0, 30.0, 1.0])]
    ext = np.concatenate(blobs + [rng.uniform(-1.9, 1.9, size=(6, 3)) + [0, 0, 5.0]]).astype(np.float32)   # + isolated noise
    frames = list(sc.frames_iter())
    records = [("extcloud", (0, 0, 0), (1, 0, 0, 0), (), (), ext)]
    for pos, q, kind, data, kw in frames:
        records.append(("depth", pos, q, (data.shape[0], data.shape[1], 1), (kw["cx"], kw["cy"], kw["fx"], kw["fy"]), data))
    fpath, ypath, out = (str(tmp_path / n) for n in ("in.gief", "cfg.yaml", "out"))
    write_frames(fpath, records)
    open(ypath, "w").write(yaml_text(sc, "is_ext_obsv_3D: %s\n" % ("true" if is_3d else "false")))
    res = subprocess.run([driver, "--frames", fpath, "--yaml", ypath, "--out", out], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    boxes = np.fromfile(out + ".boxes.f32", np.float32).reshape(-1, 7)
    want = dbscan_boxes(ext, is_3d)
    assert len(boxes) == 1 + len(want) and len(want) >= 3
    assert np.allclose(boxes[0], [-3.6, -3.2, 0.2, 4.4, 3.4, 2.6, 0])             # the fence, never active
    for b, (lo, hi) in zip(boxes[1:], want):
        assert np.allclose(b[:3], lo, atol=1e-6) and np.allclose(b[3:6], hi, atol=1e-6)
    # the far blob does not touch the local volume
    assert [int(b[6]) for b in boxes[1:4]] == [1, 1, 0]

    o = OracleMapper(sc.config())
    for pos, q, kind, data, kw in frames:
        o.set_pose(pos, q)
        o.ogm_depth(data, **kw)
        o.set_ext_boxes(boxes[:, :3].copy(), boxes[:, 3:6].copy(), boxes[:, 6].astype(np.uint8))
        o.fuse(); o.batch_edt(); o.merge()
    want_map, got = o.read_local(), load_out(out, sc.size)
    o.close()
    for key in ("type", "dist_sq", "coc"):
        assert np.array_equal(got[key], want_map[key]), key
    # voxels of allocated blocks inside an active box are occupied, never free
    pv = np.array(o_pivot(sc, frames[-1]))
    zz, yy, xx = np.meshgrid(*(np.arange(n) for n in sc.size[::-1]), indexing="ij")
    gp = np.stack([(xx + pv[0]), (yy + pv[1]), (zz + pv[2])], -1).astype(np.float32) * np.float32(sc.voxel)
    inside = ((gp >= boxes[1, :3]) & (gp <= boxes[1, 3:6])).all(-1) | ((gp >= boxes[2, :3]) & (gp <= boxes[2, 3:6])).all(-1)
    assert inside.sum() > 0
    if not is_3d:                          # the 2-D boxes span z = 0.2..2.6 and reach observed blocks
        assert (got["type"][inside] == 2).sum() > 0
    assert set(np.unique(got["type"][inside])) <= {0, 2}


def o_pivot(sc, frame):
    o = OracleMapper(sc.config())
    o.set_pose(frame[0], frame[1])
    p = o.pivot()
    o.close()
    return p


def test_rms_check_and_sets(driver, tmp_path):
    _rms_check(driver, tmp_path, (32, 32, 12))


@pytest.mark.gpu
def test_rms_check_and_csv_log_on_gpu(tmp_path):
    """The accuracy profiler (gt_checker.h:30-80: RMSE of the EDT against the nearest occupied voxel) and the
    CSV timing log (simple_logger.h:18-71) through the real binary on the MI355X, against a KD-tree in Python."""
    import __graft_entry__ as ge
    _rms_check(ge.build_host(), tmp_path, (64, 64, 24))


def _rms_check(driver, tmp_path, size):
    from scipy.spatial import cKDTree
    sc = Scenario("host_rms", size, voxel=0.1, sensor="depth", frames=3, cutoff_dist=100.0)
    frames = list(sc.frames_iter())
    records = [("depth", pos, q, (d.shape[0], d.shape[1], 1), (kw["cx"], kw["cy"], kw["fx"], kw["fy"]), d) for pos, q, _, d, kw in frames]
    fpath, out, log = (str(tmp_path / n) for n in ("in.gief", "out", "run.csv"))
    write_frames(fpath, records)
    args = [driver, "--frames", fpath, "--out", out, "--rms", "--log", log, "--set", "voxel_width=0.1", "--set", "local_size_x=%.4f" % (size[0] * 0.1 + 1e-4),
            "--set", "local_size_y=%.4f" % (size[1] * 0.1 + 1e-4), "--set", "local_size_z=%.4f" % (size[2] * 0.1 + 1e-4), "--set", "wave/cutoff_dist=100", "--set", "wave/fast_mode=false",
            "--set", "ogm/min_height=-1000", "--set", "ogm/max_height=1000", "--set", "profile_loc_rms=true"]
    res = subprocess.run(args, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got = load_out(out, sc.size)
    occ = np.argwhere(got["type"] == 2)
    known = np.argwhere(got["type"] != 0)
    d, _ = cKDTree(occ).query(known)
    err = d * sc.voxel - got["edt"][tuple(known.T)].astype(np.float64) * sc.voxel
    line = [l for l in res.stdout.split("\n") if l.startswith("rms")][0].split()
    assert abs(float(line[1]) - np.sqrt((err ** 2).mean())) < 1e-5
    assert int(line[9]) == len(known)
    rows = open(log).read().strip().split("\n")
    assert [c.strip('"') for c in rows[0].split(",")[:3]] == ["Occupancy time", "EDT time", "RMSE"] and len(rows) == 1 + len(frames)
    last = rows[-1].split(",")
    assert abs(float(last[2]) - float(line[1])) < 1e-4
    assert float(last[0]) > 0 and float(last[1]) > 0        # the two timing columns of the reference's log


def test_bad_inputs_fail_loudly(driver, tmp_path):
    bad = str(tmp_path / "bad.gief")
    open(bad, "wb").write(b"NOPE")
    assert subprocess.run([driver, "--frames", bad], capture_output=True).returncode == 1
    assert subprocess.run([driver, "--frames", str(tmp_path / "missing")], capture_output=True).returncode == 1
    assert subprocess.run([driver], capture_output=True).returncode == 2


# ------------------------------------------------------------------ the tiled C++ node (gie_tiled.hpp), ranks as processes

def _free_port_block(n):
    import socket
    for base in range(29500, 60000, 97):
        socks = []
        try:
            for k in range(n):
                s = socket.socket(); s.bind(("127.0.0.1", base + k)); socks.append(s)
            return base
        except OSError:
            continue
        finally:
            for s in socks:
                s.close()
    raise RuntimeError("no free port block")


def _tiled_replay(exe, tmp_path, world, tile, frames_n, extra=()):
    """Runs `world` ranks of gie_tiled_driver (socket transport) on the multiscan frames of test_tiling_halo and returns
    the per-rank outputs; the expectation is the in-process tiled oracle on the same frames."""
    import test_tiling_halo as T
    from gie import tiling
    frames = T._sensor_frames(frames_n)
    records = [("multiscan", pos, q, (img.shape[1], img.shape[0]), (100.0, T.KW["theta_inc"], T.KW["theta_min"], T.KW["phi_inc"], T.KW["phi_min"]), img)
               for pos, q, img in frames]
    fpath, out = str(tmp_path / "in.gief"), str(tmp_path / "out")
    write_frames(fpath, records)
    port = _free_port_block(world)
    args = ["--frames", fpath, "--world", str(world), "--port", str(port), "--out", out, "--set", "voxel_width=%g" % T.W,
            "--set", "local_size_x=%.4f" % (tile[0] * T.W + 1e-4), "--set", "local_size_y=%.4f" % (tile[1] * T.W + 1e-4),
            "--set", "local_size_z=%.4f" % (tile[2] * T.W + 1e-4), "--set", "wave/cutoff_dist=1.0", "--set", "wave/fast_mode=false",
            "--set", "ogm/min_height=-1000", "--set", "ogm/max_height=1000", "--set", "display_glb_edt=false", "--set", "display_glb_ogm=false"] + list(extra)
    procs = [subprocess.Popen([exe, "--rank", str(r)] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][1])
    got = [load_out(out + ".r%d" % r, tile) for r in range(world)]
    # expectation: the oracle's tiles in one process, exchange until stable
    grid = tiling.tile_grid(world)
    whole = tuple(grid[i] * tile[i] for i in range(3))
    cfg = gie.make_config(T.W, tile, cutoff_dist=1.0)
    ms = []
    for r in range(world):
        m = OracleMapper(cfg); m.set_tile(tiling.tile_offset_voxels(r, world, tile), whole); ms.append(m)
    rounds = 0
    try:
        for pos, q, img in frames:
            for m in ms:
                m.update(pos, q, "multiscan", img, tiled=True, **T.KW)
            rounds += tiling.exchange_until_stable_local(ms, grid)
        want = [m.read_local() for m in ms]
    finally:
        for m in ms:
            m.close()
    return got, want, rounds, [o[0] for o in outs]


@pytest.fixture(scope="module")
def tiled_driver(tmp_path_factory):
    load_emu()
    exe = str(tmp_path_factory.mktemp("host") / "gie_tiled_driver_emu")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(HOST, "gie_tiled_driver.cpp"), "-o", exe,
                           "-L" + EMU_DIR, "-lgie_emu", "-Wl,-rpath," + EMU_DIR, "-lpthread"])
    return exe


@pytest.mark.parametrize("world,tile", [(2, (32, 32, 16)), (4, (24, 24, 16))])
def test_tiled_cpp_node_ranks_as_processes(tiled_driver, tmp_path, oracle_lib, world, tile):
    """gie_host::TiledMapper with the socket transport, one process per rank (the C++ node tiles WITHOUT PyTorch): every
    rank's tile must equal the corresponding tile of the in-process tiled oracle, and the ranks must have run the same
    number of refinement rounds (the all-reduce of the seed count is the convergence test)."""
    got, want, rounds, stdout = _tiled_replay(tiled_driver, tmp_path, world, tile, 5)
    for r in range(world):
        for key in ("type", "dist_sq", "coc"):
            assert np.array_equal(got[r][key], want[r][key]), (r, key)
        assert np.allclose(got[r]["edt"], want[r]["edt"], rtol=1e-6, atol=0)
        assert stdout[r].strip().split()[-1] == str(rounds), (stdout[r], rounds)


@pytest.mark.gpu
def test_tiled_cpp_node_on_gpu_two_ranks_share_the_device(tmp_path, oracle_lib):
    """The same through the real binary on the MI355X: two rank processes on ONE GPU, socket transport with host staging
    (RCCL needs one GPU per rank; its transport is compiled by build() and exercised on a multi-GPU node only)."""
    import __graft_entry__ as ge
    got, want, rounds, stdout = _tiled_replay(ge.build_tiled_driver(), tmp_path, 2, (32, 32, 16), 5, extra=("--device", "0"))
    for r in range(2):
        for key in ("type", "dist_sq", "coc"):
            assert np.array_equal(got[r][key], want[r][key]), (r, key)
