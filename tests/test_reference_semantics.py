"""Boundary semantics with HAND-MADE expectations.

Every expected value below was typed in from a reading of the reference SOURCE (file:line given at each
check) and worked out on paper -- none is produced by the oracle, the emulation or the HIP library.  The
point: oracle and HIP path have one author, so a shared misreading is invisible to every
HIP-vs-oracle parity test; these tests fail for BOTH when the reading is wrong.  (They did: until round 4
`SeenDist.o` carried the raw voxel type 0..3 on both sides; the reference's field is a `bool`.)

The same scenes run on the oracle and the host emulation (CPU) and on the HIP library (`-m gpu`).
"""
import struct

import numpy as np
import pytest

import gie
from emu_py import EmuMapper
from oracle_py import OracleMapper

UNKNOWN, FREE, OCC, FNT = 0, 1, 2, 3
EMPTY = 999999            # EMPTY_VALUE, voxmap_utils.cuh:8

BACKENDS = [pytest.param(OracleMapper, id="oracle"), pytest.param(EmuMapper, id="emulation"),
            pytest.param(gie.Mapper, id="hip", marks=pytest.mark.gpu)]


def _cfg(size, voxel=0.5, **kw):
    kw.setdefault("cutoff_dist", 2.0)
    return gie.make_config(voxel, size, **kw)


def _update(m, labels=None, points=None, pos=(0.0, 0.0, 0.0)):
    m.set_pose(pos, (1.0, 0.0, 0.0, 0.0))
    if labels is not None:
        m.ogm_labels(labels)
    else:
        m.ogm_pointcloud(np.asarray(points, np.float32).reshape(-1, 3))
    m.fuse(); m.batch_edt(); m.merge()
    m.sync()


# --------------------------------------------------------------------------------------------------------------
# SeenDist / CostMap payload: local_batch.h:19-24 (struct { float d; bool s; bool o; }), :382-391 (convertCostMap:
# d = edt_H[i], o = glb_type_H[i] -- a char converted to bool, so 1 for every known type), msg/CostMap.msg,
# volumetric_mapper.cpp:375-389 (header: sizes, origin = coord2pos(_pvt), width, type = TYPE_EDT = 1).
# _edt_D is memset to 0 at start-up (local_batch.h:70-71) and only written for known voxels
# (unify_helper.cuh:462-463), as sqrtf(dist_sq) in VOXEL units (:496).
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("make", BACKENDS)
def test_seendist_bytes_of_a_hand_made_scene(make):
    # 8x8x8 voxels of 0.5 m around the origin: _pvt = round(0 / 0.5) - 8/2 = -4 (local_batch.h:129-142), so local = global + 4.
    # One scan: a 3x3x3 cube of FREE voxels, local (3..5)^3, with its +x face centre (5,4,4) OCCUPIED.
    lab = np.zeros((8, 8, 8), np.int8)                     # [z][y][x]
    lab[3:6, 3:6, 3:6] = FREE
    lab[4, 4, 5] = OCC
    m = make(_cfg((8, 8, 8)))
    try:
        _update(m, labels=lab)
        pay, hdr = m.read_costmap()
        raw = pay.view(np.uint8).reshape(8, 8, 8, 8)       # the bytes a planner receives, 8 per voxel, x fastest
        # (x, y, z)      type after the update                                   d (voxels)          o
        # (0,0,0)        never observed: UNKNOWN                                  0.0 (the memset)    0
        # (4,4,4)        FREE, all six neighbours observed -> stays FREE          |(5,4,4)-(4,4,4)| = 1.0   1
        # (5,4,4)        OCCUPIED (0.8*250 = 200 > 180)                           0.0                 1
        # (3,3,3)        FREE next to the unobserved (2,3,3) -> FNT               sqrt(4+1+1) = sqrt 6      1
        expect = {(0, 0, 0): struct.pack("<fBBBB", 0.0, 0, 0, 0, 0),
                  (4, 4, 4): struct.pack("<fBBBB", 1.0, 0, 1, 0, 0),
                  (5, 4, 4): struct.pack("<fBBBB", 0.0, 0, 1, 0, 0),
                  (3, 3, 3): bytes.fromhex("71c41c40") + bytes([0, 1, 0, 0])}     # 0x401cc471 = sqrtf(6)
        for (x, y, z), b in expect.items():
            assert raw[z, y, x].tobytes() == b, ((x, y, z), raw[z, y, x].tobytes().hex(), b.hex())
        # o is a bool everywhere: 27 observed voxels, nothing but 0 / 1 in the byte
        assert set(np.unique(pay["o"]).tolist()) == {0, 1} and int(pay["o"].sum()) == 27
        assert int(pay["s"].sum()) == 0
        t = m.read_local()["type"]
        assert (t[0, 0, 0], t[4, 4, 4], t[4, 4, 5], t[3, 3, 3]) == (UNKNOWN, FREE, OCC, FNT)
        assert (hdr.x_size, hdr.y_size, hdr.z_size) == (8, 8, 8)
        assert (hdr.x_origin, hdr.y_origin, hdr.z_origin) == (-2.0, -2.0, -2.0)          # coord2pos(-4) = -4 * 0.5
        assert hdr.width == 0.5 and hdr.type == 1
    finally:
        m.close()


# --------------------------------------------------------------------------------------------------------------
# "See nothing": no OCCUPIED voxel in the volume and nothing remembered.  batch EDT: coc invalid -> Mark stores
# (EMPTY_VALUE, 0xffffffff) (unify_helper.cuh:201-273) -> UpdateHashBatch (:467-475) writes
# _edt_D = _max_loc_dist_sq = X^2 + Y^2 + Z^2 (local_batch.h:47; a SQUARED distance, quirk) and `continue`s BEFORE it
# touches the hashed voxel: the stored distance / obstacle keep their defaults.  obtainFrontiers leaves such a voxel
# before it looks at any neighbour (`if(!cur_coc_in_loc) continue;`, :299-303 -- "due to limited observation (or see
# nothing)"), so NO voxel is flagged FNT, not even one on a face of the volume next to unobserved space.
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("make", BACKENDS)
def test_see_nothing_writes_the_squared_diagonal(make):
    X, Y, Z = 8, 6, 4
    m = make(_cfg((X, Y, Z)))
    try:
        _update(m, labels=np.full((Z, Y, X), FREE, np.int8))
        r = m.read_local()
        assert (r["edt"] == np.float32(8 * 8 + 6 * 6 + 4 * 4)).all()                     # 116.0
        assert (r["dist_sq"] == EMPTY).all() and (r["coc"] == EMPTY).all()
        assert (r["type"] == FREE).all()
        # the global voxels: occupancy 0.5*0 + 0.5*0 = 0 -> clamped to 1 (voxmap_utils.cuh:186-192), type FREE,
        # distance and obstacle at their constructor defaults (voxmap_utils.cuh:30-40)
        pv = m.pivot()
        assert pv == (-4, -3, -2)
        xyz = np.array([[pv[0] + x, pv[1] + y, pv[2] + z] for z in range(Z) for y in range(Y) for x in range(X)], np.int32)
        g = m.query_global(xyz)
        assert (g["occ_val"] == 1).all() and (g["vox_type"] == FREE).all()
        assert (g["dist_sq"] == EMPTY).all() and (g["coc"] == EMPTY).all()
    finally:
        m.close()


# --------------------------------------------------------------------------------------------------------------
# Occupancy filter, projective scans: updateHashOGMWithSensor (unify_helper.cuh:120-197) calls
# set_hashvoxel_occ_val(vox, 250, 0.8) for an OCCUPIED label and (vox, 0, 0.5) for a FREE one;
# set_hashvoxel_occ_val (voxmap_utils.cuh:181-200): val = a*meas + (1-a)*old (old = 0 if the voxel is UNKNOWN),
# clamp to [1, 254], truncate to unsigned char, OCCUPIED iff occ_val > threshold.
# On paper, in fp32 (1 - 0.8f = 0.19999999):
#   hits   : 200 | 200 + .2*200 = 240 | 200 + .2*240 = 248 | 200 + .2*248 = 249.6 -> 249 | 200 + .2*249 = 249.8 -> 249
#   misses : 124.5 -> 124 | 62 | 31 | 15.5 -> 15 | 7.5 -> 7 | 3.5 -> 3 | 1.5 -> 1 | 0.5 -> clamp 1
#   no label: untouched;   hit again: 200 + .2*1 = 200.2 -> 200
# --------------------------------------------------------------------------------------------------------------
SENSOR_TRAIN = [(OCC, 200, OCC), (OCC, 240, OCC), (OCC, 248, OCC), (OCC, 249, OCC), (OCC, 249, OCC),
                (FREE, 124, FREE), (FREE, 62, FREE), (FREE, 31, FREE), (FREE, 15, FREE), (FREE, 7, FREE), (FREE, 3, FREE),
                (FREE, 1, FREE), (FREE, 1, FREE), (UNKNOWN, 1, FREE), (OCC, 200, OCC)]


@pytest.mark.parametrize("make", BACKENDS)
def test_occupancy_filter_train_projective(make):
    m = make(_cfg((8, 8, 8)))
    m200 = make(_cfg((8, 8, 8), occupancy_threshold=200))
    try:
        for k, (label, occ, vtype) in enumerate(SENSOR_TRAIN):
            lab = np.zeros((8, 8, 8), np.int8)
            lab[4, 4, 4] = label                          # local (4,4,4) = global (0,0,0)
            _update(m, labels=lab)
            g = m.query_global(np.array([[0, 0, 0]], np.int32))[0]
            assert (int(g["occ_val"]), int(g["vox_type"])) == (occ, vtype), (k, label, g)
        # `>` not `>=`: with threshold 200 the first hit (200) leaves the voxel FREE, the second (240) makes it OCCUPIED
        for occ, vtype in ((200, FREE), (240, OCC)):
            lab = np.zeros((8, 8, 8), np.int8)
            lab[4, 4, 4] = OCC
            _update(m200, labels=lab)
            g = m200.query_global(np.array([[0, 0, 0]], np.int32))[0]
            assert (int(g["occ_val"]), int(g["vox_type"])) == (occ, vtype)
    finally:
        m.close(); m200.close()


# --------------------------------------------------------------------------------------------------------------
# Occupancy filter, ray-cast scans: registerLocObs (pntcld_raycast.cu:84-101) counts +1 in the voxel of a point,
# rayCastLoc (ray_cast.h:57-144) counts -1 in the sensor's voxel and in every voxel it steps into until it meets an
# OCCUPIED label; updateHashOGMWithPntCld (unify_helper.cuh:34-118): count > 0 -> (250, a = 1); count < 0 ->
# (0, a = min(1, -count/10)).
# Sensor at the origin, voxel 0.5 m, all points on the +x axis: a point at x = 1.0 lies in voxel 2, one at x = 1.5 in
# voxel 3 and its ray counts -1 in voxels 0, 1, 2.  n copies of a point give count = -n.  The voxel watched: global (2,0,0).
#   hit                 : 250                                       OCCUPIED
#   1 ray through       : (1 - .1f) * 250 = 225                     OCCUPIED
#   1 ray               : .9 * 225 = 202.5 -> 202                   OCCUPIED
#   1 ray               : .9 * 202 = 181.8 -> 181                   OCCUPIED   (181 > 180)
#   1 ray               : .9 * 181 = 162.9 -> 162                   FREE -> FNT
#   3 rays              : .7 * 162 = 113.4 -> 113                   FREE -> FNT
#   10 rays             : a = 1: 0 -> clamp 1                       FREE -> FNT
#   12 rays             : a = min(1, 1.2) = 1: 0 -> 1               FREE -> FNT
#   hit                 : 250                                       OCCUPIED
# (FREE -> FNT: the fused type is FREE; the voxel's closest obstacle (3,0,0) -- the point's own voxel -- lies inside the
#  volume and its neighbours off the axis were never observed, so obtainFrontiers (:440-443) flags it FNT and
#  UpdateHashBatch (:501-504) copies the flag into the hashed voxel, with dist_sq = 1 and coc = (3,0,0).  In the
#  projective train above the lone FREE voxel has no obstacle to point at and stays FREE, see the see-nothing test.)
# --------------------------------------------------------------------------------------------------------------
RAY_TRAIN = [((1.0, 1), 250, OCC), ((1.5, 1), 225, OCC), ((1.5, 1), 202, OCC), ((1.5, 1), 181, OCC), ((1.5, 1), 162, FNT),
             ((1.5, 3), 113, FNT), ((1.5, 10), 1, FNT), ((1.5, 12), 1, FNT), ((1.0, 1), 250, OCC)]


@pytest.mark.parametrize("make", BACKENDS)
def test_occupancy_filter_train_raycast(make):
    m = make(_cfg((16, 8, 8)))
    try:
        for k, ((x, copies), occ, vtype) in enumerate(RAY_TRAIN):
            pts = np.tile(np.array([[x, 0.0, 0.0]], np.float32), (copies, 1))
            m.set_pose((0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0))
            m.ogm_pointcloud(pts)
            cnt = m.read_ogm()["ray_count"]
            pv = m.pivot()
            assert pv == (-8, -4, -4)
            line = cnt[4, 4, 8:13].tolist()                  # global x = 0..4 on the axis
            if x == 1.0:
                assert line == [-copies, -copies, copies, 0, 0], (k, line)
            else:
                assert line == [-copies, -copies, -copies, copies, 0], (k, line)
            assert int(np.abs(cnt).sum()) == sum(abs(v) for v in line)           # nothing else was counted
            m.fuse(); m.batch_edt(); m.merge(); m.sync()
            g = m.query_global(np.array([[2, 0, 0]], np.int32))[0]
            assert (int(g["occ_val"]), int(g["vox_type"])) == (occ, vtype), (k, x, copies, g)
            if vtype == FNT:
                assert int(g["dist_sq"]) == 1 and g["coc"].tolist() == [3, 0, 0]
    finally:
        m.close()


# --------------------------------------------------------------------------------------------------------------
# Block keys and in-block order of the changed-block stream: get_VB_key (voxmap_utils.cuh:92-101) is an ARITHMETIC shift
# by 3 (the `(x & 7) < 0` correction never fires), get_voxID_in_VB (:103-109) = (x&7)*64 + (y&7)*8 + (z&7) -- z fastest.
#   voxel (-1,-1,-1)  -> block (-1,-1,-1), slot 7*64 + 7*8 + 7 = 511
#   voxel (-8, 0, 7)  -> block (-1, 0, 0), slot 0*64 + 0*8 + 7 = 7
#   voxel (-9, 8,-16) -> block (-2, 1,-2), slot 7*64 + 0*8 + 0 = 448
#   voxel ( 5, 2, 1)  -> block ( 0, 0, 0), slot 5*64 + 2*8 + 1 = 337
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("make", BACKENDS)
def test_block_keys_and_in_block_order_for_negative_coordinates(make):
    vox = {(-1, -1, -1): ((-1, -1, -1), 511), (-8, 0, 7): ((-1, 0, 0), 7), (-9, 8, -16): ((-2, 1, -2), 448),
           (5, 2, 1): ((0, 0, 0), 337)}
    X, Y, Z = 32, 32, 40                                   # _pvt = (-16, -16, -20)
    lab = np.zeros((Z, Y, X), np.int8)
    for (x, y, z) in vox:
        lab[z + 20, y + 16, x + 16] = OCC
    m = make(_cfg((X, Y, Z)))
    try:
        m.stream_enable(True)
        _update(m, labels=lab)
        assert m.pivot() == (-16, -16, -20)
        keys, blocks, n = m.stream_changed()
        got = {tuple(k): b for k, b in zip(keys.tolist(), blocks)}
        assert set(got) == {k for k, _ in vox.values()}
        for (x, y, z), (key, slot) in vox.items():
            b = got[key]
            occ = np.flatnonzero(b["vox_type"] == OCC).tolist()
            assert occ == [slot], ((x, y, z), key, occ)
            assert int(b["occ_val"][slot]) == 200 and int(b["dist_sq"][slot]) == 0 and b["coc"][slot].tolist() == [x, y, z]
            others = np.delete(np.arange(512), slot)
            assert (b["vox_type"][others] == UNKNOWN).all() and (b["dist_sq"][others] == EMPTY).all()
    finally:
        m.close()
