"""Changed-block streaming (gie_stream_enable / gie_stream_changed): the CPU mirror of the global
map the reference keeps in hash_table_H_std (glb_hash_map.cu:209-247, README.md:163-170).

Checked: the device path and the oracle flag the same blocks and deliver the same contents; a
mirror fed only by the stream equals the global map (type, distance, closest obstacle) for every
voxel, frame after frame; partial delivery keeps the rest flagged; nothing is flagged when off."""
import numpy as np
import pytest

import gie
from emu_py import EmuMapper
from oracle_py import OracleMapper
from parity import Scenario, _feed

EMPTY = 999999


def _frame(m, pos, q, kind, data, kw):
    m.set_pose(pos, q)
    _feed(m, kind, data, kw)
    m.fuse(); m.batch_edt(); m.merge()


def _by_key(keys, blocks):
    return {tuple(k): b for k, b in zip(keys.tolist(), blocks)}


def _check_mirror(m, mirror, size, rng):
    """every voxel around the local volume: the mirror (or the defaults for a block it never got)
    equals the global map"""
    pv = np.array(m.pivot())
    lo, hi = pv - 10, pv + np.array(size) + 10
    xyz = rng.integers(lo, hi, size=(6000, 3)).astype(np.int32)
    g = m.query_global(xyz)
    for i, (x, y, z) in enumerate(xyz.tolist()):
        blk = mirror.get((x >> 3, y >> 3, z >> 3))
        if blk is None:
            assert g["vox_type"][i] == 0 and g["dist_sq"][i] == EMPTY and (g["coc"][i] == EMPTY).all(), (x, y, z)
        else:
            v = blk[(x & 7) * 64 + (y & 7) * 8 + (z & 7)]        # get_voxID_in_VB
            assert v["vox_type"] == g["vox_type"][i] and v["dist_sq"] == g["dist_sq"][i] and (v["coc"] == g["coc"][i]).all(), (x, y, z)


def _run(make, sc, also_oracle=True):
    a, b = make(sc.config()), (OracleMapper(sc.config()) if also_oracle else None)
    rng = np.random.default_rng(3)
    mirror, flagged = {}, []
    try:
        a.stream_enable(True)
        if b:
            b.stream_enable(True)
        for k, (pos, q, kind, data, kw) in enumerate(sc.frames_iter()):
            _frame(a, pos, q, kind, data, kw)
            ka, ba, na = a.stream_changed()
            assert na == len(ka) and a.stream_count() == 0
            assert len({tuple(x) for x in ka.tolist()}) == len(ka)          # no duplicates
            if b:
                _frame(b, pos, q, kind, data, kw)
                kb, bb, nb = b.stream_changed()
                da, db = _by_key(ka, ba), _by_key(kb, bb)
                assert set(da) == set(db), "frame %d: flagged blocks differ (%d vs %d)" % (k, len(da), len(db))
                for key in da:
                    assert da[key].tobytes() == db[key].tobytes(), "frame %d block %s" % (k, key)
            mirror.update(_by_key(ka, ba))
            _check_mirror(a, mirror, sc.size, rng)
            flagged.append(na)
    finally:
        a.close()
        if b:
            b.close()
    return flagged


@pytest.mark.parametrize("fast", [False, True])
def test_stream_matches_oracle_and_mirror_tracks_map(fast):
    sc = Scenario("stream", (48, 40, 24), sensor="mixed", frames=6, cutoff_dist=2.0, fast_mode=fast)
    flagged = _run(EmuMapper, sc)
    assert flagged[0] > 0 and all(f > 0 for f in flagged)
    # after the first frame only part of the map changes
    assert min(flagged[1:]) < flagged[0] * 3


def test_partial_delivery_and_off():
    sc = Scenario("stream_part", (40, 40, 16), sensor="depth", frames=2)
    frames = list(sc.frames_iter())
    for make in (EmuMapper, OracleMapper):
        m = make(sc.config())
        _frame(m, *frames[0])
        assert m.stream_count() == 0                       # off by default: nothing flagged
        m.stream_enable(True)
        _frame(m, *frames[1])
        total = m.stream_count()
        assert total > 7
        seen = {}
        left = total
        while left:
            k, b, before = m.stream_changed(max_blocks=3)
            assert before == left and len(k) == min(3, left)
            assert not (set(map(tuple, k.tolist())) & set(seen))
            seen.update(_by_key(k, b))
            left -= len(k)
        assert len(seen) == total and m.stream_count() == 0
        m.stream_enable(False)
        m.close()


def test_stream_in_several_chunks(monkeypatch):
    """more flagged blocks than one trip through the staging buffers carries"""
    monkeypatch.setenv("GIE_STREAM_CHUNK_BLOCKS", "7")
    sc = Scenario("stream_chunks", (40, 40, 16), sensor="depth", frames=3)
    flagged = _run(EmuMapper, sc)
    assert max(flagged) > 3 * 7


@pytest.mark.gpu
def test_stream_on_gpu_matches_oracle(monkeypatch):
    monkeypatch.setenv("GIE_STREAM_CHUNK_BLOCKS", "500")      # several chunks: exercises the double-buffered copy
    from hooks_py import HooksMapper                           # (the switch exists in the test build of the library only)
    sc = Scenario("stream_gpu", (160, 160, 64), sensor="mixed", frames=4, cutoff_dist=2.0, img=(240, 320, 260.0), max_depth=10.0,
                  extent=(7.0, 7.0, 2.5))
    flagged = _run(HooksMapper, sc)
    assert max(flagged) > 1000


@pytest.mark.gpu
def test_stream_on_gpu_mirror_large():
    sc = Scenario("stream_gpu_big", (256, 256, 64), sensor="lidar_points", frames=3, cutoff_dist=3.0, lidar_az=1800, extent=(10.0, 10.0, 2.5))
    _run(gie.Mapper, sc, also_oracle=False)
