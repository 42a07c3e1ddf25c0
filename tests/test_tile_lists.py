"""Tile lists vs volume sweeps (fuse, Mark, obtainFrontiers, commit): every kernel picks per map
update from the length of its list; GIE_TILE_LIST forces one.  The library reads the variable once per process, so every
mode runs in its own interpreter.  CPU: the test-only emulation (which walks the same lists);
GPU: tests/test_gpu_parity.py::test_tile_list_and_sweep_modes_agree_with_the_oracle."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CODE = (
    "import sys; sys.path[:0] = [%r, %r, %r]\n"
    "import parity\n"
    "from emu_py import EmuMapper\n"
    "from oracle_py import OracleMapper\n"
    "for sensor, fast, size in (('mixed', False, (56, 48, 24)), ('lidar_points', False, (48, 48, 40)), ('depth', True, (40, 40, 16))):\n"
    "    sc = parity.Scenario('lists_' + sensor, size, sensor=sensor, frames=5, fast_mode=fast, lidar_az=360, delta_vox=6)\n"
    "    parity.run_and_compare(sc, OracleMapper, EmuMapper)\n"
    "print('ok')\n"
) % (ROOT, os.path.join(ROOT, "gie-mapping_amd"), HERE)


@pytest.mark.parametrize("mode", ["0", "1"])
def test_both_modes_match_the_oracle_on_cpu(mode):
    from emu_py import load
    load()                                            # build the emulation once, outside the children
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, GIE_TILE_LIST=mode), capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_long_run_there_and_back_on_cpu():
    """Thirty map updates, the robot travelling away and back (blocks leave and re-enter the local
    volume, tile flags and lists are reused frame after frame), kernels choosing lists or sweeps
    themselves: emulation == oracle, every array, every frame."""
    import parity
    from emu_py import EmuMapper
    from oracle_py import OracleMapper
    sc = parity.Scenario("long_run_cpu", (48, 40, 24), sensor="mixed", frames=30, delta_vox=7, yaw_deg=23.0, n_boxes=60,
                         extent=(8.0, 4.0, 1.2), toggle=0.3)
    orig = sc.frames_iter

    def there_and_back():
        fr = list(orig())
        half = len(fr) // 2
        for k, f in enumerate(fr):
            yield f if k < half else (fr[len(fr) - 1 - k][0], fr[len(fr) - 1 - k][1]) + f[2:]
    sc.frames_iter = there_and_back
    out = parity.run_and_compare(sc, OracleMapper, EmuMapper)
    assert len(out) == 30 and sum(s["visits_c"] for s in out) > 0
