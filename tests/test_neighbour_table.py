"""The neighbour table of waves A / B (gie_ctx.g_nbr, round 5): every live block's row names, for each of its six faces, the slot the
hash finds for the block across that face — or a slot whose key is not that block's, which the readers take for "no neighbour"
(a row may outlive a neighbour that block erasure took away).  A row that names NOTHING where a neighbour exists would make a
block-run of waves A / B treat that neighbour as missing; the parity tests would only see it if such a block-run happens, so the
invariant is checked here directly (test hook gie_debug_nbr_check; libgie_hip.so does not export it), after every update of
drives that erase blocks and meet them again.  The CPU emulation keeps the table with the same gie_nbr_link (it does not read it:
its waves are its own), the GPU twin checks the kernels that maintain it on the device."""
import numpy as np
import pytest

import gie
from gie import scenes
from emu_py import EmuMapper


def _drive(mapper_cls, size, retain, steps, step_vox, pool=None):
    """forth along x, a sidestep, back, forth again: erased blocks are allocated again next to blocks that stayed"""
    voxel = 0.1
    cfg = gie.make_config(voxel, size, cutoff_dist=1.0, fast_mode=False, retain_radius_blocks=retain,
                          **({"max_blocks": pool} if pool else {}))
    m = mapper_cls(cfg)
    try:
        path = list(range(steps)) + list(range(steps - 1, -1, -1)) + list(range(steps))
        for i, k in enumerate(path):
            side = (i // steps) * 5 * step_vox
            pos = (np.float32(k * step_vox * voxel), np.float32(side * voxel), np.float32(0.0))
            q = scenes.yaw_quat(0.03 * i)
            lab = scenes.hash_world_labels(scenes.local_pivot(pos, voxel, size), size, i, seed=3, p_occ=0.02, toggle_frac=0.25).astype(np.int8)
            m.update(pos, q, "labels", lab)
            assert m.debug_nbr_check() == 0, (i, k)
        return m.stats()
    finally:
        m.close()


@pytest.mark.parametrize("retain,step_vox", [(0, 5), (1, 9), (1, 24), (2, 13)])
def test_rows_agree_with_the_hash_on_a_drive_that_turns_round_emulation(retain, step_vox):
    st = _drive(EmuMapper, (24, 24, 16), retain, 6, step_vox)
    assert st["blocks_total"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("retain,step_vox,size", [(0, 5, (40, 40, 24)), (1, 9, (40, 40, 24)), (1, 40, (32, 32, 32)), (2, 13, (64, 48, 16))])
def test_rows_agree_with_the_hash_on_a_drive_that_turns_round_hip(retain, step_vox, size):
    from hooks_py import HooksMapper
    st = _drive(HooksMapper, size, retain, 7, step_vox)
    assert st["blocks_total"] > 0
