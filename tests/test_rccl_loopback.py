"""RCCL executed on ONE GPU (`RcclTransport` had only ever been compiled).

tests/gpu_helpers/rccl_loopback.cpp drives gie_host::RcclTransport (gie-mapping_amd/host/gie_tiled.hpp) with a one-rank communicator:
a mapper's +x face layer is sent to "the neighbour across -x", which is the mapper itself, grouped ncclSend / ncclRecv on the mapper's
own stream between the export and the import kernels, three map updates, the received layers compared byte for byte.  The torch side
(gie/tiling.py, what bench.py uses) gets the same treatment as far as torch allows: a one-rank "nccl" group, an all-reduce and the
stream-ordered face exchange of a mapper whose only neighbour is itself are not expressible there (torch refuses a send to the own
rank), so it is the all-reduce on the mapper's stream that is checked."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_transport_single_rank_loopback():
    import __graft_entry__ as ge
    exe = ge.build_rccl_loopback()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe, str(29500 + os.getpid() % 2000)], capture_output=True, text=True, timeout=600, env=env)
    out = (r.stdout + r.stderr).strip()
    if r.returncode == 77:
        pytest.skip(out.splitlines()[-1] if out else "RCCL refused a one-rank communicator")
    assert r.returncode == 0 and "rccl loopback OK" in out, out[-2000:]


@pytest.mark.gpu
def test_torch_nccl_single_rank_allreduce_on_the_mappers_stream():
    """torch.distributed "nccl" (= RCCL) with world_size 1: group creation + an all-reduce issued on a mapper's own HIP stream
    (torch.cuda.ExternalStream), the way tiling.exchange_until_stable_device reduces its seed counts."""
    import datetime
    import torch
    import torch.distributed as dist
    import gie
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 2000)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    try:
        g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
        m = gie.Mapper(gie.make_config(0.05, (32, 32, 16), cutoff_dist=1.0))
        try:
            dev = torch.device("cuda", 0)
            s = torch.cuda.ExternalStream(m.stream_handle(), device=dev)
            t = torch.full((4,), 7, dtype=torch.int32, device=dev)
            with torch.cuda.stream(s):
                dist.all_reduce(t, group=g)
            s.synchronize()
            assert t.tolist() == [7, 7, 7, 7]
        finally:
            m.close()
    finally:
        if created:
            dist.destroy_process_group()
