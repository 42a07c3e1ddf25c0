"""Where does pass Z spend its time?  Builds libgie_hip variants with GIE_EDTZ_ABLATE=1..3
(1: no argmin, 2: compaction only, 3: no column work) into tools/ablate/ (CPU side, before gpurun) and,
on the GPU, times the batch EDT of each on the bench workload.  Measurement aid only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "ablate")
CSRC = os.path.join(ROOT, "gie-mapping_amd", "csrc")


def build():
    os.makedirs(OUT, exist_ok=True)
    for v in (1, 2, 3):
        so = os.path.join(OUT, "libgie_hip_ab%d.so" % v)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                               "-DGIE_EDTZ_ABLATE=%d" % v, os.path.join(CSRC, "gie_hip.hip"), "-o", so])


def run_one(lib, sensor):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
    import math
    import torch
    import bench
    import gie
    from gie import mapper, scenes
    if lib:
        mapper.load_library(lib)
    rings, az, phi_min, phi_inc, bins = bench.SENSORS[sensor]
    frames = bench.make_frames(scenes, 0.05, 8, 5, sensor)
    dev = torch.device("cuda", 0)
    d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
    m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
    for i, (pos, q, pts, _) in enumerate(frames):
        if i == 3:
            m.sync(); m.profile_enable(True)
        m.set_pose(pos, q)
        if bins is None:
            m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0])
        else:
            m.ogm_multiscan_dev(d_pts[i].data_ptr(), bins, rings, 2.0 * math.pi / bins, -math.pi, math.radians(phi_inc), math.radians(phi_min))
        m.step()
    m.sync()
    prof = m.profile_read()
    print(json.dumps({"lib": os.path.basename(lib) if lib else "product", "sensor": sensor,
                      "edt_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items() if k.startswith("edt") and v[1]}}))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "one":
        run_one(sys.argv[2] if sys.argv[2] != "-" else None, sys.argv[3])
    else:
        for sensor in ("vlp16", "vlp16_projective"):
            for lib in ["-"] + [os.path.join(OUT, "libgie_hip_ab%d.so" % v) for v in (1, 2, 3)]:
                subprocess.call([sys.executable, os.path.abspath(__file__), "one", lib, sensor])
