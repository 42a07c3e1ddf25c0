"""Measurement aid: where does the segmented ray-casting kernel spend its time?
GIE_RAY_ABLATE=1: no decrements (replay + phase 1 only), =2: no type reads (replay + phase 2)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "ablate")
if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for v in (1, 2):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                               "-DGIE_RAY_ABLATE=%d" % v, os.path.join(ROOT, "gie-mapping_amd", "csrc", "gie_hip.hip"), "-o", os.path.join(OUT, "libgie_hip_ray%d.so" % v)])
    sys.exit(0)
if sys.argv[1] == "one":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
    import torch, bench, gie
    from gie import mapper, scenes
    if sys.argv[2] != "-":
        mapper.load_library(sys.argv[2])
    frames = bench.make_frames(scenes, 0.05, 8, 5, "vlp16")
    dev = torch.device("cuda", 0)
    d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
    m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
    for i, (pos, q, pts, _) in enumerate(frames):
        if i == 3:
            m.sync(); m.profile_enable(True)
        m.set_pose(pos, q); m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0]); m.step()
    m.sync(); prof = m.profile_read()
    print(json.dumps({"lib": os.path.basename(sys.argv[2]), "ray_free_ms": round(prof["ray_free"][0] / prof["ray_free"][1], 4)}))
else:
    for lib in ["-"] + [os.path.join(OUT, "libgie_hip_ray%d.so" % v) for v in (1, 2)]:
        subprocess.call([sys.executable, os.path.abspath(__file__), "one", lib])
