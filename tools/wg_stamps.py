"""Measurement aid (round 6): wall-clock stamps at the start and the end of every WAVE of one kernel, written into the middle z-plane of
the edt plane (nobody writes there on the headline workload) and read back with GIE_EDT_RAW=1 -- how many waves are resident over the
launch, how long one lives, how fast they are started.  This is what showed that the volume-order Mark sweep lasted as long as its 16 K
workgroups took to START (DESIGN.md 4 "Mark + commit (round 6)").

A kernel at its register limit is disturbed by the stamps' few scalar registers (k_markc in its walking form: 203 instead of ~190 spilled
scalar registers, three times its time) -- read the RESIDENT counts and the shape of the start curve there, not the absolute times; k_fuse_rows
is within 7 % of its undisturbed time.  profiles/r06_mark_workgroup_stamps.txt, r06_fuse_wave_stamps.txt.

    python tools/wg_stamps.py build markc|fuse      (here: patches a scratch COPY of csrc/ under /tmp, never the product sources)
    GIE_EDT_RAW=1 python tools/wg_stamps.py run markc|fuse [frames]     (on the GPU box)
"""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {   # kernel -> (the line its body starts with, waves per workgroup)
    "markc": ("    const int n = c.cnt[GIE_CNT_TL_KNOWN];\n    const int lane = threadIdx.x & 63;\n    const bool lazy = c.coc_defer && c.lazy_ok;\n", 4),
    "fuse": ("    {\n        const int n = c.cnt[GIE_CNT_TL_FUSE];\n", 4),
}
STAMP = """    /* (the stamp object holds a wave-uniform index and a time — scalar registers — not the context and nothing per lane: a reference to
     * `c` puts the kernel's argument block into memory, a per-lane pointer costs the two vector registers a kernel at its limit spills) */
    struct gie_stamp_ { float *base; unsigned w; unsigned long long t0; __device__ ~gie_stamp_() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0 && base) { base[2 * (size_t)w] = (float)(t0 & 0xffffff); base[2 * (size_t)w + 1] = (float)(wall_clock64() & 0xffffff); } } };
    const unsigned gie_stamp_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    gie_stamp_ gie_stamp = { (2 * (size_t)gie_stamp_w + 1 < (size_t)c.X * c.Y) ? c.edt + (size_t)(c.Z / 2) * c.X * c.Y : nullptr, gie_stamp_w, (unsigned long long)wall_clock64() };
"""
which = sys.argv[2] if len(sys.argv) > 2 else "markc"
LIB = os.path.join(ROOT, "tools", "ablate", "libgie_stamps_%s.so" % which)
if sys.argv[1] == "build":
    tmp = "/tmp/gie_stamps_src"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(os.path.join(tmp, "gie-mapping_amd")); os.makedirs(os.path.join(tmp, "tools"))
    shutil.copytree(os.path.join(ROOT, "gie-mapping_amd", "csrc"), os.path.join(tmp, "gie-mapping_amd", "csrc"), ignore=shutil.ignore_patterns("*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    shutil.copytree(os.path.join(ROOT, "tools", "measure"), os.path.join(tmp, "tools", "measure"))
    p = os.path.join(tmp, "gie-mapping_amd", "csrc", "gie_kernels.hip.h")
    s = open(p).read()
    anchor = KERNELS[which][0]
    assert s.count(anchor) == 1, "the kernel's first lines have changed: adjust KERNELS"
    open(p, "w").write(s.replace(anchor, STAMP + anchor))
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-no-stack-slot-sharing", "-shared", "-fPIC",
                           "-DGIE_TEST_HOOKS", os.path.join(tmp, "gie-mapping_amd", "csrc", "gie_hip.hip"), "-o", LIB])
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import mapper, scenes
mapper.load_library(LIB)
dev = torch.device("cuda", 0)
P = bench.PRESETS["c5"]; size = tuple(P["size"])
cfg = gie.make_config(P["voxel"], size, cutoff_dist=P["cutoff"], fast_mode=P["fast"])
cfg.wave_workgroups = 160
m = gie.Mapper(cfg)
feed = bench.make_feed("c5", torch, scenes, dev, P["voxel"], size, (0, 0, 0), 64)
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 6
feed.prepare(0, NF)
for i in range(NF):
    feed.step_input(m, i); m.step(); m.sync()
    if i < NF - 2:
        continue
    pl = m.read_local(vtype=False, dist_sq=False, coc=False)["edt"][size[2] // 2].reshape(-1).astype(np.float64)
    a = pl[: (pl.size // 2) * 2].reshape(-1, 2)
    t0, t1 = a[:, 0], a[:, 1]
    ok = (t0 > 0) & (t1 > 0)                        # (cells no wave has stamped hold distances or zero)
    t0, t1 = t0[ok], t1[ok]
    base = np.percentile(t0, 1)
    us = lambda t: ((t - base) % 16777216.0) / 100.0
    s, e = us(t0), us(t1)
    ok = (e >= s) & (e < 5000) & (s > -50)
    s, e = s[ok], e[ok]
    d = e - s
    print("update %d, %s: %d waves stamped; launch %.1f us from the first start to the last end; a wave lives %.2f us (median %.2f, p95 %.2f, max %.2f)"
          % (i, which, s.size, e.max(), d.mean(), np.median(d), np.percentile(d, 95), d.max()))
    step = max(5, int(e.max() / 16))
    for T in range(0, int(e.max()) + 1, step):
        print("   t = %4d us: %6d waves resident, %6d started so far" % (T, int(((s <= T) & (e > T)).sum()), int((s <= T).sum())))
