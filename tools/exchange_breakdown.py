import json, math, os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import scenes, tiling
size=(512,512,512); world=2
frames = bench.make_frames(scenes, 0.05, 8, 5, "vlp16")
dev = torch.device("cuda", 0)
d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
grid = tiling.tile_grid(world); whole = tuple(grid[i]*size[i] for i in range(3))
ms=[]
for r in range(world):
    m = gie.Mapper(gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False)); m.set_tile(tiling.tile_offset_voxels(r, world, size), whole); ms.append(m)
T={"export":0,"import":0,"refine":0,"alloc":0}; n=0
for i,(pos,q,pts,_) in enumerate(frames):
    for m in ms:
        m.set_pose(pos,q); m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0]); m.step()
    for m in ms: m.sync()
    if i < 2: continue
    for rnd in range(2):
        t0=time.perf_counter()
        layers={}
        for r,m in enumerate(ms):
            for face,nb in tiling.neighbours(r,world).items():
                t=torch.empty(m.halo_count(face)*20,dtype=torch.uint8,device=dev)
                layers[(nb,face^1)]=t
        t1=time.perf_counter()
        for r,m in enumerate(ms):
            for face,nb in tiling.neighbours(r,world).items():
                m.halo_export_dev(face, layers[(nb,face^1)].data_ptr())
        for m in ms: m.sync()
        t2=time.perf_counter()
        for (r,face),t in layers.items(): ms[r].halo_import_dev(face,t.data_ptr())
        for m in ms: m.sync()
        t3=time.perf_counter()
        s=[m.refine() for m in ms]
        t4=time.perf_counter()
        T["alloc"]+=t1-t0; T["export"]+=t2-t1; T["import"]+=t3-t2; T["refine"]+=t4-t3; n+=1
print({k: round(1e3*v/n/world,4) for k,v in T.items()}, "ms per round per tile (one face each)")
