"""Register / scratch / LDS use of every kernel of libgie_hip.so as the compiler reports it (-Rpass-analysis=kernel-resource-usage):
python tools/kernel_resources.py [extra hipcc flags].  A kernel with scratch has spilled registers — every reload waits for ALL
outstanding memory operations of the wave (vmcnt counts in order), which undoes any prefetch the kernel was written around."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-no-stack-slot-sharing", "-shared", "-fPIC",
                      "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "gie-mapping_amd", "csrc", "gie_hip.hip"), "-o", "/tmp/_kr.so"] + sys.argv[1:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = v; rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
print("%-70s %6s %8s %6s %6s %9s" % ("kernel", "VGPRs", "scratch", "vspill", "occ", "LDS"))
for name, r in rows.items():
    if name.startswith("_ZN7rocprim"):
        continue
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:70]
    print("%-70s %6s %8s %6s %6s %9s" % (short, r.get("VGPRs"), r.get("ScratchSize"), r.get("VGPRs Spill"), r.get("Occupancy"), r.get("LDS Size")))
