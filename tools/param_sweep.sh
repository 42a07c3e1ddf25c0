# measurement aid: per-kernel times of a C5 update on a fresh mapper for a few launch parameters (environment), separate allocations
run() { echo "== $1"; env $1 PROBE_CONFIGS=-2 PROBE_REPS=3 python tools/arena_probe.py 2>&1 | grep KERNELS | sed "s/KERNELS //; s/commit=0.0000 //; s/mark=0.0000 //; s/ray_[a-z]*=0.0000 //g; s/wave_[ab]=0.0000 //g"; }
for e in "$@"; do run "$e"; done
