b() { python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.2 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -2; }
echo "== fresh box"; b; b
echo "== after the 512^3 / tiling tests"; python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "512 or c2_256 or c4_slab" 2>&1 | tail -1; b; b
echo "== after the whole gpu suite"; python -m pytest tests -m gpu -q -x 2>&1 | tail -1; b; b
