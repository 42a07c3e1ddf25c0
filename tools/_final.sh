python -c "import torch" 2>/dev/null
f=0; ok=0; other=0
for i in $(seq 1 14); do
  GIE_BENCH_BACKEND=gloo GIE_BENCH_SHARE_GPU=1 timeout 60 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 2 > gpurun_out/f.log 2>&1
  if grep -q "Memory access fault" gpurun_out/f.log; then f=$((f+1)); elif grep -q '"metric"' gpurun_out/f.log; then ok=$((ok+1)); else other=$((other+1)); fi
done
echo "2-rank shared-device runs: ok $ok faults $f other $other"
(time (GIE_BENCH_BACKEND=gloo GIE_BENCH_SHARE_GPU=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --steps 4 --warmup 2)) > gpurun_out/bench8_final.log 2>&1
grep -c '"metric"' gpurun_out/bench8_final.log
bash tools/profile_round.sh r02_c5 "--steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-timed-s 0" > /dev/null 2>&1
bash tools/profile_round.sh r02_dense "--workload vlp16_projective --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-timed-s 0" > /dev/null 2>&1
bash tools/profile_round.sh r02_raycast "--workload vlp16 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-timed-s 0" > /dev/null 2>&1
python bench.py --rms > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
bash tools/pmc_run.sh > gpurun_out/sq_c5.txt 2>&1 || true
ls gpurun_out/prof_r02_c5 | head -3
