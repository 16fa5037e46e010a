mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --tb=short --no-header -p no:cacheprovider 2>&1 | tail -6
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph 2>/dev/null | tail -1
for mode in "--scaling weak" "--scaling strong --shard queries" "--scaling strong --shard entities --exchange counts" "--scaling strong --shard entities --exchange scores --materialize --batch 2048"; do
echo "== 2 ranks (gloo dry run, one GPU): $mode"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo $mode 2>&1 | grep -v "amdgpu.ids\|Setting OMP\|^\*\*\*\*\|W0" | tail -3
done
) > gpurun_out/run9.log 2>&1
cat gpurun_out/run9.log | cut -c1-1500
