TESTS="tests/test_bn_train_gpu.py tests/test_train_gpu.py tests/test_sparse_grad_gpu.py tests/test_pipeline_gpu.py tests/test_sparse_sites_gpu.py" TAIL=5 NOBENCH=1 TEST_TIMEOUT=700 bash scripts/gpu_r3_check.sh g
cd $GRAFT_REPO_ROOT
for f in "" "--no-fork"; do for st in 1 2; do
v=$(timeout -k 5 200 python -u bench.py --steps 300 --warmup 30 --cpu-frames 0 --no-roofline --no-host-io --no-sequential --streams $st $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))")
echo "streams $st fork[$f] -> $v"
done; done
timeout 300 python scripts/train_step_bench.py --steps 10 --graph 2>&1 | tail -1 | cut -c1-700
