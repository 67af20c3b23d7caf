#!/bin/bash
# round 5, GPU call 1: new tests, smoke, the full-size train -> engine-vs-oracle record, bench on random and on trained weights
O=gpurun_out/r5a
mkdir -p $O
export SESSD_BENCH_VERBOSE=1
timeout 900 python -m pytest tests/test_head_loss_gpu.py tests/test_dense_active_gpu.py tests/test_trained_gpu.py -x -q -s -m gpu > $O/tests_new.log 2>&1
echo "tests_new rc=$?" >> $O/status.txt
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "overflow or reference_loss or bookkeeping" > $O/tests_train.log 2>&1
echo "tests_train rc=$?" >> $O/status.txt
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "active or autotune" > $O/tests_pipe.log 2>&1
echo "tests_pipe rc=$?" >> $O/status.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/status.txt
timeout 900 python tests/trained_parity.py --iterations 2000 --scenes 400 --heldout 200 --workers 12 --save $O/trained_student.pt --out $O/trained_parity.json > $O/trained_parity.log 2>&1
echo "trained_parity rc=$?" >> $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "bench rc=$?" >> $O/status.txt
if [ -f $O/trained_student.pt ]; then
  timeout 600 python bench.py --steps 300 --warmup 30 --weights $O/trained_student.pt --no-train-step > $O/bench_trained.json 2> $O/bench_trained.err
  echo "bench_trained rc=$?" >> $O/status.txt
fi
cat $O/status.txt
tail -5 $O/tests_new.log
tail -30 $O/trained_parity.log
