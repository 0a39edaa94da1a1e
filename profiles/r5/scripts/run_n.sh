#!/bin/bash
# round 5, call N: bf16x3 sessions of the large graphs on the bf16-resident data flow with hi/lo operand pairs (x3r)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('eval_samples_per_s'), {k: (v['avg_us'], v['alg_tflops']) for k, v in r.items()}, d['roofline']['frac'])"; }
{
echo "== harness: hi/lo operand pairs"
cd tools/kbench
X3=1 ./bf16_gemm_test 300 200 88 3 nt 8 2 5
X3=1 ./bf16_gemm_test 333 136 77 2 nn 9 1 5
X3=1 ./bf16_gemm_test 1000 520 200 1 nn 1 3 5
X3=1 ./bf16_gemm_test 7372 1024 1843 1 nn 1 1 20
X3=1 ./bf16_gemm_test 7372 2048 1843 1 nn 4 1 20
X3=1 ./bf16_gemm_test 1843 1024 1843 4 nn 1 4 20
X3=1 ./bf16_gemm_test 1843 1843 1024 12 nt 4 1 5
./bf16_gemm_test 7372 1024 1843 1 nn 1 1 20
cd $GRAFT_REPO_ROOT
echo "== parity"
timeout 1500 python -m pytest tests -m gpu -x -q -k "(kernel_variants and 400) or (baseline_config_train and (expytky or syn8192)) or (baseline_config_full_size and (expytky or syn8192)) or (full_batch_backward and expytky) or bf16_mode_train" 2>&1 | tail -12
echo "== bench bf16x3 at the large graphs"
python bench.py --config expytky --precision bf16x3 --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky bf16x3"
python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky bf16"
python bench.py --config syn8192 --precision bf16x3 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes 2>/dev/null | q "syn8192 bf16x3"
} > $out/r5n.log 2>&1
tail -60 $out/r5n.log
