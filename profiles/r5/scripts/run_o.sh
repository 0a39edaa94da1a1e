#!/bin/bash
# round 5, call O: the library without the ping-pong tile's accumulator preload (0 B scratch again) and with the weight pool's uniform
# values back in SGPRs; x3r hi/lo pairs on the ping-pong tile; full GPU suite; every configuration's line; A/B against the spilling build
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('eval_samples_per_s'), {k: (v['avg_us'], v['alg_tflops']) for k, v in r.items()}, d['roofline']['frac'])"; }
{
echo "== harness: hi/lo operand pairs, ring and ping-pong tiles"
cd tools/kbench
X3=1 ./bf16_gemm_test 300 200 88 3 nt 8 2 5
X3=1 ./bf16_gemm_test 7372 1024 1843 1 nn 1 1 20
X3=1 ./bf16_gemm_test 7372 2048 1843 1 nn 4 1 20
X3=1 ./bf16_gemm_test 1843 1024 1843 4 nn 1 4 20
./bf16_gemm_test 7372 2176 1843 1 nn 4 1 20
./bf16_gemm_test 7372 1843 2176 12 nt 4 1 5
./bf16_gemm_test 32768 4224 8192 1 nn 4 1 5
cd $GRAFT_REPO_ROOT
echo "== full GPU suite"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "suite wall $(( $(date +%s) - t0 )) s"
echo "== lines"
python bench.py --config pemsbay --no-secondary --no-cpu-baseline --no-syn 2>/dev/null | q "pemsbay"
for rep in 1 2; do
python bench.py --config expytky --no-secondary --no-cpu-baseline --no-syn 2>/dev/null | q "expytky bf16 new"
MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_prev.so python bench.py --config expytky --no-secondary --no-cpu-baseline --no-syn 2>/dev/null | q "expytky bf16 spilling-pp"
done
python bench.py --config expytky --precision bf16x3 --no-secondary --no-cpu-baseline --no-syn 2>/dev/null | q "expytky bf16x3"
python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes --save-tiles 2>/dev/null | q "syn8192 bf16"
cp profiles/tiles/syn8192_B32_bf16.json $out/syn8192_B32_bf16.json
python bench.py --config syn8192 --precision bf16x3 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes 2>/dev/null | q "syn8192 bf16x3"
echo "== the driver's line"
t0=$(date +%s)
python bench.py 2>/dev/null > $out/r5o_default.json
echo "default line wall $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5o_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d.get(k)
    print(k, s if not isinstance(s, dict) else {a: s.get(a) for a in ("value", "ms_per_step", "dtype", "parity_tolerance", "roofline", "x3r_session")})
print("cpu", d.get("cpu_baseline"))
PY
} > $out/r5o.log 2>&1
tail -70 $out/r5o.log
