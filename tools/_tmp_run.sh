timeout 300 python -m pytest tests -m gpu -x -q -k "bf16_mode_train or half_batches" 2>&1 | tail -3 > gpurun_out/t3.log
MCRN_TUNE_LOG=1 timeout 200 python bench.py --config expytky --no-secondary --no-cpu-baseline 2>gpurun_out/tune2.log > gpurun_out/b_expy.json
