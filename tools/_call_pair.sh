mkdir -p gpurun_out/r06_pair
python -m pytest tests/test_gpu_unet.py tests/test_gpu_benched_batches.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 2" "0 2"; do set -- $cfg
  RUN_LABEL="ffhq B16 pair=$1 emit_skip=$2" DPIR_CONV5_PAIR=$1 DPIR_EMIT_SKIP=$2 python tools/forward_time.py ffhq 16 2>&1 | grep fwd
done; done | tee gpurun_out/r06_pair/ffhq.log
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 2" "0 2"; do set -- $cfg
  RUN_LABEL="imagenet256 B32 pair=$1 emit_skip=$2" DPIR_CONV5_PAIR=$1 DPIR_EMIT_SKIP=$2 python tools/forward_time.py imagenet256 32 2>&1 | grep fwd
done; done | tee gpurun_out/r06_pair/in256.log
for cfg in "1 1" "0 1" "1 2"; do set -- $cfg
  RUN_LABEL="imagenet512 B8 pair=$1 emit_skip=$2" DPIR_CONV5_PAIR=$1 DPIR_EMIT_SKIP=$2 python tools/forward_time.py imagenet512 8 512 2>&1 | grep fwd
done | tee gpurun_out/r06_pair/in512.log
