mkdir -p gpurun_out/r06_flake2
for i in $(seq 1 25); do
  python -m pytest tests/test_gpu_dps.py -m gpu -q -s -p no:cacheprovider -k "full_size_ffhq_vs_oracle" 2>&1 | grep -E "DPS_y0|passed|failed" | cut -c1-260 >> gpurun_out/r06_flake2/dps_only_25.log
done
tail -4 gpurun_out/r06_flake2/dps_only_25.log
for i in 1 2 3; do
  python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_flake2/suite_$i.log 2>&1
  tail -1 gpurun_out/r06_flake2/suite_$i.log
  grep -E "^FAILED|EXCURSION" gpurun_out/r06_flake2/suite_$i.log
done
true
