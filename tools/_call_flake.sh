mkdir -p gpurun_out/r06_flake
DPIR_DPS_POSTMORTEM=1 python -m pytest tests/test_gpu_dps.py -m gpu -q -s -k "full_size_ffhq_vs_oracle" > gpurun_out/r06_flake/dps_postmortem_forced.log 2>&1
grep -E "post-mortem|passed|failed" gpurun_out/r06_flake/dps_postmortem_forced.log
for i in 1 2 3; do
  python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_flake/suite_$i.log 2>&1
  tail -1 gpurun_out/r06_flake/suite_$i.log
  grep -E "^FAILED|EXCURSION" gpurun_out/r06_flake/suite_$i.log
done
