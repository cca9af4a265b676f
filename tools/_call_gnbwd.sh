mkdir -p gpurun_out/r06_gnbwd
python -m pytest tests/test_gpu_dps.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  RUN_LABEL="new gn_bwd" python tools/dps_time.py 8 6 2>&1 | grep "ms per NFE"
  RUN_LABEL="old gn_bwd" DIFFPIR_LIB=$PWD/diffpir_amd/csrc/libdiffpir_hip_old.so python tools/dps_time.py 8 6 2>&1 | grep "ms per NFE"
done | tee gpurun_out/r06_gnbwd/dps_time_ab.log
RUN_LABEL="new gn_bwd B16" python tools/dps_time.py 16 4 2>&1 | grep "ms per NFE" | tee -a gpurun_out/r06_gnbwd/dps_time_ab.log
RUN_LABEL="old gn_bwd B16" DIFFPIR_LIB=$PWD/diffpir_amd/csrc/libdiffpir_hip_old.so python tools/dps_time.py 16 4 2>&1 | grep "ms per NFE" | tee -a gpurun_out/r06_gnbwd/dps_time_ab.log
