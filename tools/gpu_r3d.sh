#!/bin/bash
# r3 conv5 / conv6 pipeline fixes: correctness subset, then A/B forward timings against the previous commit's build and the __syncthreads
# variant.  The two variant libraries are built by hand next to the product library before the call (they are git-ignored):
#   libdiffpir_hip_base.so        = `git archive <previous commit> diffpir_amd/csrc include | tar -x -C /tmp/base && make -C /tmp/base/diffpir_amd/csrc libdiffpir_hip.so`
#   libdiffpir_hip_syncthreads.so = the current tree built with CXXFLAGS += -DDPIR_C6_LDS_BARRIER=0
tag=${1:-r3n}
out=gpurun_out/$tag
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q > $out/pytest_subset.log 2>&1
tail -5 $out/pytest_subset.log
L=diffpir_amd/csrc
for v in base syncthreads new; do
  lib=$PWD/$L/libdiffpir_hip_$v.so; [ $v = new ] && lib=$PWD/$L/libdiffpir_hip.so
  RUN_LABEL=$v DIFFPIR_LIB=$lib timeout 300 python tools/forward_time.py ffhq 16 256 2>&1 | tail -1 | tee -a $out/forward_ab.log
done
for v in base new; do
  lib=$PWD/$L/libdiffpir_hip_$v.so; [ $v = new ] && lib=$PWD/$L/libdiffpir_hip.so
  RUN_LABEL=${v}_imagenet256_b8 DIFFPIR_LIB=$lib timeout 300 python tools/forward_time.py imagenet256 8 256 2>&1 | tail -1 | tee -a $out/forward_ab.log
done
