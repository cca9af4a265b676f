#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_unet.py -q -x -k "f16x3 or imagenet256_topology_64" > $out/pytest_unet.log 2>&1; tail -3 $out/pytest_unet.log
timeout 300 python tools/conv_compare.py 16 > $out/conv_compare.log 2>&1; cat $out/conv_compare.log
DIFFPIR_CONV=6 timeout 200 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline --nfe 20 > $out/bench6.json 2>$out/bench6.err; tail -c 900 $out/bench6.json
DIFFPIR_CONV=4 timeout 200 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline --nfe 20 > $out/bench4.json 2>$out/bench4.err; tail -c 900 $out/bench4.json
