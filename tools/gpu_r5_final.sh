#!/bin/bash
# Round-5 evidence at one commit: rocprofv3 passes (kernel trace + PMC, all four cases), the bench command under rocprofv3, the bench line.
bash tools/gpu_prof_round.sh r05
bash tools/gpu_prof_bench.sh r05
out=$PWD/gpurun_out/r05
timeout 900 python bench.py > $out/bench_1gpu.json 2> $out/bench_1gpu.err; tail -c 600 $out/bench_1gpu.json; tail -2 $out/bench_1gpu.err
