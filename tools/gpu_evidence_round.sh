#!/bin/bash
# The round's evidence set at ONE commit, in ONE gpurun call: rocprofv3 passes over the forward workload (kernel trace + one PMC pass per
# counter group, four cases), rocprofv3 over the bench command itself, and the default bench line.
# usage: tools/gpu_evidence_round.sh <tag>    -> gpurun_out/<tag>/...   (copy the .txt / .json into profiles/<tag>/, then
#        `python tools/pmc_traffic.py profiles/<tag>` refreshes profiles/pmc_traffic.json)
tag=${1:-evidence}
bash tools/gpu_prof_round.sh $tag
bash tools/gpu_prof_bench.sh $tag
out=$PWD/gpurun_out/$tag
timeout 900 python bench.py > $out/bench_1gpu.json 2> $out/bench_1gpu.err; tail -c 600 $out/bench_1gpu.json; tail -2 $out/bench_1gpu.err
