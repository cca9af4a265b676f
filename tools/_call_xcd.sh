export TMPDIR=/tmp
mkdir -p gpurun_out/r06_xcd
python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
python tools/prox_bench.py 2>&1 | grep prox | tee gpurun_out/r06_xcd/prox_bench.log
python - <<'P' 2>&1 | grep -v amdgpu | tee gpurun_out/r06_xcd/graph_timed.log
import ctypes as C, numpy as np, sys
sys.path.insert(0, ".")
import diffpir_amd
from diffpir_amd import synth, utils_sisr as sr
e = diffpir_amd.Engine(0)
for B, H, sf in ((16, 256, 1), (64, 256, 1), (32, 256, 4), (8, 512, 4)):
    rng = np.random.default_rng(0)
    y = e.to_device(rng.random((B, 3, H // sf, H // sf)).astype(np.float32))
    kk = rng.random((B, 1, 25, 25)).astype(np.float32); kk /= kk.sum(axis=(2, 3), keepdims=True)
    pre = sr.pre_calculate(y, e.to_device(kk), sf)
    x0 = e.to_device(rng.random((B, 3, H, H)).astype(np.float32) * 2 - 1)
    us = C.c_float(); best = 1e9
    for _ in range(4):
        e._check(e.lib.dpir_prox_fft_apply_timed(e.h, pre[0].spectra.handle, x0.ptr, 0.05, 1.0, 60, 1, C.byref(us))); best = min(best, us.value)
    print(f"graph-timed B={B} {H} sf={sf}: {best:.2f} us")
P
for c in "16 256 1 p256" "8 512 4 p512"; do set -- $c
  for pmc in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/px_$4_$pmc; rm -rf $d
    (cd /tmp && PROF_UNET=0 PROF_B=$1 PROF_SIZE=$2 PROF_SF=$3 rocprofv3 --kernel-trace --pmc $pmc -d $d -o px -- python $GRAFT_REPO_ROOT/tools/prof_forward.py > /dev/null 2>&1)
    python tools/rocpd_summary.py $(find $d -name "*.db" | head -1) --top 12 > gpurun_out/r06_xcd/$4_pmc_$pmc.txt 2>&1
    grep -A1 "cfft4_cols_kernel<[23]" gpurun_out/r06_xcd/$4_pmc_$pmc.txt | head -4
  done
done
