export TMPDIR=/tmp
python -m pytest tests/test_gpu_unet.py tests/test_gpu_benched_batches.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && rm -rf /tmp/pf && PROF_MODEL=ffhq PROF_B=16 rocprofv3 --kernel-trace -d /tmp/pf -o pf -- python $GRAFT_REPO_ROOT/tools/prof_forward.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $(find /tmp/pf -name "*.db" | head -1) --top 40 | grep -E "summary|gn_prm|conv6_reduce|gn_act_small" | cut -c1-170
