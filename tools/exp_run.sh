cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_hip_articulated.py tests/test_hip_smooth.py tests/test_hip_training.py tests/test_hip_fuzz.py -q -m gpu 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | head -20
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-train-leg --no-extra-legs > /tmp/log 2>&1
cd $GRAFT_REPO_ROOT && python tools/roofline_table.py /tmp/pp/*results.db | grep -v mlp
