cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_training_art.py tests/test_hip_training.py tests/test_hip_fuzz.py -q -m gpu 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | head
python tools/train_bench.py --rays 4096 --steps 10 --articulated | tail -1
python tools/train_bench.py --rays 4096 --steps 10 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_art -o x -- python $GRAFT_REPO_ROOT/tools/train_bench.py --rays 4096 --steps 10 --articulated > /tmp/log_art 2>&1
cd $GRAFT_REPO_ROOT && python tools/kstats.py /tmp/p_art/x_results.db 16 | cut -c1-150
