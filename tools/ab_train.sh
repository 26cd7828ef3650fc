cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_$1 -o x -- python $GRAFT_REPO_ROOT/tools/train_bench.py --rays 4096 --steps 10 > /tmp/log_$1 2>&1
cd $GRAFT_REPO_ROOT && python tools/kstats.py /tmp/p_$1/x_results.db 3 | tail -3 | cut -c1-140; tail -1 /tmp/log_$1 | cut -c1-120
