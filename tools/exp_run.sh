cd /tmp && export TMPDIR=/tmp
for t in "" _rpw2 _rpw4; do
rm -rf /tmp/pp; AON_HIP_LIB=$GRAFT_REPO_ROOT/articulated-object-nerf_amd/libaon_hip$t.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-train-leg --no-extra-legs > /tmp/log 2>&1
echo "== $t"; python $GRAFT_REPO_ROOT/tools/roofline_table.py /tmp/pp/*results.db | grep composite
done
