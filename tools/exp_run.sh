cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | head -20
python tools/train_bench.py --rays 4096 --steps 10 --articulated | tail -1
python tools/train_bench.py --rays 4096 --steps 10 | tail -1
