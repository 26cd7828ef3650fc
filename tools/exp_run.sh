cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_training.py -q -m gpu -k two_call 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | head
