cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_bf16x3.py -q -m gpu -s -k adversarial 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|max error|^E  " | head -30
