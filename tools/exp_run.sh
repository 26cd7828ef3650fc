cd $GRAFT_REPO_ROOT
python tests/diag/diag_masks.py | tail -2
python -m pytest tests -q -m gpu 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|AssertionError: \{|^E   .*final" | head -40
