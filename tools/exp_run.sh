cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_training.py tests/test_hip_training_art.py tests/test_hip_smooth.py tests/test_hip_fuzz.py tests/test_hip_bf16x3.py -q -m gpu 2>&1 | tail -2
for i in 1 2; do
python tools/train_bench.py --rays 4096 --steps 10 --articulated | tail -1 | cut -c1-120
python tools/train_bench.py --rays 4096 --steps 10 | tail -1 | cut -c1-120
done
