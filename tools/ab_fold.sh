cd $GRAFT_REPO_ROOT
python -m pytest tests/test_spconv.py tests/test_configs.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r05_fold_tests.txt
tail -3 gpurun_out/r05_fold_tests.txt
for i in 1 2; do
for f in 0 1; do
echo "BN_FOLD=$f" >> gpurun_out/r05_fold_ab.txt
SEEVCN_BN_FOLD=$f python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-side-modes 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))" >> gpurun_out/r05_fold_ab.txt
done; done
cat gpurun_out/r05_fold_ab.txt
