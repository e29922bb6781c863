cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_spconv.py tests/test_abi.py tests/test_head.py -x -q -m gpu 2>&1 | tail -8
for v in auto nchw auto nchw; do
  echo "=== second SEEVCN_BEV_FORMAT=$v"
  SEEVCN_BEV_FORMAT=$v timeout 600 python3 bench.py --config second --steps 10 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
for v in auto nchw; do
  echo "=== centerpoint SEEVCN_BEV_FORMAT=$v"
  SEEVCN_BEV_FORMAT=$v timeout 600 python3 bench.py --config centerpoint --steps 10 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
