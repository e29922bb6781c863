cd $GRAFT_REPO_ROOT
TOP=40 TOPC=45 timeout 600 python3 tools/host_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-170 | head -130
for v in "" "--prefetch-thread" "" "--prefetch-thread"; do
  echo "=== bench $v"
  timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-modes $v 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
done
timeout 300 python3 tools/step_hosttime.py 2>&1 | grep -v amdgpu.ids | head -30
