cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_spconv.py -x -q -m gpu 2>&1 | tail -3
for v in new adj0 pl2 new adj0 pl2; do
  echo "=== $v"
  unset SEEVCN_LIB
  if [ $v = pl2 ]; then export SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_pl2.so; fi
  if [ $v = adj0 ]; then export SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_adj0.so; fi
  MODE=fwd timeout 300 python3 tools/spconv_micro.py 2>&1 | grep -v "amdgpu.ids\|^voxelize" | cut -c1-150
done
unset SEEVCN_LIB
for v in new adj0 new adj0; do
  if [ $v = adj0 ]; then export SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_adj0.so; else unset SEEVCN_LIB; fi
  echo "=== bench $v"
  timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-modes 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('roofline_wgrad',{}).get('frac'))"
done
