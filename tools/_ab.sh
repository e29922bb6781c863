cd $GRAFT_REPO_ROOT
for v in new pl2 pl3 pl2g1 new pl2; do
  echo "=== $v"
  unset SEEVCN_LIB SEEVCN_CONV_G1_TILES
  if [ $v = pl2 ]; then export SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_pl2.so; fi
  if [ $v = pl3 ]; then export SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_pl3.so; fi
  if [ $v = pl2g1 ]; then export SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_pl2.so SEEVCN_CONV_G1_TILES=5000; fi
  MODE=fwd timeout 300 python3 tools/spconv_micro.py 2>&1 | grep -v "amdgpu.ids\|^voxelize" | cut -c1-150
done
unset SEEVCN_LIB SEEVCN_CONV_G1_TILES
for v in new pl2 new pl2; do
  if [ $v = pl2 ]; then export SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_pl2.so; else unset SEEVCN_LIB; fi
  echo "=== bench $v"
  timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-modes 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('roofline_wgrad',{}).get('frac'))"
done
