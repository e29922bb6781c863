#!/bin/bash
# usage (on the GPU box, from the repo root): tools/evidence.sh <tag>      e.g. r02_a
# One call that leaves everything profiles/ holds for a round under gpurun_out/: the default bench line, the rocprofv3 kernel stats of the
# same command (whole process + steady steps), the PMC traffic passes, the side-mode bench lines with their steady kernel tables, and the
# micro-benchmarks (per-layer sparse conv, FPS), the host-time split, the blocking-read traces and the launch sites of the PV-RCNN step.
# FIRST the gate: the driver's own command (`pytest -m gpu -x -q`) at this tree, log kept as gpurun_out/<round>_gputest.log (copy it to profiles/);
# a red gate stops the call -- numbers of a tree whose parity is not green are not evidence.  GATE=0 skips it (A/B reruns on a box that already ran it).
TAG=$1
cd $GRAFT_REPO_ROOT
if [ "${GATE:-1}" != "0" ]; then
  timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/${TAG%%_*}_gputest.log 2>&1
  rc=$?
  tail -3 gpurun_out/${TAG%%_*}_gputest.log
  [ $rc -ne 0 ] && { echo "GATE RED (rc $rc): no evidence taken"; exit $rc; }
  # the suite's worker processes (two-rank tests) may still be giving their GPU queues back: a bench started right behind them shares the GPU with
  # them and reads 3x slow for its whole timed loop (seen twice in round 5: 11.4 and 12.6 ms/step with every kernel at its normal duration)
  sleep 30
fi
timeout 900 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
TOP=8 tools/profile.sh ${TAG}_main 20 2>&1 | head -10
tools/roofline_pass.sh ${TAG} 2>&1 | tail -8
tools/traffic.sh ${TAG} 2>&1 | tail -6
tools/mfma_busy.sh ${TAG} 2>&1 | tail -8
: > gpurun_out/${TAG}_side_modes.jsonl
for c in stageA second pvrcnn centerpoint; do
  timeout 600 python3 bench.py --config $c --steps 10 --warmup 5 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_side_modes.jsonl
  TOP=6 tools/profile.sh ${TAG}_$c 5 --config $c 2>&1 | head -8
done
cut -c1-220 gpurun_out/${TAG}_side_modes.jsonl
timeout 300 python3 tools/spconv_micro.py > gpurun_out/${TAG}_spconv_micro.txt 2>&1
timeout 120 python3 tools/fps_micro.py > gpurun_out/${TAG}_fps_micro.txt 2>&1
tail -3 gpurun_out/${TAG}_spconv_micro.txt
# farthest point sampling: bucket statistics (library variant built with -DFB_STATS: tools/build_variant.sh fbstats "-DFB_STATS" fps_bucket) and the
# dependent-issue rate the rounds are made of
[ -f see-vcn_amd/lib/variants/libseevcn_hip_fbstats.so ] && SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_fbstats.so timeout 200 python3 tools/fps_stats.py > gpurun_out/${TAG}_fps_stats.txt 2>&1
timeout 200 python3 tools/vcn_gemm_trace.py > gpurun_out/${TAG}_vcn_gemm_trace.txt 2>&1
timeout 300 python3 tools/vcn_gemm_micro.py > gpurun_out/${TAG}_vcn_gemm_micro.txt 2>&1
(cd tools/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_clock valu_clock.hip 2>/dev/null && /tmp/valu_clock | head -10) > gpurun_out/${TAG}_valu_clock.txt 2>&1
timeout 300 python3 tools/host_time.py > gpurun_out/${TAG}_host_time.txt 2>&1
for c in main pvrcnn centerpoint second; do timeout 300 python3 tools/sync_trace.py $c > gpurun_out/${TAG}_sync_trace_$c.txt 2>&1; done
tools/timeline.sh ${TAG} > /dev/null 2>&1
timeout 120 python3 tools/wgrad_uniform.py > gpurun_out/${TAG}_wgrad_uniform.txt 2>&1
for d in 1 2 4 5 8 12; do echo "SEEVCN_WGRAD_DEBUG=$d" >> gpurun_out/${TAG}_wgrad_debug.txt; SEEVCN_WGRAD_PLANNED=0 SEEVCN_WGRAD_DEBUG=$d MODE=wgrad LAYER=subm3 timeout 120 python3 tools/spconv_micro.py 2>&1 | grep subm3 >> gpurun_out/${TAG}_wgrad_debug.txt; done
(cd tools/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_rate mfma_rate.hip 2>/dev/null && /tmp/mfma_rate | tail -4) > gpurun_out/${TAG}_mfma_rate.txt 2>&1
tools/wgrad_ab.sh ${TAG} > /dev/null 2>&1
timeout 300 python3 tools/launch_sites.py pvrcnn 40 > gpurun_out/${TAG}_launch_sites_pvrcnn.txt 2>&1
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_mfma gpurun_out/prof_*.log gpurun_out/tl_*.log
