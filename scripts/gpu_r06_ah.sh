#!/bin/bash
# round 6 (ah): staging packed into consecutive gaps (Cfg.w_step = 1) on the 256-row tiles: parity of the kernels that changed, then old / new builds
# alternated on one box at the headline shape (both modes), C3 (full form) and C4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
O=gpurun_out/r06
OLD=scripts/probes/ab_old/liblaser_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scheduler.py -x -q -m gpu > $O/ah_pytest.log 2>&1; echo "pytest rc=$?" >> $O/ah_pytest.log
tail -3 $O/ah_pytest.log
: > $O/ah_wstep_ab.log
for r in 1 2 3 4; do
  for mode in 0 1; do
    echo "== old $r mode $mode" >> $O/ah_wstep_ab.log; timeout 200 python scripts/with_lib.py $OLD scripts/shape_run.py 8192 8192 8192 $mode -1 0 40 >> $O/ah_wstep_ab.log 2>&1
    echo "== new $r mode $mode" >> $O/ah_wstep_ab.log; timeout 200 python scripts/shape_run.py 8192 8192 8192 $mode -1 0 40 >> $O/ah_wstep_ab.log 2>&1
  done
done
for r in 1 2 3; do
  echo "== old $r c3" >> $O/ah_wstep_ab.log; timeout 200 python scripts/with_lib.py $OLD scripts/c3_run.py 40 full 2>&1 | tail -2 >> $O/ah_wstep_ab.log
  echo "== new $r c3" >> $O/ah_wstep_ab.log; timeout 200 python scripts/c3_run.py 40 full 2>&1 | tail -2 >> $O/ah_wstep_ab.log
  echo "== old $r c4" >> $O/ah_wstep_ab.log; timeout 200 python scripts/with_lib.py $OLD scripts/conv_c4_run.py 20 >> $O/ah_wstep_ab.log 2>&1
  echo "== new $r c4" >> $O/ah_wstep_ab.log; timeout 200 python scripts/conv_c4_run.py 20 >> $O/ah_wstep_ab.log 2>&1
done
grep -v "^Hostname\|^Librccl" $O/ah_wstep_ab.log | cut -c1-200
