#!/bin/bash
# round 6 (w): walker GPU test, geometry A/B (asm with walkers vs compiler-scheduled), few-output-channel sweep
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "walk or conv" > $O/w_pytest_conv.log 2>&1; echo "pytest rc=$?" >> $O/w_pytest_conv.log
tail -5 $O/w_pytest_conv.log
timeout 600 python scripts/conv_geometry_ab.py 32 > $O/w_conv_geometry_ab.jsonl 2> $O/w_geo.err; cut -c1-420 $O/w_conv_geometry_ab.jsonl
timeout 600 python scripts/conv_geometry_ab.py 32 m64 > $O/w_conv_m64_ab.jsonl 2>> $O/w_geo.err; cut -c1-420 $O/w_conv_m64_ab.jsonl
tail -3 $O/w_geo.err
