#!/bin/bash
# round 6 (v): convolution unit walkers (pipelined transitions): parity, fuzz, A/B against the one-tile-per-workgroup launches
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/ -x -q -m gpu -k "conv" > $O/v_pytest_conv.log 2>&1; echo "pytest rc=$?" >> $O/v_pytest_conv.log
tail -3 $O/v_pytest_conv.log
timeout 900 python scripts/fuzz_conv.py 300 11 > $O/v_fuzz_conv.log 2>&1; tail -3 $O/v_fuzz_conv.log
timeout 600 python scripts/conv_walk_ab.py 5 > $O/v_conv_walk_ab.jsonl 2> $O/v_conv_walk_ab.err; cat $O/v_conv_walk_ab.jsonl | cut -c1-600; tail -3 $O/v_conv_walk_ab.err
timeout 300 python scripts/conv_c4_run.py 20 > $O/v_c4.log 2>&1; cat $O/v_c4.log
