#!/bin/bash
# the small-channel direct convolution kernels on one MI355X: parity test, per-call times of every form, kernel durations
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
T=${1:-v1}
O=$R/gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "conv" --timeout 500 2>&1 | tail -3
timeout 300 python scripts/probes/conv_direct_probe.py > $O/conv_direct_probe_$T.jsonl 2> $O/conv_direct_probe_$T.err; cat $O/conv_direct_probe_$T.jsonl; tail -2 $O/conv_direct_probe_$T.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cdp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cdp -- python $R/scripts/probes/conv_direct_probe.py > /tmp/cdp.log 2>&1
cd "$R"
f=$(find /tmp/cdp -name "*_kernel_stats.csv" | head -1); echo "stats: $f"; tail -3 /tmp/cdp.log; [ -n "$f" ] && cp "$f" $O/conv_direct_kernel_stats_$T.csv && grep -i "conv_direct" $O/conv_direct_kernel_stats_$T.csv | cut -c1-300
timeout 300 python scripts/fuzz_conv.py 300 21 small 2>&1 | tail -2
