#!/bin/bash
# round 6 (y): assembly-aware pixel cut of the convolutions + 64-row routing rule: parity, fuzz, geometry A/B
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/ -x -q -m gpu -k "walk or conv" > $O/y_pytest_conv.log 2>&1; echo "pytest rc=$?" >> $O/y_pytest_conv.log
tail -5 $O/y_pytest_conv.log
timeout 900 python scripts/fuzz_conv.py 200 31 > $O/y_fuzz_conv.log 2>&1; tail -2 $O/y_fuzz_conv.log
timeout 600 python scripts/conv_geometry_ab.py 32 > $O/y_conv_geometry_ab.jsonl 2> $O/y_geo.err
timeout 600 python scripts/conv_geometry_ab.py 32 m64 > $O/y_conv_m64_ab.jsonl 2>> $O/y_geo.err
python - <<PY
import json
for f in ("$O/y_conv_geometry_ab.jsonl", "$O/y_conv_m64_ab.jsonl"):
    for l in open(f):
        d = json.loads(l)
        print(d["ishape"], d["kshape"], d["stride"], d["mode"][:5], "asm", d["asm"]["tflops"], d["asm"]["frac"], "k", d["asm"]["kernel"], "cut", d["asm"]["cut"], "tail", d["asm"]["tail"], "| compiler", d["compiler"]["tflops"], "cut", d["compiler"]["cut"], d["asm_gain_pct"], d["bit_identical"])
PY
timeout 300 python scripts/conv_c4_run.py 20 > $O/y_c4.log 2>&1; cat $O/y_c4.log
