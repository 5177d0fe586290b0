#!/bin/bash
# price list of one filler instruction per MFMA gap: asm_probe.py on a variants file, delta vs its first line (the bare MFMA stream)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
timeout 600 python scripts/asm_probe.py "$1" --out "gpurun_out/$2" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
base=None
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    if base is None: base=d['ms_median']
    every=d['over'].get('filler_every',1)
    dt_cyc=(d['ms_median']-base)*1e-3/(8*256)*2.33e9
    print(d['variant'], d['ms_median'], d['tflops'], 'extra cycles per gap: %.1f' % (dt_cyc/(128//every)))
"
