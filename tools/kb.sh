#!/bin/bash
# compact kernel timings at the bench shape (run on the GPU box): tools/kb.sh [kernels]
python tools/bench_kernels.py --dtype bf16 --batch ${KB_BATCH:-768} --only ${1:-scan_fwd,scan_bwd,scan_idx} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()[:200]); continue
    print(d.get('kernel'), d.get('S'), d.get('dtype'), round(d.get('us', 0), 1))
"
