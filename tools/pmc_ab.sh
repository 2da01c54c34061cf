#!/bin/bash
# one PMC pass (8 counters) of the scan backward at the mixer's call pattern for each given library (run on the GPU box):
#   tools/pmc_ab.sh libA.so libB.so ...        (paths relative to csrc/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  OUT=/tmp/pmcab_$(basename $L .so); rm -rf $OUT
  DIFFMA_HIP_LIB=$R/diffma-diffusion-mamba_amd/csrc/$L rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVE_CYCLES -d $OUT -o k -- python $R/tools/bench_kernels.py --iters 3 --dtype bf16 --batch 1536 --only scan_hoist > $OUT.log 2>&1
  python - <<PY
import sqlite3, glob
cur = sqlite3.connect(glob.glob("$OUT/*.db")[0]).cursor()
rows = {r[0]: (r[1], r[2]) for r in cur.execute("select counter_name, avg(value), avg(duration) from counters_collection where kernel_name like '%scan_bwd_kernel%' group by counter_name")}
ws = 1536 * 16 * 196
print("$L", " ".join(f"{k}={v[0]/ws:.1f}/ws" for k, v in sorted(rows.items())), "dur_us=%.0f" % (list(rows.values())[0][1] / 1e3))
PY
done
