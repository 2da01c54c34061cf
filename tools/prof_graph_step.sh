#!/bin/bash
# rocprofv3 kernel census of the graphed data-parallel step on one rank (BENCH_FORCE_DDP=1), staged vs one all-reduce:
#   bash tools/prof_graph_step.sh <stages> <batch>   -> gpurun_out/prof_graph_st<stages>_b<batch>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ST=$1; B=$2
OUT=/tmp/prof_graph_$ST; rm -rf $OUT; mkdir -p $OUT
BENCH_FORCE_DDP=1 DIFFMA_GRAPH_STAGES=$ST rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --no-extras --graph --gemm-tuning frozen --batch-per-gpu $B > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace/*.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_calls desc limit 45"))
tot = sum(r[2] for r in cur.execute("select name, total_calls, total_duration from top_kernels"))
ncall = sum(r[1] for r in cur.execute("select name, total_calls from top_kernels"))
with open("$R/gpurun_out/prof_graph_st${ST}_b${B}.txt", "w") as f:
    f.write(f"stages=$ST batch=$B total kernel time {tot/1e6:.1f} (trace units), {ncall} launches (warm-up + capture + 23 replayed steps)\n")
    for n, c, t, a, p in rows:
        f.write(f"calls={c:6d} avg={a/1e3:8.1f}us tot={t/1e6:8.2f}ms  {n[:100]}\n")
PY
