#!/bin/bash
# Kernel launches PER REPLAYED STEP of the graphed training step: two rocprofv3 kernel traces of bench.py --graph with K1 and K2 timed
# steps (same warm-up, recorded GEMM solutions, no tuning); everything that happens once (warm-up, capture, the state snapshot / restore
# of GraphedTrainStep, optimizer-state allocation) cancels in the difference.   bash tools/graph_node_census.sh <batch> [K1 K2]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$1; K1=${2:-5}; K2=${3:-25}
for K in $K1 $K2; do
  OUT=/tmp/census_$K; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --batch-per-gpu $B --graph --steps $K --warmup 2 --cpu-steps 0 --no-extras --gemm-tuning frozen > $OUT/run.log 2>&1
done
python - <<PY
import sqlite3, glob
def load(k):
    cur = sqlite3.connect(glob.glob(f"/tmp/census_{k}/trace/*.db")[0]).cursor()
    return {n: (c, t) for n, c, t in cur.execute("select name, total_calls, total_duration from top_kernels")}
a, b = load($K1), load($K2)
d = $K2 - $K1
rows = []
for n in set(a) | set(b):
    ca, ta = a.get(n, (0, 0)); cb, tb = b.get(n, (0, 0))
    rows.append(((cb - ca) / d, (tb - ta) / d / 1e3, ca, cb, n))   # the view's durations are microseconds: ms per step
rows.sort(reverse=True)
per_step = sum(r[0] for r in rows)
us = sum(r[1] for r in rows)
out = [f"DiffMa-L/2 graphed training step, batch $B: kernel launches per replayed step = (trace of $K2 steps - trace of $K1 steps) / {d}",
       f"total {per_step:.1f} launches and {us:.2f} ms of kernel time per step", ""]
for r in rows:
    if abs(r[0]) >= 0.05:
        out.append(f"{r[0]:8.1f} /step {r[1]:8.3f} ms/step   (calls {r[2]} -> {r[3]})  {r[4][:110]}")
out.append("")
out.append("launches that do NOT scale with the step count (set-up: warm-up steps, snapshot / restore of the training state, optimizer state):")
for r in sorted(rows, key=lambda r: -r[2]):
    if abs(r[0]) < 0.05 and r[2] >= 50:
        out.append(f"   calls {r[2]:6d} -> {r[3]:6d}  {r[4][:110]}")
open("$R/gpurun_out/graph_census_b$B.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
PY
