#!/bin/bash
# rocprofv3 kernel-trace of the bench step (run on the GPU box): writes gpurun_out/prof_bench/top.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/prof_bench      # raw rocprof output stays on the box (gpurun_out/ is capped at 64 MiB); only top.txt travels
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 "$@" > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace/*.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 80"))
tot = sum(r[2] for r in cur.execute("select name, total_calls, total_duration from top_kernels"))
with open("$OUT/top.txt", "w") as f:
    f.write(f"total kernel time (4 steps incl. warmup): {tot/1e3:.1f} us-units\n")
    for n, c, t, a, p in rows:
        f.write(f"{p:6.2f}% calls={c:6d} avg={a:10.1f} tot={t:12.1f}  {n[:110]}\n")
print(open("$OUT/top.txt").read())
import shutil; shutil.copy("$OUT/top.txt", "$R/gpurun_out/prof_bench/top.txt")
PY
