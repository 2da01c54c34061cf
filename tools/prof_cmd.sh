#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command (run on the GPU box): the per-kernel summary goes to gpurun_out/<tag>_top.txt
#   usage: bash tools/prof_cmd.sh <tag> <command ...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- "$@" > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace/*.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 40"))
with open("$R/gpurun_out/${TAG}_top.txt", "w") as f:
    for n, c, t, a, p in rows:
        f.write(f"{p:6.2f}% calls={c:6d} avg_us={a:10.1f} tot_ms={t/1e3:10.2f}  {n[:120]}\n")
print(open("$R/gpurun_out/${TAG}_top.txt").read())
PY
