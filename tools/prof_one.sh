#!/bin/bash
# kernel-trace one script under rocprofv3 and list the top kernels: tools/prof_one.sh <tag> <cmd...>   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
( cd $R && rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- "$@" > $OUT/run.log 2>&1 )
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace/*.db")[0]
cur = sqlite3.connect(db).cursor()
for n, c, t, a, p in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 14"):
    print(f"{p:6.2f}% calls={c:5d} avg_us={a:9.1f}  {n[:100]}")
PY
