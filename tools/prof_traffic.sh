#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes) of the scan kernels at the bench shape. Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/traffic
rm -rf $OUT; mkdir -p $OUT
for dt in bf16 fp32; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f_$dt -o k -- python $R/tools/bench_kernels.py --iters 3 --dtype $dt --batch 1536 --only scan_bwd,scan_idx > $OUT/f_$dt.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w_$dt -o k -- python $R/tools/bench_kernels.py --iters 3 --dtype $dt --batch 1536 --only scan_bwd,scan_idx > $OUT/w_$dt.log 2>&1
done
python - <<PY
import sqlite3, glob, json
res = {}
for dt in ("bf16", "fp32"):
    for tag, cname in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
        dbs = glob.glob("$OUT/%s_%s/*.db" % (tag, dt))
        if not dbs: continue
        cur = sqlite3.connect(dbs[0]).cursor()
        for r in cur.execute("select kernel_name, avg(value), count(*), avg(duration) from counters_collection where counter_name='%s' and kernel_name like '%%dm::scan%%' group by kernel_name" % cname):
            key = ("scan_fwd" if "scan_fwd" in r[0] else "scan_bwd") + ("_ckpt" if ", true, true, 8>" in r[0] or "true, true, true, true" in r[0] else "")
            res.setdefault(dt, {}).setdefault(r[0][:90], {})[cname] = dict(avg=r[1], n=r[2], dur_ns=r[3])
print(json.dumps(res, indent=1))
open("$OUT/traffic.json", "w").write(json.dumps(res, indent=1))
open("$R/gpurun_out/traffic.json", "w").write(json.dumps(res, indent=1))
PY
