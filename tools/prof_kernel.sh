#!/bin/bash
# usage: tools/prof_kernel.sh <tag> <kernel-substring> <bench_kernels args...>   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; KERN=$2; shift 2
OUT=/tmp/pmc_$TAG          # raw rocprof output stays on the box; the caller redirects the printed summary into gpurun_out/
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace "$@" ; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $OUT/p1 -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/p2 -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_WAVE32_LDS -d $OUT/p3 -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/p3.log 2>&1
python - <<PY
import sqlite3, glob
for p in ("p1","p2","p3"):
    dbs = glob.glob("$OUT/%s/*.db" % p)
    if not dbs: print(p, "no db"); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    try:
        rows = list(cur.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%$KERN%' group by counter_name"))
    except Exception as e:
        print(p, "ERR", e); continue
    for r in rows: print(p, r[0], f"{r[1]:.4g}", "n=%d" % r[2], "dur_ns=%.0f" % (r[3] or 0))
PY
