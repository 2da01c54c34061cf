#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <kernel-substring> <command...>   (run on the GPU box) -- PMC counters of one kernel under any command
cd /tmp && export TMPDIR=/tmp
TAG=$1; KERN=$2; shift 2
OUT=/tmp/pmc_$TAG          # raw rocprof output stays on the box; the caller redirects the printed summary into gpurun_out/
rm -rf $OUT; mkdir -p $OUT
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $OUT/p1 -o k -- "$@" > $OUT/p1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/p2 -o k -- "$@" > $OUT/p2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/p3 -o k -- "$@" > $OUT/p3.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $OUT/p4 -o k -- "$@" > $OUT/p4.log 2>&1
python - <<PY
import sqlite3, glob
for p in ("p1","p2","p3","p4"):
    dbs = glob.glob("$OUT/%s/*.db" % p)
    if not dbs: print(p, "no db"); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    try:
        rows = list(cur.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%$KERN%' group by counter_name"))
    except Exception as e:
        print(p, "ERR", e); continue
    for r in rows: print(p, r[0], f"{r[1]:.4g}", "n=%d" % r[2], "dur_ns=%.0f" % (r[3] or 0))
PY
