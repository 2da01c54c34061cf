#!/bin/bash
# PMC passes (separate rocprofv3 runs, --kernel-trace only) of one command, per-kernel averages to gpurun_out/<tag>_pmc.txt (run on the GPU box)
#   usage: bash tools/pmc_cmd.sh <tag> <kernel-name LIKE pattern> <command ...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; LIKE=$2; shift; shift
OUT=/tmp/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
i=0
# PMC_ONE="<counters>": a single pass with these counters instead of the standard passes
if [ -n "$PMC_ONE" ]; then
  rocprofv3 --kernel-trace --pmc $PMC_ONE -d $OUT/p1 -o k -- "$@" > $OUT/p1.log 2>&1
else
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o k -- "$@" > $OUT/p$i.log 2>&1
done
fi
python - <<PY
import sqlite3, glob
out = []
for like in "$LIKE".split("|"):
    c = {}
    for db in sorted(glob.glob("$OUT/p*/*.db")):
        cur = sqlite3.connect(db).cursor()
        try:
            for r in cur.execute("select counter_name, avg(value), count(*), avg(duration), min(kernel_name) from counters_collection where kernel_name like ? group by counter_name", (like,)):
                c[r[0]] = r[1:]
        except Exception as e:
            c["ERR " + db] = (str(e), 0, 0, "")
    out.append(f"== {like}")
    for k, v in sorted(c.items()):
        out.append(f"   {k:28s} {v[0]:.6g}  n={v[1]} dur_us={v[2]/1e3:.1f}  {str(v[3])[:60]}" if isinstance(v[0], float) else f"   {k} {v}")
open("$R/gpurun_out/${TAG}_pmc.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
