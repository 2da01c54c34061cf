#!/bin/bash
# Round-6 profile set (the round-5 script with the names moved on) (run on the GPU box; the summaries land in gpurun_out/ and are copied into profiles/ by hand):
#   1. rocprofv3 --kernel-trace --stats of the default bench command            -> gpurun_out/prof_bench/top.txt
#   2. PMC passes (separate runs, --kernel-trace only) of the scan kernels as the mixer launches them since round 3 (no z, delta
#      activated) and of the stand-alone fp32 forward                            -> gpurun_out/r06_pmc_<tag>.txt + r06_valu.json
#   3. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the same launches -> gpurun_out/r06_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
[ "$1" = "nobench" ] || bash $R/tools/prof_bench.sh --no-extras --no-config-legs > $R/gpurun_out/prof_bench_r06.log 2>&1
OUT=/tmp/pmc_r06; rm -rf $OUT; mkdir -p $OUT
run_pmc() {   # tag, bench_kernels args...
  local TAG=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $OUT/${TAG}_p1 -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/${TAG}_p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/${TAG}_p2 -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/${TAG}_p2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/${TAG}_p3 -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/${TAG}_p3.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_f -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/${TAG}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_w -o k -- python $R/tools/bench_kernels.py --iters 3 "$@" > $OUT/${TAG}_w.log 2>&1
}
run_pmc hoist_bf16 --dtype bf16 --batch 1536 --only scan_hoist
run_pmc fwd_fp32 --dtype fp32 --batch 768 --only scan_fwd
python - <<PY
import sqlite3, glob, json
OUT, R = "$OUT", "$R"
L, D, N = 196, 1024, 16
def counters(tag, p, like):
    dbs = glob.glob(f"{OUT}/{tag}_{p}/*.db")
    if not dbs: return {}
    cur = sqlite3.connect(dbs[0]).cursor()
    try:
        return {r[0]: (r[1], r[2], r[3], r[4]) for r in cur.execute(
            "select counter_name, avg(value), count(*), avg(duration), min(kernel_name) from counters_collection where kernel_name like ? group by counter_name", (like,))}
    except Exception as e:
        return {"ERR": (str(e), 0, 0, "")}
valu, traffic, lines = {}, {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/prof_r06.sh), KiB per dispatch; gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests at 64 B, so reads = 2 x FETCH_SIZE; WRITE_SIZE as is.  hbm_bytes_per_launch = (2*FETCH + WRITE) * 1024.", "kernels": {}}, []
for tag, like, key, nseq, es, has_ckpt in (("hoist_bf16", "%scan_bwd_kernel%", "dm_selective_scan_bwd:bf16", 1536, 2, True),
                                             ("hoist_bf16", "%scan_fwd_kernel%", "dm_selective_scan_fwd:bf16", 1536, 2, True),
                                             ("fwd_fp32", "%scan_fwd_kernel%", "dm_selective_scan_fwd:fp32:standalone", 768, 4, False)):
    c = {}
    for p in ("p1", "p2", "p3", "f", "w"):
        c.update(counters(tag, p, like))
    lines.append(f"== {key}  nseq {nseq}  ({tag}; kernel {c.get('SQ_WAVES', ('', 0, 0, ''))[3][:110]})")
    for k, v in sorted(c.items()):
        lines.append(f"   {k} {v[0]:.5g} n={v[1]} dur_ns={v[2]:.0f}" if isinstance(v[0], float) else f"   {k} {v}")
    try:
        wave_steps = nseq * (D // 64) * L
        insts = c["SQ_INSTS_VALU"][0] / wave_steps
        busy = 4.0 * c["SQ_ACTIVE_INST_VALU"][0] / wave_steps            # SIMD-cycles the VALU pipe is occupied per wave-step
        dur = c["GRBM_GUI_ACTIVE"][2] * 1e-9
        clk = c["GRBM_GUI_ACTIVE"][0] / 8.0 / dur / 1e9                   # summed over 8 XCDs
        t100 = wave_steps / 1024.0 * busy / (clk * 1e9)                   # 1024 SIMDs, 100 % pipe utilisation
        if "bwd" in key: alg = 5 * nseq * D * L * es + 2 * nseq * N * L * 4
        elif "standalone" in key: alg = 4 * nseq * D * L * es + 2 * nseq * N * L * es + 4 * D * N + 8 * D
        else: alg = 3 * nseq * D * L * es + 2 * nseq * N * L * es + 4 * D * N + 8 * D
        valu[key] = dict(nseq=nseq, valu_insts_per_wave_step=round(insts, 1), valu_busy_cycles_per_wave_step=round(busy, 1), clock_ghz=round(clk, 3),
                         duration_us_under_profiler=round(dur * 1e6, 1), pipe_busy_frac=round(t100 / dur, 3),
                         time_at_full_pipe_us=round(t100 * 1e6, 1), algorithmic_bytes=alg,
                         valu_ceiling_frac_of_8TBps=round(alg / t100 / 8e12, 4), achieved_frac_of_8TBps=round(alg / dur / 8e12, 4))
    except KeyError as e:
        valu[key] = {"missing": str(e)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic["kernels"][key] = dict(nseq=nseq, FETCH_SIZE_KiB=c["FETCH_SIZE"][0], WRITE_SIZE_KiB=c["WRITE_SIZE"][0],
                                       hbm_bytes_per_launch=int((2 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024),
                                       avg_duration_ns_under_profiler=c["FETCH_SIZE"][2], kernel=c["FETCH_SIZE"][3][:100])
open(f"{R}/gpurun_out/r06_pmc_scan_kernels.txt", "w").write("\n".join(lines) + "\n")
open(f"{R}/gpurun_out/r06_valu.json", "w").write(json.dumps(valu, indent=1))
open(f"{R}/gpurun_out/r06_traffic.json", "w").write(json.dumps(traffic, indent=1))
print("\n".join(lines)); print(json.dumps(valu, indent=1)); print(json.dumps(traffic, indent=1))
PY
