#!/bin/bash
# soak of the opt-in two-stream mode (DIFFMA_OVERLAP_MIXERS=1; the package adds TENSILE_STREAMK_DATA_PARALLEL=1): a hang shows up
# as rc=124.  Run on the GPU box: tools/soak_two_streams.sh > gpurun_out/soak.txt
export DIFFMA_OVERLAP_MIXERS=1
n=0; bad=0
r() { n=$((n+1)); out=$(timeout -k 5 150 python bench.py --cpu-steps 0 "$@" 2>&1 | tail -1); rc=$?; v=$(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>/dev/null) || { bad=$((bad+1)); v="FAILED rc=$rc"; }; echo "$* | $v"; }
for i in 1 2 3 4 5 6; do r --steps 6; done
for b in 256 64 16; do for i in 1 2; do r --steps 6 --batch-per-gpu $b; done; done
for i in 1 2 3; do r --model DiffMa-XL/2 --use-mamba2 --batch-per-gpu 64 --steps 4 --warmup 2; done
for i in 1 2; do r --model DiffMa-XL/2 --batch-per-gpu 64 --steps 4 --warmup 2; done
for i in 1 2; do r --model DiffMa-XXL/2 --batch-per-gpu 32 --steps 3 --warmup 1; done
for i in 1 2; do r --batch-per-gpu 8 --steps 20 --graph; done
for i in 1 2; do r --mode sample --graph --batch-per-gpu 8 --steps 30; done
for i in 1 2; do BENCH_FORCE_DDP=1 r --steps 6; done
echo "runs=$n failed=$bad"
