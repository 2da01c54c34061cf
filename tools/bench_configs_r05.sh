#!/bin/bash
# Round-5 bench lines (GPU box): default (full), config 3 at the reference's batches (graphed / eager), census of the graphed step.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { local f=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/$f.json 2>gpurun_out/$f.err || tail -3 gpurun_out/$f.err; tail -1 gpurun_out/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['ms_per_step'], d['value'])"; }
run r05_bench_c3_l2_train_b1_graph --batch-per-gpu 1 --graph --steps 50 --cpu-steps 0
run r05_bench_c3_l2_train_b8_graph --batch-per-gpu 8 --graph --steps 50 --cpu-steps 0
run r05_bench_c3_l2_train_b8_eager --batch-per-gpu 8 --steps 20 --cpu-steps 0
run r05_bench_c3_l2_train_b64 --batch-per-gpu 64 --steps 15 --cpu-steps 0
run r05_bench_default
timeout 600 bash tools/graph_node_census.sh 1 > /dev/null 2>&1; head -3 gpurun_out/graph_census_b1.txt
