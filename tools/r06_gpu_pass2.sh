#!/bin/bash
# round-6 GPU pass 2: K2 (halves + single row-table loads) vs the whole-channel build, dm_ssd_bwd with its loads up front vs round 5,
# the kernels' tests, then the default bench (headline + config legs) timed as the driver would run it
mkdir -p gpurun_out/r06
export DIFFMA_TEST_REPORT_DIR=$PWD/gpurun_out/r06
L=diffma-diffusion-mamba_amd/csrc
( KB_BATCH=1536 tools/ab.sh scan_hoist $L/libdiffma_hip.so $L/lib_k2old.so ) > gpurun_out/r06/k2_ab2.txt 2>&1
for r in 1 2; do for lib in libdiffma_hip.so lib_ssdold.so; do echo "== $lib"; DIFFMA_HIP_LIB=$PWD/$L/$lib python tools/bench_ssd.py 256 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r06/ssd_ab.txt 2>&1
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "ssd or mamba2 or scan_bwd" -p no:cacheprovider > gpurun_out/r06/ssd_tests.txt 2>&1
tail -5 gpurun_out/r06/ssd_tests.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err ) 2> gpurun_out/r06/bench_default.time
grep "bench +" gpurun_out/r06/bench_default.err | tail -20; cat gpurun_out/r06/bench_default.time
python bench.py --gpus 2 --steps 3 --warmup 1; echo "rc of --gpus 2 on one GPU: $?"
cat gpurun_out/r06/k2_ab2.txt gpurun_out/r06/ssd_ab.txt
