#!/bin/bash
# build an experimental variant of the library for an A/B run (tools/ab.sh): only the named translation units are
# recompiled with the extra flags, the rest is taken from the regular build.
#   tools/exp_lib.sh <name> "<extra hipcc flags>" [unit ...]      (default unit: scan_bwd_bf16)  -> csrc/lib_<name>.so
set -e
cd "$(dirname "$0")/../diffma-diffusion-mamba_amd/csrc"
NAME=$1; FLAGS=$2; shift 2 || true
UNITS=${@:-scan_bwd_bf16}
OBJS=$(ls *.o)
for u in $UNITS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-math-errno -Wall -Wno-unused-function $FLAGS -c $u.hip -o /tmp/exp_${NAME}_$u.o &
done
wait
LINK=""
for o in $OBJS; do
  b=${o%.o}
  if [[ " $UNITS " == *" $b "* ]]; then LINK="$LINK /tmp/exp_${NAME}_$b.o"; else LINK="$LINK $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $LINK -o lib_$NAME.so
echo built lib_$NAME.so
