#!/bin/bash
# usage: scripts/mkvariant.sh <name> "<extra hipcc flags>" [dtypes: "bf16 i8 f32"]
# Builds variants/lib<name>.so: the named sweep instantiations recompiled with the extra flags, everything else
# from the production objects (make -C vllm_amd/csrc first).  Timing experiments only.
set -e
name=$1; extra=$2; dts=${3:-bf16}
cd "$(dirname "$0")/../vllm_amd/csrc"
mkdir -p build_v ../../variants
objs=""
for s in api topk_sweep sweep_inst_bf16 sweep_inst_i8 sweep_inst_f32 infonce embed_head embed_head_bwd; do
  dt=${s#sweep_inst_}
  if [[ $s == sweep_inst_* && " $dts " == *" $dt "* ]]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=200000 $extra -c $s.hip -o build_v/${name}_$s.o
    objs="$objs build_v/${name}_$s.o"
  else
    objs="$objs build/$s.o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib$name.so $objs
echo built variants/lib$name.so
