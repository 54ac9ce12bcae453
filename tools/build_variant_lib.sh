#!/bin/bash
# Build distil_whisper_amd/libdwamd_base.so from the WORKING TREE's csrc with extra compiler flags, for same-process A/B
# runs of a compile-time switch (tools/ab_step.py field 13, tools/ab_libs_gemm.py):
#     tools/build_variant_lib.sh "-DDW_EPI16=0" [output name, default libdwamd_base.so; libdwamd_base2.so = library 2 of the tools]
# The file is git-ignored.
set -e
FLAGS="$1"
OUT=${2:-libdwamd_base.so}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
mkdir -p $W/distil_whisper_amd $W/include
cp -r $ROOT/distil_whisper_amd/csrc $W/distil_whisper_amd/csrc
cp $ROOT/distil_whisper_amd/build.py $W/distil_whisper_amd/build.py
cp $ROOT/include/dwamd.h $W/include/
touch $W/distil_whisper_amd/__init__.py
(cd $W && DW_EXTRA_FLAGS="$FLAGS" python -c "
import importlib.util
spec = importlib.util.spec_from_file_location('b', 'distil_whisper_amd/build.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); print(m.build())")
cp $W/distil_whisper_amd/libdwamd.so $ROOT/distil_whisper_amd/$OUT
rm -rf $W
echo "built $OUT with $FLAGS"
