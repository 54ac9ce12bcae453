#!/bin/bash
# Build distil_whisper_amd/libdwamd_base.so from another commit's csrc (default HEAD) for same-process A/B runs
# (tools/ab_step.py field 13, tools/attn_ab_libs.py).  The file is git-ignored.
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
mkdir -p $W/distil_whisper_amd $W/include
git -C $ROOT archive $REV distil_whisper_amd/csrc distil_whisper_amd/build.py include | tar -x -C $W
touch $W/distil_whisper_amd/__init__.py
(cd $W && python -c "
import importlib.util, sys
spec = importlib.util.spec_from_file_location('b', 'distil_whisper_amd/build.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); print(m.build())")
cp $W/distil_whisper_amd/libdwamd.so $ROOT/distil_whisper_amd/libdwamd_base.so
rm -rf $W
echo "built libdwamd_base.so from $REV"
