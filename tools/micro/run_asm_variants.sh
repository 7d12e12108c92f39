#!/bin/bash
# runs tools/rr_stress.py on every tools/micro/build/libelo_asm_*.so (libraries rebuilt from hand-patched device assembly)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for lib in $ROOT/tools/micro/build/libelo_asm_*.so; do
    echo "=== $(basename $lib)"
    ELO_LIB_PATH=$lib python $ROOT/tools/rr_stress.py ${1:-10} --half-only 2>&1 | grep "runs differing"
done
