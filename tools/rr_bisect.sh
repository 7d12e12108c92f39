#!/bin/bash
# Bisect builds for DESIGN.md section 3b finding 4 (chain-kernel waves that issue no global load give wrong rows).  The
# trail that led to the cause (gpurun_out/r04_bisect, summarised in profiles/r04_rr_bisect.txt): -amdgpu-waitcnt-forcezero
# cures it -> patching the s_waitcnt of the DEVICE ASSEMBLY of that one kernel by hand (tools/micro/asm_variants/) narrows it
# to ONE `s_waitcnt lgkmcnt(1)` between a 16x16x32 MFMA and the 16x16x16 MFMA that accumulates onto its result -> not the
# LDS data but the MFMA pair itself (tools/micro/lds_read2_wait.hip, tools/micro/mfma_srcc_hazard.hip).
# Builds one library per variant (only csrc/elo_fused.hip is recompiled; the other objects are the product's), here,
# without a GPU (ORDER="a b" picks variants):
#     tools/rr_bisect.sh build
# and runs tools/rr_stress.py --where on each of them on the GPU box:
#     tools/rr_bisect.sh run [runs]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/efficientlo-net_amd
OUT=$ROOT/tools/micro/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
declare -A V
V[shipped]=""                                                                # idle waves skip their loads, mfma_shape_guard on: none differ
V[noguard]="-DELO_NO_MFMA_SHAPE_GUARD"                                       # round 3's failing form: 20 of 20 runs differ at C = 16
V[noguard_equal_loads]="-DELO_NO_MFMA_SHAPE_GUARD -DELO_RR_EQUAL_LOADS"      # round 3's shipped form: right by the luck of its schedule
V[noguard_forcezero]="-DELO_NO_MFMA_SHAPE_GUARD -mllvm -amdgpu-waitcnt-forcezero"
V[noguard_syncthreads]="-DELO_NO_MFMA_SHAPE_GUARD -DELO_RR_BARRIER=1"
V[noguard_fences]="-DELO_NO_MFMA_SHAPE_GUARD -DELO_RR_BARRIER=2"
V[noguard_stage1]="-DELO_NO_MFMA_SHAPE_GUARD -DELO_RR_STAGE=1"
ORDER="${ORDER:-shipped noguard noguard_equal_loads noguard_forcezero noguard_syncthreads noguard_fences noguard_stage1}"
case "$1" in
build)
    mkdir -p $OUT
    python $PKG/build.py > /dev/null
    XSRC=$(bash $ROOT/tools/micro/experiment_source.sh)      # elo_fused.hip with the bisect switches patched in (they are not in the product source)
    for v in $ORDER; do
        ( /opt/rocm/bin/hipcc $FLAGS ${V[$v]} -c $XSRC -o $OUT/elo_fused.$v.o &&
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/elo_fused.$v.o $(ls $PKG/build/*.hip.o | grep -v elo_fused) -o $OUT/libelo_$v.so &&
          rm $OUT/elo_fused.$v.o && echo built $v ) &
    done
    wait ;;
run)
    for v in $ORDER; do
        echo "=== $v (${V[$v]})"
        ELO_LIB_PATH=$OUT/libelo_$v.so python $ROOT/tools/rr_stress.py ${2:-20} --where --half-only 2>&1 | tail -40
    done ;;
*) echo "usage: $0 build | run [runs]"; exit 2 ;;
esac
