#!/bin/bash
# One library from HAND-PATCHED device assembly of csrc/elo_fused.hip (the unguarded build: -DELO_NO_MFMA_SHAPE_GUARD), the
# tool that narrowed DESIGN.md section 3b finding 4 from "-amdgpu-waitcnt-forcezero cures it" down to one instruction:
#     PATCHER=patch_all_waits.py   tools/micro/asm_variants/build_variant.sh A|B|C|D|E|F|G|H   (which waits of cv1_rr<16,*,HALF> become 0)
#     PATCHER=patch_chain_sites.py tools/micro/asm_variants/build_variant.sh S0_7|N1|T|P|Q|... (which of the chain's 29 counted lgkmcnt waits)
#     PATCHER=patch_site4.py       tools/micro/asm_variants/build_variant.sh S4|G0|G1|G3|G7|NB (the one wait; s_nop n behind it)
# device listing -> patch -> assemble -> link -> bundle -> host object with that fat binary -> tools/micro/build/libelo_asm_<V>.so;
# tools/micro/run_asm_variants.sh then runs tools/rr_stress.py on every such library.
set -e
V=$1; L=/opt/rocm/lib/llvm/bin
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../../.." && pwd); PKG=$ROOT/efficientlo-net_amd
WORK=${WORK:-/tmp/elo_asm_variants}; mkdir -p $WORK $ROOT/tools/micro/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DELO_NO_MFMA_SHAPE_GUARD -I$ROOT/include"
[ -f $WORK/base.s ] || /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S $PKG/csrc/elo_fused.hip -o $WORK/base.s
python $HERE/${PATCHER:-patch_all_waits.py} $WORK/base.s $WORK/mod_$V.s $V
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $WORK/mod_$V.s -o $WORK/mod_$V.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $WORK/mod_$V.out $WORK/mod_$V.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$WORK/mod_$V.out -output=$WORK/mod_$V.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $WORK/mod_$V.hipfb -c $PKG/csrc/elo_fused.hip -o $WORK/host_$V.o
python $PKG/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $WORK/host_$V.o $(ls $PKG/build/*.hip.o | grep -v elo_fused) -o $ROOT/tools/micro/build/libelo_asm_$V.so
echo built $V
