#!/usr/bin/env bash
# oracle/build_ref.sh -- build the REFERENCE's own grouping kernels for the host.
#
# TEST INFRASTRUCTURE ONLY (see oracle/elo_oracle.c header).
#
# The two reference kernels (tf_ops/2d_conv_{random,select}_k/fused_conv_g.cu)
# are plain C arithmetic inside a CUDA __global__ function; the only CUDA in
# them is the execution-space keyword, the blockIdx/threadIdx/blockDim
# built-ins, the device overload max(float,float) and the <<<>>> launcher at
# the bottom of each file.  This script compiles the kernel BODIES where they
# lie under /root/reference with g++:
#   * everything from the launcher definition on is cut (it holds the <<<>>>),
#   * `#include <cuda_runtime.h>` is dropped (nothing from it is used),
#   * a 6-line prelude supplies the CUDA *language* built-ins as host variables
#     so that one call with blockIdx.x=b, blockDim.x=1 runs the kernel's own
#     grid-stride loop over every centre of batch element b.
# No reference arithmetic is replaced.  The TF op glue (fused_conv.cpp) needs
# TensorFlow headers + libtensorflow_framework and is NOT buildable here; its
# zero-fill of the outputs (fused_conv.cpp:154-166) is done by the driver
# function below.
#
# Output: oracle/_ref/libelo_ref.so only (git-ignored; it travels with gpurun).
# The generated translation units live in a temp dir and are deleted, so no
# reference source text is ever written into the repository.
set -euo pipefail
REF=${ELO_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
[ -f "$REF/tf_ops/2d_conv_random_k/fused_conv_g.cu" ] || { echo "reference not present at $REF; keeping any prebuilt $OUT" >&2; exit 0; }
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT

cat > "$TMP/prelude.h" <<'EOF'
#include <cmath>
#define __global__
struct elo_ref_dim3 { int x, y, z; };
static thread_local elo_ref_dim3 blockIdx{0, 0, 0}, threadIdx{0, 0, 0}, blockDim{1, 1, 1};
static inline float max(float a, float b) { return a > b ? a : b; }
EOF

emit() {  # $1 = op dir, $2 = kernel symbol, $3 = exported driver name
    local src="$REF/tf_ops/$1/fused_conv_g.cu"
    {
        echo '#include "prelude.h"'
        awk '/^void FusedConv.*Launcher/ {exit} {print}' "$src" | grep -v 'cuda_runtime.h'
        cat <<EOF
#include <cstring>
extern "C" int $3(int batch, int H, int W, int npoints, int kH, int kW, int K, int flag_copy,
                  float distance, int stride_h, int stride_w, const float *xyz1, const float *xyz2,
                  const int *idx_n2, const int *random_hw, int *selected_bhw_idx, float *valid_idx,
                  float *valid_in_dis_idx, float *selected_mask, int H2, int W2)
{
    const size_t KT = (size_t)kH * kW, BN = (size_t)batch * npoints;
    std::memset(selected_bhw_idx, 0, sizeof(int) * BN * K * 3);
    std::memset(valid_idx, 0, sizeof(float) * BN * KT);
    std::memset(valid_in_dis_idx, 0, sizeof(float) * BN * KT);
    std::memset(selected_mask, 0, sizeof(float) * BN * K);
    blockDim.x = 1; threadIdx.x = 0;
    for (int b = 0; b < batch; ++b) {
        blockIdx.x = b;
        $2(batch, H, W, npoints, kH, kW, K, flag_copy, distance, stride_h, stride_w, xyz1, xyz2,
           idx_n2, random_hw, selected_bhw_idx, valid_idx, valid_in_dis_idx, selected_mask, H2, W2);
    }
    return 0;
}
EOF
    } > "$TMP/$3.cpp"
}

emit 2d_conv_random_k fused_conv_random_k_gpu elo_ref_fused_conv_random_k
emit 2d_conv_select_k fused_conv_select_k_gpu elo_ref_fused_conv_select_k

# -O2 like the reference's nvcc line (fused_conv.sh:2); no FMA contraction on
# the host (contracted and non-contracted builds gave identical
# index checksums on the probe inputs; the build's contract is "no contraction").
g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -w -I"$TMP" \
    "$TMP/elo_ref_fused_conv_random_k.cpp" "$TMP/elo_ref_fused_conv_select_k.cpp" \
    -o "$OUT/libelo_ref.so"
echo "built $OUT/libelo_ref.so"
# The same bodies WITH contraction (nvcc contracts a*b+c into fma by default, fused_conv.sh:2 passes no -fmad=false):
# tests/test_oracle_vs_ref.py checks that the goldens and the sweep do not depend on it.
g++ -O2 -std=c++17 -ffp-contract=fast -mfma -fPIC -shared -w -I"$TMP" \
    "$TMP/elo_ref_fused_conv_random_k.cpp" "$TMP/elo_ref_fused_conv_select_k.cpp" \
    -o "$OUT/libelo_ref_fma.so"
echo "built $OUT/libelo_ref_fma.so"
