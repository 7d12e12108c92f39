#!/bin/bash
# The SrcC hazard of the gfx950 matrix cores over the full shape cross-product (tools/micro/gen_mfma_srcc_matrix.py), on the GPU box:
#   bash tools/mfma_hazard_matrix.sh gpurun_out/r05/mfma_srcc_matrix.txt
OUT=${1:-gpurun_out/r05/mfma_srcc_matrix.txt}
mkdir -p "$(dirname "$OUT")"
python tools/micro/gen_mfma_srcc_matrix.py > /tmp/mfma_srcc_matrix.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 /tmp/mfma_srcc_matrix.hip -o /tmp/mfma_srcc_matrix 2>/dev/null || { echo "hipcc failed"; exit 1; }
{ echo "# $(/opt/rocm/bin/hipcc --version | grep -i 'clang version\|HIP version' | tr '\n' ' ')"; timeout 600 /tmp/mfma_srcc_matrix; } > "$OUT" 2>&1
grep -c "^PAIR" "$OUT"
