#!/bin/bash
# The product source carries no "wrong results" / bisect switches: they live in
# tools/micro/patches/elo_fused_experiments.patch -- ELO_NO_MFMA_SHAPE_GUARD, ELO_RR_EQUAL_LOADS, ELO_RR_BARRIER=1|2|3,
# ELO_RR_WHATIF_HALF_READS, ELO_CV1_STOP=1|2|3, ELO_CV1_EARLY_DESCRIPTORS.  This script copies csrc/ (and include/) to a scratch
# directory, applies the patch and prints the path of the patched elo_fused.hip: the experiment scripts (tools/rr_bisect.sh,
# tools/cv1_phases.sh, tools/cv1_phase_counters.sh) compile THAT file with their -D flags and link it with the product's other objects.
#   SRC=$(bash tools/micro/experiment_source.sh) && hipcc ... -DELO_RR_BARRIER=3 -c $SRC -o x.o
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=${ELO_EXPERIMENT_DIR:-$(mktemp -d /tmp/elo_experiment.XXXXXX)}
mkdir -p $TMP/efficientlo-net_amd $TMP/include
cp -r $ROOT/efficientlo-net_amd/csrc $TMP/efficientlo-net_amd/
cp $ROOT/include/elo.h $TMP/include/
( cd $TMP && patch -s -p1 < $ROOT/tools/micro/patches/elo_fused_experiments.patch )
echo $TMP/efficientlo-net_amd/csrc/elo_fused.hip
