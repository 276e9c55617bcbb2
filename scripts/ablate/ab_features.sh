#!/bin/bash
# Same-box A/B of library builds on the feature stage alone + the whole bench: bash scripts/ablate/ab_features.sh <steps> <rounds> <lib B> ...
STEPS=$1; R=$2; shift 2
for v in A "$@" A "$@"; do
  if [ $v = A ]; then unset REGNET_HIP_LIB; else export REGNET_HIP_LIB=$PWD/$v; fi
  echo "== features alone: $v"; REPS=30 ROWS=4 python scripts/features_alone.py 8 2>&1 | grep -v amdgpu.ids
done
unset REGNET_HIP_LIB
bash scripts/ablate/ab_libs3.sh $STEPS $R "$@"
