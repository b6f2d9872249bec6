#!/bin/bash
# scripts/ab_sgm.sh "<lib1> <lib2> ..." [reps]  -- alternate builds of the library on ONE box, SGM stage times at config B and E
LIBS=$1; REPS=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for i in $(seq $REPS); do for L in $LIBS; do
  echo "$(basename $L .so): $(WASS_GPU_LIB=$ROOT/$L python scripts/time_sgm.py 2456 2058 256 8 12 2>/dev/null | tail -1)"
done; done
for i in 1 2; do for L in $LIBS; do
  echo "$(basename $L .so): $(WASS_GPU_LIB=$ROOT/$L python scripts/time_sgm.py 3840 2160 512 8 5 2>/dev/null | tail -1)"
done; done
for L in $LIBS; do
  echo "$(basename $L .so): $(WASS_GPU_LIB=$ROOT/$L python scripts/time_sgm.py 2456 2058 256 5 12 2>/dev/null | tail -1)"
done
