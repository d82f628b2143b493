#!/bin/bash
# Tuning sweep of the engine's compile-time constants (run here to build, via gpurun to measure).
# usage: profiles/sweep.sh build | profiles/sweep.sh run
set -e
R=$(cd $(dirname $0)/.. && pwd)
D=$R/sweep_libs
VARIANTS=${VARIANTS:-"base: su4:-DCSL_SU=4 q1536:-DCSL_QMEAN=1536 q2560:-DCSL_QMEAN=2560 sct4096:-DCSL_SCT=4096 tpb2:-DCSL_TPB=2"}
REPS=${REPS:-2}
SWEEP_BENCH_ARGS=${SWEEP_BENCH_ARGS---no-kernel-timing}
if [ "$1" = build ]; then
  mkdir -p $D
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr + ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $R/include -w $flags \
      -o $D/lib_$name.so $R/occ-gnn_amd/csrc/cslicer_hip.hip $R/occ-gnn_amd/csrc/aggregate.hip $R/occ-gnn_amd/csrc/gemm_lt.hip $R/occ-gnn_amd/csrc/sage_step.hip -ldl &
  done; wait; ls $D
else
  for v in $VARIANTS; do
    name=${v%%:*}
    for rep in $(seq $REPS); do
      echo -n "$name: "
      CSLICER_LIB=$D/lib_$name.so python3 $R/bench.py --no-cpu-baseline --no-compat $SWEEP_BENCH_ARGS --e2e-steps 0 --steps 30 --warmup 5 2>/dev/null | python3 $R/profiles/show_bench.py | ${SWEEP_FILTER:-head -1}
    done
  done
fi
