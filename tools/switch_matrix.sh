#!/bin/bash
# (GPU box) every tuning / testing switch of DESIGN.md section 4.5 against a representative slice of the parity suite: the alternatives
# must stay correct, not just measurable.  One line per switch.
cd "$GRAFT_REPO_ROOT"
SEL=${SEL:-'(random_system or brick or forces_only or triclinic or bench_workload or sharded) and not mixed'}
SWITCHES=${SWITCHES:-"SNB_EWALD_ERFC=1 SNB_SCALAR_ENERGY_KERNEL=1 SNB_NO_FUSED_LISTS=1 SNB_NO_FUSED_Z=1 SNB_ZSLABS=2 SNB_NO_FIXED_SPREAD=1 SNB_FFT_TWOPASS=0 SNB_NO_INTERP_BRICKS=1 SNB_INTERP_ZSLABS=2 SNB_CONCURRENT_PME=1 SNB_OVERLAP=0 SNB_NO_GRAPH_UPDATE=1 SNB_EAGER_REBUILD_STEP=1 SNB_PLANE_DYNAMIC=1 SNB_NO_RECT_PLANES=1 SNB_BOX_PRUNE=1 SNB_NO_FUSED_FINISH=1 SNB_NO_STEP_GRAPH=1 SNB_NO_SORT_GRAPH=1 SNB_DIRECT_WGS=256 SNB_ITEM_TILES=4 SNB_NO_OWN_SPREAD=1 SNB_OWN_SLABS=3 SNB_SPREAD_MARGIN=0 SNB_INTERP_THREADS=1024 SNB_NO_KERNEL_STAMPS=1 SNB_NO_PINNED_RING=1 SNB_NB_BOX_WALK=1 SNB_NO_PLANE_FFT=1 SNB_ZMIX_NBY=4 SNB_ZMIX_NT=512 SNB_PLANE_NT=768 SNB_SIDE_REBUILD=0 SNB_SIDE_LEAD=1 SNB_NO_FUSED_ENERGY_FINISH=1"}
# (SNB_HOST_TRICLINIC=1 is left out: the triclinic tests assert that the GPU builder was used, which that switch turns off; 90 s per switch
# with the default selection -- one gpurun call holds about twelve)
# Expected deviation, stated here and not in the test: SNB_SCALAR_ENERGY_KERNEL=1 evaluates the pair energies with the A&S erfc,
# whose one-signed 1.5e-7 error sums to 1.6e-3 of the smallest cross slice of the sharded test box in single precision (the reason the
# default energy kernel uses the degree-13 polynomial): that switch runs the selection without the single-precision sharded case.
for sw in $SWITCHES; do
  sel="$SEL"
  [ "$sw" = "SNB_SCALAR_ENERGY_KERNEL=1" ] && sel="($SEL) and not (sharded and single)"
  res=$(env $sw timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "$sel" 2>&1 | tail -1)
  echo "$sw: $res"
done
