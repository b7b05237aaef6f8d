#!/bin/bash
# GPU box: the round's judged artefacts in one call -- the default bench line (replayed steps overlapped, round 4), the same command with
# SNB_OVERLAP=0 SNB_SIDE_REBUILD=0 (serial replayed steps and in-line rebuilds, rounds 1-3's form), rocprofv3 --kernel-trace --stats of the bench command with the kernels
# running ALONE (SNB_OVERLAP=0: the per-kernel durations the `roofline` objects are defined on; the stamped eager steps of the default line
# are serial too and must agree) and once more as shipped (overlapped: the pair kernel's two launches and the PME kernels beside it),
# one SQ --pmc pass and one MFMA pass (kernels alone).  Copy gpurun_out/final/* into profiles/ afterwards (tools/make_profiles.py).
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
SNB_OVERLAP=0 SNB_SIDE_REBUILD=0 python3 bench.py --no-cpu-baseline --no-double > gpurun_out/final/bench_serial.json 2> gpurun_out/final/bench_serial.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SNB_OVERLAP=0 SNB_SIDE_REBUILD=0      # (kernels alone: no PME chain and no list build beside the pair kernel)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python3 bench.py --no-cpu-baseline --no-double > gpurun_out/final/bench_under_rocprof.json 2> gpurun_out/final/rocprof.err
cp gpurun_out/final/stats/*/*kernel_stats.csv gpurun_out/final/kernel_stats.csv
bash tools/pmc_sq.sh final_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_ANY" --steps 40 --warmup 5 > gpurun_out/final/pmc_sq.txt
# matrix-core BUSY cycles of the kernel that carries the lambda mix (plane path: k_fftZInvMix, v_mfma_f32_4x4x1_16b_f32; else k_convolveX), beside the count above
bash tools/pmc_sq.sh final_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA" --steps 40 --warmup 5 > gpurun_out/final/pmc_mfma.txt
unset SNB_OVERLAP SNB_SIDE_REBUILD
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats_overlap -- python3 bench.py --no-cpu-baseline --no-double > gpurun_out/final/bench_under_rocprof_overlap.json 2> gpurun_out/final/rocprof_overlap.err
cp gpurun_out/final/stats_overlap/*/*kernel_stats.csv gpurun_out/final/kernel_stats_overlap.csv
rm -rf gpurun_out/final/stats gpurun_out/final/stats_overlap gpurun_out/final_sq gpurun_out/final_mfma
head -c 400 gpurun_out/final/bench.json; echo; head -5 gpurun_out/final/kernel_stats.csv | cut -c1-160
