#!/bin/bash
# GPU box: the round's judged artefacts in one call -- default bench line, rocprofv3 --kernel-trace --stats of the SAME command,
# one SQ --pmc pass.  Copy gpurun_out/final/* into profiles/ afterwards.
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python3 bench.py > gpurun_out/final/bench_under_rocprof.json 2> gpurun_out/final/rocprof.err
cp gpurun_out/final/stats/*/*kernel_stats.csv gpurun_out/final/kernel_stats.csv
bash tools/pmc_sq.sh final_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_ANY" --steps 40 --warmup 5 > gpurun_out/final/pmc_sq.txt
# matrix-core BUSY cycles of the kernel that carries the lambda mix (plane path: k_fftZInvMix, v_mfma_f32_4x4x1_16b_f32; else k_convolveX), beside the count above
bash tools/pmc_sq.sh final_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA" --steps 40 --warmup 5 > gpurun_out/final/pmc_mfma.txt
head -c 400 gpurun_out/final/bench.json; echo; head -5 gpurun_out/final/kernel_stats.csv | cut -c1-160
