#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh OUTDIR
# (kernel-trace + stats of the driver's bench command, PMC traffic passes -- FETCH_SIZE and WRITE_SIZE in separate runs, never
# combined with other trace domains -- for the train step and the pose-head kernels, cfg3 / cfg5 step traces).
set -o pipefail
OUT=${1:-gpurun_out/r04}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 bench.py --no-extra-configs --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1 || exit 1
echo "stats done"
for B in 256 1024 8192; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_step_b${B}_$C" -- python3 tools/prof_step.py $B 30 > "$OUT/pmc_step_${B}_$C.log" 2>&1 || exit 1
  done
done
# the throughput forms of the fused step (train_stream_kernel, wgrad_stream_kernel) at B = 8192: per-kernel time, then two SQ passes
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step_b8192_stats" -- python3 tools/prof_step.py 8192 30 > "$OUT/step_b8192_stats.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step_b1024_stats" -- python3 tools/prof_step.py 1024 60 > "$OUT/step_b1024_stats.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_stream_b8192_SQ_INSTS" -- python3 tools/prof_step.py 8192 10 > "$OUT/pmc_stream_INSTS.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_stream_b8192_SQ_CYCLES" -- python3 tools/prof_step.py 8192 10 > "$OUT/pmc_stream_CYCLES.log" 2>&1 || exit 1
echo "step pmc done"
# ONLY_STEP=1: the fused train step's kernels changed, the pose-head / GEMM / cfg3 / cfg5 kernels did not (their summaries and
# traffic stamps stay valid)
if [ -n "$ONLY_STEP" ]; then echo "all done (step only)"; exit 0; fi
for B in 256 1024 8192 16384 65536; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_head_b${B}_$C" -- python3 tools/prof_kernels.py $B 5 > "$OUT/pmc_head_${B}_$C.log" 2>&1 || exit 1
  done
done
echo "head pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5_trace" -- python3 tools/bench_poseformer.py 32 5 > "$OUT/cfg5_trace.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg3_trace" -- python3 tools/bench_seq2seq.py 512 30 graph > "$OUT/cfg3_trace.log" 2>&1 || exit 1

# K16 at 21 024 x 2 496 x 832 (NT, NN, TN) beside the library: MFMA-busy cycles and the instruction mix, two separate passes
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_gemm_MFMA" -- python3 tools/gemm_pmc.py > "$OUT/pmc_gemm_MFMA.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_gemm_INSTS" -- python3 tools/gemm_pmc.py > "$OUT/pmc_gemm_INSTS.log" 2>&1 || exit 1
echo "gemm pmc done"
echo "all done"
