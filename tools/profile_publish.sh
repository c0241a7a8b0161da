#!/bin/bash
# Copy the distilled evidence of tools/profile_round.sh <round> from gpurun_out/prof_<round>/ into profiles/ (tracked).
set -eu
R=${1:-r01}
S=gpurun_out/prof_$R
cp $S/bench_default.json profiles/${R}_bench_infer.json
cp $S/${R}_infer_kernel_trace.txt profiles/${R}_infer_kernel_trace.txt
cp $S/${R}_collect.json profiles/${R}_pmc_collect.json
cp $S/conv_traffic.json profiles/conv_traffic.json
cp $S/mfma_busy.json profiles/mfma_busy.json
cp $S/layers.txt profiles/${R}_layers_infer.txt
cp $S/${R}_layers_traffic.txt profiles/${R}_layers_traffic.txt 2>/dev/null || true
cp $S/${R}_infer_sequence.txt profiles/${R}_infer_sequence.txt
cp $S/bench_train.json profiles/${R}_bench_train.json
cp $S/${R}_train_kernel_trace.txt profiles/${R}_train_kernel_trace.txt
cp $S/${R}_train_step_kernels.txt profiles/${R}_train_step_kernels.txt
cp $S/train_fwd_ops.txt profiles/${R}_train_fwd_ops.txt
cp $S/train_bwd_ops.txt profiles/${R}_train_bwd_ops.txt
cp $S/${R}_train_step_sequence.txt profiles/${R}_train_step_sequence.txt
cp $S/bench_train_l_fp8.json profiles/${R}_bench_train_l_fp8.json
cp $S/bench_train_l_bf16.json profiles/${R}_bench_train_l_bf16.json
cp $S/${R}_train_l_fp8_step_kernels.txt profiles/${R}_train_l_fp8_step_kernels.txt
cp $S/${R}_train_bs64_step_kernels.txt profiles/${R}_train_bs64_step_kernels.txt 2>/dev/null || true
cp $S/bench_frame.json profiles/${R}_bench_frame.json
cp $S/bench_export.json profiles/${R}_bench_export.json
cp $S/train_traffic.json profiles/train_traffic.json
cp $S/${R}_conv_bench_*.txt $S/${R}_mma8_pmc_*.txt profiles/
cp $S/${R}_bn_bench.txt profiles/ 2>/dev/null || true
# the frame pipeline's kernel table + one frame in start order (tools/probe/frame_trace.sh, run after the round)
cp gpurun_out/frame_kernels.txt profiles/${R}_frame_kernels.txt 2>/dev/null || true
cp gpurun_out/frame_sequence.txt profiles/${R}_frame_sequence.txt 2>/dev/null || true
ls -la profiles/
