#!/bin/bash
# Knock-out profile of the training step (aids build: DVGR_SKIP / DVGR_SKIP_TAGS, csrc/dvgr_common.h): the step is timed with ONE
# class of launches not enqueued at a time -- the difference to the full step is what that class costs IN the step (its place on
# the critical path and its share of the chip), which the sum of kernel durations does not say.  Results of such runs are garbage.
#   tools/knockout.sh [bench args]           rows -> profiles/experiments_r03.tsv (experiment knockout_<config>)
A=dualvgr-videoqa_amd/libdvgr_hip_aids.so
NAME=knockout_${KNOCKOUT_NAME:-c2}
python tools/ab.py $NAME --rounds 2 --bench "$*" \
  full:lib=$A \
  no_S2_fwd_gemm:lib=$A:DVGR_SKIP_TAGS=2 \
  no_S4_fwd_gemm:lib=$A:DVGR_SKIP_TAGS=4 \
  no_weight_grad_gemm:lib=$A:DVGR_SKIP_TAGS=8 \
  no_dX_gemm:lib=$A:DVGR_SKIP_TAGS=16 \
  no_dz_gemm:lib=$A:DVGR_SKIP_TAGS=64 \
  no_generic_gemms:lib=$A:DVGR_SKIP_TAGS=1 \
  no_attn_fwd:lib=$A:DVGR_SKIP=k_gat_attn_fwd \
  no_attn_bwd:lib=$A:DVGR_SKIP=k_gat_attn_bwd \
  no_sf_combine_fwd:lib=$A:DVGR_SKIP=k_sf_combine_fwd \
  no_sf_bwd:lib=$A:DVGR_SKIP=k_sf_bwd \
  no_punish_bwd:lib=$A:DVGR_SKIP=k_punish_bwd \
  no_colsum:lib=$A:DVGR_SKIP=k_colsum \
  no_aux_loss:lib=$A:DVGR_SKIP=k_aux_ \
  no_query_rows:lib=$A:DVGR_SKIP=k_qa_ \
  no_small_gemms:lib=$A:DVGR_SKIP=k_small_ \
  no_split_planes:lib=$A:DVGR_SKIP=k_split_planes \
  no_tiny_kernels:lib=$A:DVGR_SKIP=k_mfb_pool,k_elu_bwd,k_zero_ranges,k_sum_and_ones,k_punish_fwd,k_zero_outputs
