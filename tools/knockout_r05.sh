#!/bin/bash
# Knock-out profile of the round-5 step (see tools/knockout.sh): the launch classes as they are since the fused forward and
# backward GAT-view kernels.  Results of such runs are garbage; only their timing is read.
#   tools/knockout_r05.sh [bench args]        rows -> profiles/experiments_r05.tsv (experiment knockout_r05_<config>)
A=dualvgr-videoqa_amd/libdvgr_hip_aids.so
NAME=knockout_r05_${KNOCKOUT_NAME:-c2}
python tools/ab.py $NAME --rounds 2 --bench "$*" \
  full:lib=$A \
  no_gat_fused_fwd:lib=$A:DVGR_SKIP=k_gat_fused_fwd \
  no_gat_fused_bwd:lib=$A:DVGR_SKIP=k_gat_fused_bwd \
  no_S2_layer0_gemm:lib=$A:DVGR_SKIP_TAGS=2 \
  no_S4_fwd_gemm:lib=$A:DVGR_SKIP_TAGS=4 \
  no_weight_grad_gemm:lib=$A:DVGR_SKIP_TAGS=8 \
  no_dX_gemm:lib=$A:DVGR_SKIP_TAGS=16 \
  no_generic_gemms:lib=$A:DVGR_SKIP_TAGS=1 \
  no_attn_fwd_layer0:lib=$A:DVGR_SKIP=k_gat_attn_fwd \
  no_sf_combine_fwd:lib=$A:DVGR_SKIP=k_sf_combine_fwd \
  no_sf_bwd:lib=$A:DVGR_SKIP=k_sf_bwd \
  no_punish_bwd:lib=$A:DVGR_SKIP=k_punish_bwd \
  no_colsum:lib=$A:DVGR_SKIP=k_colsum \
  no_aux_loss:lib=$A:DVGR_SKIP=k_aux_ \
  no_query_rows:lib=$A:DVGR_SKIP=k_qa_ \
  no_small_gemms:lib=$A:DVGR_SKIP=k_small_ \
  no_split_planes:lib=$A:DVGR_SKIP=k_split_planes \
  no_mfb:lib=$A:DVGR_SKIP=k_mfb_ \
  no_tiny_kernels:lib=$A:DVGR_SKIP=k_bwd_head,k_zero_ranges,k_punish_fwd,k_zero_outputs
