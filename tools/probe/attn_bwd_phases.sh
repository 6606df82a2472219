# time of the attention backward kernel with phases skipped (inside gpurun): bit 0 = dV, bit 1 = dS / dQ, bit 2 = dK
for a in 0 1 2 4 7; do echo "probe $a"; D2T_ATTN_BWD_PROBE=$a bash tools/profile_train.sh attnp$a 32 2 | grep "attn_train_bwd" | sed "s/(d2t::AttnTrainP)//"; done
