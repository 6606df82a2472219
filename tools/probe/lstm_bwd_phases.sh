# step time of the S0 training step with phases of attn_train_lstm_bwd_kernel skipped (inside gpurun; WRONG gradients):
# bit 1 generator^T, 2 LSTMCell input products, 4 context backward, 8 score backward, 16 query projection
for a in 0 1 2 4 8 16 31; do echo -n "probe $a: "; D2T_LSTM_BWD_PROBE=$a python tools/train_bench.py 32 3 bf16x3 S0 2>&1 | tail -1 | cut -c1-60; done
