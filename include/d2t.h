/*
 * libd2t -- C-ABI of the MI355X (gfx950) recognizer engine.
 *
 * Drop-in boundary for the doc2tex `modules.recognizers` forward pass.  The
 * reference is 100 % Python and has no FFI of its own; the entry
 * points below are what a ctypes binding for that path binds, one per reference
 * call site (citations relative to /root/reference/doc2tex/):
 *
 *   d2t_create / d2t_load_weight / d2t_finalize_weights
 *        <- Model.__init__            modules/build_model.py:8-34
 *           load_checkpoint           utils/model_utils.py:136-237 (state_dict names)
 *   d2t_encoder_shape / d2t_encode
 *        <- Model.forward_encoder     modules/build_model.py:36-43
 *           (ResNet.forward resnet.py:205-245, HybridEmbed.forward patchembed.py:115-141,
 *            ViTEncoderV3.forward vit_encoder.py:249-268, SeqModelingBuilder.forward
 *            recognizers/build_seq.py:42-83)
 *   d2t_decode_greedy
 *        <- TransformerPrediction.forward_greedy (eval)  prediction_head/tfm.py:119-143
 *   d2t_decode_attn_greedy
 *        <- Attention.forward_greedy seq2seq.py:224-331, AttentionV2.forward_greedy seq2seq_v2.py:176-293
 *   d2t_decode_beam
 *        <- TransformerPrediction.forward_beam tfm.py:145-186 + Beam tools/beam.py:38-140
 *   d2t_decode_attn_beam
 *        <- Attention.forward_beam seq2seq.py:83-222, AttentionV2.forward_beam seq2seq_v2.py:12-174
 *   d2t_train_forward / d2t_train_backward / d2t_train_grad (+ dropout, scheduled-sampling setters)
 *        <- Model.forward under module.train() and loss.backward(): forward_step / train_one_step
 *           engine/training.py:76-164 (tfm.py:103-118, seq2seq.py:224-331 with is_train)
 *   d2t_decode_greedy_async / d2t_decode_wait, d2t_set_reserved_blocks / d2t_set_decode_chains,
 *   d2t_decode_beam_batch / d2t_decode_attn_beam_batch
 *        -- serving extensions without a reference counterpart (cross-batch pipelining, batched beam search);
 *           their results equal the corresponding reference-shaped calls bit for bit
 *   d2t_op_*  -- single-kernel entry points used by the parity tests.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the boundary;
 *   - every pointer named *dev* / marked [device] is a raw HIP device pointer
 *     owned by the CALLER (e.g. a torch tensor's data_ptr()); the engine writes
 *     outputs in place and keeps its own packed copies of the weights;
 *   - all tensors fp32, token ids int64; arithmetic fp32 on the matrix cores, or (d2t_set_conv_precision) large
 *     convolutions / GEMMs as three bf16 MFMAs per product with fp32 accumulation (tokens bit-exact, logits within 1e-3);
 *   - every launch goes to the caller's hipStream_t (`stream`, may be NULL for
 *     the default stream); calls are asynchronous unless they return host
 *     scalars (d2t_decode_greedy's steps_out, d2t_decode_beam);
 *   - return value 0 = D2T_OK, otherwise an error code; d2t_last_error(ctx)
 *     gives the message.  The library never aborts and never falls back to
 *     the CPU;
 *   - a ctx is not thread-safe: one ctx per device per process.
 */
#ifndef D2T_H
#define D2T_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct d2t_ctx d2t_ctx;
typedef void* d2t_stream; /* hipStream_t */

enum {
  D2T_OK = 0,
  D2T_EINVAL = 1, /* bad argument / unsupported configuration */
  D2T_ENOMEM = 2, /* device allocation failed */
  D2T_EHIP = 3,   /* HIP runtime error */
  D2T_ESTATE = 4  /* call order (weights missing / not finalized) */
};

enum {
  D2T_ENC_RESNET = 0,        /* Feat=ResNet, Seq=None (+PositionalEncoding2D)          -> TFM decoder   */
  D2T_ENC_HYBRID_VIT = 1,    /* Seq=ViT over the ResNet backbone (HybridEmbed)                          */
  D2T_ENC_VGG_BILSTM = 2,    /* Feat=VGG,    height-mean, 2x BidirectionalLSTM          -> Attn decoder  */
  D2T_ENC_RESNET_BILSTM = 3  /* Feat=ResNet, height-mean, 2x BidirectionalLSTM                          */
};
enum { D2T_DEC_TFM = 0, D2T_DEC_ATTN = 1 };
/* d2t_config.attn_keys: which memory tokens the LSTM-attention decoder attends to / starts from */
enum {
  D2T_ATTN_KEYS_ALL_INIT_MEAN = 0,  /* Attention / AttentionV2 on a BiLSTM encoder (seq2seq.py:229-233)     */
  D2T_ATTN_KEYS_NOCLS_INIT_CLS = 1, /* AttentionV2 with seqmodel='TFM' (seq2seq_v2.py:182-199): shipped YAMLs */
  D2T_ATTN_KEYS_ALL_INIT_FIRST = 2  /* Attention (v1) on a non-BiLSTM encoder (seq2seq.py:234-236)          */
};

/* Activation codes of d2t_op_conv2d. */
enum { D2T_ACT_NONE = 0, D2T_ACT_RELU = 1, D2T_ACT_GELU = 2 };

enum { D2T_ATTN_CELL_LOCATION = 0, D2T_ATTN_CELL_BAHDANAU = 1 };
/* How the ViT encoder's position table meets a crop whose patch grid differs from max_dimension's
 * (create_vit_modeling, seq_modeling/vit_encoder.py:295-302):
 *   SINCOS_PREFIX   fix_embed: True -> ViTEncoderV3 (:229-268): frozen sincos table, flat prefix slice pos_embed[:, :N+1]
 *   LEARNED_INTERP  default -> ViTEncoder (:22-118): learned table, resized to the crop's grid by bicubic interpolation
 *                   (F.interpolate, align_corners False, scale factors (gh + 0.1) / GH, (gw + 0.1) / GW, :58-95); the table
 *                   itself when the grids agree (or the token counts agree and the padded feature map is square, :66-67)
 *   LEARNED_PREFIX  interpolate_embed: False -> ViTEncoderV2 (:207-226): learned table, flat prefix slice
 * The first and the last are the same arithmetic in inference; in training the learned tables receive gradients. */
enum { D2T_VIT_POS_SINCOS_PREFIX = 0, D2T_VIT_POS_LEARNED_INTERP = 1, D2T_VIT_POS_LEARNED_PREFIX = 2 };

typedef struct d2t_config {
  int32_t encoder;      /* D2T_ENC_*: Feat=ResNet+Seq=None  |  Seq=ViT (hybrid) */
  int32_t in_channels;  /* 1 (grey crops) */
  int32_t backbone_out; /* ResNet output_channel, 512 */
  int32_t vit_depth, vit_heads, vit_dim; /* ViT blocks / heads / hidden_size */
  int32_t patch_h, patch_w;              /* patch_size */
  int32_t max_h, max_w;                  /* max_dimension: sizes the sincos pos-embed grid */
  int32_t dec_dim, dec_heads, dec_layers, dec_ff; /* TFM d_model / nhead / layers / dim_feedforward */
  int32_t vocab;        /* num_class */
  int32_t max_seq_len;  /* Prediction.params.max_seq_len */
  /* appended in v2 of the struct: recurrent model family */
  int32_t decoder;          /* D2T_DEC_* */
  int32_t attn_hidden;      /* Attn hidden_size (= input_size = embed dim), 256 */
  int32_t attn_kernel_size; /* location filter half width: Conv1d kernel = 2*k+1 */
  int32_t attn_kernel_dim;  /* location filter channels */
  int32_t attn_keys;        /* D2T_ATTN_KEYS_* */
  int32_t attn_enc_init;    /* enc_init: initial (h,c) projected from the encoder output */
  int32_t attn_coverage;    /* 1 = attn_type 'coverage' (accumulated alignment), 0 = 'loc_aware' */
  int32_t bilstm_hidden;    /* SequenceModeling.params.hidden_size of the BiLSTM, 256 */
  int32_t batch_max_length; /* Attn decoders run batch_max_length + 1 steps */
  int32_t gcb;              /* 1: GlobalContext blocks close the four ResNet stages (gcb: True) */
  /* appended in v3: the other cells / inputs of Attention.__init__ (prediction_head/seq2seq.py:31-53) */
  int32_t attn_cell;        /* D2T_ATTN_CELL_*: location-aware (attn_type 'coverage' | 'loc_aware') or Bahdanau
                               (attention1D.py:71-118: keys "attn.i2h / attn.h2h / attn.score", no alignment memory) */
  int32_t attn_onehot;      /* 1 = embed_target False: the decoder input is the one-hot vector of the previous token
                               (seq2seq.py:72-78), rnn.weight_ih is [4H][H + num_class] and there is no embedding table */
  /* appended in v4: the ViT encoder variants beside ViTEncoderV3 */
  int32_t vit_pos;          /* D2T_VIT_POS_* */
} d2t_config;

/* ---- lifecycle ----------------------------------------------------------
 * d2t_create binds the context to the HIP device that is current for the calling thread (the reference picks its
 * device from the string opt["device"], build_pred.py:17 -- the Python layer above translates "cuda:N" into
 * hipSetDevice(N) around this call).  Every stream, event and buffer of the context lives on that device and every
 * later call switches to it on entry and restores the caller's device on return, so two contexts on two GPUs can be
 * driven from one thread.  Caller buffers on another device are rejected with D2T_EINVAL. */
int d2t_create(const d2t_config* cfg, d2t_ctx** out);
void d2t_destroy(d2t_ctx* ctx);
const char* d2t_last_error(const d2t_ctx* ctx);
/* HIP device index the context was created on (-1 for a null ctx). */
int d2t_device_of(const d2t_ctx* ctx);
/* 1 if a HIP device is usable from this process, else 0 (no ctx needed). */
int d2t_device_available(void);

/* ---- weights -------------------------------------------------------------
 * One call per state_dict entry, `name` being the reference's key (e.g.
 * "seqmodeler.SequenceModeling.blocks.0.attn.qkv.weight").  `dev` is fp32
 * [device], contiguous, `shape[ndim]` its torch shape.  Integer entries
 * (num_batches_tracked) and tables the engine rebuilds itself are ignored.
 * d2t_finalize_weights folds eval-BatchNorm into the convolutions, packs
 * everything into the engine's layouts and fails if an entry is missing.  It
 * may be called again after further d2t_load_weight calls (re-pack). */
int d2t_load_weight(d2t_ctx* ctx, const char* name, const float* dev, const int64_t* shape, int32_t ndim,
                    d2t_stream stream);
int d2t_finalize_weights(d2t_ctx* ctx, d2t_stream stream);

/* ---- encoder -------------------------------------------------------------
 * d2t_encoder_shape: host-only; token count T, feature dim d, patch grid
 * (grid_h, grid_w; the backbone feature map size for the ResNet encoder) and
 * HybridEmbed's (pad_w, pad_h) for an H x W crop.
 * d2t_encode: image [B,1,H,W] fp32 in [-1,1] [device] -> memory [B,T,d]
 * [device, caller-allocated]. */
int d2t_encoder_shape(const d2t_ctx* ctx, int32_t H, int32_t W, int32_t* T, int32_t* d, int32_t* grid_h,
                      int32_t* grid_w, int32_t* pad_w, int32_t* pad_h);
int d2t_encode(d2t_ctx* ctx, const float* image_dev, int32_t B, int32_t H, int32_t W, float* memory_dev,
               d2t_stream stream);

/* ---- greedy decode -------------------------------------------------------
 * memory [B,T,d]; start_tokens [B] int64 ([GO]).  Runs up to max_seq_len+1
 * steps with a KV cache; rows that emitted [s] keep generating until the whole
 * batch has ended (reference semantics).  If is_test != 0 the step count is
 * the first step at which every row has emitted [s] (tfm.py:138-140),
 * otherwise max_seq_len+1.
 * tokens_dev [B, max_seq_len+1] int64 and logits_dev [B, max_seq_len+1, vocab]
 * fp32 are written for steps [0, *steps_out); entries beyond are unspecified.
 * Synchronises `stream` before returning (steps_out is a host scalar). */
int d2t_decode_greedy(d2t_ctx* ctx, const float* memory_dev, int32_t B, int32_t T, const int64_t* start_tokens_dev,
                      int32_t is_test, int64_t* tokens_dev, float* logits_dev, int32_t* steps_out,
                      d2t_stream stream);

/* ---- LSTM-attention greedy decode (Attn / Attnv2 heads, eval mode) ----------
 * Attention.forward_greedy (prediction_head/seq2seq.py:224-331), AttentionV2.forward_greedy
 * (seq2seq_v2.py:176-293) with is_train = False.  memory [B,T,256]; runs batch_max_length+1 steps in one
 * launch.  tokens_dev [B,S] int64, probs_dev [B,S,V] fp32 (S = batch_max_length+1) are always full size;
 * with is_test != 0 everything after the first step at which all rows emitted [s] is zeroed, as the
 * reference leaves its pre-zeroed `probs` untouched after the early break. */
int d2t_decode_attn_greedy(d2t_ctx* ctx, const float* memory_dev, int32_t B, int32_t T, int32_t is_test,
                           int64_t* tokens_dev, float* probs_dev, int32_t* steps_out, d2t_stream stream);

/* Pipelined variant: always runs max_seq_len+1 steps (is_test = 0 semantics) and returns as soon as the
 * work is enqueued, so the caller can start encoding the next batch while this one decodes (the decode
 * loop is latency-bound and leaves most CUs idle).  tokens_dev / logits_dev / start_tokens_dev must stay
 * alive and untouched until d2t_decode_wait: it makes `stream` wait for every outstanding decode
 * (and, if host_sync != 0, blocks the host until they are complete). */
/* Decode groups: B need not be one encoder batch.  A serving loop may collect the memories of several consecutive batches
 * (same T) in one buffer and decode their rows with ONE call -- the step loop's duration hardly depends on the row count,
 * and every row's result is bit-identical to its single-batch decode (doc2tex_amd.Model.decode_group does exactly this). */
int d2t_decode_greedy_async(d2t_ctx* ctx, const float* memory_dev, int32_t B, int32_t T,
                            const int64_t* start_tokens_dev, int64_t* tokens_dev, float* logits_dev, d2t_stream stream);
int d2t_decode_wait(d2t_ctx* ctx, d2t_stream stream, int32_t host_sync);
/* Serving tickets.  Every d2t_decode_greedy_async call is numbered 1, 2, 3, ... (d2t_decode_last_ticket returns the
 * number of the most recent one, 0 before the first).  d2t_decode_wait_ticket orders `stream` (and the host, if
 * host_sync != 0) after THAT decode only -- a consumer of batch i need not wait for batches i+1, i+2 that are already
 * in flight; d2t_decode_query polls: 1 complete, 0 still running, < 0 error (-D2T_E*).  The output buffers of a decode
 * belong to the engine until its ticket is complete: a caller that recycles buffers checks the ticket first. */
/* The general asynchronous entry point: as d2t_decode_greedy_async, plus
 *   is_test != 0       the reference's early exit (tfm.py:138-140) decided ON THE DEVICE: the whole step loop is still one
 *                      graph launch, but once every batch of the group has emitted [s] in all its rows the remaining
 *                      kernels return at their first instruction; d2t_decode_steps then reports, per batch, the first step
 *                      at which all ITS rows had ended (entries of tokens / logits beyond it are unspecified);
 *   rows_per_batch     > 0: the B rows are B / rows_per_batch encoder batches of that many rows (a decode group, at most
 *                      64 batches); 0: one batch.
 * *ticket_out (optional) receives the decode's ticket. */
int d2t_decode_greedy_submit(d2t_ctx* ctx, const float* memory_dev, int32_t B, int32_t T, const int64_t* start_tokens_dev,
                             int32_t is_test, int32_t rows_per_batch, int64_t* tokens_dev, float* logits_dev,
                             d2t_stream stream, int64_t* ticket_out);
/* Step counts of the batches of one asynchronous decode (blocks until it is complete). */
int d2t_decode_steps(d2t_ctx* ctx, int64_t ticket, int32_t* steps_out, int32_t max_batches, int32_t* n_out);
int64_t d2t_decode_last_ticket(const d2t_ctx* ctx);
int d2t_decode_query(d2t_ctx* ctx, int64_t ticket);
int d2t_decode_wait_ticket(d2t_ctx* ctx, int64_t ticket, d2t_stream stream, int32_t host_sync);

/* ---- beam decode (one sample, fresh beam state per call) ------------------
 * memory [1,T,d].  seq_out: HOST buffer of max_seq_len+1 int64; *len_out its
 * used length; *score_out the hypothesis score (tools/beam.py semantics:
 * best = argmax score/len over completed hypotheses).
 * d_model 256 (round 4): Beam.advance (tools/beam.py:68-105) runs on the device -- candidate walk, completed set, row
 * compaction and a (parent, token) history per step in one single-block kernel -- and the whole max_seq_len + 1 step loop is
 * ONE captured graph per (N, T, beam); the host copies one result block back and walks the best hypothesis through the
 * history.  d2t_decode_beam is that loop with N = 1 (the same row kernels as d2t_decode_beam_batch: bit-identical).  Other
 * decoders (d_model 512) and d2t_set_beam_shared_tile(1) keep the host-side loop with one round trip per step. */
int d2t_decode_beam(d2t_ctx* ctx, const float* memory_dev, int32_t T, int32_t beam_size, int64_t* seq_out,
                    int32_t* len_out, float* score_out, d2t_stream stream);

/* d2t_decode_attn_beam for N samples in one step loop (extension, like d2t_decode_beam_batch).  memory [N][T][256];
 * seq_out (host) [N][batch_max_length + 1]; len_out, score_out (host) [N]. */
int d2t_decode_attn_beam_batch(d2t_ctx* ctx, const float* memory, int32_t N, int32_t T, int32_t beam_size, int64_t* seq_out,
                               int32_t* len_out, float* score_out, d2t_stream stream);

/* Beam search for N samples at once (not in the reference, whose forward_beam is single-sample): the same results as
 * N calls of d2t_decode_beam, with the hypotheses of all samples advanced by one shared step loop.  memory [N][T][d];
 * seq_out (host) [N][max_seq_len + 1]; len_out, score_out (host) [N]. */
int d2t_decode_beam_batch(d2t_ctx* ctx, const float* memory, int32_t N, int32_t T, int32_t beam_size, int64_t* seq_out,
                          int32_t* len_out, float* score_out, d2t_stream stream);

/* LSTM-attention beam search for ONE sample (reference: Attention.forward_beam, prediction_head/seq2seq.py:83-222;
 * AttentionV2.forward_beam, seq2seq_v2.py:12-174 -- what config/test.yaml runs with beam_size 5 / 10).  Coverage
 * attention only.  memory [1][T][256]; seq_out (host, >= batch_max_length + 1 entries) receives the token ids without
 * the leading [GO]; *score_out the reference's returned score.  1 <= beam_size <= 16. */
int d2t_decode_attn_beam(d2t_ctx* ctx, const float* memory, int32_t T, int32_t beam_size, int64_t* seq_out,
                         int32_t* len_out, float* score_out, d2t_stream stream);

/* ---- convolution arithmetic ------------------------------------------------
 * D2T_CONV_FP32   exact fp32 on v_mfma_f32_32x32x2_f32.
 * D2T_CONV_BF16X3 (default of d2t_create, as of doc2tex_amd.Model) backbone / patch-embed convolutions on the bf16 matrix cores with each fp32 operand
 *                 split into two bf16 (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate): ~2^-17 relative
 *                 product error, measured logits error 2.4e-5 against the 1e-3 budget.  Takes effect at the
 *                 next d2t_encode.
 * D2T_CONV_FP16X2 (opt-in) the backbone's feature maps are kept as fp16 (11 significant bits, rounded once where a layer stores
 *                 them) and its split-record convolutions multiply them with the weights split into fp16 hi + fp16 lo
 *                 (x*w_lo + x*w_hi, fp32 accumulate): two matrix instructions per product instead of three.  Greedy tokens
 *                 exact and |dlogit| <= 1e-3 on the reference fixtures of the HybridViT / LSTM-head stacks (measured
 *                 1.2e-4 .. 2.1e-4 on the benchmark configs), with a 5x instead of a 20x margin -- and NOT on the ResNet-only
 *                 stacks (up to 9e-3): DESIGN.md section 3.  Everything that takes fp32 input (ViT linears, decoder) is as in
 *                 D2T_CONV_BF16X3.  Needs conv kernel 3 (d2t_set_conv_kernel).
 * D2T_CONV_MIXED  (round 4, opt-in) D2T_CONV_BF16X3 with the two-MFMA fp16 arithmetic of D2T_CONV_FP16X2 in the first
 *                 D2T_MIXED_UNITS_DEFAULT of the backbone's eight plain 512 -> 512 units [layer3.1, layer3.2, layer3.3,
 *                 layer3.4, conv3, layer4.0, layer4.1, layer4.2] (resnet.py:32-48, 205-245; the K = 4608 layers) and nothing
 *                 else.  The tensors entering / leaving those units are stored as fp16 hi | fp16 lo pairs (22 bits), so the
 *                 residual stream keeps its precision and only the MFMA's activation operand is rounded to 11 bits.  The error
 *                 of the logits grows with the square root of the number of two-MFMA layers (DESIGN.md section 3,
 *                 tools/probe/fp16x2_sim.py); d2t_set_mixed_units changes the count (0 .. 8; 0 = D2T_CONV_BF16X3). */
enum { D2T_CONV_FP32 = 0, D2T_CONV_BF16X3 = 1, D2T_CONV_FP16X2 = 2, D2T_CONV_MIXED = 3 };
#define D2T_MIXED_UNITS_DEFAULT 3
int d2t_set_conv_precision(d2t_ctx* ctx, int32_t mode);
int d2t_set_mixed_units(d2t_ctx* ctx, int32_t units);

/* Pipelined serving: the split-bf16 convolution is a persistent kernel; capping its grid at
 * (2 x CUs - blocks) leaves `blocks` block slots free at all times, into which the latency-bound decode
 * kernels of the previous batch (running on the engine's high-priority stream) are placed without waiting
 * for a convolution block to retire.  0 (default) = use every slot. */
int d2t_set_reserved_blocks(d2t_ctx* ctx, int32_t blocks);

/* The split-bf16 convolution kernel of the layers with at least 128 output channels.  kind 3 (default): the pipelined
 * 256x128 kernel on v_mfma_f32_16x16x32_bf16 -- ONE persistent block per CU holding three LDS stages, LDS-DMA two K-steps
 * ahead behind counted waits, grid = (CUs - d2t_set_reserved_cus) -- with the fused max-pools / shortcuts and the fp16x2 /
 * mixed arithmetic built on it.  kind 0: the 128x128 kernel on 32x32x16 MFMAs with two blocks per CU (the kernel of the
 * narrower layers; d2t_set_reserved_blocks applies to it).  Same three products per element in the same order in both; the
 * sum inside an MFMA spans 32 k in kind 3 and 16 in kind 0, so the two agree to fp32 rounding, not bit for bit; all rows of
 * a layer always run on one kind.  (Rounds 1-3 exported five more variants here; they were A/B arms and left in round 4.) */
int d2t_set_conv_kernel(d2t_ctx* ctx, int32_t kind);
/* Validation switches (tests): pools = 0 runs the two 2x2 max-pools as their own kernels instead of inside the epilogue of the
 * convolution in front of them, shortcuts = 0 the BasicBlocks' 1x1 shortcuts as their own launches instead of inside conv2's.
 * Pooled values are bit-identical either way; the shortcut sum forms in another order (fp32 rounding).  Default: both fused. */
int d2t_set_conv_fusion(d2t_ctx* ctx, int32_t pools, int32_t shortcuts);
/* Beam search of the d_model-256 TFM decoder, cross-attention: 0 (default) one block per hypothesis row, each reading its
 * sample's memory rows; 1: one block per SAMPLE that stages the sample's memory tiles in LDS once per layer and step for
 * all its live hypotheses (beam <= 6; decode.hip beam_cross_kernel).  Same results to fp32 summation order. */
int d2t_set_beam_shared_tile(d2t_ctx* ctx, int32_t on);
int d2t_set_reserved_cus(d2t_ctx* ctx, int32_t cus);
/* Number of decode chains (1 .. 4, default 1) d2t_decode_greedy_async alternates between.  Each chain has
 * its own stream, self-attention cache and workspace, so with 2 the step loops of two consecutive batches
 * run side by side (the loop is a dependent chain of small kernels that cannot fill the chip alone). */
int d2t_set_decode_chains(d2t_ctx* ctx, int32_t chains);

/* ---- training step (engine/training.py:76-164) ----------
 * d2t_train_forward runs Model.forward under module.train() for the HybridViT + TFM stack: BatchNorm on batch
 * statistics (the engine's copies of running_mean / running_var are updated with momentum 0.1; read them back
 * with d2t_read_weight), teacher-forced decoder pass over tgt [B][L] (= text[:, :-1]; PAD = 0 keys masked,
 * causal), dropout 0.  logits [B][L][vocab] is caller memory.  Weights are the ones last given to
 * d2t_load_weight (no finalize needed).  d2t_train_backward takes dL/dlogits [B][L][vocab] and leaves the
 * gradient of every trainable parameter in engine memory; d2t_train_grad copies one out by its state_dict key.
 * d2t_train_grad may be given ANY stream: the copy is ordered (by an event) after the last backward kernel that
 * writes that gradient, so a communication stream can pick up early gradients (decoder, ViT, deep backbone layers)
 * and all-reduce them while the rest of the backward still runs on the compute stream.
 * One forward may be followed by at most one backward.  All pointers are device pointers. */
int d2t_train_forward(d2t_ctx* ctx, const float* image, int32_t B, int32_t H, int32_t W, const int64_t* tgt, int32_t L,
                      float* logits, d2t_stream stream);
int d2t_train_backward(d2t_ctx* ctx, const float* dlogits, d2t_stream stream);
int d2t_train_grad(d2t_ctx* ctx, const char* name, float* dst, int64_t numel, d2t_stream stream);
/* All of them at once (the single-process step: what loss.backward() leaves in .grad): gradient `names[i]` is written to
 * flat[offsets[i] .. offsets[i] + numels[i]) by ONE kernel ordered after the whole backward, instead of one device copy per
 * parameter (378 of them for the headline model: ~3 ms of 4-us copies).  source = 0: gradients; 1: the engine's current
 * copy of loaded tensors (the BatchNorm running statistics after a training forward, as d2t_read_weight). */
int d2t_train_gather(d2t_ctx* ctx, int32_t source, int32_t n, const char* const* names, const int64_t* offsets,
                     const int64_t* numels, float* flat, d2t_stream stream);
/* The other direction, after optimizer.step(): refresh the engine's copies of tensors that are ALREADY loaded (same names and
 * sizes as the d2t_load_weight calls that created them) from device memory, in one kernel.  Folded / packed inference
 * weights are not rebuilt: call d2t_finalize_weights before the next inference forward (the training step reads the raw copies). */
int d2t_reload_weights(d2t_ctx* ctx, int32_t n, const char* const* names, const float* const* srcs, const int64_t* numels,
                       d2t_stream stream);
/* copy the engine's current copy of a loaded tensor (e.g. BatchNorm running statistics after a training forward) */
int d2t_read_weight(d2t_ctx* ctx, const char* name, float* dst, int64_t numel, d2t_stream stream);
/* Dropout of nn.TransformerDecoderLayer(dropout=p) in the training step: on the attention probabilities of both
 * attentions, after each of the three sub-layers (dropout1/2/3) and inside the feed-forward block.  Keep masks are
 * Philox4x32-10 draws keyed by (seed, number of training forwards since the seed was set, site index): statistically
 * the same as torch's dropout, not the same random stream.  p = 0 (default) disables it.  The masks of the last
 * forward can be read back (creation order = the order torch draws them) for verification. */
int d2t_train_set_dropout(d2t_ctx* ctx, float p, uint64_t seed);
/* LSTM-attention head only.  (a) d2t_train_set_dropout's p is the head's `droprate`: dropout on the generator output
 * of every step (seq2seq.py:298), one [B][S][V] mask.  (b) Scheduled sampling (seq2seq.py:311-316): flags[t] (host
 * bytes, n = batch_max_length + 1 of them; flags[0] is ignored) says whether step t is fed the label token (1) or the
 * arg-max of step t-1's output (0).  n = 0 restores "always the label".  The flags apply to the following forwards. */
int d2t_train_set_teacher_flags(d2t_ctx* ctx, const uint8_t* flags, int32_t n);
int d2t_train_mask_count(d2t_ctx* ctx);
/* The discrete decisions of the last d2t_train_forward, in network order: one entry per ReLU (keep mask, y > 0, one byte per
 * element of the [rows][cols] output) and per max-pool (the window element kh*2+kw that holds the first maximum, one byte
 * per NHWC output element).  Test instrument: replayed in the CPU oracle they pin both sides to the same smooth function.
 * dst may be NULL to query *is_pool_out / *numel_out only. */
int d2t_train_decision_count(d2t_ctx* ctx);
int d2t_train_read_decision(d2t_ctx* ctx, int32_t index, uint8_t* dst, int64_t numel, int32_t* is_pool_out,
                            int64_t* numel_out, d2t_stream stream);
int d2t_train_read_mask(d2t_ctx* ctx, int32_t index, uint8_t* dst, int64_t numel, d2t_stream stream);
/* free the training tape, gradient buffers and workspace */
void d2t_train_release(d2t_ctx* ctx);

/* ---- in-engine kernel timing (bench.py roofline leg) ------------------------
 * While enabled, d2t_encode brackets every implicit-GEMM (MFMA) launch with a
 * pair of HIP events on the launch stream.  d2t_profile_read synchronises,
 * returns up to max_records launches in issue order as GEMM shape (M,N,K) and
 * elapsed milliseconds, and clears the log.  *n receives the number written. */
int d2t_profile_enable(d2t_ctx* ctx, int32_t on);
int d2t_profile_read(d2t_ctx* ctx, int32_t max_records, int32_t* n, int32_t* M, int32_t* N, int32_t* K, float* ms);

/* ---- single-kernel entry points (parity tests) ----------------------------
 * All tensors [device] fp32.  Activations are NHWC / row-major [rows, features].
 */
/* y = act(conv2d(x, w) + bias + residual).  x [B,H,W,Cin] NHWC, w [Cout,KH,KW,Cin]
 * (OHWI), bias [Cout] or NULL, residual [B,OH,OW,Cout] or NULL, y [B,OH,OW,Cout].
 * Out-of-range taps read as zero.  Cin must be 1 (direct kernel, 3x3 pad 1 only)
 * or a multiple of 32 (MFMA implicit GEMM). */
int d2t_op_conv2d(const float* x, const float* w, const float* bias, const float* residual, float* y, int32_t B,
                  int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW,
                  int32_t PH, int32_t PW, int32_t act, d2t_stream stream);
/* Same contract on the bf16x3 kernel (Cin % 32 == 0); splits w internally (test entry point, synchronous). */
int d2t_op_conv2d_bf16x3(const float* x, const float* w, const float* bias, const float* residual, float* y, int32_t B,
                         int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t SH,
                         int32_t SW, int32_t PH, int32_t PW, int32_t act, d2t_stream stream);
/* Same again on the split-activation kernel: x, residual and y cross the kernel boundary as bf16 hi/lo
 * planes (split / merged around the launch by this test entry point). */
/* which split-bf16 kernel d2t_op_conv2d_bf16x3_split launches (d2t_set_conv_kernel / d2t_set_reserved_cus of a context,
 * but process-wide: tests and tools only) */
int d2t_op_set_conv_kernel(int32_t kind, int32_t reserved_cus);
/* the same with the 2x2 / stride 2 max-pool that follows fused into the epilogue (ConvP::pool2; no residual): y is the pooled
 * map [B][OH/2][OW/2][Cout].  Layers with <= 64 or >= 128 output channels (the two kernels that carry the fused form). */
int d2t_op_conv2d_bf16x3_split_pool(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t H, int32_t W,
                                    int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t PH,
                                    int32_t PW, int32_t act, d2t_stream stream);
int d2t_op_conv2d_bf16x3_split(const float* x, const float* w, const float* bias, const float* residual, float* y,
                               int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW,
                               int32_t SH, int32_t SW, int32_t PH, int32_t PW, int32_t act, d2t_stream stream);
/* y[M,N] = act(x[M,K] @ w[N,K]^T + bias + residual); any M (skinny path for M<=64). */
int d2t_op_linear(const float* x, const float* w, const float* bias, const float* residual, float* y, int32_t M,
                  int32_t K, int32_t N, int32_t act, d2t_stream stream);
/* 2x2 max-pool, NHWC, stride (SH,SW), padding (PH,PW) with -inf semantics. */
int d2t_op_maxpool2x2(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t SH, int32_t SW,
                      int32_t PH, int32_t PW, d2t_stream stream);
/* y = LayerNorm(x) * gamma + beta over the last dim (D = 256 or 512). */
int d2t_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows, int32_t D,
                     float eps, d2t_stream stream);
/* ViT self-attention.  qkv [B,N,3,heads,32] -> y [B,N,heads*32]; softmax(q k^T / sqrt(32)) v. */
int d2t_op_vit_attention(const float* qkv, float* y, int32_t B, int32_t N, int32_t heads, d2t_stream stream);
/* Single-query attention (decoder step).  q [B,heads*hd]; k,v [B,heads,Lmax,hd];
 * attends over the first L keys; y [B,heads*hd].  hd = 32 or 64. */
int d2t_op_decode_attention(const float* q, const float* k, const float* v, float* y, int32_t B, int32_t heads,
                            int32_t hd, int32_t L, int32_t Lmax, d2t_stream stream);

/* ---- fused cross-entropy ------------------------------------------------------------------------------------------
 * The criterion of the reference's training step -- nn.CrossEntropyLoss(ignore_index = PAD, reduction = 'none') on
 * preds.view(-1, V) / target.view(-1) (engine/training.py:50-53, 83, 90; modules/loss/builder.py:18-24) -- as one kernel
 * each way instead of torch's log_softmax + nll_loss pair.  logits [rows][V] fp32, target [rows] int64; loss [rows] and
 * lse [rows] (log-sum-exp, kept for the backward) are written; rows whose target equals ignore_index give loss 0 and a
 * zero gradient.  dloss [rows] is the incoming gradient of the per-row losses (1 / rows for the reference's .mean()). */
int d2t_ce_forward(const float* logits, const int64_t* target, float* loss, float* lse, int32_t rows, int32_t V,
                   int64_t ignore_index, d2t_stream stream);
int d2t_ce_backward(const float* logits, const int64_t* target, const float* lse, const float* dloss, float* dlogits,
                    int32_t rows, int32_t V, int64_t ignore_index, d2t_stream stream);

/* ---- op-level TRAINING test entry points ---------------------------------------------------------------------
 * One node of the training tape, forward + backward, on caller tensors (fp32, device, row-major [rows][cols]; maps
 * NHWC; convolution weights OIHW as in the state_dict).  They run the very builders / backward functions of
 * d2t_train_forward / d2t_train_backward, so the backward kernels can be checked one op at a time.  Output pointers may
 * be NULL.  bf16x3 != 0: split-bf16 arithmetic for the convolution / data-gradient / weight-gradient GEMMs.
 * Synchronous (the call returns after the stream has drained). */
/* y = [relu]( BN_batchstats( conv(x, w) ) [+ residual] )  (gamma != NULL)   or   conv(x, w) + bias  (gamma == NULL).
 * Cin == 1 selects the stem (3x3, pad 1, BN + ReLU, 32 output channels: resnet.py:205-207). */
int d2t_op_train_conv(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                      const float* residual, const float* dy, float* y, float* dx, float* dw, float* dbias,
                      float* dgamma, float* dbeta, float* dres, int32_t B, int32_t H, int32_t W, int32_t Cin,
                      int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t PH, int32_t PW, int32_t relu,
                      int32_t bf16x3, d2t_stream stream);
/* y = [relu](x @ w^T + bias) [+ residual];  x [M][K], w [N][K] */
int d2t_op_train_linear(const float* x, const float* w, const float* bias, const float* residual, const float* dy, float* y,
                        float* dx, float* dw, float* dbias, float* dres, int32_t M, int32_t K, int32_t N, int32_t relu,
                        int32_t bf16x3, d2t_stream stream);
int d2t_op_train_layernorm(const float* x, const float* gamma, const float* beta, const float* dy, float* y, float* dx,
                           float* dgamma, float* dbeta, int32_t rows, int32_t D, float eps, d2t_stream stream);
/* nn.MultiheadAttention core: q [nb*Lq][heads*hd], kv [nb*Lk][2*heads*hd] (keys | values); keytok (optional) [nb][Lk]
 * token ids whose PAD (0) entries are masked (tgt_key_padding_mask); causal != 0 adds the causal mask. */
int d2t_op_train_attention(const float* q, const float* kv, const int64_t* keytok, const float* dy, float* y, float* dq,
                           float* dkv, int32_t nb, int32_t Lq, int32_t Lk, int32_t heads, int32_t hd, int32_t causal,
                           d2t_stream stream);
int d2t_op_train_maxpool(const float* x, const float* dy, float* y, float* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                         int32_t SH, int32_t SW, int32_t PH, int32_t PW, d2t_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* D2T_H */
