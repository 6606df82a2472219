"""Drop-in `Model(opt)` for doc2tex's recognizer (reference:
doc2tex/modules/build_model.py:7-79 and doc2tex/modules/recognizers/*).

Same constructor argument (the flat config dict), same attributes
(.stages/.featextractor/.seqmodeler/.predicter), same forward signatures and
return tuples, same state_dict key names and shapes -- but every forward runs
in libd2t (hand-written gfx950 HIP kernels) through ctypes.  There is no CPU or
eager-PyTorch fallback: on a machine without the built library or without a
HIP device the forward raises.
"""
import os

import torch
import torch.nn as nn

from . import params as P
from .engine import Engine


class FeatExtractorBuilder(nn.Module):
    """recognizers/build_feat.py:8-63 (parameter tree; ResNet or VGG; the height-mean for Seq=BiLSTM
    runs inside the engine)."""

    def __init__(self, flow, config):
        super().__init__()
        self.config = config
        self.flow = flow
        self.feat_name = flow["Feat"]
        if self.feat_name != "None":
            config["FeatureExtraction"]["params"].pop("mean_height", True)  # build_feat.py:16
            if self.feat_name == "VGG":
                self.FeatureExtraction = P.VGGFeatureExtractorParams(**config["FeatureExtraction"]["params"])
            elif self.feat_name == "ResNet":
                self.FeatureExtraction = P.ResNetFeatureExtractorParams(**config["FeatureExtraction"]["params"])
            else:
                raise NotImplementedError(f"FeatureExtraction '{self.feat_name}' is not on the accelerated path")
            self.FeatureExtraction_output = config["FeatureExtraction"]["params"]["output_channel"]
        else:
            if flow["Seq"] != "ViT":
                raise Exception("No FeatureExtraction module specified")
            self.FeatureExtraction = nn.Identity()


class SeqModelingBuilder(nn.Module):
    """recognizers/build_seq.py:7-40 (parameter tree; ViT hybrid, BiLSTM or None)."""

    def __init__(self, flow, config, FeatureExtraction_output):
        super().__init__()
        self.config = config
        self.flow = flow
        if flow["Seq"] == "ViT":
            assert config["max_dimension"] is not None, \
                "ViT encoder require exact height or max height and max width"
            sp = config["SequenceModeling"]["params"]
            bb = sp["backbone"]
            backbone = P.ResNetFeatureExtractorParams(bb["input_channel"], bb["output_channel"], bb["gcb"])
            max_dimension = ((config["imgH"], config["max_dimension"][1]) if config["imgH"]
                             else config["max_dimension"])  # vit_encoder.py:292-294
            ps = sp["patch_size"]
            ps = (ps, ps) if isinstance(ps, int) else tuple(ps)
            if sp.get("patching_style") != "2d":
                raise NotImplementedError("patching_style '1d' (TRIGBaseEncoder) crashes in the reference: HybridEmbed1D has "
                                          "no patch_size attribute for build_seq.py:63-66 to read")
            # create_vit_modeling (vit_encoder.py:295-302): fix_embed -> ViTEncoderV3 (frozen sincos table); otherwise a
            # learned table, read through bicubic interpolation (ViTEncoder) or -- interpolate_embed False -- a prefix slice
            # (ViTEncoderV2)
            self.SequenceModeling = P.ViTEncoderParams(
                img_size=tuple(max_dimension), patch_size=ps, in_chans=sp["input_channel"], depth=sp["depth"],
                embed_dim=sp["hidden_size"], num_heads=sp["num_heads"], hybrid_backbone=backbone,
                fix_embed=bool(sp.get("fix_embed", False)))
        elif flow["Seq"] == "BiLSTM":  # build_seq.py:13-25
            hidden_size = config["SequenceModeling"]["params"]["hidden_size"]
            self.SequenceModeling = nn.Sequential(
                P.BidirectionalLSTMParams(FeatureExtraction_output, hidden_size, hidden_size),
                P.BidirectionalLSTMParams(hidden_size, hidden_size, hidden_size))
            self.SequenceModeling_output = hidden_size
        elif flow["Seq"] == "None":
            if flow["Pred"] == "TFM":
                self.image_positional_encoder = P.PositionalEncoding2DParams(FeatureExtraction_output)
            self.SequenceModeling_output = FeatureExtraction_output
        else:
            raise NotImplementedError(f"SequenceModeling '{flow['Seq']}' is not on the accelerated path")


class PredictBuilder(nn.Module):
    """recognizers/build_pred.py:9-26 (parameter tree; TFM, Attn, Attnv2)."""

    def __init__(self, flow, config, SequenceModeling_output):
        super().__init__()
        self.flow = flow
        self.config = config
        if flow["Pred"] not in ("TFM", "Attn", "Attnv2"):
            raise NotImplementedError(f"Prediction '{flow['Pred']}' is not on the accelerated path")
        config["Prediction"]["params"]["num_classes"] = config["num_class"]  # build_pred.py:16-17
        config["Prediction"]["params"]["device"] = config["device"]
        if flow["Pred"] == "TFM":
            self.Prediction = P.TransformerPredictionParams(**config["Prediction"]["params"])
        else:  # Attention and AttentionV2 share parameters (seq2seq_v2.py:11)
            self.Prediction = P.AttentionParams(**config["Prediction"]["params"])


class DecodeHandle:
    """Completion handle of a pipelined forward (addition_outputs["decode"]): the forward's tokens / logits are views the
    engine's decode stream is still writing.  `wait()` orders the current stream (optionally the host) after exactly that
    decode -- launching the group first if it is still collecting batches -- and `done()` polls without blocking.
    `result()` waits and returns (prediction, logits) as the synchronous call would: with is_test they are cut at the
    first step at which every row of THIS batch had emitted [s] (tfm.py:138-140)."""

    def __init__(self, model, eng, ticket=None, index=0, tensors=None):
        self._model, self._eng, self.ticket = model, eng, ticket
        self._index, self._tensors = index, tensors  # position of this batch inside its decode group; full-size views
        self._steps = None

    def steps(self):
        """Valid length of this batch's tokens / logits (blocks the host until its decode is complete).  Cached: the
        C side keeps the step counts of the last 64 decodes only."""
        if self._steps is None:
            self._launch()
            per_batch = self._eng.decode_steps(self.ticket)
            # a decode without early exit reports ONE entry (all max_seq_len + 1 steps) whatever the group size
            self._steps = per_batch[self._index] if len(per_batch) > 1 else per_batch[0]
        return self._steps

    def result(self):
        n = self.steps()  # blocks the host until the decode is complete
        p, l = self._tensors
        return p[:, :n], l[:, :n]

    def _launch(self):
        if self.ticket is None:  # the group this forward belongs to has not been launched yet
            self._model._flush_group(self._eng)
        assert self.ticket is not None

    def done(self):
        if self.ticket is None:
            return False
        return self._eng.ticket_done(self.ticket)

    def wait(self, host_sync=False):
        self._launch()
        self._eng.wait_ticket(self.ticket, host_sync=host_sync)


class Model(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        stages = {
            "Feat": opt["FeatureExtraction"]["name"],
            "Seq": opt["SequenceModeling"]["name"],
            "Pred": opt["Prediction"]["name"],
        }
        self.stages = stages
        if stages["Seq"].__contains__("Vi"):
            assert stages["Feat"] == "None"
        self.featextractor = FeatExtractorBuilder(stages, opt)
        self.seqmodeler = SeqModelingBuilder(stages, opt, getattr(self.featextractor, "FeatureExtraction_output", None))
        self.predicter = PredictBuilder(stages, opt, getattr(self.seqmodeler, "SequenceModeling_output", None))
        self._engine = None
        # Opt-in software pipelining across consecutive forward() calls (eval, greedy, is_test=False):
        # forward returns while the latency-bound decode loop still runs on the engine's stream, so the
        # next batch's encoder overlaps it.  Results are valid after synchronize().
        self.pipelined = False
        # pipelined mode: block slots the persistent convolution leaves free for the decode stream
        self.reserved_blocks = 64
        # split-bf16 convolution kernel: 'pipelined16' (256x128 tile, one block per CU, three LDS stages, 16x16x32 MFMAs;
        # default) or 'classic' (128x128 on 32x32x16 MFMAs, two blocks per CU)
        self.conv_kernel = "pipelined16"
        # pipelined serving: compute units the pipelined kernel's grid leaves to the decode streams
        self.reserved_cus = 0
        # validation switches (tests): the 2x2 max-pools / 1x1 shortcuts inside the neighbouring convolution's launch (default)
        self.conv_fusion = (True, True)
        # beam search (TFM, d_model 256, beam <= 6): one cross-attention block per sample for all its hypotheses (default: per row)
        self.beam_shared_tile = False
        # pipelined mode: decode loops in flight side by side (1 .. 4)
        self.decode_chains = 1
        # pipelined mode: decode the rows of this many consecutive forward() calls in ONE step loop.  The decode step is a
        # chain of small latency-bound kernels whose duration barely depends on the row count, so two batches per loop halve
        # the launches (and the interference with the next encoders) per formula.  Rows are independent and every kernel is
        # dispatched by layer shape only, so each row's tokens / logits are bit-identical to an ungrouped decode.  Results
        # of a group become valid after synchronize() (which also launches a group that is still incomplete).
        self.decode_group = 1
        self._grp = None
        # Arithmetic of the backbone / large GEMMs:
        #   'bf16x3'  (default) split-bf16: 3 bf16 MFMAs per product, fp32 accumulate; max |dlogit| ~5e-5 against the 1e-3 bar
        #             on every fixture and on fresh seeds (tools/probe/fp16x2_margin.py);
        #   'fp16x2'  (opt-in, +21 % formulas/s behind a ViT encoder) fp16 feature-map records x fp16 hi / lo weights in the
        #             backbone: 2 MFMAs per product.  Tokens exact and logits within 1e-3 on every HybridViT / LSTM-head fixture
        #             (C2: ~2e-4, C4: ~1.5e-4), but the margin is 5x, not 20x: the tiny test stack T2 reaches 5e-4 .. 1.05e-3 on
        #             fresh seeds, a near-tie between two tokens flips ~5x as often, and Feat=ResNet + Seq=None stacks (their
        #             decoder reads the backbone's output directly) move by up to 9e-3.  Needs conv_kernel = 'pipelined16';
        #   'mixed'   (round 4, opt-in) 'bf16x3' with the two-MFMA arithmetic in the first `mixed_units` of the backbone's eight
        #             plain 512 -> 512 units only (layer3.1 .. layer3.4, conv3, layer4.0 .. layer4.2: the K = 4608 layers); the error
        #             grows with the square root of the number of such layers;
        #   'fp32'    exact fp32 matrix-core arithmetic.
        # D2T_CONV_PRECISION=auto|bf16x3|fp16x2|mixed|fp32 overrides the default; 'auto' is 'bf16x3'.
        # The reference's --amp (api/infer.py:120-124,157-161; engine/inferencing.py:68-72,149-153) wraps the model call in
        # torch.autocast: its convolutions and linears then run on fp16 operands and return fp16 (measured on the reference
        # itself, CPU autocast(float16): max |dlogit| 3e-3 on C2, 7e-3 on T2 against its own fp32 path).  A forward of an
        # eval()-mode HybridViT stack that the caller runs under torch.autocast("cuda") therefore takes `amp_conv_precision`
        # ('fp16x2': 2e-4 on C2, 17x closer to fp32 than the reference's AMP path; None: autocast changes nothing) -- outputs
        # stay fp32.  An explicit conv_precision / D2T_CONV_PRECISION always wins, and training steps keep their arithmetic.
        self.amp_conv_precision = "fp16x2"
        prec = os.environ.get("D2T_CONV_PRECISION", "auto")
        self._precision_auto = prec == "auto"
        if prec == "auto":
            prec = "bf16x3"
        self._conv_precision = prec
        self.mixed_units = 3  # conv_precision 'mixed': units on the two-MFMA arithmetic (0 .. 8; 3 keeps |dlogit| <= 1e-4 on C2 / C4 / S0)
        # data-parallel training: a doc2tex_amd.dist.GradSync makes loss.backward() return all-reduced (mean) gradients
        self.grad_sync = None

    def beam_search_batch(self, input, beam_size=None):
        """Extension (the reference's beam search takes one sample per call, tfm.py:146-148): encode the whole batch
        and advance the hypotheses of all samples in one step loop.  Returns [(LongTensor [1, len], score)] per
        sample, each identical to `forward(input[i:i+1], ...)` with `beam_size` set."""
        beam = int(beam_size or self.opt.get("beam_size", 1))
        memory, _, _ = self.forward_encoder(input)
        if self.stages["Pred"] == "TFM":
            return self.engine().decode_beam_batch(memory.contiguous(), beam)
        return self.engine().decode_attn_beam_batch(memory.contiguous(), beam)

    def _group_decode(self, eng, memory, start, is_test=False):
        """pipelined + decode_group > 1: collect the encoder memories of consecutive calls, launch one decode per group.
        Every group writes into fresh tensors (no ring to overrun): the views handed out stay valid for as long as the
        caller keeps them, and become readable once the group's ticket is complete."""
        import torch
        G = int(self.decode_group)
        B, T, d = memory.shape
        S, V = eng.cfg.max_seq_len + 1, eng.cfg.vocab
        key = (G, B, T, d, S, V, memory.device, bool(is_test))
        g = self._grp
        if g is not None and g["key"] != key:
            self._flush_group(eng)
            g = None
        if g is None:
            g = self._grp = {"key": key, "n": 0, "handles": [],
                             "mem": torch.empty((G * B, T, d), dtype=torch.float32, device=memory.device),
                             "start": torch.empty((G * B,), dtype=torch.int64, device=memory.device),
                             "tokens": torch.empty((G * B, S), dtype=torch.int64, device=memory.device),
                             "logits": torch.empty((G * B, S, V), dtype=torch.float32, device=memory.device)}
        k = g["n"]
        g["mem"][k * B:(k + 1) * B].copy_(memory)
        g["start"][k * B:(k + 1) * B].copy_(start.to(device=memory.device, dtype=torch.int64))
        views = g["tokens"][k * B:(k + 1) * B], g["logits"][k * B:(k + 1) * B]
        handle = DecodeHandle(self, eng, None, k, views)
        g["handles"].append(handle)
        out = views + (handle,)
        g["n"] += 1
        if g["n"] == G:
            self._flush_group(eng)
        return out

    def _flush_group(self, eng):
        g = self._grp
        if g is None or g["n"] == 0:
            return
        rows = g["n"] * g["key"][1]
        ticket = eng.decode_greedy_async_into(g["mem"][:rows], g["start"][:rows], g["tokens"][:rows], g["logits"][:rows],
                                              is_test=g["key"][-1], rows_per_batch=g["key"][1])
        for h in g["handles"]:
            h.ticket = ticket
        self._grp = None

    def synchronize(self, host_sync=True, flush=True):
        """Order the current stream (and optionally the host) after every outstanding pipelined decode.  `flush` also
        launches a decode group that is still incomplete (decode_group > 1); a serving loop that only wants to order a
        consumer stream after the groups already launched passes flush=False."""
        if self._engine is not None:
            if flush:
                self._flush_group(self._engine)
            self._engine.decode_wait(host_sync=host_sync)

    # -- engine plumbing -----------------------------------------------------
    def engine(self, finalize=True):
        """The libd2t context of this model, with the current weights uploaded (and packed for inference)."""
        # the context lives where the parameters live (model.to("cuda:1") moves the engine with them); before the first
        # .to() the reference's own device string, opt["device"] (build_pred.py:17), decides
        p0 = next(self.parameters())
        dev = p0.device if p0.is_cuda else None
        if self._engine is not None and dev is not None and self._engine.device != dev.index:
            self.synchronize()
            self._engine, self._grp = None, None  # parameters were moved to another GPU: a fresh context there
        if self._engine is None:
            self._engine = Engine(self.opt, device=dev)
        self._engine.sync_weights(self, finalize=finalize)
        want = self.reserved_blocks if self.pipelined else 0
        if getattr(self._engine, "_reserved", None) != want:
            self._engine.set_reserved_blocks(want)
            self._engine._reserved = want
        want_cus = self.reserved_cus if self.pipelined else 0
        if getattr(self._engine, "_reserved_cus", None) != want_cus:
            self._engine.set_reserved_cus(want_cus)
            self._engine._reserved_cus = want_cus
        prec = self.effective_conv_precision()
        if prec not in ("fp16x2", "mixed") and getattr(self._engine, "_precision", None) != prec:  # (before a change of the kernel)
            self._engine.set_conv_precision(prec)
            self._engine._precision = prec
        if getattr(self._engine, "_conv_kernel", None) != self.conv_kernel:
            self._engine.set_conv_kernel(self.conv_kernel)
            self._engine._conv_kernel = self.conv_kernel
        if getattr(self._engine, "_beam_shared", None) != bool(self.beam_shared_tile):
            self._engine.set_beam_shared_tile(self.beam_shared_tile)
            self._engine._beam_shared = bool(self.beam_shared_tile)
        if getattr(self._engine, "_fusion", None) != tuple(self.conv_fusion):
            self._engine.set_conv_fusion(*self.conv_fusion)
            self._engine._fusion = tuple(self.conv_fusion)
        if getattr(self._engine, "_chains", None) != self.decode_chains:
            self._engine.set_decode_chains(self.decode_chains)
            self._engine._chains = self.decode_chains
        if prec in ("fp16x2", "mixed") and getattr(self._engine, "_precision", None) != prec:  # (after the kernel choice: needs pipelined16)
            self._engine.set_conv_precision(prec)
            self._engine._precision = prec
            self._engine._mixed_units = None
        if prec == "mixed" and getattr(self._engine, "_mixed_units", None) != int(self.mixed_units):
            self._engine.set_mixed_units(int(self.mixed_units))
            self._engine._mixed_units = int(self.mixed_units)
        return self._engine

    @property
    def conv_precision(self):
        return self._conv_precision

    @conv_precision.setter
    def conv_precision(self, mode):
        if mode not in ("fp32", "bf16x3", "fp16x2", "mixed"):
            raise ValueError(f"conv_precision must be 'fp32', 'bf16x3', 'fp16x2' or 'mixed', not {mode!r}")
        self._conv_precision = mode
        self._precision_auto = False

    def _under_amp(self):
        """True when the caller's torch.autocast("cuda") asks for reduced precision and this model may follow it."""
        if not (self._precision_auto and self.amp_conv_precision) or self.training or self.stages["Seq"] != "ViT":
            return False
        import torch
        try:
            return bool(torch.is_autocast_enabled("cuda"))
        except TypeError:  # older signature: no device argument, CUDA implied
            return bool(torch.is_autocast_enabled())

    def effective_conv_precision(self):
        """The arithmetic the next forward runs in (an 'auto' fp16x2 steps back to bf16x3 for other convolution kernels;
        under the caller's torch.autocast an 'auto' model takes amp_conv_precision)."""
        if self._under_amp() and self.conv_kernel == "pipelined16":
            if self.amp_conv_precision not in ("fp16x2", "mixed", "bf16x3"):
                raise ValueError(f"amp_conv_precision must be 'fp16x2', 'mixed', 'bf16x3' or None, not {self.amp_conv_precision!r}")
            return self.amp_conv_precision
        if self._precision_auto and self._conv_precision in ("fp16x2", "mixed") and self.conv_kernel != "pipelined16":
            return "bf16x3"
        return self._conv_precision

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if self._engine is not None:
            self._engine._sig = None  # tensors moved: re-upload on next use
        return out

    # -- reference API -------------------------------------------------------
    def forward_encoder(self, input, *args, **kwargs):
        """build_model.py:36-43 -> (contextual_feature [B,T,d], output_shape, feat_pad)."""
        memory, grid, pad = self.engine().encode(input)
        if self.stages["Seq"] == "ViT":
            return memory, grid, pad  # build_seq.py:59-66
        return memory, None, None  # build_seq.py:69-76

    def forward_decoder(self, contextual_feature, text, is_train=True, is_test=False, rtl_text=None):
        """build_model.py:45-53 / build_pred.py:28-50 / tfm.py:188-195."""
        beam_size = self.opt.get("beam_size", 1)  # read on every call, build_pred.py:31
        self._luong_check()
        eng = self.engine()
        if self.stages["Pred"] in ("Attn", "Attnv2"):
            # build_pred.py:36-44 -> Attention.forward (seq2seq.py:333-347)
            if self.training or is_train:
                raise NotImplementedError(
                    "forward_decoder() is the inference entry point of the HIP engine (greedy / beam decoding of an encoder "
                    "memory): call model.eval() and pass is_train=False (engine/inferencing.py:70-76).  The teacher-forced "
                    "training pass of the LSTM-attention head runs through Model.forward() under model.train(), which keeps "
                    "the whole step on one autograd node")
            if beam_size > 1:  # seq2seq.py:333-347 -> forward_beam (one sample, returns (seq, score, None))
                prediction, logits = eng.decode_attn_beam(contextual_feature.contiguous(), beam_size)
                return prediction, logits, None, {}
            prediction, logits = eng.decode_attn_greedy(contextual_feature.contiguous(), is_test)
            return prediction, logits, None, {}
        if self.training:
            raise NotImplementedError(
                "forward_decoder() is the inference entry point of the HIP engine (greedy / beam decoding of an encoder "
                "memory): call model.eval() first.  The teacher-forced training pass (tfm.py:103-118) runs through "
                "Model.forward() under model.train(), encoder and decoder on one autograd node, so that loss.backward() "
                "reaches the backbone")
        if beam_size > 1:
            prediction, logits = eng.decode_beam(contextual_feature.contiguous(), beam_size)
        else:
            if text.dim() != 2 or text.shape[1] != 1:
                raise ValueError("eval decoding expects text = [B,1] start tokens ([GO])")
            # pipelined: the forward returns while the step loop runs on an engine stream.  With is_test the early exit is
            # taken on the device; prediction / logits are then FULL-SIZE tensors whose valid length is only known once the
            # decode is complete -- addition_outputs["decode"].result() waits and returns them cut as the reference does
            if self.pipelined and int(self.decode_group) > 1:
                prediction, logits, handle = self._group_decode(eng, contextual_feature.contiguous(), text[:, 0], is_test)
                return prediction, logits, None, {"decode": handle}
            elif self.pipelined:
                prediction, logits, ticket = eng.decode_greedy_async(contextual_feature.contiguous(), text[:, 0], is_test)
                return prediction, logits, None, {"decode": DecodeHandle(self, eng, ticket, 0, (prediction, logits))}
            else:
                prediction, logits = eng.decode_greedy(contextual_feature.contiguous(), text[:, 0], is_test)
        return prediction, logits, None, {}

    def _luong_check(self):
        """attn_type 'luong': Attention.forward_greedy / forward_beam call `self.attention_cell.reset_mem()` first thing
        (seq2seq.py:114,285; seq2seq_v2.py:65,247) and LuongAttention has no such method, so in the reference every forward
        of such a model -- training or evaluation -- ends in this AttributeError.  Same here, before any GPU work."""
        if self.stages["Pred"] in ("Attn", "Attnv2") and self.opt["Prediction"]["params"].get("attn_type", "coverage") == "luong":
            raise AttributeError("'LuongAttention' object has no attribute 'reset_mem'")

    def forward(self, input, text, is_train=True, is_test=False, rtl_text=None):
        self._luong_check()
        if self.training:
            # module.train(): teacher-forced pass with BatchNorm on batch statistics (tfm.py:103-118 / seq2seq.py:224-331
            # with is_train), one autograd node over the whole network so that loss.backward() (engine/training.py:137)
            # fills every .grad
            from .train import train_forward
            if self.stages["Pred"] == "TFM":
                if self.stages["Seq"] not in ("ViT", "None"):
                    raise NotImplementedError("the training step is implemented for the HybridViT + TFM and ResNet + TFM stacks")
            else:  # Attn / Attnv2
                if self.stages["Seq"] not in ("ViT", "BiLSTM"):
                    raise NotImplementedError("training the LSTM-attention head is implemented on the HybridViT and BiLSTM encoders")
            logits = train_forward(self, input, text)
            return logits.argmax(dim=2), logits, {}
        contextual_feature, output_shape, feat_pad = self.forward_encoder(input)
        prediction, logits, decoder_attn, addition_outputs = self.forward_decoder(
            contextual_feature, text=text, is_train=is_train, is_test=is_test, rtl_text=rtl_text)
        # decoder_attn is always None for the TFM head (build_pred.py:34,46-49)
        return prediction, logits, addition_outputs
