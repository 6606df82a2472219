"""Thin Python owner of a libd2t context: config marshalling, weight upload,
encode / decode calls on torch ROCm tensors.  No compute happens here."""
import ctypes as C

import torch

from . import _lib


def config_from_opt(opt):
    """Flat reference config dict (config/train.yaml schema) -> D2TConfig."""
    feat = opt["FeatureExtraction"]["name"]
    seq = opt["SequenceModeling"]["name"]
    pred = opt["Prediction"]
    pp = pred["params"]
    cfg = _lib.D2TConfig()
    max_dim = opt.get("max_dimension") or [0, 0]
    if opt.get("imgH"):
        max_dim = (opt["imgH"], max_dim[1])  # vit_encoder.py:292-294
    cfg.max_h, cfg.max_w = int(max_dim[0]), int(max_dim[1])
    cfg.backbone_out = 512
    cfg.in_channels = 1
    if seq == "ViT":
        sp = opt["SequenceModeling"]["params"]
        bbp = sp.get("backbone") or {}
        if bbp.get("name") != "resnet":
            raise NotImplementedError("ViT without the resnet hybrid backbone cannot run in the reference either "
                                      "(PatchEmbed returns a 3-tuple, SURVEY.md 3 notes)")
        if sp.get("patching_style") != "2d":
            raise NotImplementedError("patching_style '1d' (TRIGBaseEncoder) crashes in the reference: HybridEmbed1D has no "
                                      "patch_size attribute for build_seq.py:63-66 to read")
        # create_vit_modeling, vit_encoder.py:295-302
        if sp.get("fix_embed", False):
            cfg.vit_pos = _lib.VIT_POS_SINCOS_PREFIX       # ViTEncoderV3
        elif not sp.get("interpolate_embed", True):
            cfg.vit_pos = _lib.VIT_POS_LEARNED_PREFIX      # ViTEncoderV2
        else:
            cfg.vit_pos = _lib.VIT_POS_LEARNED_INTERP      # ViTEncoder
        cfg.gcb = int(bool(bbp.get("gcb", False)))
        cfg.encoder = _lib.ENC_HYBRID_VIT
        cfg.in_channels = int(bbp["input_channel"])
        cfg.backbone_out = int(bbp["output_channel"])
        cfg.vit_depth, cfg.vit_heads, cfg.vit_dim = int(sp["depth"]), int(sp["num_heads"]), int(sp["hidden_size"])
        ps = sp["patch_size"]
        ps = (ps, ps) if isinstance(ps, int) else tuple(ps)
        cfg.patch_h, cfg.patch_w = int(ps[0]), int(ps[1])
    elif feat in ("ResNet", "VGG") and seq in ("None", "BiLSTM"):
        fp = opt["FeatureExtraction"]["params"]
        if fp.get("gcb", False) and feat == "VGG":
            raise NotImplementedError("gcb is a ResNet option")
        cfg.gcb = int(bool(fp.get("gcb", False))) if feat == "ResNet" else 0
        cfg.in_channels = int(fp["input_channel"])
        cfg.backbone_out = int(fp["output_channel"])
        if seq == "BiLSTM":
            cfg.encoder = _lib.ENC_VGG_BILSTM if feat == "VGG" else _lib.ENC_RESNET_BILSTM
            cfg.bilstm_hidden = int(opt["SequenceModeling"]["params"]["hidden_size"])
            if opt["SequenceModeling"]["params"].get("pos_enc", False):
                raise NotImplementedError("BiLSTM pos_enc crashes in the reference (self.gated undefined, build_seq.py:55)")
        elif feat == "ResNet":
            cfg.encoder = _lib.ENC_RESNET
        else:
            raise NotImplementedError("Feat=VGG with Seq=None crashes in the reference (build_seq.py:78)")
    else:
        raise NotImplementedError(f"Feat={feat} Seq={seq} is not on the accelerated path")
    cfg.vocab = int(opt["num_class"])
    cfg.batch_max_length = int(opt.get("batch_max_length", 0))
    if pred["name"] == "TFM":
        cfg.decoder = _lib.DEC_TFM
        cfg.dec_dim, cfg.dec_heads = int(pp["d_model"]), int(pp["nhead"])
        cfg.dec_layers, cfg.dec_ff = int(pp["num_decoder_layers"]), int(pp["dim_feedforward"])
        cfg.max_seq_len = int(pp["max_seq_len"])
    elif pred["name"] in ("Attn", "Attnv2"):
        cfg.decoder = _lib.DEC_ATTN
        cfg.attn_hidden = int(pp["hidden_size"])
        cfg.attn_kernel_size, cfg.attn_kernel_dim = int(pp["kernel_size"]), int(pp["kernel_dim"])
        cfg.attn_enc_init = int(bool(pp.get("enc_init", False)))
        attn_type = pp.get("attn_type", "coverage")
        if attn_type == "luong":  # constructs in the reference, but no forward can run (build_model raises before any engine exists)
            raise AttributeError("'LuongAttention' object has no attribute 'reset_mem'")
        cfg.attn_coverage = int(attn_type == "coverage")
        cfg.attn_cell = _lib.ATTN_CELL_LOCATION if attn_type in ("coverage", "loc_aware") else _lib.ATTN_CELL_BAHDANAU
        cfg.attn_onehot = int(not pp.get("embed_target", False))
        if cfg.attn_cell == _lib.ATTN_CELL_BAHDANAU:
            cfg.attn_kernel_size, cfg.attn_kernel_dim = 0, 1  # no location filter
        sm = pp.get("seqmodel", "ViT")
        if pred["name"] == "Attnv2":  # seq2seq_v2.py:182-199
            if sm in ("BiLSTM", "VIG"):
                cfg.attn_keys = _lib.ATTN_KEYS_ALL_INIT_MEAN
            elif sm == "TFM":
                cfg.attn_keys = _lib.ATTN_KEYS_NOCLS_INIT_CLS
            else:
                raise ValueError("seqmodel must be either BiLSTM or TFM option")
        else:  # seq2seq.py:229-238
            cfg.attn_keys = _lib.ATTN_KEYS_ALL_INIT_MEAN if sm == "BiLSTM" else _lib.ATTN_KEYS_ALL_INIT_FIRST
    else:
        raise NotImplementedError(f"Prediction '{pred['name']}' is not on the accelerated path")
    return cfg


def device_index(device):
    """opt["device"] as the reference passes it around ('cuda', 'cuda:1', a torch.device, None) -> HIP device index."""
    if device is None or (isinstance(device, str) and device in ("", "cpu")):
        return torch.cuda.current_device()  # the engine has no CPU path; 'cpu' configs get the current GPU
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"doc2tex_amd: device '{device}' is not a ROCm (cuda) device; the engine has no CPU path")
    return torch.cuda.current_device() if d.index is None else int(d.index)


class Engine:
    def __init__(self, opt, device=None):
        self.lib = _lib.require_device()
        self.cfg = config_from_opt(opt)
        self.ctx = C.c_void_p()
        # the context lives on ONE device: opt["device"] (build_pred.py:17) unless the module has been moved since
        self.device = device_index(opt.get("device") if device is None else device)
        if not 0 <= self.device < torch.cuda.device_count():
            raise RuntimeError(f"doc2tex_amd: no HIP device {self.device} (visible: {torch.cuda.device_count()})")
        with torch.cuda.device(self.device):
            rc = self.lib.d2t_create(C.byref(self.cfg), C.byref(self.ctx))
        if rc != _lib.D2T_OK:
            msg = self.lib.d2t_last_error(self.ctx).decode() if self.ctx else ""
            if self.ctx:
                self.lib.d2t_destroy(self.ctx)
            self.ctx = None
            raise RuntimeError(f"d2t_create failed (code {rc}): {msg}")
        assert self.lib.d2t_device_of(self.ctx) == self.device
        self._sig = None

    def _on_device(self, t, what):
        """Tensors handed to the context must live on its device (the C side checks the raw pointers too)."""
        if not t.is_cuda:
            raise RuntimeError(f"doc2tex_amd: {what} must be a ROCm (cuda) tensor; the engine has no CPU path")
        if t.device.index != self.device:
            raise RuntimeError(f"doc2tex_amd: {what} is on {t.device} but this model's engine lives on cuda:{self.device}; "
                               "move the input, or the Model with .to(device), so that they agree")

    def __del__(self):
        ctx, self.ctx = getattr(self, "ctx", None), None
        if ctx:
            try:
                self.lib.d2t_destroy(ctx)
            except Exception:
                pass

    def _check(self, rc, what):
        _lib.check(rc, self.ctx, what)

    # ---- weights ---------------------------------------------------------
    def sync_weights(self, module, finalize=True):
        """Upload the module's state if any tensor changed since last time; finalize=True also folds / packs it
        for inference (the training step reads the raw copies and skips that)."""
        items = []
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.is_floating_point() or name.endswith("image_positional_encoder.pe"):
                continue
            items.append((name, t))
        sig = tuple((n, t.data_ptr(), t._version, tuple(t.shape)) for n, t in items)
        if sig == self._sig:
            if finalize and getattr(self, "_fin_sig", None) != sig:
                self._check(self.lib.d2t_finalize_weights(self.ctx, _lib.stream_of(items[0][1])), "finalize_weights")
                self._fin_sig = sig
            return
        stream = None
        tds = []
        for name, t in items:
            if not t.is_cuda:
                raise RuntimeError(f"doc2tex_amd: parameter '{name}' is on {t.device}; move the Model to the GPU")
            self._on_device(t, f"parameter '{name}'")
            td = t.detach()
            if td.dtype != torch.float32 or not td.is_contiguous():
                td = td.float().contiguous()
            tds.append(td)
            stream = _lib.stream_of(td)
        shapes = tuple((n, tuple(t.shape)) for n, t in items)
        if shapes == getattr(self, "_loaded_shapes", None):
            # same tensors as last time with new contents (an optimizer step): one copy kernel instead of one copy each
            n = len(items)
            self._check(self.lib.d2t_reload_weights(
                self.ctx, n, (C.c_char_p * n)(*[nm.encode() for nm, _ in items]),
                (C.c_void_p * n)(*[td.data_ptr() for td in tds]), (C.c_int64 * n)(*[td.numel() for td in tds]), stream),
                "reload_weights")
        else:
            for (name, _), td in zip(items, tds):
                shape = (C.c_int64 * td.dim())(*td.shape)
                self._check(self.lib.d2t_load_weight(self.ctx, name.encode(), _lib.ptr(td), shape, td.dim(), stream),
                            f"load_weight({name})")
            self._loaded_shapes = shapes
        self._fin_sig = None
        if finalize:
            self._check(self.lib.d2t_finalize_weights(self.ctx, stream), "finalize_weights")
            self._fin_sig = sig
        self._sig = sig

    # ---- training step -------------------------------------------------------
    def train_forward(self, image, tgt):
        """Model.forward under module.train(): logits [B,L,V] of the teacher-forced pass (BatchNorm on batch
        statistics; the engine's running statistics are updated, see read_weight)."""
        self._on_device(image, "input")
        image = image.float().contiguous()
        tgt = tgt.to(device=image.device, dtype=torch.int64).contiguous()
        B, _, H, W = image.shape
        L = tgt.shape[1]
        logits = torch.empty((B, L, self.cfg.vocab), dtype=torch.float32, device=image.device)
        self._train_keep = (image, tgt)  # the backward pass reads them
        self._check(self.lib.d2t_train_forward(self.ctx, _lib.ptr(image), B, H, W, _lib.ptr(tgt), L, _lib.ptr(logits),
                                               _lib.stream_of(image)), "train_forward")
        return logits

    def set_dropout(self, p, seed):
        """Dropout probability of the decoder layers in the training step and the seed of its Philox masks."""
        self._check(self.lib.d2t_train_set_dropout(self.ctx, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF), "train_set_dropout")

    def set_teacher_flags(self, flags):
        """Scheduled sampling of the LSTM-attention head: flags[t] = 1 feeds the label at step t, 0 the model's own
        arg-max (seq2seq.py:311-316).  None / empty = always the label."""
        data = bytes(bytearray(int(bool(f)) for f in flags)) if flags else b""
        self._check(self.lib.d2t_train_set_teacher_flags(self.ctx, data, len(data)), "train_set_teacher_flags")

    def mask_count(self):
        return int(self.lib.d2t_train_mask_count(self.ctx))

    def read_mask(self, index, numel):
        m = torch.empty(int(numel), dtype=torch.uint8, device=f"cuda:{self.device}")
        self._check(self.lib.d2t_train_read_mask(self.ctx, int(index), _lib.ptr(m), int(numel), _lib.stream_of(m)), "train_read_mask")
        return m

    def train_backward(self, dlogits):
        dlogits = dlogits.float().contiguous()
        self._check(self.lib.d2t_train_backward(self.ctx, _lib.ptr(dlogits), _lib.stream_of(dlogits)), "train_backward")
        self._train_keep = None

    def train_grad(self, name, like):
        g = torch.empty_like(like, dtype=torch.float32, memory_format=torch.contiguous_format)
        self._check(self.lib.d2t_train_grad(self.ctx, name.encode(), _lib.ptr(g), g.numel(), _lib.stream_of(g)),
                    f"train_grad({name})")
        return g

    def train_gather(self, names, likes, source="grad"):
        """All gradients (source "grad") or the engine's copies of loaded tensors (source "weight": the BatchNorm running
        statistics after a training forward) in ONE kernel: returns fp32 tensors shaped like `likes`, views of one flat
        buffer (as DistributedDataParallel's gradient_as_bucket_view hands them out)."""
        n = len(names)
        numels = [int(t.numel()) for t in likes]
        offs, total = [], 0
        for m in numels:  # 16-byte aligned starts: the copy kernel moves float4s
            offs.append(total)
            total += (m + 3) & ~3
        if n == 0:
            return []
        flat = torch.empty(max(total, 1), dtype=torch.float32, device=likes[0].device)
        arr = (C.c_char_p * n)(*[s.encode() for s in names])
        self._check(self.lib.d2t_train_gather(self.ctx, 0 if source == "grad" else 1, n, arr, (C.c_int64 * n)(*offs),
                                              (C.c_int64 * n)(*numels), _lib.ptr(flat), _lib.stream_of(flat)),
                    "train_gather")
        return [flat[o:o + m].view(t.shape) for o, m, t in zip(offs, numels, likes)]

    def train_grad_into(self, name, dst):
        """Copy the gradient of `name` into the flat fp32 view `dst` on the current stream (ordered after the
        kernels that produce it, whichever stream that is)."""
        self._check(self.lib.d2t_train_grad(self.ctx, name.encode(), _lib.ptr(dst), dst.numel(), _lib.stream_of(dst)),
                    f"train_grad({name})")

    def read_weight(self, name, dst):
        """Copy the engine's current copy of a loaded tensor into `dst` (fp32, contiguous, same size)."""
        self._check(self.lib.d2t_read_weight(self.ctx, name.encode(), _lib.ptr(dst), dst.numel(), _lib.stream_of(dst)),
                    f"read_weight({name})")

    # ---- encoder -----------------------------------------------------------
    def encoder_shape(self, H, W):
        v = [C.c_int32() for _ in range(6)]
        self._check(self.lib.d2t_encoder_shape(self.ctx, H, W, *[C.byref(x) for x in v]), "encoder_shape")
        T, d, gh, gw, pw, ph = [x.value for x in v]
        return T, d, gh, gw, pw, ph

    def encode(self, image):
        self._on_device(image, "input")
        if image.dim() != 4 or image.shape[1] != 1:
            raise ValueError(f"expected image [B,1,H,W], got {tuple(image.shape)}")
        image = image.float().contiguous()
        B, _, H, W = image.shape
        T, d, gh, gw, pw, ph = self.encoder_shape(H, W)
        memory = torch.empty((B, T, d), dtype=torch.float32, device=image.device)
        self._check(self.lib.d2t_encode(self.ctx, _lib.ptr(image), B, H, W, _lib.ptr(memory),
                                        _lib.stream_of(image)), "encode")
        return memory, (gh, gw), (pw, ph)

    def set_conv_precision(self, mode):
        """'fp32' (exact), 'bf16x3' (split-bf16 matrix-core path for the convolutions: three products per element) or 'fp16x2'
        (fp16 feature maps times fp16 hi / lo weights in the backbone: two products per element), or 'mixed' ('bf16x3' with the
        'fp16x2' arithmetic in the first few 512 -> 512 units of the backbone only: set_mixed_units)."""
        code = {"fp32": _lib.CONV_FP32, "bf16x3": _lib.CONV_BF16X3, "fp16x2": _lib.CONV_FP16X2, "mixed": _lib.CONV_MIXED}[mode]
        self._check(self.lib.d2t_set_conv_precision(self.ctx, code), "set_conv_precision")

    def set_mixed_units(self, units):
        """'mixed' precision: how many of the backbone's eight plain 512 -> 512 units run the two-MFMA fp16 arithmetic (0 .. 8)."""
        self._check(self.lib.d2t_set_mixed_units(self.ctx, int(units)), "set_mixed_units")

    def set_reserved_blocks(self, blocks):
        self._check(self.lib.d2t_set_reserved_blocks(self.ctx, int(blocks)), "set_reserved_blocks")

    def set_reserved_cus(self, cus):
        self._check(self.lib.d2t_set_reserved_cus(self.ctx, int(cus)), "set_reserved_cus")

    def set_conv_kernel(self, kind):
        """'pipelined16' (256x128 tile, one block per CU, three LDS stages, 16x16x32 MFMAs; default) or 'classic' (128x128 on
        32x32x16 MFMAs, two blocks per CU)."""
        if kind not in ("classic", "pipelined16"):
            raise ValueError(f"conv kernel must be 'pipelined16' or 'classic', not {kind!r}")
        self._check(self.lib.d2t_set_conv_kernel(self.ctx, {"classic": 0, "pipelined16": 3}[kind]), "set_conv_kernel")

    def set_beam_shared_tile(self, on):
        """Beam search: one cross-attention block per sample serving all its hypotheses from one staged memory tile (beam <= 6)."""
        self._check(self.lib.d2t_set_beam_shared_tile(self.ctx, int(bool(on))), "set_beam_shared_tile")

    def set_conv_fusion(self, pools=True, shortcuts=True):
        """Validation switches: run the 2x2 max-pools / the BasicBlocks' 1x1 shortcuts as their own kernels (False) instead of
        inside the neighbouring convolution's launch (True, default)."""
        self._check(self.lib.d2t_set_conv_fusion(self.ctx, int(bool(pools)), int(bool(shortcuts))), "set_conv_fusion")

    def set_decode_chains(self, chains):
        self._check(self.lib.d2t_set_decode_chains(self.ctx, int(chains)), "set_decode_chains")

    # ---- kernel timing -------------------------------------------------------
    def profile(self, on):
        self._check(self.lib.d2t_profile_enable(self.ctx, int(bool(on))), "profile_enable")

    def profile_read(self, max_records=4096):
        """[(M, N, K, ms)] for every MFMA implicit-GEMM launch since the last read."""
        n = C.c_int32(0)
        M, N, K = [(C.c_int32 * max_records)() for _ in range(3)]
        ms = (C.c_float * max_records)()
        self._check(self.lib.d2t_profile_read(self.ctx, max_records, C.byref(n), M, N, K, ms), "profile_read")
        return [(M[i], N[i], K[i], ms[i]) for i in range(n.value)]

    # ---- decoder -----------------------------------------------------------
    def decode_greedy(self, memory, start_tokens, is_test):
        self._on_device(memory, "memory")
        memory = memory.float().contiguous()
        B, T, _ = memory.shape
        S, V = self.cfg.max_seq_len + 1, self.cfg.vocab
        start = start_tokens.to(device=memory.device, dtype=torch.int64).contiguous()
        tokens = torch.zeros((B, S), dtype=torch.int64, device=memory.device)
        logits = torch.zeros((B, S, V), dtype=torch.float32, device=memory.device)
        steps = C.c_int32(0)
        self._check(self.lib.d2t_decode_greedy(self.ctx, _lib.ptr(memory), B, T, _lib.ptr(start), int(bool(is_test)),
                                               _lib.ptr(tokens), _lib.ptr(logits), C.byref(steps),
                                               _lib.stream_of(memory)), "decode_greedy")
        s = steps.value
        return tokens[:, :s], logits[:, :s]

    def decode_attn_greedy(self, memory, is_test):
        """Attention/AttentionV2.forward_greedy in eval mode: full-size (preds_index [B,S], probs [B,S,V])."""
        memory = memory.float().contiguous()
        B, T, _ = memory.shape
        S, V = self.cfg.batch_max_length + 1, self.cfg.vocab
        tokens = torch.zeros((B, S), dtype=torch.int64, device=memory.device)
        probs = torch.zeros((B, S, V), dtype=torch.float32, device=memory.device)
        steps = C.c_int32(0)
        self._check(self.lib.d2t_decode_attn_greedy(self.ctx, _lib.ptr(memory), B, T, int(bool(is_test)),
                                                    _lib.ptr(tokens), _lib.ptr(probs), C.byref(steps),
                                                    _lib.stream_of(memory)), "decode_attn_greedy")
        return tokens, probs

    def decode_greedy_async(self, memory, start_tokens, is_test=False):
        """Pipelined greedy decode (max_seq_len+1 steps; with is_test the device stops early): returns (tokens, logits, ticket).  The tensors are fresh
        allocations written by the engine's decode stream; they are valid once `ticket` is complete (wait_ticket /
        decode_wait) and are never recycled behind the caller's back -- the engine keeps them alive until then."""
        self._on_device(memory, "memory")
        memory = memory.float().contiguous()
        B, T, _ = memory.shape
        S, V = self.cfg.max_seq_len + 1, self.cfg.vocab
        start = start_tokens.to(device=memory.device, dtype=torch.int64).contiguous()
        tokens = torch.empty((B, S), dtype=torch.int64, device=memory.device)
        logits = torch.empty((B, S, V), dtype=torch.float32, device=memory.device)
        ticket = self.decode_greedy_async_into(memory, start, tokens, logits, is_test=is_test)
        return tokens, logits, ticket

    def decode_greedy_async_into(self, memory, start, tokens, logits, is_test=False, rows_per_batch=0):
        """d2t_decode_greedy_submit on caller-held buffers; returns the decode's ticket.  The buffers are referenced here
        until the ticket completes, so dropping them early cannot hand their memory to another allocation mid-write.
        rows_per_batch: the rows are several encoder batches of that size (a decode group)."""
        B, T, _ = memory.shape
        t = C.c_int64(0)
        self._check(self.lib.d2t_decode_greedy_submit(self.ctx, _lib.ptr(memory), B, T, _lib.ptr(start), int(bool(is_test)),
                                                      int(rows_per_batch), _lib.ptr(tokens), _lib.ptr(logits),
                                                      _lib.stream_of(memory), C.byref(t)), "decode_greedy_submit")
        self._wait_dev = memory.device
        ticket = int(t.value)
        held = getattr(self, "_held", None)
        if held is None:
            held = self._held = []
        held.append((ticket, (memory, start, tokens, logits)))
        while held and self.ticket_done(held[0][0]):  # tickets complete in order per chain; at most a few stay pending
            self.decode_steps(held.pop(0)[0])  # keep its step counts past the C side's 64-entry ring
        if len(held) > 48:  # the C side keeps 64 ticket events: never let a live ticket fall out of its ring
            self.wait_ticket(held[0][0], host_sync=True)
        return ticket

    def decode_steps(self, ticket):
        """Per-batch step counts of an asynchronous decode (waits for it): [steps of batch 0, batch 1, ...]; ONE entry
        (max_seq_len + 1) for a decode without early exit.  Remembered on this side, so a handle consumed more than 64
        decodes later still learns its length."""
        cache = self.__dict__.setdefault("_steps_cache", {})
        ticket = int(ticket)
        if ticket not in cache:
            out = (C.c_int32 * 64)()
            n = C.c_int32(0)
            self._check(self.lib.d2t_decode_steps(self.ctx, ticket, out, 64, C.byref(n)), "decode_steps")
            cache[ticket] = [int(out[i]) for i in range(n.value)]
            for old in [t for t in cache if t <= ticket - 4096]:
                del cache[old]
        return cache[ticket]

    def ticket_done(self, ticket):
        r = int(self.lib.d2t_decode_query(self.ctx, int(ticket)))
        if r < 0:
            self._check(-r, "decode_query")
        return r == 1

    def wait_ticket(self, ticket, host_sync=False):
        """Order the current stream (and the host, if asked) after the decode with this ticket -- and only that one."""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._check(self.lib.d2t_decode_wait_ticket(self.ctx, int(ticket), stream, int(bool(host_sync))), "decode_wait_ticket")

    def decode_wait(self, host_sync=False):
        dev = getattr(self, "_wait_dev", None)
        if dev is None:
            return
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self._check(self.lib.d2t_decode_wait(self.ctx, stream, int(bool(host_sync))), "decode_wait")
        if host_sync:
            for t, _ in getattr(self, "_held", None) or []:
                self.decode_steps(t)
            self._held = []

    def decode_attn_beam(self, memory, beam_size):
        """Attention.forward_beam / AttentionV2.forward_beam for one sample: (LongTensor [1, n] on the CPU, score)."""
        memory = memory.float().contiguous()
        assert memory.shape[0] == 1  # seq2seq.py:90
        S = self.cfg.batch_max_length + 1
        seq = (C.c_int64 * S)()
        n = C.c_int32(0)
        score = C.c_float(0.0)
        self._check(self.lib.d2t_decode_attn_beam(self.ctx, _lib.ptr(memory), memory.shape[1], int(beam_size), seq,
                                                  C.byref(n), C.byref(score), _lib.stream_of(memory)), "decode_attn_beam")
        return torch.LongTensor(list(seq[: n.value])).unsqueeze(0), torch.tensor(float(score.value))

    def decode_attn_beam_batch(self, memory, beam_size):
        """LSTM-attention beam search for every sample of memory [N,T,256] in one step loop."""
        memory = memory.float().contiguous()
        N, S = memory.shape[0], self.cfg.batch_max_length + 1
        seq = (C.c_int64 * (N * S))()
        n = (C.c_int32 * N)()
        score = (C.c_float * N)()
        self._check(self.lib.d2t_decode_attn_beam_batch(self.ctx, _lib.ptr(memory), N, memory.shape[1], int(beam_size), seq,
                                                        n, score, _lib.stream_of(memory)), "decode_attn_beam_batch")
        return [(torch.LongTensor(list(seq[i * S: i * S + n[i]])).unsqueeze(0), torch.tensor(float(score[i])))
                for i in range(N)]

    def decode_beam_batch(self, memory, beam_size):
        """Beam search for every sample of memory [N,T,d] in one shared step loop: [(LongTensor [1,len], score)] * N,
        each equal to decode_beam on that sample alone."""
        memory = memory.float().contiguous()
        N, S = memory.shape[0], self.cfg.max_seq_len + 1
        seq = (C.c_int64 * (N * S))()
        n = (C.c_int32 * N)()
        score = (C.c_float * N)()
        self._check(self.lib.d2t_decode_beam_batch(self.ctx, _lib.ptr(memory), N, memory.shape[1], int(beam_size), seq, n,
                                                   score, _lib.stream_of(memory)), "decode_beam_batch")
        return [(torch.LongTensor(list(seq[i * S: i * S + n[i]])).unsqueeze(0), float(score[i])) for i in range(N)]

    def decode_beam(self, memory, beam_size):
        memory = memory.float().contiguous()
        if memory.shape[0] != 1:
            raise AssertionError(
                f"beam search should only have signle source, encounter with batch size: {memory.shape[0]}")
        S = self.cfg.max_seq_len + 1
        seq = (C.c_int64 * S)()
        n = C.c_int32(0)
        score = C.c_float(0.0)
        self._check(self.lib.d2t_decode_beam(self.ctx, _lib.ptr(memory), memory.shape[1], int(beam_size), seq,
                                             C.byref(n), C.byref(score), _lib.stream_of(memory)), "decode_beam")
        return torch.LongTensor(list(seq[: n.value])).unsqueeze(0), float(score.value)
