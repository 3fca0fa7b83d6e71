"""NativeClip / NativeSigLip — CLIP-family foundation models whose towers run on the package's own HIP kernels.

The reference's ``OpenClip.encode_image`` / ``encode_text`` (foundation_models/clip.py:103-135) call straight
into the third-party ``open_clip`` torch model.  ``NativeClip`` wraps any such ``AbstractVLM`` (tokenizer and
preprocessing stay the wrapped object's), reads the weights out of its torch modules once, and runs the
pre-LN transformer towers through the C ABI (K11: ``sl_linear`` = fp32-input MFMA GEMM with bias / GELU /
residual / patch-scatter epilogues, ``sl_layernorm``, ``sl_attention``, ``sl_patchify``, ``sl_embed_tokens``).
``gemm="bf16x3"`` (default) runs the linear layers as split-bf16 GEMMs on the bf16 matrix cores (three MFMAs per
product, fp32 accumulate: fp32-class accuracy, ~2.4x the fp32-MFMA rate) with weights split once at construction and
activations emitted in split form by the producing kernel; ``gemm="f32"`` uses the fp32-input MFMA GEMM.  Either
way features agree with the torch modules to ~1e-5 relative, and the object plugs into ``Lens`` and
``ActivationComponentVisualizer`` like any other ``AbstractVLM``.

Module layouts read (probed by attribute; anything else is refused with a ``TypeError`` that names what was found):

* **open_clip ``CLIP``** (what ``OpenClip(url)`` builds for the ViT-B/32, B/16, L/14 ... families, clip.py:52-62):
  ``visual`` = ``VisionTransformer`` {``conv1`` (no bias), ``class_embedding``, ``positional_embedding``, ``ln_pre``
  (or Identity), ``transformer.resblocks[i]`` {``ln_1``, ``attn`` (``nn.MultiheadAttention``), ``ls_1``, ``ln_2``,
  ``mlp`` (``c_fc``, ``gelu``, ``c_proj``), ``ls_2``}, ``ln_post``, ``proj``, ``pool_type`` in {``tok``, ``avg``},
  ``final_ln_after_pool``}; text members on the model itself: ``token_embedding``, ``positional_embedding``,
  ``transformer``, ``ln_final``, ``text_projection`` (matrix or ``nn.Linear``), ``attn_mask``, ``text_pool_type`` in
  {``argmax``, ``first``, ``last``}.  LayerScale (``ls_1`` / ``ls_2`` with a ``gamma``) is folded into the weights of
  the projection it follows.  Attention pooling (CoCa) and hybrid / convolutional towers (MobileCLIP) are refused.
* **open_clip ``CLIP`` with a ResNet image tower** (``OpenClip("RN50", ...)`` — BASELINE configs[0]'s embed model; RN101, RN50x4 ...):
  ``visual`` = ``ModifiedResNet`` {three-conv stem, ``layer1..4``, ``attnpool`` = ``AttentionPool2d`` {``positional_embedding``,
  ``q_proj``, ``k_proj``, ``v_proj``, ``c_proj``, ``num_heads``}}.  The convolutional trunk stays on PyTorch (MIOpen), like a probed
  model's forward; the attention pool and the projection to the joint space run on the kernels (``NativeResNetVision``:
  ``sl_tokens_from_map``, two GEMMs, ``sl_attention_pool_q``, ``c_proj``); the text tower natively as above.
* **open_clip ``CustomTextCLIP`` with a timm trunk** (what ``SigLipV2()`` builds, clip.py:190-211,
  ``hf-hub:timm/ViT-B-16-SigLIP2``): ``visual`` = ``TimmModel`` {``trunk`` = timm ``VisionTransformer`` with
  ``patch_embed.proj`` (bias), ``pos_embed``, no class token, ``blocks[i]`` {``norm1``, ``attn.qkv`` / ``attn.proj``,
  ``norm2``, ``mlp.fc1`` / ``act`` / ``fc2``}, ``norm``, ``attn_pool`` (``AttentionPoolLatent``: ``latent``, ``q``, ``kv``,
  ``proj``, ``norm``, ``mlp``), ``head``}; ``text`` = open_clip ``TextTransformer`` (non-causal, ``pool_type="last"``,
  ``text_projection`` = ``nn.Linear`` with bias).  ``NativeSigLip``.
* **transformers ``SiglipModel``** (``vision_model`` / ``text_model``; the SigLIP-so400m geometry of BASELINE configs[3]).
  ``NativeSigLip``.
* ``synth.SyntheticClip`` (the bench's random-init ViT-B/32).

head_dim 32, 64, 72, 80, 88, 96, 104 or 128 in the towers (any multiple of 4 up to 128 in the pooling heads).  open_clip / timm are not installed in this image, so the open_clip layouts are exercised
against ``tests/openclip_like.py`` — torch modules carrying open_clip 3.0's attribute tree — and parity for this row
stays build-vs-torch-module (the reference's own tests pin shapes only, tests/foundation_models/test_clip.py:32-85).
"""
from __future__ import annotations

import torch
from torch import nn

from semanticlens_amd import _native as N
from semanticlens_amd.foundation_models.base import AbstractVLM


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _first(obj, *names):
    for n in names:
        cur = obj
        ok = True
        for part in n.split("."):
            if not hasattr(cur, part):
                ok = False
                break
            cur = getattr(cur, part)
        if ok and cur is not None:
            return cur
    raise AttributeError(f"none of {names} found on {type(obj).__name__}")


_HEAD_DIMS = (32, 64, 72, 80, 88, 96, 104, 128)  # exactly what csrc/encoder.hip instantiates sl_attention / sl_attention_bf16x3 for


def _ln(mod: nn.Module, device):
    """(gamma, beta, eps) of a LayerNorm-like module (open_clip's ``LayerNorm`` / ``LayerNormFp32`` subclass nn.LayerNorm)."""
    if not isinstance(mod, nn.LayerNorm) or mod.weight is None or mod.bias is None:
        raise TypeError(f"expected an affine LayerNorm, found {type(mod).__name__}")
    return _f32(mod.weight, device), _f32(mod.bias, device), float(mod.eps)


def _is_identity(mod) -> bool:
    return mod is None or isinstance(mod, (nn.Identity, nn.Dropout))


def _act_code(act) -> int:
    """Activation of an MLP: a module (nn.GELU with ``approximate``, open_clip's QuickGELU, timm's GELUTanh / QuickGELU) or a
    config string (transformers: ``gelu``, ``gelu_pytorch_tanh``, ``quick_gelu``)."""
    if isinstance(act, nn.Module):
        name = type(act).__name__.lower()
        if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "tanh":
            return N.SL_ACT_GELU_TANH
    else:
        name = (act or "gelu").lower()
    if "quick" in name:
        return N.SL_ACT_QUICKGELU
    if "tanh" in name:
        return N.SL_ACT_GELU_TANH
    if "gelu" in name:
        return N.SL_ACT_GELU
    raise TypeError(f"unsupported MLP activation {name!r} (GELU, tanh-GELU and QuickGELU are built)")


def _layer_scale(mod, device):
    """LayerScale gamma of a residual branch (open_clip ``ls_1`` / ``ls_2``, timm ``ls1`` / ``ls2``), or None."""
    if _is_identity(mod):
        return None
    gamma = getattr(mod, "gamma", None)
    if gamma is None:
        raise TypeError(f"residual-branch scale {type(mod).__name__} has no `gamma`")
    return _f32(gamma, device)


class _BlockWeights:
    """Weights of one pre-LN residual block as flat fp32 device tensors, whatever module layout they were read from:
    ``x += W_o attn(LN1(x) W_qkv^T) ; x += W_pr act(W_fc LN2(x))`` with the in-projection packed ``[q | k | v]``."""

    def __init__(self, width: int, heads: int, act: int):
        self.width, self.heads, self.act = int(width), int(heads), act
        self.head_dim = self.width // self.heads
        if self.head_dim not in _HEAD_DIMS or self.head_dim * self.heads != self.width:
            raise ValueError(f"head_dim {self.head_dim} is not supported (built: {_HEAD_DIMS})")

    def scale_branches(self, g_attn, g_mlp):
        """Fold LayerScale into the projection it follows: gamma * (W x + b) = (gamma W) x + gamma b."""
        if g_attn is not None:
            self.w_o, self.b_o = (self.w_o * g_attn[:, None]).contiguous(), (self.b_o * g_attn).contiguous()
        if g_mlp is not None:
            self.w_pr, self.b_pr = (self.w_pr * g_mlp[:, None]).contiguous(), (self.b_pr * g_mlp).contiguous()
        return self

    def finish(self, split: bool):
        if split and not hasattr(self, "s_qkv"):
            self.s_qkv, self.s_o = N.Split.of(self.w_qkv), N.Split.of(self.w_o)
            self.s_fc, self.s_pr = N.Split.of(self.w_fc), N.Split.of(self.w_pr)
        return self


def _bias(lin: nn.Linear, device):
    return _f32(lin.bias, device) if lin.bias is not None else torch.zeros(lin.out_features, dtype=torch.float32, device=device)


def _read_mha_block(blk: nn.Module, device) -> _BlockWeights:
    """open_clip ``ResidualAttentionBlock`` (and ``synth._Block``): ``ln_1``, ``attn`` = nn.MultiheadAttention with the packed
    in-projection, ``ls_1``, ``ln_2``, ``mlp`` = (Linear, activation, Linear), ``ls_2``."""
    attn = getattr(blk, "attn", None)
    if not isinstance(attn, nn.MultiheadAttention) or attn.in_proj_weight is None or attn.in_proj_bias is None:
        raise TypeError("expected torch.nn.MultiheadAttention blocks with a packed in_proj_weight and bias")
    if hasattr(blk, "ln_1_kv") or getattr(blk, "ln_attn", None) is not None and not _is_identity(blk.ln_attn):
        raise TypeError("cross-attention / scaled-attention blocks are not built")
    linears = [m for m in blk.mlp.modules() if isinstance(m, nn.Linear)]
    others = [m for m in blk.mlp.children() if not isinstance(m, nn.Linear) and not _is_identity(m)]
    if len(linears) != 2 or len(others) != 1:
        raise TypeError("expected an MLP of Linear, activation, Linear")
    w = _BlockWeights(attn.embed_dim, attn.num_heads, _act_code(others[0]))
    w.ln1, w.ln2 = _ln(blk.ln_1, device), _ln(blk.ln_2, device)
    w.w_qkv, w.b_qkv = _f32(attn.in_proj_weight, device), _f32(attn.in_proj_bias, device)
    w.w_o, w.b_o = _f32(attn.out_proj.weight, device), _bias(attn.out_proj, device)
    w.w_fc, w.b_fc = _f32(linears[0].weight, device), _bias(linears[0], device)
    w.w_pr, w.b_pr = _f32(linears[1].weight, device), _bias(linears[1], device)
    return w.scale_branches(_layer_scale(getattr(blk, "ls_1", None), device), _layer_scale(getattr(blk, "ls_2", None), device))


def _read_hf_siglip_block(blk: nn.Module, heads: int, act: int, device) -> _BlockWeights:
    """transformers ``SiglipEncoderLayer``: ``layer_norm1``, ``self_attn`` with separate ``q_proj`` / ``k_proj`` / ``v_proj`` /
    ``out_proj`` (packed here into the ``[q | k | v]`` in-projection the attention kernel reads), ``layer_norm2``,
    ``mlp.fc1`` / ``mlp.fc2``."""
    a = blk.self_attn
    w = _BlockWeights(a.q_proj.in_features, heads, act)
    w.ln1, w.ln2 = _ln(blk.layer_norm1, device), _ln(blk.layer_norm2, device)
    w.w_qkv = torch.cat([_f32(p.weight, device) for p in (a.q_proj, a.k_proj, a.v_proj)], 0).contiguous()
    w.b_qkv = torch.cat([_bias(p, device) for p in (a.q_proj, a.k_proj, a.v_proj)], 0).contiguous()
    w.w_o, w.b_o = _f32(a.out_proj.weight, device), _bias(a.out_proj, device)
    w.w_fc, w.b_fc = _f32(blk.mlp.fc1.weight, device), _bias(blk.mlp.fc1, device)
    w.w_pr, w.b_pr = _f32(blk.mlp.fc2.weight, device), _bias(blk.mlp.fc2, device)
    return w


def _read_timm_block(blk: nn.Module, device) -> _BlockWeights:
    """timm ``vision_transformer.Block``: ``norm1``, ``attn`` {``qkv`` (packed Linear), ``q_norm``, ``k_norm``, ``proj``,
    ``num_heads``}, ``ls1``, ``norm2``, ``mlp`` {``fc1``, ``act``, ``norm``, ``fc2``}, ``ls2``."""
    a = blk.attn
    if not isinstance(getattr(a, "qkv", None), nn.Linear) or not isinstance(getattr(a, "proj", None), nn.Linear):
        raise TypeError("expected timm Attention with packed `qkv` and `proj` Linear layers")
    if not (_is_identity(getattr(a, "q_norm", None)) and _is_identity(getattr(a, "k_norm", None))):
        raise TypeError("timm Attention with q/k normalisation is not built")
    if not _is_identity(getattr(blk.mlp, "norm", None)):
        raise TypeError("timm Mlp with an inner norm is not built")
    w = _BlockWeights(a.qkv.in_features, a.num_heads, _act_code(blk.mlp.act))
    w.ln1, w.ln2 = _ln(blk.norm1, device), _ln(blk.norm2, device)
    w.w_qkv, w.b_qkv = _f32(a.qkv.weight, device), _bias(a.qkv, device)
    w.w_o, w.b_o = _f32(a.proj.weight, device), _bias(a.proj, device)
    w.w_fc, w.b_fc = _f32(blk.mlp.fc1.weight, device), _bias(blk.mlp.fc1, device)
    w.w_pr, w.b_pr = _f32(blk.mlp.fc2.weight, device), _bias(blk.mlp.fc2, device)
    return w.scale_branches(_layer_scale(getattr(blk, "ls1", None), device), _layer_scale(getattr(blk, "ls2", None), device))


def _project(pooled: torch.Tensor, w: torch.Tensor | None, b: torch.Tensor | None, s_w) -> torch.Tensor:
    """The tower's output projection of the pooled rows.  With ``s_w`` (the weight as a split matrix: split mode) it runs in the
    arithmetic of the blocks (``linear3``: a (B, W) x (D, W) product is a few tiles, which the small-grid kernel walks in
    ~6 us; the fp32-MFMA kernel took 65)."""
    if w is None:
        return pooled
    if s_w is None:
        return N.linear(pooled, w, b)
    return N.linear3(N.Split.of(pooled), s_w, b)


class _Tower:
    """L residual blocks over a (B*T, W) fp32 token matrix (the residual stream stays fp32 in both modes)."""

    def __init__(self, blocks: list, split: bool):
        """``blocks``: `_BlockWeights` in execution order (built by one of the layout readers below)."""
        if not blocks:
            raise TypeError("the tower has no residual blocks")
        self.split = split
        self.blocks = [b.finish(split) for b in blocks]
        self.width = self.blocks[0].width
        self.heads = self.blocks[0].heads

    def forward(self, x: torch.Tensor, B: int, T: int, causal: bool, pool_rows: torch.Tensor | None = None) -> torch.Tensor:
        """Runs the blocks over ``x`` in place.  ``pool_rows`` (B int64 row indices): only those rows of the final
        residual stream are needed (class token / end-of-text token), so the last block stops being computed for
        the other rows after its attention (keys and values still come from every token) and ``(B, W)`` is returned.
        Each GEMM row depends on its own input row only, so the pooled rows are bit-identical either way."""
        M, W = x.shape
        F = self.blocks[0].w_fc.shape[0]
        qkv = torch.empty((M, 3 * W), dtype=torch.float32, device=x.device)
        last = len(self.blocks) - 1
        if self.split:
            h, att, hid = N.Split(M, W, x.device), N.Split(M, W, x.device), N.Split(M, F, x.device)
            for i, blk in enumerate(self.blocks):
                N.layernorm(x, *blk.ln1, out_split=h)
                N.linear3(h, blk.s_qkv, blk.b_qkv, out=qkv)
                if i == last and pool_rows is not None:
                    att32 = N.attention(qkv, B, T, blk.heads, blk.head_dim, causal, bf16x3=True)
                    xp = N.gather_rows(x, pool_rows, check=False)
                    N.linear3(N.Split.of(N.gather_rows(att32, pool_rows, check=False)), blk.s_o, blk.b_o, residual=xp, out=xp)
                    hp = N.layernorm(xp, *blk.ln2, out_split=N.Split(B, W, x.device))
                    hidp = N.linear3(hp, blk.s_fc, blk.b_fc, act=blk.act, out_split=N.Split(B, F, x.device))
                    N.linear3(hidp, blk.s_pr, blk.b_pr, residual=xp, out=xp)
                    return xp
                N.attention(qkv, B, T, blk.heads, blk.head_dim, causal, out_split=att, bf16x3=True)  # both products split-bf16 x3
                N.linear3(att, blk.s_o, blk.b_o, residual=x, out=x)  # x += out_proj(attn)
                N.layernorm(x, *blk.ln2, out_split=h)
                N.linear3(h, blk.s_fc, blk.b_fc, act=blk.act, out_split=hid)  # GELU output leaves as split bf16
                N.linear3(hid, blk.s_pr, blk.b_pr, residual=x, out=x)  # x += c_proj(gelu(c_fc))
            return x if pool_rows is None else N.gather_rows(x, pool_rows, check=False)
        h = torch.empty_like(x)
        att = torch.empty_like(x)
        hid = torch.empty((M, F), dtype=torch.float32, device=x.device)
        for i, blk in enumerate(self.blocks):
            N.layernorm(x, *blk.ln1, out=h)
            N.linear(h, blk.w_qkv, blk.b_qkv, out=qkv)
            N.attention(qkv, B, T, blk.heads, blk.head_dim, causal, out=att)
            if i == last and pool_rows is not None:
                xp = N.gather_rows(x, pool_rows, check=False)
                N.linear(N.gather_rows(att, pool_rows, check=False), blk.w_o, blk.b_o, residual=xp, out=xp)
                hp = N.layernorm(xp, *blk.ln2)
                N.linear(N.linear(hp, blk.w_fc, blk.b_fc, act=blk.act), blk.w_pr, blk.b_pr, residual=xp, out=xp)
                return xp
            N.linear(att, blk.w_o, blk.b_o, residual=x, out=x)
            N.layernorm(x, *blk.ln2, out=h)
            N.linear(h, blk.w_fc, blk.b_fc, act=blk.act, out=hid)
            N.linear(hid, blk.w_pr, blk.b_pr, residual=x, out=x)
        return x if pool_rows is None else N.gather_rows(x, pool_rows, check=False)


# ---- a batch as several chunks on several HIP streams (off by default) ---------------------------------------------------
# At B = 256 the ViT-B/32 tower's GEMMs have 150-600 tiles of 256 x 256 for 256 CUs: the out-projection and fc2 fill 59 %
# of the chip, fc1 78 % of its last round.  The samples of a batch are independent all the way through the tower (GEMM
# rows, attention per image), so the batch can be cut in SL_ENC_STREAMS chunks that each run the whole tower on a stream
# of their own, the idle CUs of one chunk's kernel running another's.  Every feature keeps its bits
# (tests/test_gpu_native_clip.py).  Measured (tools/encoder_bench.py, B = 256): 8.56 -> 8.11 ms in round 2, but 7.99 ->
# 8.25 ms once the residual epilogue stopped serialising its loads (round 3): a half batch's GEMMs are 75-300 tiles and run
# the 128 x 128 kernel (o-proj 65 us for HALF the rows against 63 us for all of them).  Default: one stream.
_SIDE_STREAMS: dict = {}


def _enc_streams() -> int:
    import os

    return max(1, min(4, int(os.environ.get("SL_ENC_STREAMS", "1"))))


def _in_chunks(fn, batch: torch.Tensor, rows_per_sample: int, min_rows: int = 4096) -> torch.Tensor:
    """``fn(batch)`` computed as ``SL_ENC_STREAMS`` (default 1 = off) chunks on side streams when every chunk still has at least
    ``min_rows`` token rows; otherwise one call on the current stream."""
    n = _enc_streams()
    B = batch.shape[0]
    if n == 1 or B < 2 * n or (B // n) * rows_per_sample < min_rows:
        return fn(batch)
    dev = batch.device
    key = (dev.index, n)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(dev) for _ in range(n)]
    cur = torch.cuda.current_stream(dev)
    outs = []
    for st, part in zip(_SIDE_STREAMS[key], batch.chunk(n)):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            part.record_stream(st)
            out = fn(part)
        out.record_stream(cur)
        outs.append(out)
    for st in _SIDE_STREAMS[key]:
        cur.wait_stream(st)
    return torch.cat(outs)


class NativeResNetVision:
    """CLIP's ResNet image towers (open_clip ``ModifiedResNet``: ``OpenClip("RN50", ...)``, BASELINE configs[0]'s embed model).
    The convolutional trunk (stem + ``layer1..4``: the user's torch modules, MIOpen) stays on PyTorch like any probed model's
    forward; what follows it runs on the package's kernels: the NCHW map becomes ``(B, HW + 1, C)`` token rows with the mean
    token in front (``sl_tokens_from_map``, positions added), ONE GEMM projects every token to ``[k | v]`` and one the mean
    tokens to the queries, ``sl_attention_pool_q`` attends with one query per image and head, and ``c_proj`` maps to the
    joint space.  Same arithmetic as ``attnpool.forward`` (multi_head_attention_forward with separate projection weights)."""

    def __init__(self, visual: nn.Module, device, split: bool):
        pool = visual.attnpool
        for name in ("positional_embedding", "q_proj", "k_proj", "v_proj", "c_proj", "num_heads"):
            if not hasattr(pool, name):
                raise TypeError(f"visual.attnpool has no `{name}`: not open_clip's AttentionPool2d")
        self.visual = visual
        W = pool.q_proj.in_features
        self.heads, self.width = int(pool.num_heads), W
        self.head_dim = W // self.heads
        if self.head_dim * self.heads != W or self.head_dim % 4 or self.head_dim > 128:
            raise ValueError(f"attention pool: head_dim {self.head_dim} (a multiple of 4 up to 128)")
        self.pos = _f32(pool.positional_embedding, device)  # (HW + 1, W)
        self.w_kv = torch.cat([_f32(pool.k_proj.weight, device), _f32(pool.v_proj.weight, device)]).contiguous()  # (2W, W): rows [k | v]
        self.b_kv = torch.cat([_bias(pool.k_proj, device), _bias(pool.v_proj, device)]).contiguous()
        self.w_q, self.b_q = _f32(pool.q_proj.weight, device), _bias(pool.q_proj, device)
        self.w_c, self.b_c = _f32(pool.c_proj.weight, device), _bias(pool.c_proj, device)
        # The head's four projections run on the fp32-MFMA GEMM (`N.linear`) in EVERY `gemm` mode.  A trunk hands the pool
        # tokens of magnitude ~20, its softmax logits are in the hundreds, and a 3-product bf16 split (per-product relative
        # error 2^-16..2^-17) then moves the attention weights by ~0.5 %: 1.1e-3 absolute on features of scale 22, 6x torch's
        # own fp32 distance from float64 (round-4 driver run).  The GEMMs are (B*50) x 4096 x 2048 at most: microseconds
        # beside the convolutional trunk, so the split buys nothing here.
        self.split = False
        del split

    def _linear(self, x, w, b):
        return N.linear(x, w, b)

    def trunk(self, img: torch.Tensor) -> torch.Tensor:
        """``(B, C, h, w)`` output of ``layer4`` — the model's own forward with the attention pool taken out."""
        v = self.visual
        pool, v.attnpool = v.attnpool, nn.Identity()
        try:
            return v(img)
        finally:
            v.attnpool = pool

    @torch.no_grad()
    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        img = N.to_device(img).to(next(self.visual.parameters()).dtype)
        return self.head(self.trunk(img))

    @torch.no_grad()
    def head(self, fmap: torch.Tensor) -> torch.Tensor:
        """The attention pool + projection of a ``(B, C, h, w)`` trunk output, on the kernels."""
        fmap = N.to_device(fmap).to(torch.float32).contiguous()
        B, C = fmap.shape[:2]
        T = fmap[0, 0].numel() + 1
        if C != self.width or T != self.pos.shape[0]:
            raise ValueError(f"trunk output {tuple(fmap.shape)} does not match the attention pool ({self.pos.shape[0] - 1} positions of width {self.width})")
        tokens = N.tokens_from_map(fmap, self.pos)  # (B, T, W), row 0 = mean token
        rows = tokens.reshape(B * T, C)
        kv = self._linear(rows, self.w_kv, self.b_kv)  # (B*T, 2W)
        q = self._linear(tokens[:, 0].contiguous(), self.w_q, self.b_q)  # (B, W)
        pooled = N.attention_pool_q(q, kv, B, T, self.heads, self.head_dim)
        return self._linear(pooled, self.w_c, self.b_c)


class NativeVisionTower:
    """open_clip ``VisionTransformer`` forward (class token + learned positions, pre-LN blocks) on the kernels.  ``pool``:
    ``"tok"`` (class token) or ``"avg"`` (mean of the patch tokens); ``ln_after_pool`` = open_clip's ``final_ln_after_pool``;
    ``ln_pre`` may be absent (``no_ln_pre``), ``proj`` too."""

    def __init__(self, visual: nn.Module, blocks, device, split: bool, pool: str = "tok", ln_after_pool: bool = False):
        conv = visual.conv1
        if conv.bias is not None or conv.kernel_size != conv.stride:
            raise TypeError("NativeClip expects a bias-free patch embedding with stride == kernel size")
        if pool not in ("tok", "avg"):
            raise TypeError(f"visual.pool_type={pool!r}: only 'tok' and 'avg' are built")
        self.patch = conv.kernel_size[0]
        self.width = conv.out_channels
        self.w_patch = _f32(conv.weight.reshape(self.width, -1), device)  # (W, C*P*P)
        self.cls = _f32(visual.class_embedding, device)
        self.pos = _f32(_first(visual, "positional_embedding", "positional_embedding_v"), device)
        self.ln_pre = None if _is_identity(visual.ln_pre) else _ln(visual.ln_pre, device)
        self.ln_post = _ln(visual.ln_post, device)
        proj = getattr(visual, "proj", None)
        if proj is None:
            proj = getattr(visual, "proj_v", None)
        self.w_proj = _f32(proj.t(), device) if proj is not None else None  # features = x @ proj -> Linear weight (D, W)
        self.s_proj = N.Split.of(self.w_proj) if split and self.w_proj is not None else None
        self.split = split
        if split:
            self.s_patch = N.Split.of(self.w_patch)
        self.tower = _Tower([_read_mha_block(b, device) for b in blocks], split)
        self.pool, self.ln_after_pool = pool, bool(ln_after_pool)
        self.pool_shortcut = True  # last block: out-proj / MLP for the class-token rows only (see _Tower.forward)

    @torch.no_grad()
    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        img = N.to_device(img).to(torch.float32).contiguous()
        return _in_chunks(self._encode, img, self.chunk_rows(img))

    def chunk_rows(self, img) -> int:
        """Token rows one image contributes (what `_in_chunks` sizes its split by)."""
        return (img.shape[2] // self.patch) * (img.shape[3] // self.patch) + 1

    def _encode(self, img: torch.Tensor) -> torch.Tensor:
        B = img.shape[0]
        n_patch = (img.shape[2] // self.patch) * (img.shape[3] // self.patch)
        T = n_patch + 1
        if T != self.pos.shape[0]:
            raise ValueError(f"image gives {T} tokens, positional embedding has {self.pos.shape[0]}")
        W = self.width
        x = torch.empty((B * T, W), dtype=torch.float32, device=img.device)
        # patch embedding GEMM; its epilogue scatters row (b, p) to token row b*T + 1 + p and adds pos[1 + p]
        if self.split:
            patches = N.patchify(img, self.patch, out_split=N.Split(B * n_patch, self.w_patch.shape[1], img.device))
            N.linear3(patches, self.s_patch, out=x, scatter=(n_patch, T, 1), rowadd=self.pos)
        else:
            patches = N.patchify(img, self.patch)
            N.linear(patches, self.w_patch, out=x, scatter=(n_patch, T, 1), rowadd=self.pos)
        N.broadcast_row(self.cls, self.pos[0], B, T * W, x)  # token 0 = class embedding + pos[0]
        h = N.layernorm(x, *self.ln_pre) if self.ln_pre is not None else x
        if self.pool == "tok":
            # LayerNorm is per row, so ln_post before or after picking the class token gives the same bits
            if self.pool_shortcut:
                cls_rows = torch.arange(B, device=img.device, dtype=torch.int64) * T
                pooled = N.layernorm(self.tower.forward(h, B, T, causal=False, pool_rows=cls_rows), *self.ln_post)
            else:
                h = self.tower.forward(h, B, T, causal=False)
                pooled = N.layernorm(h, *self.ln_post, rows=B, x_row_stride=T * W)  # class-token rows only
        else:  # "avg": mean over the patch tokens (K2's token mean on the (B, T-1, W) view), ln_post on either side of it
            h = self.tower.forward(h, B, T, causal=False)
            if not self.ln_after_pool:
                h = N.layernorm(h, *self.ln_post)
            pooled = torch.empty((B, W), dtype=torch.float32, device=img.device)
            N.reduce_tokens(h.view(B, T, W)[:, 1:], N.SL_TOK_MEAN, 0, None, pooled)
            if self.ln_after_pool:
                pooled = N.layernorm(pooled, *self.ln_post)
        return _project(pooled, self.w_proj, None, self.s_proj)


class NativeTextTower:
    """open_clip text tower (``CLIP``'s flattened members or a ``TextTransformer``) on the kernels: token + position
    embeddings, pre-LN blocks (causal iff the module carries an ``attn_mask``), ``ln_final``, pooling at the end-of-text
    token (``"argmax"``: CLIP's tokenizer gives it the highest id), the first or the last position, then the projection
    (a matrix, an ``nn.Linear`` with bias, or none)."""

    def __init__(self, model: nn.Module, blocks, device, split: bool, pool: str = "argmax", causal: bool = True):
        if pool not in ("argmax", "first", "last"):
            raise TypeError(f"text pool_type={pool!r}: only 'argmax', 'first' and 'last' are built")
        if getattr(model, "cls_emb", None) is not None:
            raise TypeError("text towers with a learned class embedding (CoCa) are not built")
        self.table = _f32(model.token_embedding.weight, device)
        self.pos = _f32(_first(model, "positional_embedding", "positional_embedding_t"), device)
        self.ln_final = _ln(model.ln_final, device)
        proj = getattr(model, "text_projection", None)
        if proj is None:
            proj = getattr(model, "proj_t", None)
        self.b_proj = None
        if isinstance(proj, nn.Linear):
            self.w_proj, self.b_proj = _f32(proj.weight, device), (_f32(proj.bias, device) if proj.bias is not None else None)
        else:
            self.w_proj = _f32(proj.t(), device) if proj is not None else None
        self.s_proj = N.Split.of(self.w_proj) if split and self.w_proj is not None else None
        self.tower = _Tower([_read_mha_block(b, device) for b in blocks], split)
        self.pool, self.causal = pool, bool(causal)
        self.truncate = True  # skip the positions after the batch's last end-of-text token (see __call__)
        self.pool_shortcut = True  # last block: out-proj / MLP for the pooled rows only

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = N.to_device(tokens).to(torch.int64).contiguous()
        B, T = tokens.shape
        if T > self.pos.shape[0]:
            raise ValueError(f"{T} tokens exceed the text tower's context length {self.pos.shape[0]}")
        if self.pool == "argmax":
            at = tokens.argmax(dim=-1)  # end-of-text token (highest id): the position CLIP's text tower pools
            if self.truncate and self.causal and B > 0:
                # the tower is causal: position t only sees positions <= t, so nothing after the last end-of-text token of
                # the batch can reach a pooled feature.  Dropping those (padding) positions leaves every pooled value
                # bit-identical and shrinks the work from context_length to the longest prompt (one scalar readback).
                t_eff = int(at.max().item()) + 1
                if t_eff < T:
                    tokens, T = tokens[:, :t_eff].contiguous(), t_eff
        elif self.pool == "first":
            at = torch.zeros(B, dtype=torch.int64, device=tokens.device)
        else:
            at = torch.full((B,), T - 1, dtype=torch.int64, device=tokens.device)
        x = N.embed_tokens(self.table, tokens, self.pos[:T].contiguous())
        rows = torch.arange(B, device=tokens.device) * T + at
        if self.pool_shortcut:
            picked = self.tower.forward(x, B, T, causal=self.causal, pool_rows=rows)
        else:
            picked = N.gather_rows(self.tower.forward(x, B, T, causal=self.causal), rows, check=False)
        pooled = N.layernorm(picked, *self.ln_final)
        return _project(pooled, self.w_proj, self.b_proj, self.s_proj)


def _require_clip_layout(model):
    """open_clip builds many variants behind the same attribute names.  What the native towers implement is listed in the
    module docstring; anything else is refused here instead of silently producing different features."""
    v = model.visual
    problems = []
    if getattr(v, "attn_pool", None) is not None:
        problems.append("visual.attn_pool is set (attention pooling)")
    if getattr(v, "pool_type", "tok") not in ("tok", "avg"):
        problems.append(f"visual.pool_type={getattr(v, 'pool_type', None)!r} (only 'tok' and 'avg')")
    if getattr(model, "text_pool_type", "argmax") not in ("argmax", "first", "last"):
        problems.append(f"text_pool_type={getattr(model, 'text_pool_type', None)!r} (only 'argmax', 'first', 'last')")
    if problems:
        raise TypeError("NativeClip does not implement this open_clip variant: " + "; ".join(sorted(set(problems))))


class NativeClip(AbstractVLM):
    """``AbstractVLM`` running ``base``'s CLIP towers on HIP kernels; ``base`` keeps tokenizer + preprocessing."""

    # The embed stage of the concept-DB build may hold preprocessed batches back until this many images are there: the
    # towers' GEMMs reach their rate from a few thousand rows, and an embedding does not depend on its batch (bit for bit).
    embed_accumulate = 256

    def __init__(self, base, device=None, gemm: str = "bf16x3", preprocess=None):
        """``preprocess``: ``None`` keeps ``base.preprocess`` (host transform, as upstream); a
        ``DevicePreprocess`` (or ``"device"`` to derive one from ``base.preprocessor``) runs resize / crop /
        normalise for the whole batch on the device (K12)."""
        if gemm not in ("bf16x3", "f32"):
            raise ValueError("gemm must be 'bf16x3' or 'f32'")
        split = gemm == "bf16x3"
        self.base = base
        model = base.model
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            dev = N.default_device()
        self._device = dev
        base.to(dev)
        visual = getattr(model, "visual", None)
        if visual is not None and hasattr(visual, "attnpool") and hasattr(visual, "layer4"):
            # CLIP ResNet (open_clip ModifiedResNet): conv trunk on PyTorch, attention pool + projection on the kernels;
            # the text tower in open_clip's layout (members on the model) or synth's (`model.text.blocks`)
            self.vision = NativeResNetVision(visual, dev, split)
            if hasattr(model, "transformer") and hasattr(model.transformer, "resblocks"):
                pool = getattr(model, "text_pool_type", "argmax")
                self.text = NativeTextTower(model, model.transformer.resblocks, dev, split, pool=pool,
                                            causal=getattr(model, "attn_mask", None) is not None)
            elif hasattr(getattr(model, "text", None), "blocks"):
                self.text = NativeTextTower(model, model.text.blocks, dev, split)
            else:
                raise TypeError("NativeClip: CLIP-ResNet image tower with a text tower that is neither open_clip's nor synth's layout")
        elif visual is not None and hasattr(visual, "conv1") and hasattr(visual, "transformer"):
            # open_clip CLIP: `visual` is the whole image tower, the text tower's members sit on the model itself
            # (CustomTextCLIP keeps them under `.text`)
            _require_clip_layout(model)
            self.vision = NativeVisionTower(visual, visual.transformer.resblocks, dev, split, pool=getattr(visual, "pool_type", "tok"),
                                            ln_after_pool=getattr(visual, "final_ln_after_pool", False))
            text = model.text if hasattr(model, "text") and hasattr(model.text, "transformer") else model
            if not hasattr(getattr(text, "transformer", None), "resblocks"):
                # e.g. open_clip's CustomTextCLIP around a Hugging Face text encoder: `.text.transformer` is an HF model
                raise TypeError(f"NativeClip reads open_clip's TextTransformer layout (`transformer.resblocks`); the text tower is "
                                f"{type(text).__name__} around {type(getattr(text, 'transformer', None)).__name__} — run this model "
                                "through its own torch modules (OpenClip without .native())")
            pool = getattr(text, "pool_type", None) or getattr(model, "text_pool_type", "argmax")
            self.text = NativeTextTower(text, text.transformer.resblocks, dev, split, pool=pool,
                                        causal=getattr(text, "attn_mask", None) is not None)
        elif visual is not None and hasattr(model, "conv1") and hasattr(visual, "blocks"):
            # synth._ClipModel: embedding members on the model, block stacks in .visual / .text
            self.vision = NativeVisionTower(_SynthVisual(model), visual.blocks, dev, split)
            self.text = NativeTextTower(model, model.text.blocks, dev, split)
        else:
            found = type(visual).__name__ if visual is not None else "no `visual` member"
            raise TypeError(f"NativeClip reads open_clip's CLIP / VisionTransformer layout; {type(model).__name__} has {found} "
                            "(timm trunks: NativeSigLip; convolutional / hybrid towers such as MobileCLIP are not built)")
        self.name = f"native-{gemm}-" + getattr(base, "name", type(base).__name__)
        if preprocess == "device":
            from semanticlens_amd.foundation_models.preprocess import DevicePreprocess

            preprocess = DevicePreprocess.from_transform(base.preprocessor)
        self._preprocess = preprocess.to(dev) if preprocess is not None else None

    @property
    def device(self):
        return self._device

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise N.NativeLibraryError("NativeClip runs on a HIP device only")
        return self

    def encode_image(self, img):
        return self.vision(img)

    def encode_text(self, tokens):
        return self.text(tokens)

    def preprocess(self, img):
        if self._preprocess is not None:
            out = self._preprocess(img)
            return out.unsqueeze(0) if out.ndim == 3 else out  # clip.py:160-162: a single image gets a batch axis
        return self.base.preprocess(img)

    def tokenize(self, txt, *args, **kwargs):
        return self.base.tokenize(txt, *args, **kwargs)


class _SynthVisual:
    """Adapter giving ``synth._ClipModel``'s image-tower members the names of open_clip's ``visual`` module."""

    def __init__(self, m):
        self.conv1, self.class_embedding = m.conv1, m.class_embedding
        self.positional_embedding = m.positional_embedding_v
        self.ln_pre, self.ln_post, self.proj = m.ln_pre, m.ln_post, m.proj_v


# ------------------------------------------------------------------------------------------------------------------
# SigLIP-layout towers (reference: foundation_models/clip.py:190-211 `SigLipV2`; BASELINE configs[3] names SigLIP-so400m)
# ------------------------------------------------------------------------------------------------------------------
class _MapHead:
    """Weights of a MAP head (one learned probe attends over all tokens, out-projection, LayerNorm + MLP residual branch):
    transformers' ``SiglipMultiheadAttentionPoolingHead`` and timm's ``AttentionPoolLatent`` are the same computation."""

    def __init__(self, probe, wq, bq, wkv, bkv, w_o, b_o, ln, w1, b1, w2, b2, heads: int, act: int):
        W = probe.numel()
        self.heads, self.head_dim, self.act = int(heads), W // int(heads), act
        if self.head_dim not in _HEAD_DIMS or self.head_dim * self.heads != W:
            raise ValueError(f"MAP head: head_dim {self.head_dim} is not supported (built: {_HEAD_DIMS})")
        # the probe is the same for every image: its query projection is a constant of the model
        self.q_probe = N.linear(probe.reshape(1, W).contiguous(), wq, bq).reshape(W).contiguous()
        self.w_kv, self.b_kv = wkv, bkv  # (2W, W): rows [k | v]
        self.w_o, self.b_o, self.ln, self.w1, self.b1, self.w2, self.b2 = w_o, b_o, ln, w1, b1, w2, b2

    @classmethod
    def from_transformers(cls, head: nn.Module, act: int, device):
        mha: nn.MultiheadAttention = head.attention
        W = mha.embed_dim
        ipw, ipb = _f32(mha.in_proj_weight, device), _f32(mha.in_proj_bias, device)
        return cls(_f32(head.probe, device), ipw[:W].contiguous(), ipb[:W].contiguous(), ipw[W:].contiguous(), ipb[W:].contiguous(),
                   _f32(mha.out_proj.weight, device), _bias(mha.out_proj, device), _ln(head.layernorm, device),
                   _f32(head.mlp.fc1.weight, device), _bias(head.mlp.fc1, device), _f32(head.mlp.fc2.weight, device),
                   _bias(head.mlp.fc2, device), mha.num_heads, act)

    @classmethod
    def from_timm(cls, pool: nn.Module, device):
        """timm ``AttentionPoolLatent``: ``latent`` (1, 1, W), ``q``, ``kv`` (outputs ``[k | v]``), ``proj``, ``norm``, ``mlp``."""
        if getattr(pool, "latent_len", 1) != 1 or getattr(pool, "pos_embed", None) is not None:
            raise TypeError("timm AttentionPoolLatent with several latents or its own position embedding is not built")
        if not (_is_identity(getattr(pool, "q_norm", None)) and _is_identity(getattr(pool, "k_norm", None))):
            raise TypeError("timm AttentionPoolLatent with q/k normalisation is not built")
        if getattr(pool, "pool", "token") != "token":
            raise TypeError(f"timm AttentionPoolLatent pool={pool.pool!r}: only 'token'")
        return cls(_f32(pool.latent, device), _f32(pool.q.weight, device), _bias(pool.q, device), _f32(pool.kv.weight, device),
                   _bias(pool.kv, device), _f32(pool.proj.weight, device), _bias(pool.proj, device), _ln(pool.norm, device),
                   _f32(pool.mlp.fc1.weight, device), _bias(pool.mlp.fc1, device), _f32(pool.mlp.fc2.weight, device),
                   _bias(pool.mlp.fc2, device), pool.num_heads, _act_code(pool.mlp.act))


class NativeSigLipVision:
    """Image tower: patch embedding WITH bias, learned positions, no class token, non-causal blocks, a final LayerNorm,
    then the MAP head (``sl_attention_pool``); the pooled token is the image feature, optionally followed by a Linear
    (open_clip ``TimmModel.head.proj``)."""

    def __init__(self, conv: nn.Conv2d, pos: torch.Tensor, blocks: list, ln_post, head: _MapHead, device, split: bool,
                 final_proj: nn.Linear | None = None):
        if conv.kernel_size != conv.stride:
            raise TypeError("NativeSigLip expects a patch embedding with stride == kernel size")
        self.patch = conv.kernel_size[0]
        self.width = conv.out_channels
        self.w_patch = _f32(conv.weight.reshape(self.width, -1), device)
        self.b_patch = _f32(conv.bias, device) if conv.bias is not None else None
        self.pos = _f32(pos.reshape(-1, self.width), device)  # (n_patches, W)
        self.split = split
        if split:
            self.s_patch = N.Split.of(self.w_patch)
        self.tower = _Tower(blocks, split)
        self.ln_post = ln_post
        self.head = head
        self.w_final = _f32(final_proj.weight, device) if final_proj is not None else None
        self.b_final = _f32(final_proj.bias, device) if final_proj is not None and final_proj.bias is not None else None
        if split:  # the head's GEMMs in the arithmetic of the blocks: its (B, W) products are small grids (gemm_skinny.hpp)
            self.s_kv = N.Split.of(head.w_kv)
            self.s_o, self.s_1, self.s_2 = N.Split.of(head.w_o), N.Split.of(head.w1), N.Split.of(head.w2)
            self.s_final = N.Split.of(self.w_final) if self.w_final is not None else None

    @classmethod
    def from_transformers(cls, vm: nn.Module, cfg, device, split: bool):
        """transformers ``SiglipVisionTransformer``: ``embeddings``, ``encoder.layers``, ``post_layernorm``, ``head``."""
        act = _act_code(getattr(cfg, "hidden_act", "gelu_pytorch_tanh"))
        blocks = [_read_hf_siglip_block(b, cfg.num_attention_heads, act, device) for b in vm.encoder.layers]
        return cls(vm.embeddings.patch_embedding, vm.embeddings.position_embedding.weight, blocks, _ln(vm.post_layernorm, device),
                   _MapHead.from_transformers(vm.head, act, device), device, split)

    @classmethod
    def from_open_clip_timm(cls, visual: nn.Module, device, split: bool):
        """open_clip ``TimmModel``: ``trunk`` = timm ``VisionTransformer`` built with ``global_pool="map"``, ``head`` = the
        (possibly empty) projection ``nn.Sequential``."""
        t = visual.trunk
        problems = []
        if getattr(t, "global_pool", "map") != "map" or getattr(t, "attn_pool", None) is None:
            problems.append(f"trunk.global_pool={getattr(t, 'global_pool', None)!r} (only the MAP head 'map')")
        if getattr(t, "cls_token", None) is not None or getattr(t, "reg_token", None) is not None:
            problems.append("the trunk has class / register tokens")
        for name in ("norm_pre", "fc_norm"):
            if not _is_identity(getattr(t, name, None)):
                problems.append(f"trunk.{name} is not Identity")
        pe = t.patch_embed
        if not _is_identity(getattr(pe, "norm", None)) or not isinstance(getattr(pe, "proj", None), nn.Conv2d):
            problems.append("patch_embed is not a plain strided convolution")
        if not _is_identity(getattr(t, "head", None)):
            problems.append("trunk.head is not Identity (the classifier of the timm model)")
        final = None
        for name, m in (visual.head.named_children() if hasattr(visual, "head") else ()):
            if isinstance(m, nn.Linear) and final is None:
                final = m
            elif not _is_identity(m):
                problems.append(f"visual.head.{name} is a {type(m).__name__}")
        if problems:
            raise TypeError("NativeSigLip does not implement this timm trunk: " + "; ".join(problems))
        blocks = [_read_timm_block(b, device) for b in t.blocks]
        return cls(pe.proj, t.pos_embed, blocks, _ln(t.norm, device), _MapHead.from_timm(t.attn_pool, device), device, split, final)

    @torch.no_grad()
    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        img = N.to_device(img).to(torch.float32).contiguous()
        return _in_chunks(self._encode, img, (img.shape[2] // self.patch) * (img.shape[3] // self.patch))

    def _encode(self, img: torch.Tensor) -> torch.Tensor:
        B = img.shape[0]
        T = (img.shape[2] // self.patch) * (img.shape[3] // self.patch)
        if T != self.pos.shape[0]:
            raise ValueError(f"image gives {T} patches, the positional embedding has {self.pos.shape[0]}")
        W = self.width
        hd = self.head
        x = torch.empty((B * T, W), dtype=torch.float32, device=img.device)
        # patch embedding GEMM (+ bias); its epilogue writes row (b, p) to token row b*T + p and adds pos[p]
        if self.split:
            patches = N.patchify(img, self.patch, out_split=N.Split(B * T, self.w_patch.shape[1], img.device))
            N.linear3(patches, self.s_patch, self.b_patch, out=x, scatter=(T, T, 0), rowadd=self.pos)
        else:
            N.linear(N.patchify(img, self.patch), self.w_patch, self.b_patch, out=x, scatter=(T, T, 0), rowadd=self.pos)
        x = self.tower.forward(x, B, T, causal=False)
        if self.split:
            h = N.layernorm(x, *self.ln_post, out_split=N.Split(B * T, W, img.device))
            kv = N.linear3(h, self.s_kv, hd.b_kv)
        else:
            kv = N.linear(N.layernorm(x, *self.ln_post), hd.w_kv, hd.b_kv)
        pooled = N.attention_pool(hd.q_probe, kv, B, T, hd.heads, hd.head_dim)  # (B, W)
        if self.split:
            dev = img.device
            res = N.linear3(N.Split.of(pooled), self.s_o, hd.b_o)
            hid = N.linear3(N.layernorm(res, *hd.ln, out_split=N.Split(B, W, dev)), self.s_1, hd.b1, act=hd.act,
                            out_split=N.Split(B, hd.w1.shape[0], dev))
            out = N.linear3(hid, self.s_2, hd.b2, residual=res, out=res)
            return N.linear3(N.Split.of(out), self.s_final, self.b_final) if self.w_final is not None else out
        res = N.linear(pooled, hd.w_o, hd.b_o)
        hid = N.linear(N.layernorm(res, *hd.ln), hd.w1, hd.b1, act=hd.act)
        out = N.linear(hid, hd.w2, hd.b2, residual=res, out=res)
        return N.linear(out, self.w_final, self.b_final) if self.w_final is not None else out


class NativeSigLipText:
    """transformers ``SiglipTextTransformer``: token + position embeddings, NON-causal blocks (SigLIP pads to the context
    length and does not mask), ``final_layer_norm``, the LAST position pooled, then the ``head`` Linear (with bias)."""

    def __init__(self, tm: nn.Module, cfg, device, split: bool):
        self.table = _f32(tm.embeddings.token_embedding.weight, device)
        self.pos = _f32(tm.embeddings.position_embedding.weight, device)
        act = _act_code(getattr(cfg, "hidden_act", ""))
        self.tower = _Tower([_read_hf_siglip_block(b, cfg.num_attention_heads, act, device) for b in tm.encoder.layers], split)
        self.ln_final = _ln(tm.final_layer_norm, device)
        self.w_head, self.b_head = _f32(tm.head.weight, device), _f32(tm.head.bias, device)
        self.s_head = N.Split.of(self.w_head) if split else None

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = N.to_device(tokens).to(torch.int64).contiguous()
        B, T = tokens.shape
        if T > self.pos.shape[0]:
            raise ValueError(f"{T} tokens exceed the {self.pos.shape[0]} positions of the text tower")
        x = N.embed_tokens(self.table, tokens, self.pos[:T].contiguous())
        rows = torch.arange(B, device=tokens.device, dtype=torch.int64) * T + (T - 1)
        picked = self.tower.forward(x, B, T, causal=False, pool_rows=rows)
        return _project(N.layernorm(picked, *self.ln_final), self.w_head, self.b_head, self.s_head)


def friendly_batch(tokens: int, width: int, mlp: int, device=None, lo: int = 64, hi: int = 256) -> int:
    """Images per encoder call (``embed_accumulate``) whose token rows fill whole rounds of 256 x 256 GEMM tiles on the part.

    A GEMM of ``tm x tn`` tiles takes ``ceil(tm tn / CUs)`` rounds whatever its last round holds: SigLIP-so400m at 64 images
    (16 384 rows) runs its out-projection and fc2 (N = 1152: 5 tile columns) as 320 tiles = 2 rounds for 1.25 rounds of work
    (296 of the 411 TFLOP/s the same kernels reach at 256 images, where every GEMM is a whole number of rounds:
    ``profiles/r04_siglip_b64_kernel_stats.csv`` / ``_b256_``).  An embedding does not depend on the batch it is computed in, so
    the embed stage is free to pick the count: the one in ``[lo, hi]`` (multiples of 8) with the fewest tile rounds per image over the four
    GEMMs of a block."""
    cus = 256
    if device is not None and torch.cuda.is_available():
        cus = torch.cuda.get_device_properties(device).multi_processor_count or 256
    tns = [-(-3 * width // 256), -(-width // 256), -(-mlp // 256), -(-width // 256)]
    best, best_cost = hi, float("inf")
    for b in range(lo, hi + 1, 8):
        tm = -(-b * tokens // 256)
        cost = sum(-(-tm * tn // cus) for tn in tns) / b
        if cost < best_cost - 1e-12:
            best, best_cost = b, cost
    return best


class NativeSigLip(AbstractVLM):
    """``AbstractVLM`` running the towers of a SigLIP-layout model on HIP kernels.

    ``base.model`` is either what the reference's ``SigLipV2`` constructs (foundation_models/clip.py:190-211: open_clip's
    ``CustomTextCLIP`` — ``visual`` = ``TimmModel`` around a timm ViT trunk with a MAP head, ``text`` = open_clip's
    ``TextTransformer``, non-causal, pooled at the last position, projection with bias) or transformers' ``SiglipModel``
    (``vision_model`` / ``text_model``; the geometry of SigLIP-so400m — width 1152, head_dim 72 — is what BASELINE
    configs[3] names).  Tokenizer and host preprocessing stay ``base``'s; ``preprocess`` as for :class:`NativeClip`."""

    embed_accumulate = 64  # see NativeClip; replaced per instance in __init__ (tile-friendly row count for the tower's geometry)

    def __init__(self, base, device=None, gemm: str = "bf16x3", preprocess=None):
        if gemm not in ("bf16x3", "f32"):
            raise ValueError("gemm must be 'bf16x3' or 'f32'")
        model = base.model
        hf = hasattr(model, "vision_model") and hasattr(model, "text_model") and hasattr(model.vision_model, "head")
        oc = hasattr(model, "visual") and hasattr(model.visual, "trunk") and hasattr(model, "text") and hasattr(model.text, "transformer")
        if not (hf or oc):
            raise TypeError("NativeSigLip expects transformers' SiglipModel layout (`vision_model` with a MAP `head`, `text_model`) "
                            "or open_clip's CustomTextCLIP layout (`visual.trunk` = timm ViT, `text` = TextTransformer)")
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            dev = N.default_device()
        self._device = dev
        self.base = base
        base.to(dev)
        split = gemm == "bf16x3"
        if hf:
            self.vision = NativeSigLipVision.from_transformers(model.vision_model, model.config.vision_config, dev, split)
            self.text = NativeSigLipText(model.text_model, model.config.text_config, dev, split)
        else:
            self.vision = NativeSigLipVision.from_open_clip_timm(model.visual, dev, split)
            text = model.text
            self.text = NativeTextTower(text, text.transformer.resblocks, dev, split, pool=getattr(text, "pool_type", "argmax"),
                                        causal=getattr(text, "attn_mask", None) is not None)
        self.name = f"native-{gemm}-" + getattr(base, "name", type(base).__name__)
        self.embed_accumulate = friendly_batch(self.vision.pos.shape[0], self.vision.width, self.vision.tower.blocks[0].w_fc.shape[0], dev)
        if preprocess == "device":
            from semanticlens_amd.foundation_models.preprocess import DevicePreprocess

            preprocess = DevicePreprocess.from_transform(base.preprocessor)
        self._preprocess = preprocess.to(dev) if preprocess is not None else None

    @property
    def device(self):
        return self._device

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise N.NativeLibraryError("NativeSigLip runs on a HIP device only")
        return self

    def encode_image(self, img):
        return self.vision(img)

    def encode_text(self, tokens):
        return self.text(tokens)

    def preprocess(self, img):
        if self._preprocess is not None:
            out = self._preprocess(img)
            return out.unsqueeze(0) if out.ndim == 3 else out
        return self.base.preprocess(img)

    def tokenize(self, txt, *args, **kwargs):
        return self.base.tokenize(txt, *args, **kwargs)


class NativeTextClip(AbstractVLM):
    """The wrapped model with its TEXT tower on the kernels and its image tower left to PyTorch — for CLIP models whose image
    tower the native classes do not read: MobileCLIP (``ClipMobile``, foundation_models/clip.py:214-247 — the reference
    tutorial's foundation model: a FastViT hybrid of re-parameterised convolutions behind open_clip's ``TimmModel``) and any
    other convolutional / hybrid tower.  ``base.model`` is open_clip's ``CustomTextCLIP`` (``text`` = ``TextTransformer``) or
    ``CLIP`` (text members on the model).  Text probing — ``Lens.text_probing``'s 10 000-prompt sweeps — runs natively;
    ``encode_image`` is ``base``'s own (the embed stage treats it like any user ``AbstractVLM``).  ``preprocess`` as for
    :class:`NativeClip`."""

    def __init__(self, base, device=None, gemm: str = "bf16x3", preprocess=None):
        if gemm not in ("bf16x3", "f32"):
            raise ValueError("gemm must be 'bf16x3' or 'f32'")
        model = base.model
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            dev = N.default_device()
        self._device, self.base = dev, base
        base.to(dev)
        text = model.text if hasattr(getattr(model, "text", None), "transformer") else model
        if not hasattr(getattr(text, "transformer", None), "resblocks") or not hasattr(text, "token_embedding"):
            raise TypeError(f"NativeTextClip reads open_clip's TextTransformer layout (`token_embedding`, `transformer.resblocks`, "
                            f"`ln_final`, `text_projection`); found {type(text).__name__}")
        pool = getattr(text, "pool_type", None) or getattr(model, "text_pool_type", "argmax")
        if pool not in ("argmax", "first", "last"):
            raise TypeError(f"text pool_type={pool!r} (only 'argmax', 'first', 'last')")
        self.text = NativeTextTower(text, text.transformer.resblocks, dev, gemm == "bf16x3", pool=pool,
                                    causal=getattr(text, "attn_mask", None) is not None)
        self.name = f"native-text-{gemm}-" + getattr(base, "name", type(base).__name__)
        if preprocess == "device":
            from semanticlens_amd.foundation_models.preprocess import DevicePreprocess

            preprocess = DevicePreprocess.from_transform(base.preprocessor)
        self._preprocess = preprocess.to(dev) if preprocess is not None else None

    @property
    def device(self):
        return self._device

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise N.NativeLibraryError("NativeTextClip runs on a HIP device only")
        return self

    def encode_image(self, img):
        return self.base.encode_image(img)

    def encode_text(self, tokens):
        return self.text(tokens)

    def preprocess(self, img):
        if self._preprocess is not None:
            out = self._preprocess(img)
            return out.unsqueeze(0) if out.ndim == 3 else out
        return self.base.preprocess(img)

    def tokenize(self, txt, *args, **kwargs):
        return self.base.tokenize(txt, *args, **kwargs)


def native_model(base, gemm: str = "bf16x3", preprocess=None, image_tower: str = "native"):
    """The native counterpart of a wrapped open_clip / transformers model, chosen by its module layout.  ``image_tower``:
    ``"native"`` (refuse a tower the native classes do not read), ``"auto"`` (fall back to :class:`NativeTextClip` — torch image
    tower, native text tower — when only the image tower is refused), ``"torch"`` (always that)."""
    if image_tower not in ("native", "auto", "torch"):
        raise ValueError("image_tower must be 'native', 'auto' or 'torch'")
    if image_tower == "torch":
        return NativeTextClip(base, gemm=gemm, preprocess=preprocess)
    model = base.model
    try:
        if hasattr(model, "vision_model") or hasattr(getattr(model, "visual", None), "trunk"):
            return NativeSigLip(base, gemm=gemm, preprocess=preprocess)
        return NativeClip(base, gemm=gemm, preprocess=preprocess)
    except TypeError as err:
        if image_tower != "auto":
            raise
        try:
            return NativeTextClip(base, gemm=gemm, preprocess=preprocess)
        except TypeError:
            raise err from None
