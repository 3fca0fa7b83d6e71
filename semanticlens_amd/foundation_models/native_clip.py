"""NativeClip — a CLIP-family foundation model whose towers run on the package's own HIP kernels.

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

Supported layout (probed by attribute, the names open_clip's ``VisionTransformer`` / ``TextTransformer`` and
``synth.SyntheticClip`` use): ``conv1`` (patch embedding, no bias), ``class_embedding``, positional embedding,
``ln_pre``, blocks with ``ln_1`` / ``attn`` (``torch.nn.MultiheadAttention``) / ``ln_2`` / ``mlp`` (Linear, GELU or
QuickGELU, Linear), ``ln_post`` / ``ln_final``, projection matrices; head_dim 32 / 64 / 72 / 80 / 88 / 96 / 104 / 128.
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


class _Block:
    """Weights of one pre-LN residual block, as flat fp32 device tensors."""

    def __init__(self, blk: nn.Module, device, split: bool):
        attn: nn.MultiheadAttention = blk.attn
        if not isinstance(attn, nn.MultiheadAttention) or attn.in_proj_weight is None:
            raise TypeError("NativeClip expects torch.nn.MultiheadAttention blocks with a packed in_proj_weight")
        self.heads = attn.num_heads
        self.width = attn.embed_dim
        self.head_dim = self.width // self.heads
        if self.head_dim not in (32, 64, 72, 80, 88, 96, 104, 128) or self.head_dim * self.heads != self.width:
            raise ValueError(f"head_dim {self.head_dim} is not supported (32, 64, 72, 80, 88, 96, 104, 128)")
        self.ln1 = (_f32(blk.ln_1.weight, device), _f32(blk.ln_1.bias, device), blk.ln_1.eps)
        self.ln2 = (_f32(blk.ln_2.weight, device), _f32(blk.ln_2.bias, device), blk.ln_2.eps)
        self.w_qkv, self.b_qkv = _f32(attn.in_proj_weight, device), _f32(attn.in_proj_bias, device)
        self.w_o, self.b_o = _f32(attn.out_proj.weight, device), _f32(attn.out_proj.bias, device)
        linears = [m for m in blk.mlp.modules() if isinstance(m, nn.Linear)]
        if len(linears) != 2:
            raise TypeError("NativeClip expects an MLP of two Linear layers")
        self.w_fc, self.b_fc = _f32(linears[0].weight, device), _f32(linears[0].bias, device)
        self.w_pr, self.b_pr = _f32(linears[1].weight, device), _f32(linears[1].bias, device)
        acts = [m for m in blk.mlp.modules() if not isinstance(m, (nn.Linear, nn.Sequential)) and m is not blk.mlp]
        name = type(acts[0]).__name__.lower() if acts else "gelu"
        self.act = N.SL_ACT_QUICKGELU if "quick" in name else N.SL_ACT_GELU
        if split:
            self.s_qkv, self.s_o = N.Split.of(self.w_qkv), N.Split.of(self.w_o)
            self.s_fc, self.s_pr = N.Split.of(self.w_fc), N.Split.of(self.w_pr)


class _Tower:
    """L residual blocks over a (B*T, W) fp32 token matrix (the residual stream stays fp32 in both modes)."""

    def __init__(self, blocks, device, split: bool):
        self.split = split
        self.blocks = [_Block(b, device, split) for b in blocks]
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
                    att32 = N.attention(qkv, B, T, blk.heads, blk.head_dim, causal)
                    xp = N.gather_rows(x, pool_rows, check=False)
                    N.linear3(N.Split.of(N.gather_rows(att32, pool_rows, check=False)), blk.s_o, blk.b_o, residual=xp, out=xp)
                    hp = N.layernorm(xp, *blk.ln2, out_split=N.Split(B, W, x.device))
                    hidp = N.linear3(hp, blk.s_fc, blk.b_fc, act=blk.act, out_split=N.Split(B, F, x.device))
                    N.linear3(hidp, blk.s_pr, blk.b_pr, residual=xp, out=xp)
                    return xp
                N.attention(qkv, B, T, blk.heads, blk.head_dim, causal, out_split=att)
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


class NativeVisionTower:
    def __init__(self, visual: nn.Module, blocks, device, split: bool):
        conv = visual.conv1
        if conv.bias is not None or conv.kernel_size != conv.stride:
            raise TypeError("NativeClip expects a bias-free patch embedding with stride == kernel size")
        self.patch = conv.kernel_size[0]
        self.width = conv.out_channels
        self.w_patch = _f32(conv.weight.reshape(self.width, -1), device)  # (W, C*P*P)
        self.cls = _f32(visual.class_embedding, device)
        self.pos = _f32(_first(visual, "positional_embedding", "positional_embedding_v"), device)
        ln_pre, ln_post = visual.ln_pre, visual.ln_post
        self.ln_pre = (_f32(ln_pre.weight, device), _f32(ln_pre.bias, device), ln_pre.eps)
        self.ln_post = (_f32(ln_post.weight, device), _f32(ln_post.bias, device), ln_post.eps)
        proj = _first(visual, "proj", "proj_v")
        self.w_proj = _f32(proj.t(), device)  # features = x @ proj  ->  Linear weight (D, W)
        self.split = split
        if split:
            self.s_patch = N.Split.of(self.w_patch)
        self.tower = _Tower(blocks, device, split)
        self.pool_shortcut = True  # last block: out-proj / MLP for the class-token rows only (see _Tower.forward)

    @torch.no_grad()
    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        img = N.to_device(img).to(torch.float32).contiguous()
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
        h = N.layernorm(x, *self.ln_pre)
        if self.pool_shortcut:
            cls_rows = torch.arange(B, device=img.device, dtype=torch.int64) * T
            pooled = N.layernorm(self.tower.forward(h, B, T, causal=False, pool_rows=cls_rows), *self.ln_post)
        else:
            h = self.tower.forward(h, B, T, causal=False)
            pooled = N.layernorm(h, *self.ln_post, rows=B, x_row_stride=T * W)  # class-token rows only
        return N.linear(pooled, self.w_proj)


class NativeTextTower:
    def __init__(self, model: nn.Module, blocks, device, split: bool):
        self.table = _f32(model.token_embedding.weight, device)
        self.pos = _f32(_first(model, "positional_embedding", "positional_embedding_t"), device)
        ln = model.ln_final
        self.ln_final = (_f32(ln.weight, device), _f32(ln.bias, device), ln.eps)
        proj = _first(model, "text_projection", "proj_t")
        self.w_proj = _f32(proj.t(), device)
        self.tower = _Tower(blocks, device, split)
        self.truncate = True  # skip the positions after the batch's last end-of-text token (see __call__)
        self.pool_shortcut = True  # last block: out-proj / MLP for the end-of-text rows only

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = N.to_device(tokens).to(torch.int64).contiguous()
        B, T = tokens.shape
        if T > self.pos.shape[0]:
            raise ValueError(f"{T} tokens exceed the text tower's context length {self.pos.shape[0]}")
        eot = tokens.argmax(dim=-1)  # end-of-text token (highest id): the position CLIP's text tower pools
        if self.truncate and B > 0:
            # the tower is causal: position t only sees positions <= t, so nothing after the last end-of-text token of
            # the batch can reach a pooled feature.  Dropping those (padding) positions leaves every pooled value
            # bit-identical and shrinks the work from context_length to the longest prompt (one scalar readback).
            t_eff = int(eot.max().item()) + 1
            if t_eff < T:
                tokens, T = tokens[:, :t_eff].contiguous(), t_eff
        x = N.embed_tokens(self.table, tokens, self.pos[:T].contiguous())
        rows = torch.arange(B, device=tokens.device) * T + eot
        if self.pool_shortcut:
            picked = self.tower.forward(x, B, T, causal=True, pool_rows=rows)
        else:
            picked = N.gather_rows(self.tower.forward(x, B, T, causal=True), rows, check=False)
        pooled = N.layernorm(picked, *self.ln_final)
        return N.linear(pooled, self.w_proj)


def _require_clip_layout(model):
    """open_clip builds many variants behind the same attribute names.  The native towers implement exactly one: class-token
    pooling of the image tower, a causal text tower pooled at the end-of-text (argmax) token, no LayerScale, no attention
    pooling.  Anything else is refused here instead of silently producing different features."""
    v = model.visual
    problems = []
    if getattr(v, "attn_pool", None) is not None:
        problems.append("visual.attn_pool is set (attention pooling)")
    if getattr(v, "pool_type", "tok") not in ("tok",):
        problems.append(f"visual.pool_type={getattr(v, 'pool_type', None)!r} (only 'tok')")
    if getattr(model, "text_pool_type", "argmax") not in ("argmax",):
        problems.append(f"text_pool_type={getattr(model, 'text_pool_type', None)!r} (only 'argmax')")
    for tower in (getattr(v, "transformer", None), getattr(model, "transformer", None)):
        for blk in getattr(tower, "resblocks", []):
            for ls in ("ls_1", "ls_2"):
                if hasattr(blk, ls) and not isinstance(getattr(blk, ls), nn.Identity):
                    problems.append(f"{ls} is not Identity (LayerScale)")
    if getattr(model, "attn_mask", None) is None and hasattr(model, "attn_mask"):
        problems.append("the text tower has no causal attn_mask")
    if problems:
        raise TypeError("NativeClip does not implement this open_clip variant: " + "; ".join(sorted(set(problems))))


class NativeClip(AbstractVLM):
    """``AbstractVLM`` running ``base``'s CLIP towers on HIP kernels; ``base`` keeps tokenizer + preprocessing."""

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
        if hasattr(model, "visual") and hasattr(model.visual, "conv1"):  # open_clip: CLIP.visual is the whole image tower
            visual, vblocks = model.visual, _first(model.visual, "transformer.resblocks")
            tblocks = _first(model, "transformer.resblocks")
            _require_clip_layout(model)
        else:  # synth._ClipModel: embedding members on the model, block stacks in .visual / .text
            visual, vblocks = _SynthVisual(model), model.visual.blocks
            tblocks = model.text.blocks
        self.vision = NativeVisionTower(visual, vblocks, dev, split)
        try:
            self.text = NativeTextTower(model, tblocks, dev, split)
        except (AttributeError, TypeError, ValueError):
            self.text = None  # text tower layout not recognised: encode_text stays on the wrapped torch model
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
        if self.text is None:
            return self.base.encode_text(tokens)
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
def _act_code(name: str) -> int:
    name = (name or "gelu").lower()
    if "quick" in name:
        return N.SL_ACT_QUICKGELU
    if "tanh" in name:
        return N.SL_ACT_GELU_TANH
    return N.SL_ACT_GELU


class _SigLipBlock:
    """One pre-LN block of a SigLIP encoder in `_Block`'s field layout.  Source layout: ``layer_norm1``, ``self_attn`` with
    separate ``q_proj`` / ``k_proj`` / ``v_proj`` / ``out_proj`` Linear layers (packed here into the ``[q | k | v]``
    in-projection the attention kernel reads), ``layer_norm2``, ``mlp.fc1`` / ``mlp.fc2`` (transformers' ``SiglipEncoderLayer``)."""

    def __init__(self, blk: nn.Module, heads: int, act: int, device, split: bool):
        a = blk.self_attn
        self.width = a.q_proj.in_features
        self.heads = heads
        self.head_dim = self.width // heads
        if self.head_dim not in (32, 64, 72, 80, 88, 96, 104, 128) or self.head_dim * heads != self.width:
            raise ValueError(f"head_dim {self.head_dim} is not supported (32, 64, 72, 80, 88, 96, 104, 128)")
        self.ln1 = (_f32(blk.layer_norm1.weight, device), _f32(blk.layer_norm1.bias, device), blk.layer_norm1.eps)
        self.ln2 = (_f32(blk.layer_norm2.weight, device), _f32(blk.layer_norm2.bias, device), blk.layer_norm2.eps)
        self.w_qkv = torch.cat([_f32(p.weight, device) for p in (a.q_proj, a.k_proj, a.v_proj)], 0).contiguous()
        self.b_qkv = torch.cat([_f32(p.bias, device) for p in (a.q_proj, a.k_proj, a.v_proj)], 0).contiguous()
        self.w_o, self.b_o = _f32(a.out_proj.weight, device), _f32(a.out_proj.bias, device)
        self.w_fc, self.b_fc = _f32(blk.mlp.fc1.weight, device), _f32(blk.mlp.fc1.bias, device)
        self.w_pr, self.b_pr = _f32(blk.mlp.fc2.weight, device), _f32(blk.mlp.fc2.bias, device)
        self.act = act
        if split:
            self.s_qkv, self.s_o = N.Split.of(self.w_qkv), N.Split.of(self.w_o)
            self.s_fc, self.s_pr = N.Split.of(self.w_fc), N.Split.of(self.w_pr)


class _SigLipStack(_Tower):
    def __init__(self, layers, heads: int, act: int, device, split: bool):
        self.split = split
        self.blocks = [_SigLipBlock(b, heads, act, device, split) for b in layers]
        self.width = self.blocks[0].width
        self.heads = heads


class NativeSigLipVision:
    """Image tower: patch embedding WITH bias, learned positions, no class token, non-causal blocks, ``post_layernorm``,
    then the MAP head: one learned probe attends over all tokens (``sl_attention_pool``), out-projection, and a
    LayerNorm + MLP residual branch; the pooled token is the image feature (no further projection)."""

    def __init__(self, vm: nn.Module, cfg, device, split: bool):
        emb = vm.embeddings
        conv = emb.patch_embedding
        if conv.kernel_size != conv.stride:
            raise TypeError("NativeSigLip expects a patch embedding with stride == kernel size")
        self.patch = conv.kernel_size[0]
        self.width = conv.out_channels
        self.w_patch = _f32(conv.weight.reshape(self.width, -1), device)
        self.b_patch = _f32(conv.bias, device) if conv.bias is not None else None
        self.pos = _f32(emb.position_embedding.weight, device)  # (n_patches, W)
        act = _act_code(getattr(cfg, "hidden_act", "gelu_pytorch_tanh"))
        self.split = split
        if split:
            self.s_patch = N.Split.of(self.w_patch)
        self.tower = _SigLipStack(vm.encoder.layers, cfg.num_attention_heads, act, device, split)
        ln = vm.post_layernorm
        self.ln_post = (_f32(ln.weight, device), _f32(ln.bias, device), ln.eps)
        head = vm.head
        mha: nn.MultiheadAttention = head.attention
        W = self.width
        self.heads = mha.num_heads
        self.head_dim = W // self.heads
        wq, wk, wv = mha.in_proj_weight[:W], mha.in_proj_weight[W:2 * W], mha.in_proj_weight[2 * W:]
        bq, bkv = mha.in_proj_bias[:W], mha.in_proj_bias[W:]
        # the probe is the same for every image: its query projection is a constant of the model
        probe = _f32(head.probe.reshape(1, W), device)
        self.q_probe = N.linear(probe, _f32(wq, device), _f32(bq, device)).reshape(W).contiguous()
        self.w_kv = torch.cat([_f32(wk, device), _f32(wv, device)], 0).contiguous()  # (2W, W): rows [k | v]
        self.b_kv = _f32(bkv, device)
        self.w_ho, self.b_ho = _f32(mha.out_proj.weight, device), _f32(mha.out_proj.bias, device)
        self.ln_head = (_f32(head.layernorm.weight, device), _f32(head.layernorm.bias, device), head.layernorm.eps)
        self.w_h1, self.b_h1 = _f32(head.mlp.fc1.weight, device), _f32(head.mlp.fc1.bias, device)
        self.w_h2, self.b_h2 = _f32(head.mlp.fc2.weight, device), _f32(head.mlp.fc2.bias, device)
        self.head_act = act
        if split:
            self.s_kv = N.Split.of(self.w_kv)

    @torch.no_grad()
    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        img = N.to_device(img).to(torch.float32).contiguous()
        B = img.shape[0]
        T = (img.shape[2] // self.patch) * (img.shape[3] // self.patch)
        if T != self.pos.shape[0]:
            raise ValueError(f"image gives {T} patches, the positional embedding has {self.pos.shape[0]}")
        W = self.width
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
            kv = N.linear3(h, self.s_kv, self.b_kv)
        else:
            kv = N.linear(N.layernorm(x, *self.ln_post), self.w_kv, self.b_kv)
        pooled = N.attention_pool(self.q_probe, kv, B, T, self.heads, self.head_dim)  # (B, W)
        res = N.linear(pooled, self.w_ho, self.b_ho)
        hid = N.linear(N.layernorm(res, *self.ln_head), self.w_h1, self.b_h1, act=self.head_act)
        return N.linear(hid, self.w_h2, self.b_h2, residual=res, out=res)


class NativeSigLipText:
    """Text tower: token + position embeddings, NON-causal blocks (SigLIP pads to the context length and does not mask),
    ``final_layer_norm``, the LAST position pooled, then the ``head`` Linear (with bias)."""

    def __init__(self, tm: nn.Module, cfg, device, split: bool):
        self.table = _f32(tm.embeddings.token_embedding.weight, device)
        self.pos = _f32(tm.embeddings.position_embedding.weight, device)
        self.tower = _SigLipStack(tm.encoder.layers, cfg.num_attention_heads, _act_code(getattr(cfg, "hidden_act", "")), device, split)
        ln = tm.final_layer_norm
        self.ln_final = (_f32(ln.weight, device), _f32(ln.bias, device), ln.eps)
        self.w_head, self.b_head = _f32(tm.head.weight, device), _f32(tm.head.bias, device)

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = N.to_device(tokens).to(torch.int64).contiguous()
        B, T = tokens.shape
        if T > self.pos.shape[0]:
            raise ValueError(f"{T} tokens exceed the {self.pos.shape[0]} positions of the text tower")
        x = N.embed_tokens(self.table, tokens, self.pos[:T].contiguous())
        rows = torch.arange(B, device=tokens.device, dtype=torch.int64) * T + (T - 1)
        picked = self.tower.forward(x, B, T, causal=False, pool_rows=rows)
        return N.linear(N.layernorm(picked, *self.ln_final), self.w_head, self.b_head)


class NativeSigLip(AbstractVLM):
    """``AbstractVLM`` running the towers of a SigLIP-layout model on HIP kernels.

    ``base`` is an ``AbstractVLM`` whose ``.model`` follows transformers' ``SiglipModel`` layout (``vision_model`` /
    ``text_model``; the geometry of SigLIP-so400m — width 1152, head_dim 72, MAP pooling — is what BASELINE configs[3]
    names).  Tokenizer and host preprocessing stay ``base``'s; ``preprocess`` as for :class:`NativeClip`.  open_clip's own
    timm-based SigLIP modules use other attribute names and are not mapped (open_clip / timm are absent here, so such a
    mapping could not be tested)."""

    def __init__(self, base, device=None, gemm: str = "bf16x3", preprocess=None):
        if gemm not in ("bf16x3", "f32"):
            raise ValueError("gemm must be 'bf16x3' or 'f32'")
        model = base.model
        if not (hasattr(model, "vision_model") and hasattr(model, "text_model") and hasattr(model.vision_model, "head")):
            raise TypeError("NativeSigLip expects a model with `vision_model` (with a MAP `head`) and `text_model`")
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            dev = N.default_device()
        self._device = dev
        self.base = base
        base.to(dev)
        split = gemm == "bf16x3"
        self.vision = NativeSigLipVision(model.vision_model, model.config.vision_config, dev, split)
        self.text = NativeSigLipText(model.text_model, model.config.text_config, dev, split)
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
