"""Torch modules that carry the ATTRIBUTE TREE of the models the reference's wrappers construct — test fixtures.

``semanticlens/foundation_models/clip.py:52-62`` builds ``open_clip.create_model_and_transforms(url)``; open-clip-torch 3.0.0
(the reference's lock file) and timm are not installed in this image and cannot be vendored.  ``NativeClip`` /
``NativeSigLip`` read weights by attribute name, so what they need to be exercised against is a model whose members are
named and composed like open_clip's — not open_clip's code.  This file restates that layout from the published structure
of the two packages (module and parameter names, constructor options and their forward semantics), written from scratch:

* ``CLIP``: ``visual`` = ``VisionTransformer`` {``conv1``, ``class_embedding``, ``positional_embedding``, ``patch_dropout``,
  ``ln_pre``, ``transformer.resblocks[i]`` {``ln_1``, ``attn``, ``ls_1``, ``ln_2``, ``mlp`` (``c_fc``, ``gelu``, ``c_proj``), ``ls_2``},
  ``attn_pool``, ``pool_type``, ``final_ln_after_pool``, ``ln_post``, ``proj``}; the text tower's members flattened onto the
  model: ``transformer``, ``token_embedding``, ``positional_embedding``, ``ln_final``, ``text_projection``, ``attn_mask``,
  ``text_pool_type``, ``context_length``, ``vocab_size``.
* ``CustomTextCLIP``: ``visual`` + ``text`` = ``TextTransformer`` (same members, plus ``pool_type``, ``cls_emb``).
* ``TimmModel`` {``trunk``, ``head``} around a timm-style ``TimmViT`` {``patch_embed.proj``, ``cls_token``, ``reg_token``,
  ``pos_embed``, ``norm_pre``, ``blocks[i]`` {``norm1``, ``attn`` {``qkv``, ``q_norm``, ``k_norm``, ``proj``}, ``ls1``, ``norm2``,
  ``mlp`` {``fc1``, ``act``, ``norm``, ``fc2``}, ``ls2``}, ``norm``, ``attn_pool`` = ``AttentionPoolLatent`` {``latent``, ``q``,
  ``kv``, ``proj``, ``norm``, ``mlp``}, ``fc_norm``, ``head``} — what ``hf-hub:timm/ViT-B-16-SigLIP2`` (``SigLipV2``,
  clip.py:190-211) resolves to: ``timm_pool="map"``, ``timm_proj="none"``, text tower with ``no_causal_mask``,
  ``pool_type="last"``, ``proj_bias``, LayerNorm eps 1e-6 and tanh-GELU.

Parity against these modules is parity against a same-layout torch model, not against open_clip's weights: the row stays
"unpinned" in DESIGN.md.  ``fake_open_clip_module`` packages the builders as an importable ``open_clip`` stand-in so that
the reference-API wrappers (``OpenClip`` / ``SigLipV2``) and their ``.native()`` can be driven end to end.
"""
from __future__ import annotations

import math
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class LayerScale(nn.Module):
    def __init__(self, dim, init_value):
        super().__init__()
        self.gamma = nn.Parameter(init_value * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


# ------------------------------------------------------------------------------------------- open_clip's own transformer
class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, mlp_ratio=4.0, ls_init_value=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.ln_1 = norm_layer(d_model)
        self.attn = nn.MultiheadAttention(d_model, n_head, batch_first=True)
        self.ls_1 = LayerScale(d_model, ls_init_value) if ls_init_value is not None else nn.Identity()
        self.ln_2 = norm_layer(d_model)
        hidden = int(d_model * mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, hidden)), ("gelu", act_layer()), ("c_proj", nn.Linear(hidden, d_model))]))
        self.ls_2 = LayerScale(d_model, ls_init_value) if ls_init_value is not None else nn.Identity()

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.ls_1(self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0])
        return x + self.ls_2(self.mlp(self.ln_2(x)))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio=4.0, ls_init_value=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio, ls_init_value, act_layer, norm_layer)
                                        for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for blk in self.resblocks:
            x = blk(x, attn_mask=attn_mask)
        return x


class VisionTransformer(nn.Module):
    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio=4.0, ls_init_value=None, output_dim=512,
                 no_ln_pre=False, pool_type="tok", final_ln_after_pool=False, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 attentional_pool=False):
        super().__init__()
        self.image_size, self.patch_size = (image_size, image_size), (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.final_ln_after_pool = final_ln_after_pool
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.patch_dropout = nn.Identity()
        self.ln_pre = nn.Identity() if no_ln_pre else norm_layer(width)
        self.transformer = Transformer(width, layers, heads, mlp_ratio, ls_init_value, act_layer, norm_layer)
        # CoCa-style attentional pooling is only ever *detected* (and refused) by the native reader
        self.attn_pool = nn.MultiheadAttention(width, heads, batch_first=True) if attentional_pool else None
        self.pool_type = pool_type
        self.ln_post = norm_layer(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def _global_pool(self, x):
        if self.pool_type == "avg":
            return x[:, 1:].mean(dim=1)
        if self.pool_type == "tok":
            return x[:, 0]
        return x

    def forward(self, x):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = self.class_embedding.view(1, 1, -1).expand(x.shape[0], -1, -1).to(x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(self.patch_dropout(x))
        x = self.transformer(x)
        if self.attn_pool is not None:
            raise NotImplementedError("attentional pooling is a refusal fixture")
        if self.final_ln_after_pool:
            pooled = self.ln_post(self._global_pool(x))
        else:
            pooled = self._global_pool(self.ln_post(x))
        return pooled @ self.proj if self.proj is not None else pooled


def text_global_pool(x, text, pool_type):
    if pool_type == "first":
        return x[:, 0]
    if pool_type == "last":
        return x[:, -1]
    if pool_type == "argmax":
        return x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)]
    return x


class TextTransformer(nn.Module):
    def __init__(self, context_length=77, vocab_size=49408, width=512, heads=8, layers=12, mlp_ratio=4.0, ls_init_value=None,
                 output_dim=512, no_causal_mask=False, pool_type="argmax", proj_bias=False, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.context_length = self.num_pos = context_length
        self.vocab_size, self.width, self.output_dim, self.heads = vocab_size, width, output_dim, heads
        self.pool_type = pool_type
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.cls_emb = None
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(self.num_pos, width))
        self.transformer = Transformer(width, layers, heads, mlp_ratio, ls_init_value, act_layer, norm_layer)
        self.ln_final = norm_layer(width)
        if no_causal_mask:
            self.attn_mask = None
        else:
            self.register_buffer("attn_mask", torch.full((self.num_pos, self.num_pos), float("-inf")).triu_(1), persistent=False)
        if proj_bias:
            self.text_projection = nn.Linear(width, output_dim)
        else:
            self.text_projection = nn.Parameter(width ** -0.5 * torch.randn(width, output_dim))

    def forward(self, text):
        T = text.shape[1]
        x = self.token_embedding(text) + self.positional_embedding[:T]
        mask = self.attn_mask[:T, :T] if self.attn_mask is not None else None
        x = self.ln_final(self.transformer(x, attn_mask=mask))
        pooled = text_global_pool(x, text, self.pool_type)
        if isinstance(self.text_projection, nn.Linear):
            return self.text_projection(pooled)
        return pooled @ self.text_projection


class CLIP(nn.Module):
    """open_clip.CLIP: the image tower under ``visual``, the text tower's members on the model itself."""

    def __init__(self, embed_dim, vision_cfg: dict, text_cfg: dict, quick_gelu=False):
        super().__init__()
        act = QuickGELU if quick_gelu else nn.GELU
        self.visual = VisionTransformer(output_dim=embed_dim, act_layer=act, **vision_cfg)
        text = TextTransformer(output_dim=embed_dim, act_layer=act, **text_cfg)
        self.transformer = text.transformer
        self.context_length, self.vocab_size = text.context_length, text.vocab_size
        self.token_embedding = text.token_embedding
        self.positional_embedding = text.positional_embedding
        self.ln_final = text.ln_final
        self.text_projection = text.text_projection
        self.text_pool_type = text.pool_type
        self.register_buffer("attn_mask", text.attn_mask, persistent=False)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def encode_image(self, image, normalize=False):
        f = self.visual(image)
        return F.normalize(f, dim=-1) if normalize else f

    def encode_text(self, text, normalize=False):
        T = text.shape[1]
        x = self.token_embedding(text) + self.positional_embedding[:T]
        mask = self.attn_mask[:T, :T] if self.attn_mask is not None else None
        x = self.ln_final(self.transformer(x, attn_mask=mask))
        x = text_global_pool(x, text, self.text_pool_type)
        x = self.text_projection(x) if isinstance(self.text_projection, nn.Linear) else x @ self.text_projection
        return F.normalize(x, dim=-1) if normalize else x


class CustomTextCLIP(nn.Module):
    def __init__(self, visual: nn.Module, text: TextTransformer):
        super().__init__()
        self.visual, self.text = visual, text
        self.context_length, self.vocab_size = text.context_length, text.vocab_size
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(10.0))
        self.logit_bias = nn.Parameter(torch.ones([]) * -10.0)

    def encode_image(self, image, normalize=False):
        f = self.visual(image)
        return F.normalize(f, dim=-1) if normalize else f

    def encode_text(self, text, normalize=False):
        f = self.text(text)
        return F.normalize(f, dim=-1) if normalize else f


# ------------------------------------------------------------------------------------------------ timm-style ViT trunk
class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class TimmAttention(nn.Module):
    def __init__(self, dim, num_heads, qk_norm=False):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.q_norm = nn.LayerNorm(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = nn.LayerNorm(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)

    def forward(self, x):
        B, T, C = x.shape
        q, k, v = self.qkv(x).reshape(B, T, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        a = (self.q_norm(q) * self.scale) @ self.k_norm(k).transpose(-2, -1)
        x = (a.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, C)
        return self.proj_drop(self.proj(x))


class TimmMlp(nn.Module):
    def __init__(self, dim, hidden, act_layer):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = act_layer()
        self.drop1 = nn.Dropout(0.0)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Dropout(0.0)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class TimmBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, act_layer, norm_layer, init_values=None, qk_norm=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = TimmAttention(dim, num_heads, qk_norm)
        self.ls1 = LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = TimmMlp(dim, int(dim * mlp_ratio), act_layer)
        self.ls2 = LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))


class AttentionPoolLatent(nn.Module):
    """timm's MAP head: one learned latent queries all tokens; projection; LayerNorm + MLP residual branch."""

    def __init__(self, dim, num_heads, mlp_ratio, norm_layer, act_layer):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.pool = "token"
        self.pos_embed = None
        self.latent_dim, self.latent_len = dim, 1
        self.latent = nn.Parameter(torch.randn(1, 1, dim) * dim ** -0.5)
        self.q = nn.Linear(dim, dim)
        self.kv = nn.Linear(dim, dim * 2)
        self.q_norm, self.k_norm = nn.Identity(), nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)
        self.norm = norm_layer(dim)
        self.mlp = TimmMlp(dim, int(dim * mlp_ratio), act_layer)

    def forward(self, x):
        B, T, C = x.shape
        q = self.q(self.latent.expand(B, -1, -1)).reshape(B, 1, self.num_heads, self.head_dim).transpose(1, 2)
        k, v = self.kv(x).reshape(B, T, 2, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        a = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        x = self.proj_drop(self.proj((a @ v).transpose(1, 2).reshape(B, 1, C)))
        x = x + self.mlp(self.norm(x))
        return x[:, 0]


class GELUTanh(nn.Module):
    def forward(self, x):
        return F.gelu(x, approximate="tanh")


class TimmViT(nn.Module):
    """timm VisionTransformer as built for the SigLIP checkpoints: no class token, learned positions, MAP pooling."""

    def __init__(self, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, global_pool="map",
                 class_token=False, act_layer=GELUTanh, eps=1e-6, init_values=None, qk_norm=False):
        super().__init__()
        norm_layer = lambda d: nn.LayerNorm(d, eps=eps)  # noqa: E731
        self.global_pool, self.embed_dim, self.num_features = global_pool, embed_dim, embed_dim
        self.num_prefix_tokens = 1 if class_token else 0
        self.patch_embed = PatchEmbed(img_size, patch_size, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if class_token else None
        self.reg_token = None
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + self.num_prefix_tokens, embed_dim) * 0.02)
        self.pos_drop = nn.Dropout(0.0)
        self.patch_drop = nn.Identity()
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[TimmBlock(embed_dim, num_heads, mlp_ratio, act_layer, norm_layer, init_values, qk_norm)
                                      for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.attn_pool = AttentionPoolLatent(embed_dim, num_heads, mlp_ratio, norm_layer, act_layer) if global_pool == "map" else None
        self.fc_norm = nn.Identity()
        self.head_drop = nn.Dropout(0.0)
        self.head = nn.Identity()

    def forward(self, x):
        x = self.patch_embed(x)
        if self.cls_token is not None:
            x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        x = self.norm(self.blocks(self.norm_pre(self.patch_drop(self.pos_drop(x + self.pos_embed)))))
        if self.attn_pool is not None:
            x = self.attn_pool(x)
        elif self.global_pool == "avg":
            x = x[:, self.num_prefix_tokens:].mean(dim=1)
        else:
            x = x[:, 0]
        return self.head(self.head_drop(self.fc_norm(x)))


class TimmModel(nn.Module):
    """open_clip's adapter around a timm trunk: ``head`` holds the projection (empty for ``timm_proj="none"``)."""

    def __init__(self, trunk: TimmViT, embed_dim=None, proj="none"):
        super().__init__()
        self.trunk = trunk
        layers = OrderedDict()
        if proj == "linear":
            layers["drop"] = nn.Dropout(0.0)
            layers["proj"] = nn.Linear(trunk.num_features, embed_dim, bias=False)
        self.head = nn.Sequential(layers)

    def forward(self, x):
        return self.head(self.trunk(x))


# ------------------------------------------------------------------------------------------------------ model builders
def build_clip_vit(embed_dim=512, image_size=224, patch=32, v_width=768, v_layers=12, v_heads=12, ctx=77, vocab=49408, t_width=512,
                   t_layers=12, t_heads=8, quick_gelu=False, seed=0, **variant) -> CLIP:
    """``open_clip.create_model("ViT-B-32")`` geometry by default.  ``variant``: ``pool_type``, ``no_ln_pre``,
    ``final_ln_after_pool``, ``ls_init_value``, ``attentional_pool`` (vision); ``text_pool_type``, ``no_causal_mask``,
    ``proj_bias`` (text)."""
    torch.manual_seed(seed)
    vkeys = ("pool_type", "no_ln_pre", "final_ln_after_pool", "ls_init_value", "attentional_pool")
    vcfg = dict(image_size=image_size, patch_size=patch, width=v_width, layers=v_layers, heads=v_heads,
                **{k: variant[k] for k in vkeys if k in variant})
    tcfg = dict(context_length=ctx, vocab_size=vocab, width=t_width, heads=t_heads, layers=t_layers,
                pool_type=variant.get("text_pool_type", "argmax"), no_causal_mask=variant.get("no_causal_mask", False),
                proj_bias=variant.get("proj_bias", False), ls_init_value=variant.get("ls_init_value"))
    model = CLIP(embed_dim, vcfg, tcfg, quick_gelu=quick_gelu).eval()
    _randomize(model, seed)
    return model


def build_siglip2(embed_dim=768, image_size=224, patch=16, width=768, layers=12, heads=12, ctx=64, vocab=256000, t_layers=None,
                  proj="none", seed=0, **trunk_variant) -> CustomTextCLIP:
    """``hf-hub:timm/ViT-B-16-SigLIP2`` geometry by default (``SigLipV2``, clip.py:190-211)."""
    torch.manual_seed(seed)
    ln = lambda d: nn.LayerNorm(d, eps=1e-6)  # noqa: E731
    trunk = TimmViT(image_size, patch, width, layers, heads, **trunk_variant)
    visual = TimmModel(trunk, embed_dim, proj)
    text = TextTransformer(context_length=ctx, vocab_size=vocab, width=width, heads=heads, layers=t_layers or layers, output_dim=embed_dim,
                           no_causal_mask=True, pool_type="last", proj_bias=True, act_layer=GELUTanh, norm_layer=ln)
    model = CustomTextCLIP(visual, text).eval()
    _randomize(model, seed)
    return model


def _randomize(model: nn.Module, seed: int):
    """Non-trivial values everywhere a fresh module holds zeros / ones (biases, LayerNorm affine, LayerScale), so that a reader
    that drops or swaps one of them cannot pass."""
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim == 1 and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1 and name.endswith("weight"):  # LayerNorm gamma
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("gamma"):
                p.copy_(0.5 + 0.5 * torch.rand(p.shape, generator=g))
            elif name.endswith("in_proj_weight") or (p.ndim == 2 and "token_embedding" not in name and "positional" not in name):
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / math.sqrt(p.shape[-1])))


# ------------------------------------------------------------------------------------------- importable open_clip stand-in
def _eval_transform(size, mean, std):
    """PIL image -> normalised (3, size, size) tensor, squash-resized with bilinear (enough for plumbing tests)."""

    def tf(img):
        arr = np.asarray(img.convert("RGB").resize((size, size)), dtype=np.float32) / 255.0
        t = torch.from_numpy(arr).permute(2, 0, 1)
        return (t - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]

    return tf


def _tokenizer(vocab, eot=None, pad=0):
    """Deterministic word-hash tokenizer with open_clip's call signature ``tokenizer(texts, context_length=...)``."""

    def tok(texts, context_length=77):
        texts = [texts] if isinstance(texts, str) else list(texts)
        out = torch.full((len(texts), context_length), pad, dtype=torch.int64)
        for r, s in enumerate(texts):
            ids = [1 + (sum(ord(c) * 131 ** i for i, c in enumerate(w)) % (vocab - 3)) for w in s.lower().split()]
            if eot is not None:
                ids = ids[: context_length - 1] + [eot]
            out[r, : min(len(ids), context_length)] = torch.tensor(ids[:context_length], dtype=torch.int64)
        return out

    return tok


def fake_open_clip_module(registry: dict):
    """A module object exposing ``create_model_and_transforms`` / ``get_tokenizer`` over ``registry``:
    ``{model_name: (builder() -> nn.Module, image_size, mean, std, tokenizer)}``; records its calls in ``mod.calls``."""
    mod = types.ModuleType("open_clip")
    mod.calls = []

    def create_model_and_transforms(model_name, pretrained=None, **kwargs):
        mod.calls.append(("create", model_name, pretrained, kwargs))
        build, size, mean, std, _ = registry[model_name]
        tf = _eval_transform(size, mean, std)
        return build(), tf, tf

    def get_tokenizer(model_name, **kwargs):
        mod.calls.append(("tokenizer", model_name))
        return registry[model_name][4]

    mod.create_model_and_transforms, mod.get_tokenizer = create_model_and_transforms, get_tokenizer
    return mod


CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
HALF = (0.5, 0.5, 0.5)
