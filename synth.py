"""Synthetic workloads for bench.py / tests: the BASELINE.json configurations with random-init
models and counter-based synthetic images (there is no network for datasets or checkpoints).

Not part of the product package: plain-torch definitions of the probed models (ResNet-18/50,
torchvision module names so the reference's layer names ``layer2/3/4`` apply), a CLIP-shaped
foundation model behind ``AbstractVLM`` (ViT-B/32 image tower, 12x512 text tower, 512-d joint
space — the OpenCLIP ViT-B/32 architecture with random weights), and a dataset whose sample
``i`` is a pure function of ``(seed, i)`` so any shard reproduces any sample.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from semanticlens_amd.foundation_models.base import AbstractVLM

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------------------------------------
# counter-based synthetic images
# ------------------------------------------------------------------------------------------------
def _mix(x: torch.Tensor) -> torch.Tensor:
    """32-bit integer hash (murmur3 finaliser) on int64 tensors holding values < 2**32."""
    m = 0xFFFFFFFF
    x = x & m
    x = ((x ^ (x >> 16)) * 0x85EBCA6B) & m
    x = ((x ^ (x >> 13)) * 0xC2B2AE35) & m
    return (x ^ (x >> 16)) & m


def synth_images_u8(indices: torch.Tensor, size: int = 224, seed: int = 0) -> torch.Tensor:
    """uint8 images (n, 3, size, size) for dataset indices ``indices`` (any device), a pure function
    of (seed, index): hashed pixel noise modulated by a per-image low-frequency pattern and gain."""
    dev = indices.device
    idx = indices.to(torch.int64).reshape(-1, 1, 1, 1)
    c = torch.arange(3, device=dev, dtype=torch.int64).reshape(1, 3, 1, 1)
    y = torch.arange(size, device=dev, dtype=torch.int64).reshape(1, 1, size, 1)
    x = torch.arange(size, device=dev, dtype=torch.int64).reshape(1, 1, 1, size)
    key = _mix(idx * 0x9E3779B1 + seed * 0x7F4A7C15 + 1)
    noise = _mix(key + ((c * size + y) * size + x) * 0x27D4EB2F) & 0xFF
    # per-image parameters from the key
    gain = 0.25 + 3.75 * ((_mix(key + 11) & 0xFFFF).to(torch.float32) / 65535.0)
    fx = 1 + (_mix(key + 23) & 7).to(torch.float32)
    fy = 1 + (_mix(key + 37) & 7).to(torch.float32)
    ph = (_mix(key + 41) & 0xFFFF).to(torch.float32) / 65535.0 * (2 * math.pi)
    wave = 0.5 + 0.5 * torch.sin(2 * math.pi * (fx * x.to(torch.float32) + fy * y.to(torch.float32)) / size + ph + c.to(torch.float32))
    img = noise.to(torch.float32) * (0.35 + 0.65 * wave) * gain / 2.0
    return img.clamp_(0, 255).to(torch.uint8)


_NORM_CONSTS: dict = {}


def normalize_u8(u8: torch.Tensor, mean, std) -> torch.Tensor:
    # the constants are uploaded once per device: `torch.tensor(list, device=cuda)` is a blocking copy, i.e. a stream
    # synchronisation per call, which kept the host from running ahead of the device in the bench loop
    key = (str(u8.device), tuple(mean), tuple(std))
    if key not in _NORM_CONSTS:
        _NORM_CONSTS[key] = (torch.tensor(mean, device=u8.device, dtype=torch.float32).reshape(1, 3, 1, 1),
                             torch.tensor(std, device=u8.device, dtype=torch.float32).reshape(1, 3, 1, 1))
    m, s = _NORM_CONSTS[key]
    return (u8.to(torch.float32) / 255.0 - m) / s


class SyntheticImageDataset(torch.utils.data.Dataset):
    """``mode='model'`` yields ``(normalised fp32 tensor, 0)`` like the reference's ``dataset_model``;
    ``mode='fm'`` yields the raw uint8 image for the foundation model's own preprocessing."""

    def __init__(self, n: int, mode: str = "model", size: int = 224, seed: int = 0):
        self.n, self.mode, self.size, self.seed = n, mode, size, seed
        self.name = f"synthetic-{n}-s{seed}"

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        u8 = synth_images_u8(torch.tensor([i]), self.size, self.seed)[0]
        if self.mode == "model":
            return normalize_u8(u8[None], IMAGENET_MEAN, IMAGENET_STD)[0], 0
        return u8


# ------------------------------------------------------------------------------------------------
# probed models: ResNet-18 / ResNet-50 (He et al.), torchvision naming
# ------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make(self, block, planes, n, stride):
        blocks = [block(self.inplanes, planes, stride)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(seed: int = 0) -> nn.Module:
    torch.manual_seed(seed)
    m = ResNet(BasicBlock, [2, 2, 2, 2]).eval()
    m.name = "resnet18-random"
    return m


def resnet50(seed: int = 0) -> nn.Module:
    torch.manual_seed(seed)
    m = ResNet(Bottleneck, [3, 4, 6, 3]).eval()
    m.name = "resnet50-random"
    return m


# ------------------------------------------------------------------------------------------------
# CLIP-shaped foundation model (ViT-B/32 architecture, random init)
# ------------------------------------------------------------------------------------------------
class _Block(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(nn.Linear(width, width * 4), nn.GELU(), nn.Linear(width * 4, width))

    def forward(self, x, mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=mask)[0]
        return x + self.mlp(self.ln_2(x))


class _Tower(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.blocks = nn.ModuleList([_Block(width, heads) for _ in range(layers)])

    def forward(self, x, mask=None):
        for b in self.blocks:
            x = b(x, mask)
        return x


class _ClipModel(nn.Module):
    def __init__(self, embed_dim=512, image_size=224, patch=32, v_width=768, v_layers=12, v_heads=12, ctx=77,
                 vocab=49408, t_width=512, t_layers=12, t_heads=8):
        super().__init__()
        self.context_length = ctx
        self.conv1 = nn.Conv2d(3, v_width, patch, patch, bias=False)
        n_tok = (image_size // patch) ** 2 + 1
        self.class_embedding = nn.Parameter(v_width**-0.5 * torch.randn(v_width))
        self.positional_embedding_v = nn.Parameter(v_width**-0.5 * torch.randn(n_tok, v_width))
        self.ln_pre = nn.LayerNorm(v_width)
        self.visual = _Tower(v_width, v_layers, v_heads)
        self.ln_post = nn.LayerNorm(v_width)
        self.proj_v = nn.Parameter(v_width**-0.5 * torch.randn(v_width, embed_dim))
        self.token_embedding = nn.Embedding(vocab, t_width)
        self.positional_embedding_t = nn.Parameter(0.01 * torch.randn(ctx, t_width))
        self.text = _Tower(t_width, t_layers, t_heads)
        self.ln_final = nn.LayerNorm(t_width)
        self.proj_t = nn.Parameter(t_width**-0.5 * torch.randn(t_width, embed_dim))
        self.register_buffer("causal", torch.full((ctx, ctx), float("-inf")).triu_(1), persistent=False)

    def encode_image(self, img):
        x = self.conv1(img).flatten(2).transpose(1, 2)
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = self.ln_pre(torch.cat([cls, x], 1) + self.positional_embedding_v)
        x = self.visual(x)
        return self.ln_post(x[:, 0]) @ self.proj_v

    def encode_text(self, tokens):
        x = self.token_embedding(tokens) + self.positional_embedding_t[: tokens.shape[1]]
        x = self.ln_final(self.text(x, self.causal[: tokens.shape[1], : tokens.shape[1]]))
        eot = tokens.argmax(dim=-1)  # end-of-text carries the highest id, as in CLIP's tokenizer
        return x[torch.arange(x.shape[0], device=x.device), eot] @ self.proj_t


class SyntheticClip(AbstractVLM):
    """Random-init CLIP ViT-B/32 behind the ``AbstractVLM`` seam (what ``OpenClip('ViT-B-32')`` would be
    without weights).  ``preprocess`` accepts uint8 tensors (or PIL images) and normalises on the device;
    ``tokenize`` is a deterministic word-hash tokenizer (the BPE vocabulary is not available offline)."""

    def __init__(self, device="cpu", seed: int = 1, embed_dim: int = 512, **arch):
        torch.manual_seed(seed)
        self.model = _ClipModel(embed_dim=embed_dim, **arch).eval().to(device)
        self.name = f"synthetic-clip-vitb32-d{embed_dim}"

    @property
    def device(self):
        return next(self.model.parameters()).device

    def to(self, device):
        return self.model.to(device)

    @torch.no_grad()
    def encode_image(self, img):
        return self.model.encode_image(img)

    @torch.no_grad()
    def encode_text(self, tokens):
        return self.model.encode_text(tokens)

    def preprocess(self, img):
        import numpy as np

        def one(i):
            if isinstance(i, torch.Tensor):
                return i
            return torch.from_numpy(np.asarray(i.convert("RGB"))).permute(2, 0, 1)

        batch = torch.stack([one(i) for i in img]) if isinstance(img, (list, tuple)) else one(img)
        if batch.ndim == 3:
            batch = batch.unsqueeze(0)
        return normalize_u8(batch.to(self.device), CLIP_MEAN, CLIP_STD)

    def tokenize(self, txt, context_length=None):
        ctx = context_length or self.model.context_length
        if isinstance(txt, str):
            txt = [txt]
        out = torch.zeros(len(txt), ctx, dtype=torch.int64)
        for r, s in enumerate(txt):
            words = s.lower().split()[: ctx - 2]
            ids = [49406] + [1 + (hash_str(w) % 49000) for w in words] + [49407]
            out[r, : len(ids)] = torch.tensor(ids)
        return out.to(self.device)


def hash_str(s: str) -> int:
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


# ------------------------------------------------------------------------------------------------
# SigLIP-shaped foundation model (transformers' SiglipModel, random init) — BASELINE configs[3] names SigLIP-so400m
# ------------------------------------------------------------------------------------------------
SIGLIP_SO400M = dict(width=1152, layers=27, heads=16, mlp=4304, image_size=224, patch=14, ctx=64, vocab=32000)
SIGLIP_MEAN = SIGLIP_STD = (0.5, 0.5, 0.5)


class SyntheticSigLip(AbstractVLM):
    """Random-init SigLIP behind the ``AbstractVLM`` seam: MAP-pooled image tower without a class token, non-causal text
    tower pooled at the last position (what ``SigLipV2`` / SigLIP-so400m would be without weights).  ``preprocess`` and
    ``tokenize`` as for :class:`SyntheticClip` (SigLIP pads with token 1 up to the context length)."""

    def __init__(self, device="cpu", seed: int = 1, width=1152, layers=27, heads=16, mlp=4304, image_size=224, patch=14, ctx=64,
                 vocab=32000, t_layers=None):
        from transformers import SiglipConfig, SiglipModel

        torch.manual_seed(seed)
        common = dict(hidden_size=width, intermediate_size=mlp, num_attention_heads=heads)
        cfg = SiglipConfig(
            vision_config=dict(num_hidden_layers=layers, image_size=image_size, patch_size=patch, **common),
            text_config=dict(num_hidden_layers=t_layers or layers, vocab_size=vocab, max_position_embeddings=ctx, projection_size=width,
                             bos_token_id=None, eos_token_id=None, pad_token_id=1, **common))
        self.model = SiglipModel(cfg).eval().to(device)
        self.ctx, self.vocab = ctx, vocab
        self.name = f"synthetic-siglip-w{width}-l{layers}"

    @property
    def device(self):
        return next(self.model.parameters()).device

    def to(self, device):
        return self.model.to(device)

    @torch.no_grad()
    def encode_image(self, img):
        return self.model.vision_model(pixel_values=img).pooler_output

    @torch.no_grad()
    def encode_text(self, tokens):
        return self.model.text_model(input_ids=tokens).pooler_output

    def preprocess(self, img):
        import numpy as np

        def one(i):
            if isinstance(i, torch.Tensor):
                return i
            return torch.from_numpy(np.asarray(i.convert("RGB"))).permute(2, 0, 1)

        batch = torch.stack([one(i) for i in img]) if isinstance(img, (list, tuple)) else one(img)
        if batch.ndim == 3:
            batch = batch.unsqueeze(0)
        return normalize_u8(batch.to(self.device), SIGLIP_MEAN, SIGLIP_STD)

    def tokenize(self, txt, context_length=None):
        ctx = context_length or self.ctx
        if isinstance(txt, str):
            txt = [txt]
        out = torch.ones(len(txt), ctx, dtype=torch.int64)  # pad id 1
        for r, s_ in enumerate(txt):
            ids = [2 + (hash_str(w) % (self.vocab - 2)) for w in s_.lower().split()][:ctx]
            out[r, : len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out.to(self.device)


# ------------------------------------------------------------------------------------------------
# probed model of BASELINE configs[3]: ViT-B/16 (random init), encoder blocks named ``blocks.<i>``
# ------------------------------------------------------------------------------------------------
class VisionTransformer(nn.Module):
    """ViT-B/16 classifier geometry: 16x16 patches of a 224x224 image + class token = 197 tokens of width 768, 12 pre-LN
    blocks (12 heads, MLP 3072), LayerNorm, linear head.  Every ``blocks.<i>`` outputs ``(B, 197, 768)``."""

    def __init__(self, image_size=224, patch=16, width=768, layers=12, heads=12, num_classes=1000):
        super().__init__()
        self.patch_embed = nn.Conv2d(3, width, patch, patch)
        n_tok = (image_size // patch) ** 2 + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, width))
        self.pos_embed = nn.Parameter(0.02 * torch.randn(1, n_tok, width))
        self.blocks = nn.ModuleList([_Block(width, heads) for _ in range(layers)])
        self.norm = nn.LayerNorm(width)
        self.head = nn.Linear(width, num_classes)

    def forward(self, x):
        x = self.patch_embed(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], 1) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return self.head(self.norm(x[:, 0]))


def vit_b16(seed: int = 0, **arch) -> nn.Module:
    torch.manual_seed(seed)
    m = VisionTransformer(**arch).eval()
    m.name = "vit-b16-random"
    return m


# ------------------------------------------------------------------------------------------------
# probed model of BASELINE configs[4]: ConvNeXt-L (random init), stages named ``stages.<i>``
# ------------------------------------------------------------------------------------------------
class _LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel axis of an NCHW map (ConvNeXt's channels_first norm)."""

    def forward(self, x):
        return super().forward(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class ConvNeXtBlock(nn.Module):
    """Liu et al. 2022: 7x7 depthwise conv -> LayerNorm -> 1x1 (4x) -> GELU -> 1x1 -> layer scale -> residual."""

    def __init__(self, dim, layer_scale=1e-6, nchw_out=False):
        super().__init__()
        self.nchw_out = nchw_out
        self.dwconv = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale * torch.ones(dim))

    def forward(self, x):
        h = self.dwconv(x).permute(0, 2, 3, 1)
        h = self.pwconv2(self.act(self.pwconv1(self.norm(h)))) * self.gamma
        if self.nchw_out:
            # `x + h.permute(...)` takes the permuted branch's layout: every block output — and so every later depthwise
            # convolution's input — is channels_last-strided (torchvision's and timm's blocks behave the same), which on ROCm sends
            # the 7x7 depthwise convolutions to MIOpen's naive NHWC kernel (60 % of the forward).  This variant writes the sum
            # NCHW-contiguous instead (one transposing add), keeping the convolutions on their NCHW kernels.
            return torch.add(x, h.permute(0, 3, 1, 2), out=torch.empty_like(x, memory_format=torch.contiguous_format))
        return x + h.permute(0, 3, 1, 2)


class ConvNeXt(nn.Module):
    """ConvNeXt classifier: 4x4/4 stem, four stages of ``depths`` blocks at ``dims`` channels with 2x2/2 downsampling in
    front of stages 1-3, global average pool, LayerNorm, linear head.  ``stages.<i>`` outputs ``(B, dims[i], 56 >> i, 56 >> i)``
    at 224 x 224: ConvNeXt-L = depths (3, 3, 27, 3), dims (192, 384, 768, 1536) -> S = 3136 / 784 / 196 / 49."""

    def __init__(self, depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536), num_classes=1000, layer_scale=1e-6, nchw_out=False):
        super().__init__()
        self.downsample_layers = nn.ModuleList([nn.Sequential(nn.Conv2d(3, dims[0], 4, 4), _LayerNorm2d(dims[0], eps=1e-6))])
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(_LayerNorm2d(dims[i], eps=1e-6), nn.Conv2d(dims[i], dims[i + 1], 2, 2)))
        self.stages = nn.ModuleList([nn.Sequential(*[ConvNeXtBlock(d, layer_scale, nchw_out) for _ in range(n)])
                                     for n, d in zip(depths, dims)])
        self.norm = nn.LayerNorm(dims[-1], eps=1e-6)
        self.head = nn.Linear(dims[-1], num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        for down, stage in zip(self.downsample_layers, self.stages):
            x = stage(down(x))
        return self.head(self.norm(x.mean((-2, -1))))


def convnext_l(seed: int = 0, layer_scale: float = 1.0, **arch) -> nn.Module:
    """ConvNeXt-L, random init.  ``layer_scale`` defaults to 1.0 instead of the training-time initial value 1e-6: with
    random weights and gamma = 1e-6 every block would be the identity and all blocks of a stage would emit the same map."""
    torch.manual_seed(seed)
    m = ConvNeXt(layer_scale=layer_scale, **arch).eval()
    m.name = "convnext-l-random"
    return m


# ------------------------------------------------------------------------------------------------
# CLIP-ResNet image tower (BASELINE configs[0] names "OpenClip RN50 embed"): the attribute tree open_clip's
# `ModifiedResNet` exposes (three-conv stem, anti-aliased bottlenecks, attention pool), random init
# ------------------------------------------------------------------------------------------------
class ClipBottleneck(nn.Module):
    """Bottleneck of CLIP's ResNets: strides are average pools (after conv2 and in front of the downsample conv)."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.act3 = nn.ReLU(inplace=True)
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            from collections import OrderedDict

            self.downsample = nn.Sequential(OrderedDict([("-1", nn.AvgPool2d(stride)), ("0", nn.Conv2d(inplanes, planes * 4, 1, bias=False)),
                                                         ("1", nn.BatchNorm2d(planes * 4))]))

    def forward(self, x):
        out = self.act1(self.bn1(self.conv1(x)))
        out = self.avgpool(self.act2(self.bn2(self.conv2(out))))
        out = self.bn3(self.conv3(out))
        return self.act3(out + (x if self.downsample is None else self.downsample(x)))


class AttentionPool2d(nn.Module):
    """CLIP's attention pool: the mean token (plus its position) queries all HW + 1 tokens; `c_proj` maps to the joint space."""

    def __init__(self, spacial_dim: int, embed_dim: int, num_heads: int, output_dim: int):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim**2 + 1, embed_dim) / embed_dim**0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim)
        self.num_heads = num_heads

    def forward(self, x):
        x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3]).permute(2, 0, 1)  # NCHW -> (HW)NC
        x = torch.cat([x.mean(dim=0, keepdim=True), x], dim=0)
        x = x + self.positional_embedding[:, None, :].to(x.dtype)
        x, _ = nn.functional.multi_head_attention_forward(
            query=x[:1], key=x, value=x, embed_dim_to_check=x.shape[-1], num_heads=self.num_heads, q_proj_weight=self.q_proj.weight,
            k_proj_weight=self.k_proj.weight, v_proj_weight=self.v_proj.weight, in_proj_weight=None,
            in_proj_bias=torch.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]), bias_k=None, bias_v=None, add_zero_attn=False,
            dropout_p=0.0, out_proj_weight=self.c_proj.weight, out_proj_bias=self.c_proj.bias, use_separate_proj_weight=True,
            training=False, need_weights=False)
        return x[0]


class ModifiedResNet(nn.Module):
    """CLIP RN50 by default: layers (3, 4, 6, 3), width 64 -> 2048 channels at 7 x 7, 32 heads of 64, 1024-d output."""

    def __init__(self, layers=(3, 4, 6, 3), output_dim=1024, heads=32, image_size=224, width=64):
        super().__init__()
        self.output_dim, self.image_size = output_dim, image_size
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.act2 = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.act3 = nn.ReLU(inplace=True)
        self.avgpool = nn.AvgPool2d(2)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)
        self.attnpool = AttentionPool2d(image_size // 32, width * 32, heads, output_dim)

    def _make_layer(self, planes, blocks, stride=1):
        out = [ClipBottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * 4
        out += [ClipBottleneck(self._inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*out)

    def stem(self, x):
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.act2(self.bn2(self.conv2(x)))
        x = self.act3(self.bn3(self.conv3(x)))
        return self.avgpool(x)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(self.stem(x)))))
        return self.attnpool(x)


class _ClipRN50Model(_ClipModel):
    """RN50-CLIP: `visual` is the ModifiedResNet (the whole image tower), the text tower as in `_ClipModel` (12 x 512, ctx 77)."""

    def __init__(self, embed_dim=1024, image_size=224, layers=(3, 4, 6, 3), width=64, **text_arch):
        super().__init__(embed_dim=embed_dim, image_size=image_size, patch=32, v_width=64, v_layers=0, v_heads=1, **text_arch)
        for name in ("conv1", "class_embedding", "positional_embedding_v", "ln_pre", "ln_post", "proj_v"):
            delattr(self, name)
        self.visual = ModifiedResNet(layers, embed_dim, width * 32 // 64, image_size, width)
        for m in self.visual.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def encode_image(self, img):
        return self.visual(img)


class SyntheticClipRN50(SyntheticClip):
    """Random-init CLIP RN50 behind the ``AbstractVLM`` seam (what ``OpenClip("RN50", ...)`` would be without weights):
    ModifiedResNet image tower (38 M parameters, 2048 x 7 x 7 -> attention pool -> 1024), 12 x 512 text tower, 1024-d joint space."""

    def __init__(self, device="cpu", seed: int = 1, embed_dim: int = 1024, **arch):
        torch.manual_seed(seed)
        self.model = _ClipRN50Model(embed_dim=embed_dim, **arch).eval().to(device)
        self.name = f"synthetic-clip-rn50-d{embed_dim}"
