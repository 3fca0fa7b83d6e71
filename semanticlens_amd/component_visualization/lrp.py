"""Layer-wise relevance propagation for the relevance visualizer — the ``EpsilonPlusFlat`` rule set.

The reference's ``RelevanceComponentVisualizer`` (``semanticlens/component_visualization/relevance_based.py:19,104-149``)
hands the attribution to zennit-crp with zennit's ``EpsilonPlusFlat`` composite: the z+ rule for convolutions, the epsilon
rule for dense layers, the flat rule for the first layer, and pass-through for activations and batch norm.  zennit and
zennit-crp are third-party packages that are neither vendored nor installed here, so **parity for this row is unpinned**;
this module restates the published rules on PyTorch autograd (the probed model's forward and backward stay PyTorch's, as
for the activation visualizer) so that the relevance fed to K1 (sum) / ``sl_abs_norm_rows`` / K3 is LRP relevance and not
only gradient x activation:

* every rule is applied the way zennit does, as a modified *gradient*: a module's forward is left untouched and its
  backward is replaced so that the "gradient" that arrives at a tensor IS that tensor's relevance.  For a linear module
  (convolution or dense layer) with input ``a`` and incoming relevance ``R_out``:
  ``z = sum_m f_m(a; w_m)``, ``s = R_out / stabilise(z)``, ``R_in = sum_m a_m * d(z_m . s)/d a_m`` over the rule's modified
  inputs / weights ``m``:

  =========  ==============================================================  ==================================
  rule       modified passes                                                 used for
  =========  ==============================================================  ==================================
  epsilon    ``(a, w, b)``; ``stabilise(z) = z + eps * sign(z)``, eps 1e-6   ``nn.Linear``
  z+         ``(a+, w+, b+)`` and ``(a-, w-, 0)``                            ``nn.Conv*``
  flat       ``(1, 1)``: ``R_out`` spread evenly over the receptive field    the first linear module (module order)
  norm       ``(a,)`` through the module itself                              average pooling
  pass       ``R_in = R_out``                                                activations, batch norm, dropout (layer / group
                                                                             norm only with ``norm_pass=True``)
  =========  ==============================================================  ==================================

  Max pooling and tensor additions keep PyTorch's own gradient (winner-takes-all; an un-canonised residual ``x + f(x)``
  hands ``R_out`` to both branches, which is also what zennit does without its ResNet canoniser).
* the relevance of a hooked layer is what arrives at its OUTPUT, started from ``R = logit[target]`` at the model's output
  (crp's ``CondAttribution`` with ``{"y": target}``).

``lrp_epsilon_plus_flat`` has the signature of ``relevance_based.gradient_x_activation`` and returns
``{layer: (activation, relevance)}`` per batch.
"""
from __future__ import annotations

from contextlib import contextmanager

import torch
from torch import nn

_CONVS = (nn.Conv1d, nn.Conv2d, nn.Conv3d)
# zennit's EpsilonPlusFlat maps only BatchNorm to `Pass` and leaves nn.LayerNorm / nn.GroupNorm to plain autograd: that is the
# DEFAULT here too (zennit parity; changing it changes the relevance top-k ids of every ViT / ConvNeXt).  Autograd's LayerNorm
# Jacobian scales the incoming relevance by gamma / sigma per block, and on ConvNeXt-L (36 LayerNorm blocks, BASELINE
# configs[4]) the relevance of the early stages overflows to inf / NaN (tests/test_lrp.py).  `norm_pass=True` is the explicit
# opt-in that passes relevance through layer / group norm like batch norm and stays finite there; the visualizer names it in
# its composite ("..._normpass") so cache directories of the two variants never mix.
_PASS = (nn.ReLU, nn.ReLU6, nn.LeakyReLU, nn.ELU, nn.GELU, nn.SiLU, nn.Sigmoid, nn.Tanh, nn.Hardswish, nn.Hardtanh, nn.Softplus,
         nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.Dropout, nn.Dropout2d, nn.Identity)
_NORM_PASS = (nn.LayerNorm, nn.GroupNorm)
_AVGPOOL = (nn.AvgPool1d, nn.AvgPool2d, nn.AvgPool3d, nn.AdaptiveAvgPool1d, nn.AdaptiveAvgPool2d, nn.AdaptiveAvgPool3d)


_IN_RULE = [False]  # a rule's backward is re-running a module: its forward hooks must stay out of the way


def _stabilise(z: torch.Tensor, eps: float) -> torch.Tensor:
    return z + eps * torch.where(z >= 0, torch.ones_like(z), -torch.ones_like(z))


def _alias(out: torch.Tensor) -> torch.Tensor:
    """What a rule's forward returns: the module's own output, as a NON-view alias.  ``out.view_as(out)`` made inside a custom
    Function is a view autograd refuses to see modified in place, and torchvision-style residual blocks do exactly that
    (``out = bn2(conv2(x)); out += identity``, functional in-place ReLUs).  ``detach()`` shares storage and version counter
    with ``out`` — so an in-place edit of a tensor another node saved is still caught — without being a view."""
    return out.detach()


def _functional(module: nn.Module):
    """``f(a, w, b)`` computing the module's linear map with other weights."""
    if isinstance(module, nn.Linear):
        return lambda a, w, b: nn.functional.linear(a, w, b)
    fn = {1: nn.functional.conv1d, 2: nn.functional.conv2d, 3: nn.functional.conv3d}[module.weight.ndim - 2]
    if module.padding_mode != "zeros":
        raise NotImplementedError(f"LRP for {type(module).__name__} with padding_mode={module.padding_mode!r}")
    return lambda a, w, b: fn(a, w, b, module.stride, module.padding, module.dilation, module.groups)


class _LinearRule(torch.autograd.Function):
    """Forward: the module's own output.  Backward: the rule's relevance redistribution (see the module docstring)."""

    @staticmethod
    def forward(ctx, a, out, module, rule, eps):
        ctx.module, ctx.rule, ctx.eps = module, rule, eps
        ctx.save_for_backward(a)
        return _alias(out)

    @staticmethod
    def backward(ctx, r_out):
        (a,) = ctx.saved_tensors
        module, rule, eps = ctx.module, ctx.rule, ctx.eps
        f = _functional(module)
        w = module.weight.detach()
        b = module.bias.detach() if module.bias is not None else None
        if rule == "epsilon":
            passes = [(a.detach(), w, b)]
        elif rule == "zplus":
            # zennit's ZPlus: ClampMod(min=0) on weight AND bias in the (a+, w+) pass, the bias zeroed in the (a-, w-) pass
            bp = b.clamp(min=0) if b is not None else None
            passes = [(a.detach().clamp(min=0), w.clamp(min=0), bp), (a.detach().clamp(max=0), w.clamp(max=0), None)]
        elif rule == "flat":
            passes = [(torch.ones_like(a), torch.ones_like(w), None)]
        else:
            raise ValueError(rule)
        with torch.enable_grad():
            ins = [p[0].requires_grad_(True) for p in passes]
            z = sum(f(x, pw, pb) for x, (_, pw, pb) in zip(ins, passes))
            s = (r_out / _stabilise(z, eps)).detach()
            grads = torch.autograd.grad((z * s).sum(), ins)
        r_in = sum(x.detach() * g for x, g in zip(ins, grads))
        return r_in, None, None, None, None


class _NormRule(torch.autograd.Function):
    """Average pooling: ``R_in = a * pool^T(R_out / pool(a))`` (zennit ``Norm``)."""

    @staticmethod
    def forward(ctx, a, out, module, eps):
        ctx.module, ctx.eps = module, eps
        ctx.save_for_backward(a)
        return _alias(out)

    @staticmethod
    def backward(ctx, r_out):
        (a,) = ctx.saved_tensors
        _IN_RULE[0] = True
        try:
            with torch.enable_grad():
                x = a.detach().requires_grad_(True)
                z = ctx.module(x)
                s = (r_out / _stabilise(z, ctx.eps)).detach()
                (g,) = torch.autograd.grad((z * s).sum(), x)
        finally:
            _IN_RULE[0] = False
        return x.detach() * g, None, None, None


class _PassRule(torch.autograd.Function):
    """Activations / batch norm: the relevance goes through unchanged."""

    @staticmethod
    def forward(ctx, a, out):
        return _alias(out)

    @staticmethod
    def backward(ctx, r_out):
        return r_out, None


@contextmanager
def epsilon_plus_flat(model: nn.Module, epsilon: float = 1e-6, first_layer_flat: bool = True, norm_pass: bool = False):
    """While active, the backward pass of ``model`` propagates EpsilonPlusFlat relevance instead of gradients.
    ``norm_pass=True`` (not zennit's behaviour, opt-in) also passes relevance through ``nn.LayerNorm`` / ``nn.GroupNorm``.

    Rules attach to leaf modules by type (table in the module docstring).  In-place activations are switched to
    out-of-place for the duration (their input is needed by the rule of the module in front of them)."""
    handles, restored = [], []
    # zennit's SpecialFirstLayerMapComposite picks the first linear module in MODULE order (registration order of
    # `model.modules()`), not the first one the forward pass happens to call
    first_linear = next((m for m in model.modules() if not len(list(m.children())) and isinstance(m, (nn.Linear,) + _CONVS)), None)

    def hook_for(kind):
        def hook(module, inputs, output):
            a = inputs[0]
            if _IN_RULE[0] or not (torch.is_tensor(a) and torch.is_tensor(output)) or not torch.is_grad_enabled():
                return None
            if kind == "linear":
                _functional(module)  # refuses what the rules cannot re-run (padding modes) now, not in the backward pass
                if not a.requires_grad:  # the model's input: keep the graph connected so that the flat rule can run
                    a = a.detach().requires_grad_(True)
                rule = "epsilon" if isinstance(module, nn.Linear) else "zplus"
                if module is first_linear and first_layer_flat:
                    rule = "flat"
                return _LinearRule.apply(a, output, module, rule, epsilon)
            if not a.requires_grad:
                return None
            if kind == "norm":
                return _NormRule.apply(a, output, module, epsilon)
            return _PassRule.apply(a, output)

        return hook

    for module in model.modules():
        if len(list(module.children())):
            continue
        if getattr(module, "inplace", False):
            module.inplace = False
            restored.append(module)
        if isinstance(module, (nn.Linear,) + _CONVS):
            handles.append(module.register_forward_hook(hook_for("linear")))
        elif isinstance(module, _AVGPOOL):
            handles.append(module.register_forward_hook(hook_for("norm")))
        elif isinstance(module, _PASS) or (norm_pass and isinstance(module, _NORM_PASS)):
            handles.append(module.register_forward_hook(hook_for("pass")))
    try:
        yield
    finally:
        for h in handles:
            h.remove()
        for module in restored:
            module.inplace = True


def lrp_epsilon_plus_flat(model: nn.Module, layers: dict[str, nn.Module], images: torch.Tensor, targets: torch.Tensor | None,
                          epsilon: float = 1e-6, norm_pass: bool = False):
    """``{layer: (activation, relevance)}`` for one batch under the EpsilonPlusFlat rules, relevance started from the target
    logit (``targets`` None = the model's own prediction).  Same contract as ``gradient_x_activation``.  ``norm_pass``: see
    :func:`epsilon_plus_flat` (default False = zennit's rule set)."""
    kept: dict[str, torch.Tensor] = {}

    def keep(name):
        def hook(module, ins, out):
            kept[name] = out

        return hook

    with epsilon_plus_flat(model, epsilon, norm_pass=norm_pass):
        # registered AFTER the rule hooks, so `out` is the tensor the rule's autograd node produced
        handles = [m.register_forward_hook(keep(n)) for n, m in layers.items()]
        try:
            with torch.enable_grad():
                x = images.detach().requires_grad_(True)
                logits = model(x)
                if logits.ndim != 2:
                    raise ValueError(f"the probed model must return (B, n_classes) logits, got shape {tuple(logits.shape)}")
                if targets is None:
                    targets = logits.argmax(dim=1)
                start = torch.zeros_like(logits).scatter_(1, targets.reshape(-1, 1).to(logits.device), 1.0) * logits.detach()
                names = [n for n, a in kept.items() if a.requires_grad]
                grads = torch.autograd.grad(logits, [kept[n] for n in names], grad_outputs=start, allow_unused=True) if names else ()
        finally:
            for h in handles:
                h.remove()
    by_name = dict(zip(names, grads))
    out = {}
    for name, act in kept.items():
        rel = by_name.get(name)
        out[name] = (act.detach(), (rel if rel is not None else torch.zeros_like(act)).detach())
    return out
