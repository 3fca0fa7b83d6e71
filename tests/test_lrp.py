"""`semanticlens_amd.component_visualization.lrp` — the EpsilonPlusFlat rule set restated on PyTorch autograd.

zennit / zennit-crp (what the reference's relevance visualizer wraps, relevance_based.py:16-19) are absent, so these tests
pin the rules against their published definitions computed by hand, and against properties LRP guarantees: conservation on
bias-free networks and LRP-0 == gradient x input on ReLU networks.  Pure PyTorch: runs on the CPU."""
import pytest
import torch
from torch import nn

from semanticlens_amd.component_visualization.lrp import epsilon_plus_flat, lrp_epsilon_plus_flat
from semanticlens_amd.component_visualization.relevance_based import gradient_x_activation


class Net(nn.Module):
    def __init__(self, bias=False):
        super().__init__()
        self.c1 = nn.Conv2d(3, 4, 3, bias=bias)
        self.r1 = nn.ReLU(inplace=True)
        self.c2 = nn.Conv2d(4, 5, 3, bias=bias)
        self.bn = nn.BatchNorm2d(5)
        self.r2 = nn.ReLU()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(5, 3, bias=bias)

    def forward(self, x):
        return self.fc(self.pool(self.r2(self.bn(self.c2(self.r1(self.c1(x)))))).flatten(1))


def _net(bias=False, seed=0):
    torch.manual_seed(seed)
    m = Net(bias).eval()
    with torch.no_grad():  # identity batch norm so that the pass-through rule stays conservative in this test
        m.bn.weight.fill_(1.0), m.bn.bias.zero_(), m.bn.running_mean.zero_(), m.bn.running_var.fill_(1.0 - m.bn.eps)
    return m


def test_relevance_is_conserved_and_matches_the_rules_written_out():
    m = _net()
    x = torch.rand(3, 3, 8, 8)
    layers = {n: mod for n, mod in m.named_modules() if n in ("r1", "c2", "r2", "fc")}
    out = lrp_epsilon_plus_flat(m, layers, x, None)
    assert m.r1.inplace and all(p.grad is None for p in m.parameters())  # nothing of the caller's model is left modified
    with torch.no_grad():
        a1 = m.c1(x).clamp(min=0)
        a2 = m.c2(a1).clamp(min=0)
        p = a2.mean((2, 3))
        y = p @ m.fc.weight.T
    t = y.argmax(1)
    idx = torch.arange(3)
    for name, (act, rel) in out.items():  # bias-free: every layer carries exactly the target logit
        assert act.shape == rel.shape
        assert torch.allclose(rel.flatten(1).sum(1), y[idx, t], rtol=1e-3, atol=1e-6), name
    assert torch.equal(out["r1"][0], a1) and torch.allclose(out["fc"][0], y)
    # the rules, written out: epsilon at fc, norm at the pooling, z+ at c2
    sgn = lambda z: torch.where(z >= 0, torch.ones_like(z), -torch.ones_like(z))  # noqa: E731
    r_y = torch.zeros_like(y)
    r_y[idx, t] = y[idx, t]
    assert torch.allclose(out["fc"][1], r_y)
    r_p = p * ((r_y / (y + 1e-6 * sgn(y))) @ m.fc.weight)
    r_a2 = a2 * (r_p / (p + 1e-6))[:, :, None, None] / (a2.shape[2] * a2.shape[3])
    assert torch.allclose(out["r2"][1], r_a2, rtol=1e-5, atol=1e-8)
    wp = m.c2.weight.detach().clamp(min=0)
    z = nn.functional.conv2d(a1, wp)  # a1 >= 0: the (a-, w-) pass contributes nothing
    r_a1 = a1 * nn.functional.conv_transpose2d(r_a2 / (z + 1e-6 * sgn(z)), wp)
    assert torch.allclose(out["r1"][1], r_a1, rtol=1e-5, atol=1e-8)
    # explicit targets condition on the label
    lab = torch.tensor([0, 1, 2])
    out_l = lrp_epsilon_plus_flat(m, {"fc": m.fc}, x, lab)
    assert torch.allclose(out_l["fc"][1].sum(1), y[idx, lab])


def test_flat_rule_spreads_relevance_evenly_over_the_first_layers_receptive_fields():
    m = _net()
    x = torch.rand(2, 3, 8, 8)
    with epsilon_plus_flat(m):
        xx = x.clone().requires_grad_(True)
        y = m(xx)
        start = torch.zeros_like(y)
        start[:, 0] = y[:, 0].detach()
        (r_x,) = torch.autograd.grad(y, xx, grad_outputs=start)
    assert torch.allclose(r_x.flatten(1).sum(1), y[:, 0].detach(), rtol=1e-2, atol=1e-6)  # the 1e-6 stabilisers absorb a little
    assert torch.allclose(r_x[:, 0], r_x[:, 1]) and torch.allclose(r_x[:, 0], r_x[:, 2])  # independent of the pixel values / channel
    # outside the context the model differentiates normally again
    xx = x.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(m(xx)[:, 0].sum(), xx)
    with torch.no_grad():
        eps_ = 1e-3
        d = torch.zeros_like(x)
        d[0, 1, 3, 4] = eps_
        fd = (m(x + d)[0, 0] - m(x - d)[0, 0]) / (2 * eps_)
    assert abs(g[0, 1, 3, 4].item() - fd.item()) < 1e-3


def test_epsilon_rule_everywhere_equals_gradient_x_activation_on_a_relu_network():
    """LRP-0 / LRP-epsilon with eps -> 0 on a ReLU network is gradient x input (Ancona et al. 2018): dense ReLU stack,
    epsilon rule on every layer (an all-`nn.Linear` network, no flat first layer) against plain autograd."""
    torch.manual_seed(1)
    m = nn.Sequential(nn.Linear(7, 9), nn.ReLU(), nn.Linear(9, 6), nn.ReLU(), nn.Linear(6, 4)).eval()
    x = torch.randn(5, 7)
    layers = {"1": m[1], "3": m[3]}
    want = gradient_x_activation(m, layers, x, None)
    kept = {}
    hooks = []
    with epsilon_plus_flat(m, epsilon=1e-9, first_layer_flat=False):
        hooks = [mod.register_forward_hook(lambda mo, i, o, n=n: kept.__setitem__(n, o)) for n, mod in layers.items()]
        xx = x.clone().requires_grad_(True)
        y = m(xx)
        t = y.argmax(1)
        start = torch.zeros_like(y).scatter_(1, t[:, None], 1.0) * y.detach()
        grads = torch.autograd.grad(y, [kept["1"], kept["3"]], grad_outputs=start)
    for h in hooks:
        h.remove()
    # biases absorb relevance under the epsilon rule, so compare on the bias-free part: R = a * grad holds when b = 0
    with torch.no_grad():
        for mod in m:
            if isinstance(mod, nn.Linear):
                mod.bias.zero_()
    want = gradient_x_activation(m, layers, x, None)
    kept.clear()
    with epsilon_plus_flat(m, epsilon=1e-9, first_layer_flat=False):
        hooks = [mod.register_forward_hook(lambda mo, i, o, n=n: kept.__setitem__(n, o)) for n, mod in layers.items()]
        xx = x.clone().requires_grad_(True)
        y = m(xx)
        t = y.argmax(1)
        start = torch.zeros_like(y).scatter_(1, t[:, None], 1.0) * y.detach()
        grads = torch.autograd.grad(y, [kept["1"], kept["3"]], grad_outputs=start)
    for h in hooks:
        h.remove()
    for name, g in zip(("1", "3"), grads):
        assert torch.allclose(g, want[name][1], rtol=1e-4, atol=1e-6), name


def test_unsupported_padding_mode_is_reported():
    m = nn.Sequential(nn.Conv2d(3, 2, 3, padding=1, padding_mode="reflect"), nn.ReLU(), nn.Flatten(), nn.Linear(2 * 16, 2)).eval()
    with pytest.raises(NotImplementedError, match="padding_mode"):
        lrp_epsilon_plus_flat(m, {"1": m[1]}, torch.rand(1, 3, 4, 4), None)


class _BasicBlock(nn.Module):
    """torchvision's BasicBlock pattern: in-place ReLU modules and an in-place residual add on a hooked leaf's output."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out += identity  # in-place on the output of a rule's autograd node
        return nn.functional.relu(out, inplace=True)  # functional in-place activation: not a module, no rule


def test_residual_block_with_in_place_add_and_in_place_relu():
    """rule outputs were views made inside a custom Function; `out += identity` raised
    'Output 0 of _PassRuleBackward is a view and is being modified inplace' on every ResNet-family model."""
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(3, 6, 3, padding=1, bias=False), nn.ReLU(inplace=True), _BasicBlock(6), _BasicBlock(6),
                      nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(6, 4, bias=False)).eval()
    x = torch.rand(2, 3, 8, 8)
    layers = {"2": m[2], "3": m[3], "6": m[6]}
    out = lrp_epsilon_plus_flat(m, layers, x, None)
    with torch.no_grad():
        y = m(x)
    t = y.argmax(1)
    assert torch.allclose(out["6"][1].sum(1), y[torch.arange(2), t])
    for name in ("2", "3"):
        act, rel = out[name]
        assert act.shape == rel.shape == (2, 6, 8, 8) and torch.isfinite(rel).all() and rel.abs().sum() > 0
    assert torch.allclose(out["3"][0], m[:4](x))  # the hooked activations are the model's own
    assert m[1].inplace and m[2].relu.inplace  # and the in-place flags are back


def test_zplus_keeps_the_positive_part_of_the_bias_in_its_first_pass():
    """zennit's ZPlus clamps weight AND bias at zero from below in the (a+, w+) pass and zeroes the bias in the (a-, w-)
    pass: z = conv(a+, w+) + b+ + conv(a-, w-)."""
    torch.manual_seed(2)
    m = nn.Sequential(nn.Conv2d(2, 3, 3, bias=True), nn.ReLU(), nn.Conv2d(3, 4, 3, bias=True), nn.ReLU(), nn.AdaptiveAvgPool2d(1),
                      nn.Flatten(), nn.Linear(4, 2, bias=False)).eval()
    with torch.no_grad():
        m[2].bias.copy_(torch.tensor([0.5, -0.7, 0.25, -0.1]))
    x = torch.rand(2, 2, 7, 7)
    out = lrp_epsilon_plus_flat(m, {"1": m[1], "3": m[3]}, x, None)
    a1, r_a2 = out["1"][0], out["3"][1]
    wp, bp = m[2].weight.detach().clamp(min=0), m[2].bias.detach().clamp(min=0)
    z = nn.functional.conv2d(a1, wp, bp)  # a1 >= 0 after the ReLU: the (a-, w-) pass is zero
    want = a1 * nn.functional.conv_transpose2d(r_a2 / (z + 1e-6 * torch.where(z >= 0, 1.0, -1.0)), wp)
    assert torch.allclose(out["1"][1], want, rtol=1e-5, atol=1e-8)


def test_first_layer_is_the_first_linear_module_in_module_order():
    """zennit's SpecialFirstLayerMapComposite: module order, not call order."""

    class Swapped(nn.Module):
        def __init__(self):
            super().__init__()
            self.late = nn.Linear(5, 3, bias=False)  # registered first, called second
            self.early = nn.Linear(4, 5, bias=False)

        def forward(self, x):
            return self.late(torch.relu(self.early(x)))

    torch.manual_seed(3)
    m = Swapped().eval()
    x = torch.rand(2, 4) + 0.1
    with epsilon_plus_flat(m):
        xx = x.clone().requires_grad_(True)
        y = m(xx)
        (r_x,) = torch.autograd.grad(y, xx, grad_outputs=y.detach())
    # `early` runs the epsilon rule (relevance depends on x), it is not the flat one
    assert not torch.allclose(r_x[:, 0], r_x[:, 1])


def test_layer_norm_passes_relevance_through():
    """A ConvNeXt-style stack (depthwise conv -> LayerNorm -> Linear -> GELU -> Linear -> residual, BASELINE configs[4]'s probed
    model): with LayerNorm left to autograd the relevance grew ~100x per block (3.6e19 at stage 0 of this 9-block toy) and
    overflowed on ConvNeXt-L; passed through, what is left is the doubling at every un-canonised residual add and the epsilon
    rule's 1 / z on signed pre-activations (~4x per block: finite in fp32 for ConvNeXt-L's 36 blocks)."""
    import synth

    torch.manual_seed(0)
    m = synth.ConvNeXt(depths=(2, 2, 3, 2), dims=(16, 32, 64, 128), layer_scale=1.0).eval()
    x = torch.randn(2, 3, 64, 64)
    layers = {f"stages.{i}": m.stages[i] for i in range(4)}
    out = lrp_epsilon_plus_flat(m, layers, x, None, norm_pass=True)
    with torch.no_grad():
        y = m(x)
    top = y.max(1).values.abs().max().item()
    for name, (act, rel) in out.items():
        assert torch.isfinite(rel).all(), name
        assert 0 < rel.abs().max().item() < 1e8 * max(top, 1.0), (name, rel.abs().max().item(), top)
    # a LayerNorm alone: relevance in == relevance out
    ln = nn.Sequential(nn.Linear(6, 6, bias=False), nn.LayerNorm(6), nn.Linear(6, 3, bias=False)).eval()
    xin = torch.rand(4, 6) + 0.5
    res = lrp_epsilon_plus_flat(ln, {"0": ln[0], "1": ln[1]}, xin, None, norm_pass=True)
    assert torch.allclose(res["0"][1], res["1"][1])
    # the DEFAULT is zennit's EpsilonPlusFlat: LayerNorm is left to autograd, so the relevance at its input is the relevance at
    # its output pushed through the LayerNorm Jacobian (and differs from the pass-through variant)
    res0 = lrp_epsilon_plus_flat(ln, {"0": ln[0], "1": ln[1]}, xin, None)
    assert torch.allclose(res0["1"][1], res["1"][1])  # downstream of the norm nothing changes
    h = ln[0](xin).detach().requires_grad_(True)
    (want,) = torch.autograd.grad(ln[1](h), h, grad_outputs=res0["1"][1])
    assert torch.allclose(res0["0"][1], want, atol=1e-6) and not torch.allclose(res0["0"][1], res["0"][1])
