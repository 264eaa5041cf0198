"""The README's other published models (README.rst:259-267: DenseNet-121 / 161 next to the ResNets): KFAC on an ImageNet
DenseNet-121, N = 32, through the API.  A dense block's layer geometries are not a ResNet's - 1x1 convolutions whose input
width grows by 32 per unit (64, 96, ... 1024: most of them no multiple of 128), 58 3x3 convolutions with 128 input channels
at four resolutions, 1x1 transitions - so the factor build is checked against the fp64 oracle on one layer of each class at
the full batch, the whole model through identities that hold at any size, invert / sample through their defining identity."""
import pytest
import torch

from conftest import identity_residual_bound, rel_fro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def densenet_kfac(gpu):
    from curvature_amd import models
    from curvature_amd.curvatures import KFAC
    torch.manual_seed(0)
    model = models.densenet121().to(gpu).train()
    kfac = KFAC(model)
    x = torch.randn(32, 3, 224, 224, device=gpu)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    kfac.update(batch_size=32)
    torch.cuda.synchronize()
    return model, kfac


def test_layer_table_and_factor_identities(gpu, densenet_kfac):
    model, kfac = densenet_kfac
    layers = kfac._layers()
    assert len(layers) == 121 and list(kfac.state.keys()) == layers           # 120 convolutions + classifier, modules() order
    widths = [l.in_channels for l in layers if l.__class__.__name__ == "Conv2d" and l.kernel_size == (1, 1)]
    assert widths[:7] == [64, 96, 128, 160, 192, 224, 256] and max(widths) == 1024 and len(widths) == 61   # (1024: transition 3)
    for layer in layers:
        A, G = kfac.state[layer]
        assert torch.equal(A, A.t()) and torch.equal(G, G.t())
        assert torch.isfinite(A).all() and torch.isfinite(G).all()
        x, g = kfac.record[layer]
        N = x.shape[0]
        L = g.shape[2] * g.shape[3] if g.dim() == 4 else 1
        tr_g = float(g.detach().double().pow(2).sum()) * N / L
        assert abs(float(torch.trace(G.double())) - tr_g) <= 1e-5 * abs(tr_g)
        if layer.__class__.__name__ == "Conv2d" and layer.kernel_size == (1, 1):
            tr_a = float(x.detach().double().pow(2).sum()) / (N * L)
            assert abs(float(torch.trace(A.double())) - tr_a) <= 1e-5 * abs(tr_a)


@pytest.mark.parametrize("name", [
    "features.0",            # 7x7 / stride 2 stem
    "features.5.conv1",      # block 1, unit 2: 1x1, C = 96 at 56x56
    "features.9.conv1",      # block 1, last unit: 1x1, C = 224
    "features.9.conv2",      # 3x3, C = 128 at 56x56
    "features.12",           # transition 1: 1x1, 256 -> 128 at 56x56
    "features.25.conv1",     # block 2, last unit: 1x1, C = 480 at 28x28
    "features.25.conv2",     # 3x3, C = 128 at 28x28
    "features.73.conv1",     # block 4, last unit: 1x1, C = 992 at 7x7
    "features.73.conv2",     # 3x3, C = 128 at 7x7
])
def test_factor_build_matches_the_oracle_at_densenet_geometry(gpu, densenet_kfac, name):
    import oracle.curvature_oracle as o
    model, kfac = densenet_kfac
    layer = dict(model.named_modules())[name]
    assert layer.__class__.__name__ == "Conv2d", name
    x, g = kfac.record[layer]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))
    try:
        A, G = o.kfac_factors(x.detach().double().cpu(), g.detach().double().cpu(), has_bias=False, **o.layer_geometry(layer))
    finally:
        torch.set_num_threads(threads)
    ea, eg = rel_fro(kfac.state[layer][0], A), rel_fro(kfac.state[layer][1], G)
    assert ea < 1e-4 and eg < 1e-4, (name, ea, eg)


def test_invert_and_sample(gpu, densenet_kfac):
    model, kfac = densenet_kfac
    kfac.invert(add=1.0, multiply=1000.0)
    layers = kfac._layers()
    for layer in sorted(layers, key=lambda l: -kfac.state[l][0].shape[0])[:2] + [layers[0], layers[-1]]:
        for F, Lf in zip(kfac.state[layer], kfac.inv_state[layer]):
            n = F.shape[0]
            assert torch.equal(Lf, torch.tril(Lf))
            M = (1000.0 ** 0.5) * F.double() + torch.eye(n, device=gpu, dtype=torch.float64)
            M = (M + M.t()) / 2
            R = (Lf.double() @ Lf.double().t()) @ M - torch.eye(n, device=gpu, dtype=torch.float64)
            assert float(torch.linalg.norm(R)) / n ** 0.5 < identity_residual_bound(M), n
    mean = {k: v.clone() for k, v in kfac.model_state.items()}
    kfac.sample_and_replace()
    torch.cuda.synchronize()
    state = model.state_dict()
    assert all(torch.isfinite(v).all() for v in state.values())
    changed = sum(int(not torch.equal(state[k], mean[k])) for k in state)
    assert changed == 121 + 1                          # 121 weights and the classifier's bias; BatchNorm tensors restored
