"""Plugin-API behaviour that needs no GPU: layer selection, hyper-parameter indexing, error convention
(mirrors curvature/curvatures.py:38-65, :361-365 of the reference)."""
import numpy as np
import pytest
import torch

from curvature_amd import models
from curvature_amd.curvatures import Curvature, Diagonal, KFAC


def test_layer_types_normalisation():
    m = models.lenet5()
    assert KFAC(m).layer_types == ['Linear', 'Conv2d', 'MultiheadAttention']
    assert KFAC(m, []).layer_types == ['Linear', 'Conv2d', 'MultiheadAttention']
    assert KFAC(m, 'Conv2d').layer_types == ['Conv2d']
    assert [l.__class__.__name__ for l in KFAC(m, 'Conv2d')._layers()] == ['Conv2d', 'Conv2d']
    assert len(KFAC(m, ['Linear'])._layers()) == 3
    with pytest.raises(AssertionError):
        KFAC(m, ['Conv3d'])
    with pytest.raises(TypeError):
        KFAC(m, 3)


def test_layer_order_is_modules_order():
    m = models.resnet18()
    layers = Diagonal(m)._layers()
    ref = [l for l in m.modules() if l.__class__.__name__ in ('Linear', 'Conv2d')]
    assert layers == ref and len(layers) == 21
    # the downsample 1x1 conv comes after the block's main convs
    names = {mod: n for n, mod in m.named_modules()}
    assert names[layers[7]] == 'layer2.0.downsample.0'


def test_hyper_indexing():
    h = Curvature._hyper
    assert h(0.5, 1, 3, 5) == (0.5, 1.0)                              # scalars (int accepted)
    assert h(np.float32(0.5), np.float64(2), 0, 5) == (0.5, 2.0)      # numpy scalars (superset of the reference)
    assert h([1, 2, 3], (4, 5, 6), 1, 3) == (2.0, 5.0)                # both lists -> per layer
    with pytest.raises(TypeError):
        h([1, 2, 3], 7.0, 1, 3)                                        # mixed: float(list), as in the reference
    with pytest.raises(AssertionError):
        h([1, 2], [3, 4], 0, 3)                                        # wrong length


def test_mha_projections_are_layers_of_kfac_efb_inf():
    """The reference raises NotImplementedError for MultiheadAttention in KFAC / EFB / INF (curvatures.py:303-304); here
    its two projections are Linear-like layers, tapped off the attention forward's F.linear calls."""
    import torch.nn.functional as F
    from curvature_amd.curvatures import AttentionProjection, BlockDiagonal
    m = torch.nn.Sequential(torch.nn.Linear(4, 4))
    m.add_module("attn", torch.nn.MultiheadAttention(4, 2))
    k = KFAC(m)
    layers = k._layers()
    assert [type(l).__name__ for l in layers] == ["Linear", "AttentionProjection", "AttentionProjection"]
    assert layers[1:] == list(AttentionProjection.of(m.attn)) and layers[1].kind == "attn_in"
    assert (layers[1].in_features, layers[1].out_features, layers[2].out_features) == (4, 12, 4)
    assert k._global_index() == {layers[0]: 0, layers[1]: 1, layers[2]: 2}
    original = F.linear
    x = torch.randn(5, 3, 4)
    y, _ = m.attn(x, x, x)
    y.sum().backward()
    assert F.linear is original                                       # the tap lives for the attention forward only
    assert tuple(k.record[layers[1]][0].shape) == (5, 3, 4) and tuple(k.record[layers[1]][1].shape) == (5, 3, 12)
    assert tuple(k.record[layers[2]][1].shape) == (15, 4)
    assert len(KFAC(m, 'Linear')._layers()) == 1                      # not selected: untouched
    with pytest.raises(NotImplementedError):
        BlockDiagonal(m)._layers()                                    # (BlockDiagonal: as in the reference)
    kdim = torch.nn.MultiheadAttention(4, 2, kdim=6, vdim=6)
    with pytest.raises(NotImplementedError):
        KFAC(kdim)


def test_dilated_conv_rejected():
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, dilation=2))
    with pytest.raises(NotImplementedError):
        KFAC(m)


def test_invert_and_sample_need_state():
    k = KFAC(models.lenet5())
    with pytest.raises(AssertionError):
        k.invert()
    with pytest.raises(AssertionError):
        k.sample(k._layers()[0])


def test_model_state_is_a_deep_copy():
    m = models.lenet5()
    k = KFAC(m)
    w0 = k.model_state['0.weight'].clone()
    with torch.no_grad():
        m[0].weight.add_(1.0)
    assert torch.equal(k.model_state['0.weight'], w0)
    assert k.model_state_of(m[0], 'weight') is k.model_state['0.weight']


def test_named_state_round_trip(tmp_path):
    """curvature_amd.io: estimator state keyed by layer name, loaded into another model instance, and the
    reference's module-keyed format matched by position."""
    import torch
    from curvature_amd import io, models
    from curvature_amd.curvatures import KFAC
    m1, m2 = models.lenet5(), models.lenet5()
    k1, k2 = KFAC(m1), KFAC(m2)
    layers1 = [l for l in m1.modules() if l.__class__.__name__ in ("Linear", "Conv2d")]
    layers2 = [l for l in m2.modules() if l.__class__.__name__ in ("Linear", "Conv2d")]
    torch.manual_seed(0)
    k1.state = {l: [torch.randn(3, 3), torch.randn(2, 2)] for l in layers1}
    path = str(tmp_path / "kfac_state.pth")
    io.save_state(k1, path)
    io.load_state(k2, path)
    assert list(k2.state.keys()) == layers2
    for a, b in zip(layers1, layers2):
        assert torch.equal(k1.state[a][0], k2.state[b][0]) and torch.equal(k1.state[a][1], k2.state[b][1])
    names = io.named_state(k1)
    assert list(names.keys()) == [n for n, mod in m1.named_modules() if mod in layers1]
    k3 = KFAC(models.lenet5())
    io.load_state(k3, k1.state)                      # module-keyed dict of ANOTHER instance: by position
    assert torch.equal(list(k3.state.values())[4][1], k1.state[layers1[4]][1])


def test_partition_layers_resnet50():
    """Layer partition under the non-additive rank cost: every layer owned once, identical on every call, the
    4608-wide layers alone on their ranks at 8 GPUs (their serial chain is the step time there)."""
    from curvature_amd import models, sharding
    rows = models.layer_table(models.resnet50(), (3, 224, 224))
    dims = [(r["n"], r["m"], 32 * r["L"]) for r in rows]
    for world in (1, 2, 4, 8):
        owner = sharding.partition_layers(dims, world)
        assert owner == sharding.partition_layers(dims, world) and len(owner) == len(dims)
        assert set(owner) == set(range(world))
        est = [sharding.rank_cost([d for d, o in zip(dims, owner) if o == r]) for r in range(world)]
        additive = sharding.lpt_partition([sharding.rank_cost([d]) for d in dims], world)
        est_add = [sharding.rank_cost([d for d, o in zip(dims, additive) if o == r]) for r in range(world)]
        assert max(est) <= max(est_add) * 1.001
    owner8 = sharding.partition_layers(dims, 8)
    for i, d in enumerate(dims):
        if d[0] == 4608:
            assert sum(1 for o in owner8 if o == owner8[i]) == 1


def test_global_layer_index_and_shard_kwargs():
    """Per-layer hyper-parameter lists are indexed by the position among ALL selected layers in modules() order,
    also on a rank that owns only some of them; estimators accept `shard=` at construction."""
    from curvature_amd import sharding
    from curvature_amd.curvatures import EFB, INF
    m = models.lenet5()
    layers = [l for l in m.modules() if l.__class__.__name__ in ('Linear', 'Conv2d')]
    shard = sharding.Shard([0, 1, 0, 1, 1], rank=1, world=2)
    d = Diagonal(m, shard=shard)
    assert d._global_index() == {l: i for i, l in enumerate(layers)}
    assert [i for i, _ in d._owned()] == [1, 3, 4]
    k = KFAC(m, 'Linear', shard=sharding.Shard([0, 1, 1], rank=1, world=2))
    assert [i for i, _ in k._owned()] == [1, 2] and list(k._global_index().values()) == [0, 1, 2]
    # EFB / INF with given eigenvectors do no device work at construction
    eig = {l: (torch.eye(2), torch.eye(2)) for l in layers}
    e = EFB(m, {}, shard=shard, eigvecs=eig)
    assert e.eigvecs is eig and [layers.index(l) for l in e._mine()] == [1, 3, 4]
    facs = {l: [torch.eye(2), torch.eye(2)] for l in layers}
    i = INF(m, {l: torch.ones(2, 2) for l in layers}, facs, {l: torch.ones(2, 2) for l in layers}, shard=shard, eigvecs=eig)
    assert [layers.index(l) for l in i.diags.keys()] == [1, 3, 4] and list(i.lambdas.keys()) == list(i.diags.keys())
    with pytest.raises(AssertionError):
        INF(m, {}, facs, {l: torch.ones(2, 2) for l in layers}, eigvecs=eig)


def test_diagonal_state_order_with_attention():
    m = torch.nn.Sequential(torch.nn.Linear(4, 4))
    m.add_module("attn", torch.nn.MultiheadAttention(4, 2))
    m.add_module("attn2", torch.nn.MultiheadAttention(4, 2))
    m.add_module("out", torch.nn.Linear(4, 2))
    d = Diagonal(m)
    assert list(d._global_index().keys()) == [m[0], 'attn_in', 'attn_out', m.out]
    assert len(d._attention()) == 2 and len(d._layers()) == 2        # out_proj is not a selected 'Linear'


def test_noise_seed_is_lazy_and_per_instance():
    m = models.lenet5()
    a, b = KFAC(m), KFAC(m)
    assert a.noise_seed is None and b.noise_seed is None            # nothing drawn at construction
    torch.manual_seed(5)
    sa, sb = a._seed(), b._seed()
    assert sa != sb and a._seed() == sa                              # two draws from torch's generator; then pinned
    torch.manual_seed(5)
    assert KFAC(m)._seed() == sa


def test_block_diagonal_api():
    """BlockDiagonal has the reference's surface (curvatures.py:196-261); MultiheadAttention is rejected (the
    reference's branch for it raises inside torch.cat); nothing computes on the CPU."""
    from curvature_amd.curvatures import BlockDiagonal, Curvature
    assert issubclass(BlockDiagonal, Curvature)
    m = torch.nn.Sequential(torch.nn.Linear(4, 3))
    est = BlockDiagonal(m)
    with pytest.raises(AssertionError):
        est.invert()
    with pytest.raises(AssertionError):
        est.sample(m[0])
    m.add_module("attn", torch.nn.MultiheadAttention(4, 2))
    with pytest.raises(NotImplementedError):
        BlockDiagonal(m).update(batch_size=1)
    m[0].weight.grad = torch.zeros(3, 4)
    m[0].bias.grad = torch.zeros(3)
    with pytest.raises(RuntimeError):
        BlockDiagonal(m, 'Linear').update(batch_size=1)           # CPU tensors: no fallback


def test_layer_list_is_walked_once_and_survives_layer_type_changes():
    """`_layers()` walks the module tree once per estimator (every phase of a step calls it: 0.1 ms per walk on a
    ResNet-50, with the GPU waiting behind invert()'s read-back); the cache is keyed on the model object and on
    `layer_types`, and callers get their own list."""
    m = models.lenet5()
    k = KFAC(m)
    first = k._layers()
    assert k._layers() == first and k._layers() is not first
    first.clear()                                             # a caller's list is its own
    assert len(k._layers()) == 5
    calls = []
    orig = m.modules
    m.modules = lambda: (calls.append(1), orig())[1]
    assert len(k._layers()) == 5 and not calls                # no further walk
    k.layer_types = ['Linear']
    assert len(k._layers()) == 3 and calls                    # a changed selection is a new walk


def test_two_estimators_tapping_one_attention_module_leave_f_linear_alone():
    """Both estimators record the projections' inputs / gradients, and F.linear is the original function again after
    the forward (a per-estimator patch restored in registration order would leave the first tap's wrapper behind)."""
    import torch.nn.functional as F
    attn = torch.nn.MultiheadAttention(8, 2, batch_first=True)
    original = F.linear
    k1, k2 = KFAC(attn), KFAC(attn)
    x = torch.randn(3, 5, 8)
    attn(x, x, x, need_weights=False)[0].sum().backward()
    assert F.linear is original
    for k in (k1, k2):
        for layer in k._layers():
            assert k.record[layer][0] is not None and k.record[layer][1] is not None
