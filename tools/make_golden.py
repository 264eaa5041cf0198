#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container: it imports the reference package from /root/reference (read-only,
never copied) with the one shim SURVEY.md section 8(c) documents (torch.symeig was removed from
torch; the reference calls it at curvature/utils.py:57-58).  The outputs are plain data (.npz /
.json): inputs and the reference's results for them.  Nothing from /root/reference travels.

    python tools/make_golden.py            # rewrites tests/golden/*.npz, *.json
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

warnings.filterwarnings("ignore")
# torch.symeig still exists as a stub that raises; replace it unconditionally
torch.symeig = lambda A, eigenvectors=False, upper=True: torch.linalg.eigh(A, UPLO="U" if upper else "L")

# Second torch-compatibility shim (same class as symeig): INF._dim_reduction indexes tensors with Python
# lists of 0-dim tensors (curvature/curvatures.py:643-645).  torch>=1.6 of the reference's era converted
# such a list to one LongTensor index (numpy semantics); torch 2.10 reads it as a multi-dimensional index
# and raises.  Restore the old meaning for exactly that pattern.
_orig_getitem = torch.Tensor.__getitem__


def _is_scalar_tensor_list(idx):
    return isinstance(idx, list) and len(idx) > 0 and all(isinstance(i, torch.Tensor) and i.dim() == 0 for i in idx)


def _compat_getitem(self, idx):
    if _is_scalar_tensor_list(idx):
        idx = torch.stack(idx).long()
    elif isinstance(idx, tuple) and any(_is_scalar_tensor_list(i) for i in idx):
        idx = tuple(torch.stack(i).long() if _is_scalar_tensor_list(i) else i for i in idx)
    return _orig_getitem(self, idx)


torch.Tensor.__getitem__ = _compat_getitem

# Third shim, same cause: torch.tensor(list of 0-dim tensors) (curvatures.py:634-635) raises in torch 2.10.
_orig_tensor = torch.tensor


def _compat_tensor(data, *args, **kwargs):
    if _is_scalar_tensor_list(data):
        data = [d.item() for d in data]
    return _orig_tensor(data, *args, **kwargs)


torch.tensor = _compat_tensor

sys.path.insert(0, REF)
os.chdir(REF)  # lenet5(pretrained=...) resolves its checkpoint relative to cwd (curvature/lenet5.py:27)
from curvature.curvatures import KFAC, EFB, INF, Diagonal, BlockDiagonal  # noqa: E402
from curvature.lenet5 import lenet5  # noqa: E402
from curvature import resnet as ref_resnet  # noqa: E402
from curvature.utils import kron as ref_kron  # noqa: E402

torch.set_num_threads(4)


def npf(t):
    return t.detach().cpu().numpy().copy()   # copy: the reference mutates its state in place (+=)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrays)} arrays")


def fwd_bwd(model, x, seed):
    """scripts/test.py:33-45: labels drawn from the model's own predictive distribution."""
    logits = model(x)
    torch.manual_seed(seed)
    labels = torch.distributions.Categorical(logits=logits).sample()
    loss = torch.nn.functional.cross_entropy(logits, labels)
    model.zero_grad()
    loss.backward()
    return labels


def layers_of(est):
    return [l for l in est.model.modules() if l.__class__.__name__ in ("Linear", "Conv2d")]


def gen_lenet():
    N = 8
    model = lenet5(pretrained="mnist", device="cpu").eval()
    kfac = KFAC(model)
    diag = Diagonal(model)
    arrays = {}
    batches = []
    for b in range(3):
        torch.manual_seed(100 + b)
        x = torch.rand(N, 1, 28, 28)
        labels = fwd_bwd(model, x, 200 + b)
        kfac.update(batch_size=N)
        diag.update(batch_size=N)
        arrays[f"b{b}_x"] = npf(x)
        arrays[f"b{b}_labels"] = npf(labels)
        for li, layer in enumerate(layers_of(kfac)):
            xin, g = kfac.record[layer]
            arrays[f"b{b}_l{li}_x"] = npf(xin)
            arrays[f"b{b}_l{li}_g"] = npf(g / N)          # raw grad_output (N = 8: exact division)
            arrays[f"b{b}_l{li}_gw"] = npf(layer.weight.grad)
            arrays[f"b{b}_l{li}_gb"] = npf(layer.bias.grad)
        if b in (0, 2):
            for li, layer in enumerate(layers_of(kfac)):
                arrays[f"A_after{b + 1}_l{li}"] = npf(kfac.state[layer][0])
                arrays[f"G_after{b + 1}_l{li}"] = npf(kfac.state[layer][1])
                arrays[f"diag_after{b + 1}_l{li}"] = npf(diag.state[layer])
    for li, layer in enumerate(layers_of(kfac)):
        arrays[f"w_l{li}"] = npf(layer.weight)
        arrays[f"bias_l{li}"] = npf(layer.bias)
    save("g1_kfac_lenet.npz", **arrays)

    # ---- G3: invert ----
    inv = {}
    per_layer_add = [0.5, 1.0, 2.0, 0.25, 3.0]
    per_layer_mul = [1.0, 10.0, 100.0, 5.0, 1000.0]
    for tag, (add, mul) in {"a": (0.5, 1), "b": (1.0, 1000.0), "c": (per_layer_add, per_layer_mul)}.items():
        kfac.inv_state = {}
        kfac.invert(add=add, multiply=mul)
        for li, layer in enumerate(layers_of(kfac)):
            inv[f"{tag}_LA_l{li}"] = npf(kfac.inv_state[layer][0])
            inv[f"{tag}_LG_l{li}"] = npf(kfac.inv_state[layer][1])
    # fp64 twin of sets "a", "b" and "c": the reference's own code run on float64 factors
    state32 = kfac.state
    kfac.state = {l: [A.double(), G.double()] for l, (A, G) in state32.items()}
    for tag, (add, mul) in {"a": (0.5, 1), "b": (1.0, 1000.0), "c": (per_layer_add, per_layer_mul)}.items():
        kfac.inv_state = {}
        kfac.invert(add=add, multiply=mul)
        for li, layer in enumerate(layers_of(kfac)):
            inv[f"{tag}64_LA_l{li}"] = npf(kfac.inv_state[layer][0]).astype(np.float32)
            inv[f"{tag}64_LG_l{li}"] = npf(kfac.inv_state[layer][1]).astype(np.float32)
    kfac.state = state32
    inv["c_add"] = np.array(per_layer_add)
    inv["c_mul"] = np.array(per_layer_mul)
    save("g3_kfac_invert.npz", **inv)

    # ---- G4: sample_and_replace with reproducible noise ----
    kfac.inv_state = {}
    kfac.invert(add=0.5, multiply=1)
    smp = {}
    torch.manual_seed(4242)
    for li, layer in enumerate(layers_of(kfac)):
        n, m = kfac.inv_state[layer][0].size(0), kfac.inv_state[layer][1].size(0)
        smp[f"z_l{li}"] = npf(torch.randn(n, m))
    torch.manual_seed(4242)
    for li, layer in enumerate(layers_of(kfac)):
        smp[f"sample_l{li}"] = npf(kfac.sample(layer))
    torch.manual_seed(4242)
    kfac.sample_and_replace()
    for li, layer in enumerate(layers_of(kfac)):
        smp[f"w_new_l{li}"] = npf(layer.weight)
        smp[f"b_new_l{li}"] = npf(layer.bias)
    model.load_state_dict(kfac.model_state)
    save("g4_kfac_sample.npz", **smp)

    # ---- G5 / G6: eigenvectors and EFB ----
    efb = EFB(model, kfac.state)
    e = {}
    for li, layer in enumerate(layers_of(kfac)):
        e[f"UA_l{li}"] = npf(efb.eigvecs[layer][0])
        e[f"UG_l{li}"] = npf(efb.eigvecs[layer][1])
    save("g5_eigvecs_lenet.npz", **e)

    g6 = {}
    for b in range(2):
        torch.manual_seed(100 + b)
        x = torch.rand(N, 1, 28, 28)
        fwd_bwd(model, x, 200 + b)        # same batches/labels as G1 b0, b1 -> same grads
        efb.update(batch_size=N)
    for li, layer in enumerate(layers_of(kfac)):
        g6[f"lambda_l{li}"] = npf(efb.state[layer])
        g6[f"diags_l{li}"] = npf(efb.diags[layer])
    efb.invert(add=0.5, multiply=2.0)
    for li, layer in enumerate(layers_of(kfac)):
        g6[f"inv_l{li}"] = npf(efb.inv_state[layer])
    torch.manual_seed(77)
    for li, layer in enumerate(layers_of(kfac)):
        n, m = efb.eigvecs[layer][0].size(0), efb.eigvecs[layer][1].size(0)
        g6[f"z_l{li}"] = npf(torch.randn(n, m))
    torch.manual_seed(77)
    for li, layer in enumerate(layers_of(kfac)):
        g6[f"sample_l{li}"] = npf(efb.sample(layer))
    save("g6_efb_lenet.npz", **g6)

    # ---- G7 / G8 / G9: INF ----
    g7 = {}
    for rank in (10, 100, 10 ** 9):
        inf = INF(model, efb.diags, kfac.state, efb.state)
        inf.eigvecs = efb.eigvecs     # same eigenvectors as G5 (the ctor recomputes identical ones)
        inf.update(rank=rank)
        tag = {10: "r10", 100: "r100", 10 ** 9: "rall"}[rank]
        for li, layer in enumerate(layers_of(kfac)):
            ua, ug, lam, D = inf.state[layer]
            UA, UG = efb.eigvecs[layer]
            m = UG.shape[1]
            if rank < lam.numel() or ua.shape[1] < UA.shape[1] or ug.shape[1] < UG.shape[1]:
                I = np.array([int(np.flatnonzero((npf(UA) == npf(ua)[:, [c]]).all(0))[0]) for c in range(ua.shape[1])])
                J = np.array([int(np.flatnonzero((npf(UG) == npf(ug)[:, [c]]).all(0))[0]) for c in range(ug.shape[1])])
            else:
                I, J = np.arange(UA.shape[1]), np.arange(UG.shape[1])
            g7[f"{tag}_I_l{li}"] = I.astype(np.int64)
            g7[f"{tag}_J_l{li}"] = J.astype(np.int64)
            if tag != "rall":
                g7[f"{tag}_lam_l{li}"] = npf(lam)
                g7[f"{tag}_D_l{li}"] = npf(D)
            elif li in (0, 4):
                g7[f"{tag}_D_l{li}"] = npf(D)
        if rank == 10:
            inf10 = inf
    save("g7_inf_update.npz", **g7)

    g8 = {}
    add, mul = 10.0, 50.0
    # fp64 twin first (invert clamps `state` in place, curvatures.py:523)
    st32 = inf10.state
    inf10.state = {l: tuple(t.double() for t in v) for l, v in st32.items()}
    inf10.invert(add=add, multiply=mul)
    for li, layer in enumerate(layers_of(kfac)):
        g8[f"Pc64_l{li}"] = npf(inf10.inv_state[layer][3]).astype(np.float32)
    inv64 = dict(inf10.inv_state)
    inf10.state = {l: tuple(t.clone() for t in v) for l, v in st32.items()}
    inf10.inv_state = {}
    inf10.invert(add=add, multiply=mul)
    for li, layer in enumerate(layers_of(kfac)):
        ua, ug, r, Pc = inf10.inv_state[layer]
        lam = inf10.state[layer][2]
        sigma = (mul * lam).sqrt()
        V_s = r.contiguous().view(-1, 1) * ref_kron(ua, ug) @ torch.diag(sigma)
        vtv = V_s.t() @ V_s
        g8[f"r_l{li}"] = npf(r)
        g8[f"sigma_l{li}"] = npf(sigma)
        g8[f"vtv_l{li}"] = npf((vtv + vtv.t()) / 2.0)
        g8[f"Pc_l{li}"] = npf(Pc)
        g8[f"Dclamped_l{li}"] = npf(inf10.state[layer][3])
    g8["add"] = np.array(add)
    g8["mul"] = np.array(mul)
    save("g8_inf_invert.npz", **g8)

    g9 = {}
    torch.manual_seed(99)
    for li, layer in enumerate(layers_of(kfac)):
        ua, ug = inf10.inv_state[layer][0], inf10.inv_state[layer][1]
        g9[f"X_l{li}"] = npf(torch.randn(ua.shape[0] * ug.shape[0]))
    torch.manual_seed(99)
    for li, layer in enumerate(layers_of(kfac)):
        g9[f"sample_l{li}"] = npf(inf10.sample(layer))
    # fp64 twin: the reference's own sampler on its float64 inverse state with the SAME noise.  The sampler
    # draws X itself (torch.randn(..., dtype=eigvecs.dtype), curvatures.py:590), and a float64 draw is a
    # different stream, so torch.randn is pinned to the recorded X for the duration of each call.
    inv32 = inf10.inv_state
    inf10.inv_state = inv64
    real_randn = torch.randn
    try:
        for li, layer in enumerate(layers_of(kfac)):
            X64 = torch.from_numpy(g9[f"X_l{li}"]).double()
            torch.randn = lambda *a, _x=X64, **k: _x.clone()
            g9[f"sample64_l{li}"] = npf(inf10.sample(layer)).astype(np.float32)
    finally:
        torch.randn = real_randn
        inf10.inv_state = inv32
    save("g9_inf_sample.npz", **g9)


def gen_conv_shapes():
    """G2: hand-picked ResNet layer shapes at small spatial size, through the reference's KFAC."""
    torch.manual_seed(5)
    model = torch.nn.Sequential(
        torch.nn.Conv2d(3, 8, 7, 2, 3, bias=False), torch.nn.ReLU(),
        torch.nn.Conv2d(8, 16, 3, 1, 1, bias=False), torch.nn.ReLU(),
        torch.nn.Conv2d(16, 16, 3, 2, 1, bias=False), torch.nn.ReLU(),
        torch.nn.Conv2d(16, 32, 1, 2, 0, bias=False), torch.nn.ReLU(),
        torch.nn.AdaptiveAvgPool2d((1, 1)), torch.nn.Flatten(),
        torch.nn.Linear(32, 10)).eval()
    kfac = KFAC(model)
    N = 2
    x = torch.randn(N, 3, 30, 30)
    fwd_bwd(model, x, 6)
    kfac.update(batch_size=N)
    arrays = {"x": npf(x)}
    for li, layer in enumerate(layers_of(kfac)):
        xin, g = kfac.record[layer]
        arrays[f"l{li}_x"] = npf(xin)
        arrays[f"l{li}_g"] = npf(g / N)
        arrays[f"l{li}_A"] = npf(kfac.state[layer][0])
        arrays[f"l{li}_G"] = npf(kfac.state[layer][1])
        if layer.__class__.__name__ == "Conv2d":
            arrays[f"l{li}_geom"] = np.array(list(layer.kernel_size) + list(layer.stride) + list(layer.padding))
        arrays[f"l{li}_bias"] = np.array(int(layer.bias is not None))
    save("g2_kfac_convshapes.npz", **arrays)


def imagenet_resnet(block, layers):
    """the reference's ResNet with the ImageNet stem reproduces torchvision's layer list."""
    model = ref_resnet.ResNet(block=block, layers=layers, num_classes=1000)
    model.conv1 = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    return model


def gen_layer_tables():
    """G11: layer order / shapes (bit-exact indexing contract)."""
    tables = {}
    specs = {
        "lenet5": (lenet5(), (1, 1, 28, 28)),
        "resnet18": (imagenet_resnet(ref_resnet.BasicBlock, [2, 2, 2, 2]), (1, 3, 224, 224)),
        "resnet50": (imagenet_resnet(ref_resnet.Bottleneck, [3, 4, 6, 3]), (1, 3, 224, 224)),
    }
    for name, (model, shape) in specs.items():
        model.eval()
        kfac = KFAC(model)
        with torch.no_grad():
            pass
        out = model(torch.zeros(*shape))
        out.sum().backward()
        kfac.update(batch_size=1)
        names = {m: n for n, m in model.named_modules()}
        rows = []
        for idx, (layer, (A, G)) in enumerate(kfac.state.items()):
            xin, g = kfac.record[layer]
            L = int(g.numel() // (g.shape[0] * g.shape[1]))
            rows.append({"index": idx, "name": names[layer], "kind": layer.__class__.__name__,
                         "n": int(A.shape[0]), "m": int(G.shape[0]), "L": L,
                         "has_bias": layer.bias is not None})
        tables[name] = rows
        print(name, len(rows), "layers")
    with open(os.path.join(OUT, "g11_layer_tables.json"), "w") as fh:
        json.dump(tables, fh, indent=0)


def gen_mc_fisher():
    """G12: the MC-Fisher outer loop of scripts/factors.py:47-61 (one forward, `samples` label draws from
    Categorical(logits), backward(retain_graph=True) + update per draw), driven with the reference's KFAC
    class.  The script itself imports torchvision (absent here), so its loop is restated; labels are fixed
    by seeding right before each draw and recorded."""
    N, samples = 8, 3
    model = lenet5(pretrained="mnist", device="cpu").train()
    criterion = torch.nn.CrossEntropyLoss()
    est = KFAC(model)
    arrays = {"samples": np.array(samples)}
    for b in range(2):
        torch.manual_seed(400 + b)
        images = torch.rand(N, 1, 28, 28)
        arrays[f"b{b}_x"] = npf(images)
        logits = model(images)
        dist = torch.distributions.Categorical(logits=logits)
        for smp in range(samples):
            torch.manual_seed(500 + 10 * b + smp)
            labels = dist.sample()
            arrays[f"b{b}_s{smp}_labels"] = npf(labels)
            loss = criterion(logits, labels)
            model.zero_grad()
            loss.backward(retain_graph=True)
            est.update(images.size(0))
    for li, layer in enumerate(layers_of(est)):
        arrays[f"A_l{li}"] = npf(est.state[layer][0])
        arrays[f"G_l{li}"] = npf(est.state[layer][1])
    save("g12_mc_fisher_lenet.npz", **arrays)


def gen_kron():
    """G10: the reference's only known-answer test (curvature/utils.py:301-309)."""
    a = torch.tensor([[1, 2], [3, 4]])
    b = torch.tensor([[0, 5], [6, 7]])
    torch.manual_seed(1)
    c, d = torch.randn(3, 2), torch.randn(4, 5)
    save("g10_kron.npz", a=npf(a), b=npf(b), ab=npf(ref_kron(a, b)), c=npf(c), d=npf(d), cd=npf(ref_kron(c, d)))


def gen_block_diagonal():
    """G13: BlockDiagonal (curvature/curvatures.py:196-261) on a small conv + linear net: per-layer P x P state after
    one and two batches, the inverse factors for a scalar and a per-layer hyper-parameter pair, and samples of the
    Linear layers (the reference's `sample` cannot handle Conv2d: it concatenates a 4-D view with a 2-D column)."""
    N = 4
    torch.manual_seed(13)
    model = torch.nn.Sequential(torch.nn.Conv2d(1, 2, 3), torch.nn.ReLU(), torch.nn.Flatten(),
                                torch.nn.Linear(2 * 4 * 4, 6), torch.nn.ReLU(), torch.nn.Linear(6, 4)).eval()
    est = BlockDiagonal(model)
    g = {}
    for li, layer in enumerate(layers_of(est)):
        g[f"w_l{li}"] = npf(layer.weight)
        g[f"bias_l{li}"] = npf(layer.bias)
    for b in range(2):
        torch.manual_seed(600 + b)
        x = torch.rand(N, 1, 6, 6)
        labels = fwd_bwd(model, x, 700 + b)
        est.update(batch_size=N)
        g[f"b{b}_x"] = npf(x)
        g[f"b{b}_labels"] = npf(labels)
        for li, layer in enumerate(layers_of(est)):
            g[f"b{b}_l{li}_gw"] = npf(layer.weight.grad)
            g[f"b{b}_l{li}_gb"] = npf(layer.bias.grad)
            g[f"state_after{b + 1}_l{li}"] = npf(est.state[layer])
    per_layer_add = [0.5, 1.0, 2.0]
    per_layer_mul = [1.0, 10.0, 100.0]
    for tag, (add, mul) in {"a": (0.5, 2.0), "b": (per_layer_add, per_layer_mul)}.items():
        est.inv_state = {}
        est.invert(add=add, multiply=mul)
        for li, layer in enumerate(layers_of(est)):
            g[f"{tag}_inv_l{li}"] = npf(est.inv_state[layer])
    g["b_add"], g["b_mul"] = np.array(per_layer_add), np.array(per_layer_mul)
    est.inv_state = {}
    est.invert(add=0.5, multiply=2.0)
    torch.manual_seed(1313)
    for li, layer in enumerate(layers_of(est)):
        g[f"z_l{li}"] = npf(torch.randn(est.inv_state[layer].shape[0]))       # the stream `.normal_()` draws from
    torch.manual_seed(1313)
    for li, layer in enumerate(layers_of(est)):
        if isinstance(layer, torch.nn.Linear):
            g[f"sample_l{li}"] = npf(est.sample(layer))
        else:
            est.inv_state[layer].new(est.inv_state[layer].shape[0]).normal_()   # keep the stream aligned with z_l*
    save("g13_block_diagonal.npz", **g)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]                      # e.g. `python tools/make_golden.py mc_fisher` regenerates one set
    gens = {"kron": gen_kron, "conv_shapes": gen_conv_shapes, "lenet": gen_lenet, "layer_tables": gen_layer_tables,
            "mc_fisher": gen_mc_fisher, "block_diagonal": gen_block_diagonal}
    for name, fn in gens.items():
        if not only or name in only:
            fn()
