#!/usr/bin/env python
"""One convolution of a ResNet-50 at a time through KFAC.update (N = 32), for counter passes that attribute the LDS-DMA
kernel's memory-side fetches to a class of factor (tools/traffic_by_class.sh); `--model` prints the launch plan's operand
bytes for the same layer instead (tools/traffic_model.py; host only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

CLASSES = {   # name: (Cin, Cout, k, stride, pad, H)
    "corr64": (64, 64, 3, 1, 1, 56), "corr128": (128, 128, 3, 1, 1, 28), "corr256": (256, 256, 3, 1, 1, 14),
    "corr512": (512, 512, 3, 1, 1, 7), "unf1152": (128, 128, 3, 2, 1, 56), "unf2304": (256, 256, 3, 2, 1, 28),
    "unf4608": (512, 512, 3, 2, 1, 14), "g256": (64, 256, 1, 1, 0, 56), "a256": (256, 64, 1, 1, 0, 56),
    "a1024": (1024, 256, 1, 1, 0, 14), "a2048": (2048, 512, 1, 1, 0, 7),
}


def build(name):
    ci, co, k, s, p, H = CLASSES[name]
    return torch.nn.Sequential(torch.nn.Conv2d(ci, co, k, s, p, bias=False)), (32, ci, H, H)


def main():
    name = sys.argv[1]
    model, shape = build(name)
    if "--model" in sys.argv:
        import traffic_model
        geoms = traffic_model.geometries(model, 32, shape[1:])
        once, streamed = traffic_model.operand_bytes(geoms, traffic_model.plan(geoms), 32)[:2]
        print("%s: once %.1f MB, streamed by the work items %.1f MB" % (name, once / 1e6, streamed / 1e6))
        return
    from curvature_amd.curvatures import KFAC
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = model.to(dev)
    kfac = KFAC(model)
    x = torch.randn(*shape, device=dev)
    model(x).square().sum().backward()
    for _ in range(4):
        kfac.update(32)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
