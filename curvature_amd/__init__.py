"""curvature_amd: MI355X-native (gfx950) KFAC / EFB / INF curvature hot path, drop-in for the
``Curvature.update() / invert() / sample_and_replace()`` plugin API of DLR-RM/curvature."""
__version__ = "0.1.0"
