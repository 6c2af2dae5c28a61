"""MI355X-native (gfx950 / CDNA4) implementation of GO-SLAM's two data-parallel hot paths.

Tracking path : correlation pyramid build + lookup, reprojection, dense bundle adjustment
                (the reference's `droid_backends` extension + its Python callers).
Mapping path  : multi-resolution hash-grid NeuS renderer (the reference's tinycudann use).

Every compute entry point goes through the C-ABI shared library built from
`go_slam_amd/csrc/*.hip` (declared in `include/goslam_hip.h`); there is no CPU fallback.
"""
__version__ = "0.1.0"
