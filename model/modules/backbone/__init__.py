from unipose_amd.modules import build_backbone  # noqa: F401
