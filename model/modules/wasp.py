from unipose_amd.modules import WASP as wasp, build_wasp  # noqa: F401
