from unipose_amd.modules import WASP as wasp  # noqa: F401


def build_wasp(backbone, output_stride, BatchNorm):
    return wasp(backbone, output_stride, BatchNorm, video=True)
