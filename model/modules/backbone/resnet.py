from unipose_amd.modules import Bottleneck, ResNet  # noqa: F401


def ResNet101(output_stride, BatchNorm, pretrained=False):
    return ResNet((3, 4, 23, 3), output_stride, BatchNorm)
