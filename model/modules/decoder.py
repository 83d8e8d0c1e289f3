from unipose_amd.modules import Decoder, build_decoder  # noqa: F401
