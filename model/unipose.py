from unipose_amd.unipose import unipose  # noqa: F401  (reference: model/unipose.py:8)
