# the reference driver imports this module name, which its tree does not contain (uniposeLSTM.py:26)
from unipose_amd.uniposeLSTM import unipose, unipose_lstm  # noqa: F401
