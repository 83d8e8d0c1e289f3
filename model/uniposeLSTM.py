from unipose_amd.uniposeLSTM import unipose, unipose_lstm  # noqa: F401  (reference: model/uniposeLSTM.py:67)
