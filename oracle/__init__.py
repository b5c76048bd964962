"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the reference hot path of fatchord/WaveRNN
(`models/fatchord_version.py:169-264`).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import anything from this package; the product path
(`wavernn_amd/`) never does.
"""
