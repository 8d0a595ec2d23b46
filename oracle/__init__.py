"""ORACLE -- test infrastructure only (see oracle/kernels.py, oracle/model.py headers).

CPU restatement of the reference's AG2Video training step, pinned against golden vectors produced by the unmodified
reference (tests/golden).  Never imported by the product package ag2video_amd.
"""
