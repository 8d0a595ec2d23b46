"""TEST-ONLY: ctypes binding of ag2video_amd/csrc/libag2v_test.so (test_aids.hip) -- kernels that exist for the tests and are not part
of the product library or its header."""
import ctypes
import os

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ag2video_amd', 'csrc', 'libag2v_test.so')
_LIB = [None]


def _lib():
    if _LIB[0] is None:
        import torch  # noqa: F401  (its HIP runtime first, as ag2video_amd._lib.load does)
        if not os.path.exists(_PATH):
            raise RuntimeError("%s not found: make -C ag2video_amd/csrc" % _PATH)
        lib = ctypes.CDLL(_PATH)
        lib.ag2v_test_hold_cus.restype = ctypes.c_int
        lib.ag2v_test_hold_cus.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _LIB[0] = lib
    return _LIB[0]


def hold_cus(n_wgs, ms, stream):
    """Occupy n_wgs CUs (1024 threads + 128 KB of LDS each) for ms <= 500 milliseconds of wall time on torch stream `stream`, the way
    RCCL's kernels or another stream's persistent kernel do; bounded by the clock."""
    rc = _lib().ag2v_test_hold_cus(int(n_wgs), float(ms), ctypes.c_void_p(stream.cuda_stream))
    if rc:
        raise RuntimeError("ag2v_test_hold_cus failed: %d" % rc)
