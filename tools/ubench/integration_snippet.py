# extracted from INTEGRATION.md section C (kept runnable: tools/ubench/integration_snippet.py)
import ctypes, torch
lib = ctypes.CDLL("diffma-diffusion-mamba_amd/csrc/libdiffma_hip.so")      # after `import torch`

class dm_merge_args(ctypes.Structure):                                      # mirrors include/diffma_hip.h
    _fields_ = [("nin", ctypes.c_int32), ("batch", ctypes.c_int32), ("seqlen", ctypes.c_int32), ("dim", ctypes.c_int32),
                ("io_dtype", ctypes.c_int32), ("out_dtype", ctypes.c_int32),
                ("in", ctypes.c_void_p), ("row_index", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("in_sk", ctypes.c_int64), ("in_sb", ctypes.c_int64), ("in_sl", ctypes.c_int64),
                ("o_sb", ctypes.c_int64), ("o_sl", ctypes.c_int64)]

lib.dm_token_merge.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.dm_last_error.restype = ctypes.c_char_p

ys = torch.randn(3, 4, 196, 1024, device="cuda")        # CrossMerge input, token-major
out = torch.empty(4, 196, 1024, device="cuda")
a = dm_merge_args(3, 4, 196, 1024, 0, 0, ys.data_ptr(), None, out.data_ptr(), *ys.stride()[:3], *out.stride()[:2])
rc = lib.dm_token_merge(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
if rc != 0:
    raise RuntimeError(lib.dm_last_error().decode())

torch.cuda.synchronize()
ref = sum(ys[k] for k in range(3))
print('max abs err', float((out - ref).abs().max()))
