"""Diagnostic only: per-segment cycle shares inside sumcheck_round_kernel (libzkhip_diag.so, -DZK_STAMPS)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zk_cryptography_amd import _native as N
N.LIB_PATH = os.path.join(N.CSRC, "libzkhip_diag.so")
import zk_cryptography_amd as zk
n = 1 << 24
t = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda")
poly = zk.Multilinear(t)
for _ in range(3):
    sc = zk.Sumcheck(poly); sc.poly_sum(); sc.prove()
buf = np.zeros((64, 8), dtype=np.uint64)
N.lib().zkhip_debug_read_stamps(N.Context.get().handle, buf.ctypes.data_as(C.c_void_p))
for r in range(14):
    s = buf[r].astype(np.int64)
    print("round %2d  reduce %6d  load_state %6d  transcript %6d cycles" % (r, s[1] - s[0], s[2] - s[1], s[3] - s[2]))
