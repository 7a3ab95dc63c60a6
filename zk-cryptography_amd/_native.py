"""ctypes loader for csrc/libzkhip.so (the C ABI of include/zkhip.h) + context handling.

PyTorch supplies device memory (int64 CUDA tensors viewed as uint64 limbs), the HIP
stream and torch.distributed; it is plumbing only -- all arithmetic happens in libzkhip.
"""
import ctypes as C
import os
import subprocess
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libzkhip.so")

ZKHIP_OK, ERR_HIP, ERR_SHAPE, ERR_INDEX, ERR_ARG, ERR_NOMEM, ERR_BUSY, ERR_PEER, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7, -8


class ZkhipError(RuntimeError):
    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


class ZkhipPeerError(ZkhipError):
    """a sharded prover: ANOTHER rank failed and said so through the exchange (ZKHIP_ERR_PEER); this rank's outputs are void"""


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "zkhip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"] + (["-B"] if force else []) + ["libzkhip.so"])
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZkhipError("libzkhip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)")
        # torch wheels bundle their own libamdhip64.so.7; import torch FIRST so that libzkhip binds to the
        # same HIP runtime instance (we exchange device pointers and streams with torch).
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        _lib.zkhip_status_string.restype = C.c_char_p
        _lib.zkhip_sumcheck_plan_log_blocks.argtypes = [C.c_size_t]
        # the prover's hot calls take plain integers for their pointers (no ctypes object per argument: host time between a
        # proof and the next call's first launch is GPU idle time)
        vp = C.c_void_p
        _lib.zkhip_mle_block_sums_deferred.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp]
        _lib.zkhip_mle_block_sums.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp, vp]
        _lib.zkhip_sumcheck_prove.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, C.c_uint32, vp, vp, vp]
        _lib.zkhip_sumcheck_prove_begin.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, C.c_uint32, vp]
        _lib.zkhip_sumcheck_prove_end.argtypes = [vp, C.c_uint32, vp, vp, vp]
    return _lib


def check(status, what=""):
    if status == ZKHIP_OK:
        return
    msg = lib().zkhip_status_string(status).decode()
    if status == ERR_HIP:
        msg += " (hipError %d)" % lib().zkhip_last_hip_error(None)
    if status == ERR_SHAPE:
        raise AssertionError("%s: %s" % (what, msg))      # the reference panics (assert!/assert_eq!)
    if status == ERR_INDEX:
        raise IndexError("%s: %s" % (what, msg))
    raise (ZkhipPeerError if status == ERR_PEER else ZkhipError)("%s: %s (status %d)" % (what, msg, status), status)


class Context:
    """One libzkhip context per (host thread, device), enqueueing on torch's current stream (include/zkhip.h: "one context per
    host thread; contexts are independent" -- the mirror classes reach theirs through Context.get, so two Python threads that prove
    at once never share one).  The per-thread table is thread-LOCAL storage: it goes away with its thread (a thread identifier can
    be reused by a later thread; a context must not be), and a context nobody refers to any more is destroyed."""

    _tls = threading.local()

    def __init__(self, device_index):
        import torch
        if not torch.cuda.is_available():
            raise ZkhipError("no GPU visible: zk_cryptography_amd has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.handle = C.c_void_p()
        self._comms = []                 # distributed.Comm objects created on this context: closed before the context goes
        self._circuits = []              # gkr._DeviceCircuit objects (zkhip_circuit handles refer to their context): destroyed first, too
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check(lib().zkhip_ctx_create(C.byref(self.handle), C.c_int(device_index), C.c_void_p(stream)), "ctx_create")
        self._stream = stream

    @classmethod
    def get(cls, device_index=None):
        import torch
        if device_index is None:
            device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
        table = cls._tls.__dict__.setdefault("contexts", {})
        ctx = table.get(device_index)
        if ctx is None or not ctx.handle:
            ctx = table[device_index] = Context(device_index)
        ctx.sync_stream()
        return ctx

    def destroy(self):
        """zkhip_ctx_destroy: closes the communicators and device circuits created on this context (both refer to it), waits for the
        context's streams and returns every buffer it holds"""
        if self.handle:
            for m in list(self._comms):
                m.close()
            for d in list(self._circuits):
                d.close()
            h, self.handle = self.handle, None
            table = Context._tls.__dict__.get("contexts", {})
            for k, v in list(table.items()):
                if v is self:
                    del table[k]
            check(lib().zkhip_ctx_destroy(h), "ctx_destroy")

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter shutdown: the HIP runtime may already be gone; the process ends anyway
            return
        try:
            self.destroy()
        except Exception:
            pass

    def sync_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            check(lib().zkhip_ctx_set_stream(self.handle, C.c_void_p(s)), "set_stream")
            self._stream = s

    def synchronize(self):
        check(lib().zkhip_ctx_synchronize(self.handle), "synchronize")


def ptr(t):
    """device pointer of a contiguous torch tensor"""
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())
