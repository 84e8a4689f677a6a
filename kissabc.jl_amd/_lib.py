"""Loader of the C-ABI library (kissabc.jl_amd/lib/libkabc_hip.so).

There is NO fallback: if the HIP extension is missing or no gfx950 device is
visible, the product path raises.  (The CPU oracle lives under oracle/ and is
only ever used by tests/, smoke() and bench.py's cpu_baseline leg.)
"""
import ctypes as C
import os
import threading

from . import _cdefs

_HERE = os.path.dirname(os.path.abspath(__file__))
# KABC_PROBES=1 selects the build with the timing probes of tools/ compiled into the AIS
# kernel (make -C kissabc.jl_amd/csrc PROBES=1; built on first use)
PROBES = os.environ.get("KABC_PROBES") == "1"
LIB_PATH = os.path.join(_HERE, "lib", "libkabc_hip_probes.so" if PROBES else "libkabc_hip.so")
# KABC_LIB: an experimental build of the library (make VARIANT=...), for A/B runs on one box
LIB_PATH = os.environ.get("KABC_LIB") or LIB_PATH
_lib = None


class KabcError(RuntimeError):
    """Raised for every non-zero kabc_status_t; .status holds the code and the
    message is the library's (for reference-defined errors: the reference's text)."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def _preload_torch_hip():
    # KABC_NO_TORCH_PRELOAD=1: hosts that never import torch (bench.py, the RCCL tests) keep
    # the process on the system ROCm runtime alone
    if os.environ.get("KABC_NO_TORCH_PRELOAD") == "1":
        return
    _preload_torch_hip_impl()


def _preload_torch_hip_impl():
    """PyTorch wheels bundle their own ROCm runtime (libamdhip64 & co).  If this
    library pulls in the system libamdhip64 first, a later `import torch` binds to a
    mixed set of ROCm libraries and reports "No HIP GPUs are available".  Loading
    torch first makes both use torch's copy (same soname).  Only done when torch is
    installed; the C ABI itself never needs torch."""
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def load():
    global _lib
    if _lib is None:
        _preload_torch_hip()
        if PROBES and not os.path.exists(LIB_PATH):
            import subprocess
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "PROBES=1",
                            f"-j{min(32, os.cpu_count() or 8)}"], check=True)
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C kissabc.jl_amd/csrc`). The KissABC MI355X path has no "
                "CPU fallback.")
        _lib = _cdefs.bind(C.CDLL(LIB_PATH))
    return _lib


def check(status):
    if status != 0:
        msg = load().kabc_last_error()
        raise KabcError(status, msg.decode("utf-8", "replace") if msg else f"kabc status {status}")


class Context:
    """kabc_ctx_t: one GPU + one HIP stream."""

    def __init__(self, device=0, stream=None, _borrowed=None):
        lib = load()
        self._owned = _borrowed is None
        if _borrowed is not None:   # a context owned by a communicator (kabc_comm_init_all)
            self._h = C.c_void_p(_borrowed)
        else:
            self._h = C.c_void_p()
            check(lib.kabc_ctx_create(int(device), C.c_void_p(stream) if stream else None,
                                      C.byref(self._h)))
        self.device = device

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(load().kabc_ctx_synchronize(self._h))

    def close(self):
        if self._h and self._owned:
            release_pinned_cache()   # (page-locked result blocks kept for reuse go with the context)
            load().kabc_ctx_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


# Page-locked blocks whose arrays have died, kept for the next result of the same size: pinning
# 268 MB (an smc result at 2 M particles x 16) costs as much as copying into it ten times.
_pinned_free = {}            # nbytes -> [address, ...]
_pinned_free_bytes = 0
_pinned_lock = threading.RLock()   # (re-entrant: a GC pass inside the locked region may finalise another block)


def _pinned_cache_cap():
    """KABC_PINNED_CACHE_MB (default 1024): page-locked bytes kept on the free list"""
    return int(float(os.environ.get("KABC_PINNED_CACHE_MB", "1024")) * (1 << 20))


def release_pinned_cache():
    """Hand every cached page-locked block back to the driver (also done when a Context closes)."""
    global _pinned_free_bytes
    with _pinned_lock:
        blocks = [p for lst in _pinned_free.values() for p in lst]
        _pinned_free.clear()
        _pinned_free_bytes = 0
    if _lib is not None:
        for p in blocks:
            _lib.kabc_host_free(C.c_void_p(p))


class _PinnedBlock:
    """One kabc_host_alloc allocation; when the last numpy view dies it goes back to the free list
    (or to the driver, beyond KABC_PINNED_CACHE_MB)."""

    def __init__(self, nbytes):
        global _pinned_free_bytes
        with _pinned_lock:
            lst = _pinned_free.get(nbytes)
            if lst:
                self.ptr, self.nbytes = lst.pop(), nbytes
                _pinned_free_bytes -= nbytes
                return
        p = C.c_void_p()
        check(load().kabc_host_alloc(C.c_size_t(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes

    def __del__(self):
        global _pinned_free_bytes
        try:
            if self.ptr and _lib is not None:
                with _pinned_lock:
                    keep = _pinned_free_bytes + self.nbytes <= _pinned_cache_cap()
                    if keep:
                        _pinned_free.setdefault(self.nbytes, []).append(self.ptr)
                        _pinned_free_bytes += self.nbytes
                if not keep:
                    _lib.kabc_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass
        self.ptr = None


def pinned_empty(shape, dtype="float64"):
    """numpy array over page-locked host memory (kabc_host_alloc): the destination of
    the sample trace, so that kabc_ais_advance can DMA it while the kernels run, and of large
    smc / ABCDE results.  Falls back to ordinary memory when pinning fails or KABC_PINNED_TRACE=0."""
    import numpy as np
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    if n == 0 or os.environ.get("KABC_PINNED_TRACE", "1") == "0":
        return np.empty(shape, dtype=dt)
    try:
        blk = _PinnedBlock(n)
    except KabcError:
        return np.empty(shape, dtype=dt)
    buf = (C.c_char * n).from_address(blk.ptr)
    buf._kabc_block = blk
    return np.frombuffer(buf, dtype=dt).reshape(shape)


def result_empty(shape, dtype="float64"):
    """destination of a result copy: page-locked from 1 MiB on (the copy then runs at the PCIe
    rate instead of faulting fresh pages in one by one), ordinary memory below"""
    import numpy as np
    dt = np.dtype(dtype)
    if int(np.prod(shape)) * dt.itemsize >= (1 << 20):
        return pinned_empty(shape, dt)
    return np.empty(shape, dtype=dt)


MATH_FN = {"log": 0, "exp": 1, "log1p": 2, "lgamma": 3, "sincos2pi": 4, "sqrt": 5, "rint": 6,
           "log_pn": 7, "sqrt_pn": 8, "u01": 9, "normal_pair": 10, "index32": 11,
           "exp_bounded": 12}


def math_probe(name, x, ctx=None):
    """kabc_math_probe: one include/kabc_math.h function evaluated on the device
    (verification only; 64-bit words travel as the bit patterns of doubles)."""
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size // (2 if name in ("normal_pair", "index32") else 1)
    ow = 2 if name in ("sincos2pi", "normal_pair") else 1
    out = np.empty(n * ow)
    ctx = ctx or default_context()
    check(load().kabc_math_probe(ctx.handle, MATH_FN[name], n,
                                 x.ctypes.data_as(_cdefs.c_double_p),
                                 out.ctypes.data_as(_cdefs.c_double_p)))
    return out.reshape(-1, 2) if ow == 2 else out
