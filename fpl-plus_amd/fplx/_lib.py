"""ctypes binding of libfplx.so (C ABI declared in include/fplx.h).

The prototypes are parsed from the header itself, so the Python side can never drift from
the C side, and `declared_symbols()` lets the tests check that the library exports exactly
what the header declares.  There is NO fallback: if the library is missing or a call fails
the product path raises (RuntimeError / ValueError, mirroring the reference's exception
style: ValueError for bad names / shapes, e.g. PyMIC/pymic/net_run_dsbn/dsbn.py:61-64).
"""
import ctypes
import os
import re

# Two switches of the HIP runtime, both read ONCE when the runtime initialises (the first GPU call of the process); an explicit
# setting of the user always wins:
#  GPU_MAX_HW_QUEUES=8      HIP maps streams onto hardware queues (default 4) round-robin.  fplx runs backward on two streams
#                           (data gradients | weight gradients); once a process group exists RCCL adds streams of its own, and
#                           with 4 queues the two fplx streams end up sharing one - serialised, the whole overlap (about 1 ms of
#                           the train step) is lost (profiles/r02_rccl_single_rank.txt).
#  HIP_FORCE_DEV_KERNARG=1  kernel arguments in DEVICE memory: by default the kernarg segment lives in host memory and every
#                           dispatch reads it across the host link before its first wave starts - about 1.1 us per launch.  The
#                           train step is 211 dependent launches, a quarter of them shorter than 10 us: 8.62 -> 8.38 ms (-2.8 %,
#                           three alternating pairs of processes on one box, profiles/r05_kernel_ab.txt section 19).
# Set here they only help if fplx is imported BEFORE the first GPU call.  A host program that initialised the GPU first (PyMIC
# imports torch, builds its device, and reaches fplx through the registry later - INTEGRATION.md section 1) would lose them
# SILENTLY: that case raises a RuntimeWarning below.
_RUNTIME_ENV = (("GPU_MAX_HW_QUEUES", "8"), ("HIP_FORCE_DEV_KERNARG", "1"))
_env_preset = {k: os.environ.get(k) for k, _ in _RUNTIME_ENV}      # what the user / launcher had set before this import
for _k, _v in _RUNTIME_ENV:
    os.environ.setdefault(_k, _v)

# torch first: libfplx.so must bind to the SAME HIP runtime (libamdhip64.so.7) the process uses
# for its device memory and streams.  PyTorch-ROCm ships its own copy; whichever copy is loaded
# first serves both, and the ROCm-7.2 system copy does not see the devices torch opened.
import torch  # noqa: F401,E402


def runtime_env_report():
    """-> {variable: (value now in os.environ, 'preset' | 'fplx' | 'too late')}: 'too late' = the GPU was already initialised
    when fplx was imported and the variable was not in the environment, so the HIP runtime never saw it"""
    return dict(_env_report)


_gpu_was_up = bool(torch.cuda.is_initialized())
_env_report = {}
for _k, _v in _RUNTIME_ENV:
    _env_report[_k] = (os.environ.get(_k), "preset" if _env_preset[_k] is not None else ("too late" if _gpu_was_up else "fplx"))
_late = [k for k, (_, how) in _env_report.items() if how == "too late"]
if _late:
    import warnings
    warnings.warn(
        "fplx was imported after the GPU had been initialised and %s %s not in the environment: the HIP runtime reads %s only "
        "at its start, so this process runs without %s (about 3-4 %% of the train step: launch latency and stream overlap; "
        "results are unaffected).  Import fplx before the first GPU call, or export %s." % (
            " / ".join(_late), "was" if len(_late) == 1 else "were", "it" if len(_late) == 1 else "them",
            "it" if len(_late) == 1 else "them", " ".join("%s=%s" % (k, dict(_RUNTIME_ENV)[k]) for k in _late)),
        RuntimeWarning, stacklevel=2)

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
HEADER = os.path.join(_ROOT, "include", "fplx.h")
LIB_PATH = os.path.join(_HERE, "libfplx.so")

F32, BF16 = 0, 1

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
    "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32, "fplx_stream_t": ctypes.c_void_p,
}


def _ctype(decl):
    decl = decl.replace("const", " ").strip()
    if "*" in decl:
        return ctypes.c_void_p
    t = decl.split()[0]
    return _CT[t]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(fplx_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                argtypes.append(_ctype(a))
        out[name] = (_CT[ret], argtypes)
    return out


def declared_symbols():
    return sorted(parse_header().keys())


class FplxError(RuntimeError):
    pass


_lib = None
_protos = None


def lib():
    """Load libfplx.so (once).  Raises RuntimeError if it has not been built."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "fplx: %s not found - the HIP extension is not built (run __graft_entry__.build() or "
            "`make -C fpl-plus_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    _lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, argtypes) in _protos.items():
        fn = getattr(_lib, name)
        fn.restype = ret
        fn.argtypes = argtypes
    _apply_env_tuning()
    return _lib


def tuning_keys():
    """names of the library's tuning knobs (fplx_tuning_key)"""
    keys, buf, i = [], ctypes.create_string_buffer(64), 0
    while lib().fplx_tuning_key(i, buf, 64) >= 0:
        keys.append(buf.value.decode())
        i += 1
    return keys


tuning_epoch = 0            # bumped by every set_tuning: host-side caches built under the old knob values (Engine's packs, folds and
                            # split plans) are dropped when they see a new epoch


def set_tuning(key, value):
    """fplx_set_tuning: A/B knob of the kernel dispatchers (benchmarks and tests only; every knob selects among kernels computing the same function - some in another order of fp32 additions).
    The knob table is process-wide and read by the size queries and the launches independently: flip knobs between steps,
    never between a query (fplx_*_stats_rows, fplx_*_ws_bytes, plan queries) and the launch it sizes."""
    global tuning_epoch
    check(lib().fplx_set_tuning(key.encode(), int(value)))
    tuning_epoch += 1


def get_tuning(key):
    v = ctypes.c_int64()
    check(lib().fplx_get_tuning(key.encode(), ctypes.byref(v)))
    return v.value


def _apply_env_tuning():
    """The library itself never reads the environment; the benchmark tools' FPLX_<KEY>=<int> variables (FPLX_BRICK=0,
    FPLX_MARCH32_V2=1, ...) are translated here, once, when the library is loaded.  A bad value leaves NO half-configured
    library behind: the handle is dropped before the error is raised."""
    global _lib
    buf, i = ctypes.create_string_buffer(64), 0
    try:
        while _lib.fplx_tuning_key(i, buf, 64) >= 0:
            key = buf.value.decode()
            env = os.environ.get("FPLX_" + key.upper())
            if env is not None:
                try:
                    val = int(env)
                except ValueError:
                    raise ValueError("fplx: FPLX_%s=%r is not an integer" % (key.upper(), env))
                rc = _lib.fplx_set_tuning(key.encode(), val)
                if rc != 0:
                    raise FplxError("fplx (%d): fplx_set_tuning(%s, %d) failed" % (rc, key, val))
            i += 1
    except Exception:
        _lib = None
        raise


# ---- host-side A/B switches (benchmarks and bisection tools only).  The ONE place product Python translates FPLX_<KEY>
# environment variables: Engine / TrainStep read their defaults from here, tools flip the attributes they set.
_HOST_KNOBS = {
    "pack_reuse": 1,        # Engine.allow_pack_reuse: the second domain's forward of an iteration reuses the first one's packs
    "side_stream": 1,       # Engine.use_side_stream: weight gradients on a second HIP stream (0: one stream, clean profiles)
    "split_cat": 1,         # Engine.use_split_cat: level-0 skip || up as two tensors
    "fused_pool": 1,        # Engine.use_fused_pool: DownBlock tails in one pass each way
    "eval_fuse": 1,         # Engine.use_eval_fusion: eval-mode BatchNorm folded into the packs
    "stem_wg_main": 1,      # Engine.stem_wgrad_on_main
    "outconv_fuse": 1,      # Engine.use_outconv_fusion: out_conv fused with the last site's BatchNorm + PReLU passes
    "stem_wgrad_bn": 1,     # Engine.use_stem_wgrad_bn: the stem's weight gradient forms dy from y and d(a) itself (no apply pass)
    "outconv_wgrad_bn": 1,  # Engine.use_outconv_wgrad_bn: out_conv's weight gradient from y (fplx_outconv_wgrad_bn); the last site's activation is never stored
    "bucket_elems": 1 << 21,    # TrainStep: gradient all-reduce bucket size (elements)
    "pack_small_multi": 1,  # Engine._pack: the transposed-convolution and out_conv packs of a step in one launch (fplx_pack_weights_multi)
    "adam_pack": 1,         # FusedAdam.step_flat: the shared segment's Adam and the 3x3x3 weight packs in one launch (fplx_adam_pack_step)
    "ddp_overlap_all": 1,   # TrainStep.step_all under data parallelism: buckets folded + all-reduced during the last domain's backward
}
_host_vals = {}


def host_knob(key):
    """value of a host-side A/B switch: FPLX_<KEY> from the environment (read once), else the shipped default"""
    if key not in _HOST_KNOBS:
        raise ValueError("fplx: unknown host knob %r" % (key,))
    if key not in _host_vals:
        env = os.environ.get("FPLX_" + key.upper())
        try:
            _host_vals[key] = _HOST_KNOBS[key] if env is None else int(env)
        except ValueError:
            raise ValueError("fplx: FPLX_%s=%r is not an integer" % (key.upper(), env))
    return _host_vals[key]


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().fplx_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc):
    if rc == 0:
        return
    msg = last_error()
    if rc in (-1, -2, -5):
        raise ValueError("fplx: " + msg)
    raise FplxError("fplx (%d): %s" % (rc, msg))


def call(name, *args):
    check(getattr(lib(), name)(*args))
