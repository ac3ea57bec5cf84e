"""ctypes binding of the linearizability checker (include/lincheck.h).

Mirrors the reference's porcupine entry points (src/porcupine/porcupine.go:5-39) for the key/value
model of src/models/kv.go: `check_operations(history)`, `check_operations_timeout(history, seconds)`,
`check_events(history)`.  Results are the strings "Ok", "Illegal", "Unknown" (src/porcupine/model.go:71-77).
"""
import collections
import ctypes as C
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "liblincheck.so")
GET, PUT, APPEND = 0, 1, 2
OK, ILLEGAL, UNKNOWN = "Ok", "Illegal", "Unknown"
_RESULT = {0: ILLEGAL, 1: OK, 2: UNKNOWN}

# porcupine.Operation with KvInput / KvOutput flattened (src/porcupine/model.go:5-11, src/models/kv.go:7-15)
Operation = collections.namedtuple("Operation", "client_id op key value output call ret")
# porcupine.Event (src/porcupine/model.go:20-25); a return carries its output in `value`
Event = collections.namedtuple("Event", "client_id is_return id op key value")


class _Op(C.Structure):
    _fields_ = [("client_id", C.c_int32), ("op", C.c_int32), ("key", C.c_char_p), ("value", C.c_char_p),
                ("output", C.c_char_p), ("call", C.c_int64), ("ret", C.c_int64)]


class _Ev(C.Structure):
    _fields_ = [("client_id", C.c_int32), ("is_return", C.c_int32), ("id", C.c_int32), ("op", C.c_int32),
                ("key", C.c_char_p), ("value", C.c_char_p)]


EXPORTS = ["lc_check_kv_operations", "lc_check_kv_events", "lc_bitset_hash", "lc_last_cache_size"]
_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("checker library missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.lc_check_kv_operations.argtypes = [C.POINTER(_Op), C.c_int64, C.c_int64, C.POINTER(C.c_int32)]
        lib.lc_check_kv_events.argtypes = [C.POINTER(_Ev), C.c_int64, C.c_int64, C.POINTER(C.c_int32)]
        lib.lc_bitset_hash.argtypes = [C.POINTER(C.c_uint64), C.c_int64]
        lib.lc_bitset_hash.restype = C.c_uint64
        lib.lc_last_cache_size.restype = C.c_int64
        _lib = lib
    return _lib


def _b(s):
    return s.encode() if isinstance(s, str) else s


def check_operations_timeout(history, seconds):
    lib = load_library()
    arr = (_Op * max(1, len(history)))()
    for i, o in enumerate(history):
        arr[i] = _Op(o.client_id, o.op, _b(o.key), _b(o.value), _b(o.output), o.call, o.ret)
    res = C.c_int32(-1)
    if lib.lc_check_kv_operations(arr, len(history), int(seconds * 1000), C.byref(res)) != 0:
        raise ValueError("malformed history")
    return _RESULT[res.value]


def check_operations(history):
    return check_operations_timeout(history, 0) == OK


def check_events_timeout(history, seconds):
    lib = load_library()
    arr = (_Ev * max(1, len(history)))()
    for i, e in enumerate(history):
        arr[i] = _Ev(e.client_id, int(bool(e.is_return)), e.id, e.op, _b(e.key), _b(e.value))
    res = C.c_int32(-1)
    if lib.lc_check_kv_events(arr, len(history), int(seconds * 1000), C.byref(res)) != 0:
        raise ValueError("malformed history")
    return _RESULT[res.value]


def check_events(history):
    return check_events_timeout(history, 0) == OK


def bitset_hash(words):
    lib = load_library()
    arr = (C.c_uint64 * max(1, len(words)))(*words)
    return int(lib.lc_bitset_hash(arr, len(words)))


def last_cache_size():
    return int(load_library().lc_last_cache_size())
