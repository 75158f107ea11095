"""Reads a `torch.save`d state_dict WITHOUT torch (numpy only): the zip container torch >= 1.6 writes -- `<name>/data.pkl`
(a protocol-2 pickle whose tensors are `torch._utils._rebuild_tensor_v2(storage, offset, size, stride, ...)` calls over
persistent ids `('storage', <torch.XStorage>, key, location, numel)`) plus one raw little-endian file per storage,
`<name>/data/<key>`.  Only what a state_dict needs is understood (dicts / OrderedDicts of dense CPU tensors, nn.Parameter
wrappers); anything else raises `Unsupported`, and the caller falls back to `torch.load`.  It exists so that the tools can
read `model.pt` on their start-up thread while the interpreter is still importing torch (pfann_amd/prewarm.py)."""
import collections
import pickle
import zipfile

import numpy as np


class Unsupported(Exception):
    pass


_DTYPES = {"FloatStorage": "<f4", "DoubleStorage": "<f8", "HalfStorage": "<f2", "LongStorage": "<i8", "IntStorage": "<i4",
           "ShortStorage": "<i2", "CharStorage": "i1", "ByteStorage": "u1", "BoolStorage": "?"}


class _StorageType:
    def __init__(self, name):
        if name not in _DTYPES:
            raise Unsupported("storage type %s" % name)
        self.dtype = np.dtype(_DTYPES[name])


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    flat, dtype = storage
    size, stride = tuple(size), tuple(stride)
    if len(size) == 0:
        return flat[storage_offset:storage_offset + 1].reshape(())
    need = storage_offset + sum((n - 1) * s for n, s in zip(size, stride)) + 1 if all(n > 0 for n in size) else 0
    if need > flat.shape[0]:
        raise Unsupported("tensor outside its storage")
    return np.lib.stride_tricks.as_strided(flat[storage_offset:], shape=size, strides=tuple(s * dtype.itemsize for s in stride),
                                           writeable=False)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


class _Unpickler(pickle.Unpickler):
    def __init__(self, f, zf, prefix):
        super().__init__(f)
        self.zf, self.prefix, self.cache = zf, prefix, {}

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return collections.OrderedDict
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name.endswith("Storage"):
            return _StorageType(name)
        raise Unsupported("global %s.%s" % (module, name))

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise Unsupported("persistent id %r" % (pid,))
        _, st, key, location, numel = pid[:5]
        if str(location) != "cpu" and not str(location).startswith("cuda"):
            raise Unsupported("location %r" % (location,))
        if key not in self.cache:
            raw = self.zf.read("%s/data/%s" % (self.prefix, key))
            flat = np.frombuffer(raw, dtype=st.dtype)
            if flat.shape[0] < int(numel):
                raise Unsupported("storage %s shorter than declared" % key)
            self.cache[key] = (flat, st.dtype)
        return self.cache[key]


def load_state_dict_numpy(path):
    """-> OrderedDict name -> numpy array (read-only views of the file's storages); raises Unsupported / OSError / ..."""
    with zipfile.ZipFile(path) as zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise Unsupported("not a torch.save zip archive")
        prefix = pkl[0][: -len("/data.pkl")]
        order = [n for n in zf.namelist() if n == prefix + "/byteorder"]
        if order and zf.read(order[0]).strip() not in (b"little", b""):
            raise Unsupported("big-endian archive")
        import io
        obj = _Unpickler(io.BytesIO(zf.read(pkl[0])), zf, prefix).load()
    if not isinstance(obj, dict):
        raise Unsupported("top-level object is %s, not a state_dict" % type(obj).__name__)
    out = collections.OrderedDict()
    for k, v in obj.items():
        if not isinstance(k, str) or not isinstance(v, np.ndarray):
            raise Unsupported("entry %r is not a tensor" % (k,))
        out[k] = v
    return out
