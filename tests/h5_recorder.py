"""A minimal in-memory stand-in for ``h5py.File`` (h5py is not on the image): groups with ``attrs``,
``create_group`` (nested paths, ``track_order``), dataset assignment / in-place update / ``flush``,
``in`` / ``del`` / nested ``[]``.  It records what a writer does so that tests can check group,
dataset and attribute names, shapes and values of the reference's DataHandler layout
(tdgl/solver/runner.py:104-183) exactly."""

import numpy as np

OPENED = {}  # path -> RecorderFile (kept after close, for inspection)


class Dataset:
    def __init__(self, value):
        self.value = np.array(value)
        self.flushes = 0

    @property
    def shape(self):
        return self.value.shape

    def __setitem__(self, key, value):
        self.value[key] = value

    def __getitem__(self, key):
        return self.value[key]

    def __array__(self, dtype=None, copy=None):
        return self.value if dtype is None else self.value.astype(dtype)

    def flush(self):
        self.flushes += 1


class Group:
    def __init__(self, track_order=False):
        self.attrs = {}
        self.items_ = {}
        self.track_order = track_order

    def _walk(self, path, create=False):
        node = self
        parts = [p for p in str(path).split("/") if p]
        for p in parts[:-1]:
            if p not in node.items_:
                if not create:
                    raise KeyError(path)
                node.items_[p] = Group()
            node = node.items_[p]
        return node, parts[-1]

    def create_group(self, name, track_order=False):
        node, leaf = self._walk(name, create=True)
        if leaf in node.items_:
            raise ValueError(f"group {name!r} exists")
        node.items_[leaf] = Group(track_order)
        return node.items_[leaf]

    def require_group(self, name):
        return self[name] if name in self else self.create_group(name)

    def __setitem__(self, key, value):
        node, leaf = self._walk(key, create=True)
        if leaf in node.items_:
            raise ValueError(f"dataset {key!r} exists (h5py refuses to overwrite)")
        node.items_[leaf] = Dataset(value)

    def __getitem__(self, key):
        node, leaf = self._walk(key)
        return node.items_[leaf]

    def __delitem__(self, key):
        node, leaf = self._walk(key)
        del node.items_[leaf]

    def __contains__(self, key):
        try:
            node, leaf = self._walk(key)
        except KeyError:
            return False
        return leaf in node.items_

    def __iter__(self):
        return iter(self.items_)

    def keys(self):
        return self.items_.keys()


class RecorderFile(Group):
    def __init__(self, path, mode="x", **kw):
        super().__init__()
        self.path, self.mode, self.kw = path, mode, kw
        self.closed = False
        self.swmr_mode = False
        self.file_flushes = 0
        if mode == "x" and path in OPENED and not OPENED[path].closed:
            raise FileExistsError(path)
        if mode == "x":  # like h5py, leave a file behind (readers check that it exists)
            try:
                open(path, "ab").close()
            except OSError:
                pass
        OPENED[path] = self

    def flush(self):
        self.file_flushes += 1

    def close(self):
        self.closed = True


def open_file(path, mode="x", **kw):
    """Factory with h5py.File's signature: "x" creates, "r" re-opens what was written to ``path``."""
    if mode == "r":
        f = OPENED[str(path)]
        f.closed = False
        return f
    return RecorderFile(str(path), mode, **kw)
