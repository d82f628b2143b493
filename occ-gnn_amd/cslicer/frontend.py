"""Python surface of the reference's pybind11 module `cslicer`, over the HIP engine.

Mirrors cslicer/pyfrontend.cpp:116-148 name for name:

    cslicer.cslicer(name, queue_size, no_worker_threads, number_of_epochs, minibatch_size)
        .getSample() -> cslicer.sample        .getNoSamples() -> int
    cslicer.sample.layers          list[n_layers][n_parts] of cslicer.bipatite
    cslicer.bipatite               in_nodes, indptr, out_nodes, owned_out_nodes, indices,
                                   from_ids, to_ids, self_ids_in, self_ids_out, gpu_id
    cslicer.test_pyfront(), cslicer.test_list(l)

Differences, all additive (keyword arguments with the reference's constants as
defaults): data_root (reference: "/data/sandeep/", pyfrontend.cpp:24; here also
$CSLICER_DATA_ROOT), fanout (10,10,10), n_parts (4), device, seed (5489).
`no_worker_threads` becomes the number of engine streams: each stream is one
reference worker with its own mt19937(5489); minibatch b of an epoch goes to
stream b % S (the reference's assignment is a thread race, WorkerPool.cpp:29-33).
Samples come back in batch order.  getSample() never holds results hostage: the
next round is already running on the GPU while the caller consumes this one.
"""
import ctypes as C
import os

import numpy as np

from . import _abi, l0

DEFAULT_DATA_ROOT = "/data/sandeep/"


def _libc_rand():
    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    return libc.rand


def epoch_shuffle(nodes):
    """std::random_shuffle(first, last) as libstdc++ implements it
    (bits/stl_algo.h: `j = first + std::rand() % ((i - first) + 1); iter_swap(i, j)`),
    driven by glibc rand(), which the reference never seeds (WorkerPool.cpp:40).
    In place on a numpy int64 array.  Pinned against the libstdc++ call itself in
    tests/test_frontend_cpu.py."""
    rand = _libc_rand()
    n = nodes.shape[0]
    for i in range(1, n):
        j = rand() % (i + 1)
        if i != j:
            t = nodes[i]
            nodes[i] = nodes[j]
            nodes[j] = t
    return nodes


class bipatite(object):  # spelling is the reference's API (pyfrontend.cpp:128)
    """PyBipartite (pybipartite.h:10-34). Attribute reads return fresh lists, as
    pybind11's STL casters do for def_readwrite members."""
    _NAMES = ("in_nodes", "indptr", "out_nodes", "owned_out_nodes", "indices",
              "self_ids_in", "self_ids_out")

    def __init__(self, gpu_id=-1, n_parts=4):
        self.__dict__["_d"] = {n: [] for n in self._NAMES}
        self._d["from_ids"] = [[] for _ in range(n_parts)]
        self._d["to_ids"] = [[] for _ in range(n_parts)]
        self._d["gpu_id"] = int(gpu_id)

    def __getattr__(self, name):
        d = self.__dict__["_d"]
        if name not in d:
            raise AttributeError(name)
        v = d[name]
        if name == "gpu_id":
            return v
        if name in ("from_ids", "to_ids"):
            return [list(x) for x in v]
        return list(v)

    def __setattr__(self, name, value):
        d = self.__dict__["_d"]
        if name not in d:
            raise AttributeError(name)
        if name == "gpu_id":
            d[name] = int(value)
        elif name in ("from_ids", "to_ids"):
            d[name] = [[int(x) for x in row] for row in value]
        else:
            d[name] = [int(x) for x in value]


class sample(object):
    """PySample (pybipartite.h:36-47): `layers[l][g]`, l = hop from the seeds."""

    def __init__(self, layers=None):
        self.__dict__["_layers"] = layers if layers is not None else []

    @property
    def layers(self):
        return [list(row) for row in self._layers]

    @layers.setter
    def layers(self, value):
        self.__dict__["_layers"] = [list(row) for row in value]


def test_pyfront():
    """testpysample (pyfrontend.cpp:111-114): an empty 3x4 sample."""
    return sample([[bipatite(g, 4) for g in range(4)] for _ in range(3)])


def test_list(l):
    """testlist (pyfrontend.cpp:94-109): returns [1,2,3,4] and appends 10 to the CALLER's list (`py::list l` is
    a handle to the argument, pyfrontend.cpp:107; verified on the compiled reference: [7,8] -> [7,8,10]).
    Like pybind11's py::list parameter it accepts a list only."""
    if not isinstance(l, list):
        raise TypeError("test_list(): incompatible function arguments. The following argument types are supported:\n"
                        "    1. (arg0: list) -> list")
    l.append(10)
    return [1, 2, 3, 4]


def _sample_from_engine(eng, stream, slot):
    d = eng.sample_dict(stream, slot)
    layers = []
    for parts in d["layers"]:
        row = []
        for g, bp in enumerate(parts):
            b = bipatite(g, eng.n_parts)
            for n in bipatite._NAMES:
                b._d[n] = bp[n].tolist()
            b._d["from_ids"] = [x.tolist() for x in bp["from_ids"]]
            b._d["to_ids"] = [x.tolist() for x in bp["to_ids"]]
            row.append(b)
        layers.append(row)
    return sample(layers)


class cslicer(object):
    """CSlicer (pyfrontend.cpp:25-89)."""

    def __init__(self, name, queue_size, no_worker_threads, number_of_epochs, minibatch_size,
                 data_root=None, fanout=(10, 10, 10), n_parts=4, device=0, seed=5489,
                 shuffle=True, partition="mod"):
        root = data_root or os.environ.get("CSLICER_DATA_ROOT", DEFAULT_DATA_ROOT)
        self.name = os.path.join(root, name)
        if not os.path.isdir(self.name):
            # the reference reads garbage here (no error translation, dataset.cpp:18-36)
            raise FileNotFoundError("L0 dataset directory %s not found" % self.name)
        self.queue_size = int(queue_size)  # stored and ignored, as in WorkerPool.cpp:23-24
        self.no_worker_threads = int(no_worker_threads)
        self.number_of_epochs = int(number_of_epochs)
        self.minibatch_size = int(minibatch_size)
        indptr, indices, meta = l0.read_l0(self.name, mmap=False, check=True)
        self.num_nodes = int(meta["num_nodes"])
        self._shuffle = bool(shuffle)
        S = max(1, self.no_worker_threads)
        # the reference loads partition_map_opt.bin (dataset.cpp:59-67) but slices by v % 4
        # (pyfrontend.cpp:57); partition="file" uses the map (METIS output of python/utils/metis.py)
        workload = None
        if partition == "file":
            workload = np.fromfile(os.path.join(self.name, "partition_map_opt.bin"), dtype=np.int32,
                                   count=self.num_nodes)
            if workload.shape[0] != self.num_nodes:
                raise ValueError("partition_map_opt.bin is shorter than num_nodes")
        elif partition != "mod":
            raise ValueError("partition must be 'mod' or 'file'")
        self._eng = _abi.Engine(indptr, indices, n_parts=n_parts, fanouts=tuple(fanout),
                                max_batch=self.minibatch_size, n_streams=S, n_slots=2,
                                device=device, rng_seed=seed, workload=workload)
        self._S = S
        self._training_nodes = np.arange(self.num_nodes, dtype=np.int64)  # WorkerPool.cpp:12-16
        self._batches_per_epoch = (self.num_nodes - 1) // self.minibatch_size + 1
        self._epoch = -1
        self._next_round = 0          # next round (within the epoch) to submit
        self._rounds_per_epoch = (self._batches_per_epoch + S - 1) // S
        self._pending = []            # submitted rounds: (slot, n_batches)
        self._ready = []              # (slot, stream) of the round being handed out
        self._slot = 0
        self._handed = 0
        self._prefetch()

    # -- WorkerPool::run (WorkerPool.cpp:37-60), pull-driven
    def _submit_next(self):
        if self._epoch >= self.number_of_epochs:
            return False
        if self._epoch < 0 or self._next_round >= self._rounds_per_epoch:
            self._epoch += 1
            self._next_round = 0
            if self._epoch >= self.number_of_epochs:
                return False
            if self._shuffle:
                epoch_shuffle(self._training_nodes)
            self._eng.set_nodes(self._training_nodes)
        first = self._next_round * self._S
        nb = min(self._S, self._batches_per_epoch - first)
        self._eng.submit_round(first, self.minibatch_size, nb, slot=self._slot)
        self._pending.append((self._slot, nb))
        self._slot ^= 1
        self._next_round += 1
        return True

    def _prefetch(self):
        # keep at most two rounds in flight (two result slots); the epoch's
        # node order may only be replaced once its rounds have been handed out
        while len(self._pending) < 2:
            if self._next_round >= self._rounds_per_epoch and self._pending:
                break  # epoch boundary: set_nodes must wait for the queued rounds
            if not self._submit_next():
                break

    def getSample(self):
        """WorkerPool::pop_object (WorkerPool.h:39-44); blocks until the sample exists."""
        if self._handed >= self.getNoSamples():
            raise RuntimeError("all %d samples already consumed (the reference would block forever)"
                               % self.getNoSamples())
        if not self._ready:
            slot, nb = self._pending.pop(0)
            self._ready = [(slot, s) for s in range(nb)]
            self._cur_left = nb
        slot, s = self._ready.pop(0)
        out = _sample_from_engine(self._eng, s, slot)
        self._handed += 1
        if not self._ready:
            self._prefetch()
        return out

    def getNoSamples(self):
        """CSlicer::expected_number_of_samples (pyfrontend.cpp:80-83)."""
        return self._batches_per_epoch * self.number_of_epochs

    def close(self):
        if getattr(self, "_eng", None) is not None:
            self._eng.close()
            self._eng = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
