"""``SymMatrix`` and ``matrix_de_novo``: the reference's matrix.py surface over ndarray storage.

``matrix_de_novo(genomes, func, cpus, as_distance=True)`` is the drop-in boundary
(reference matrix.py:432-497, called from scripts/phamclust.py:258).  When ``func`` is one
of the six ``METRICS`` callables the whole N x N fill runs on the GPU through
``libphamclust_hip.so`` (one upload, one ``pc_fill``); any other callable takes the generic
per-pair loop with the reference's semantics.  There is no CPU fallback for the six
metrics: a missing library or a failing call raises.

``SymMatrix`` keeps the reference's method surface (matrix.py:33-405) but stores a dense
float64 array (NaN = unset) plus a name -> slot map instead of a dict of dicts, so the
downstream clustering step gets its ndarray without N^2 Python calls.  Behaviour kept on
purpose: weights are rounded to 6 places and range-checked on write (matrix.py:316-323),
``__iter__`` yields the upper half incl. the diagonal in current node order
(matrix.py:368-379), ``medoid`` counts the diagonal twice and breaks ties by node order
(matrix.py:80-104), ``lock()`` makes writes fail (matrix.py:308-309).
"""

import logging
import os

import numpy as np
from scipy.cluster.hierarchy import dendrogram, linkage
from scipy.spatial.distance import squareform

from phamclust_amd.statistics import average, skewness, standard_deviation


class SymMatrix:
    def __init__(self, nodes, is_distance=False):
        self._nodes = nodes
        self._slot = {name: k for k, name in enumerate(nodes)}
        n = len(nodes)
        self._data = np.full((n, n), np.nan, dtype=np.float64)
        self._is_distance = is_distance
        self._locked = False

    # -- bulk constructors (new; used by the GPU fill) -----------------------------
    @classmethod
    def from_condensed(cls, nodes, condensed, is_distance=True):
        """Build from a scipy-condensed vector whose values are already rounded to 6 places;
        the diagonal is preset to ``1.0 - is_distance`` as matrix_de_novo does (matrix.py:467-468)."""
        nodes = list(nodes)
        n = len(nodes)
        condensed = np.ascontiguousarray(condensed, dtype=np.float64)
        if condensed.shape != (n * (n - 1) // 2,):
            raise ValueError(f"need {n * (n - 1) // 2} condensed values but got {condensed.shape}")
        if condensed.size and not (np.nanmin(condensed) >= 0.0 and np.nanmax(condensed) <= 1.0):
            raise ValueError("weight not in [0.0, 1.0]")
        self = cls.__new__(cls)
        self._nodes = nodes
        self._slot = {name: k for k, name in enumerate(nodes)}
        # scipy expands the vector in C: no N^2/2 index arrays (two of them were 800 MB at N = 10,000)
        self._data = squareform(condensed, force="tomatrix", checks=False) if n > 1 else np.zeros((n, n), dtype=np.float64)
        self._is_distance = is_distance
        self._locked = False
        if not is_distance:
            np.fill_diagonal(self._data, 1.0)
        return self

    # -- properties ------------------------------------------------------------------
    @property
    def is_distance(self):
        return self._is_distance

    @property
    def nodes(self):
        return self._nodes[:]

    def _order(self):
        return np.fromiter((self._slot[name] for name in self._nodes), dtype=np.int64, count=len(self._nodes))

    def _ordered(self):
        """The weights in current node order.  READ-ONLY for callers: when the traversal order is still the storage
        order this is the storage itself, not a copy (an N = 10,000 matrix is 800 MB)."""
        order = self._order()
        if order.size == self._data.shape[0] and (order == np.arange(order.size)).all():
            return self._data
        return self._data[np.ix_(order, order)]

    @property
    def diameter(self):
        if not self.is_distance:
            raise ValueError("cannot compute diameter for similarity matrix")
        n = len(self)
        if n < 2:
            return 0.0
        off = squareform(self._ordered(), force="tovector", checks=False)
        top = np.nanmax(off) if off.size else 0.0
        return round(float(max(top, 0.0)), 6)

    def _central(self):
        """Per node, the mean of its N+1 incident weights.  The reference gathers them by walking the upper half
        (matrix.py:84-97), so node k's list reads w(0,k) .. w(k-1,k), w(k,k) TWICE, w(k,k+1) .. and is summed left
        to right; sweeping the columns in order and adding the diagonal a second time right after its own column
        adds the same numbers in the same order for every node at once (ties between nodes fall as they do there)."""
        n = len(self)
        m = self._ordered()
        totals = np.zeros(n, dtype=np.float64)
        for j in range(n):
            totals += m[j]                       # column j == row j (symmetric); contiguous
            totals[j] += m[j, j]
        return [(name, float(total) / (n + 1)) for name, total in zip(self._nodes, totals)]

    @property
    def medoid(self):
        scored = sorted(self._central(), key=lambda item: item[1])
        return scored[0] if self.is_distance else scored[-1]

    @property
    def anti_medoid(self):
        scored = sorted(self._central(), key=lambda item: item[1])
        return scored[-1] if self.is_distance else scored[0]

    @property
    def statistics(self):
        if len(self) == 1:
            node = self._nodes[0]
            return self.get_weight(node, node), 0.0, 0.0
        edges = squareform(self._ordered(), force="tovector", checks=False).tolist()    # upper half, row by row
        mean = average(edges)
        std_dev = standard_deviation(edges, mean)
        skew = 0.0 if std_dev == 0.0 else skewness(edges, mean)
        return mean, std_dev, skew

    # -- structure -------------------------------------------------------------------
    def extract_submatrix(self, nodes):
        for name in nodes:
            if name not in self:
                raise KeyError(f"node '{name}' not in matrix")
        sub = SymMatrix(nodes, self.is_distance)
        idx = np.fromiter((self._slot[name] for name in nodes), dtype=np.int64, count=len(nodes))
        sub._data = self._data[np.ix_(idx, idx)].copy()
        return sub

    def append_node(self, source, data):
        """Grow by one node whose weights to every node (itself included) come in ``data`` (matrix.py:169-213):
        the new name must be new, ``data`` must cover exactly the current nodes plus the new one, and the self-edge
        must be what the diagonal of this kind of matrix holds."""
        if self._locked:
            raise AttributeError("matrix is marked as read-only")
        if source in self:
            raise KeyError(f"node '{source}' is already in this matrix")
        if source not in data:
            raise KeyError(f"incoming data lacks an self-edge for '{source}'")
        on_diagonal = 0.0 if self.is_distance else 1.0
        if data[source] != on_diagonal:
            kind = "distance" if self.is_distance else "similarity"
            raise ValueError(f"nonsense value {data[source]} for self-edge on {kind} matrix")
        given = set(data) - {source}
        absent = set(self._slot) - given
        if absent:
            raise KeyError(f"missing edge(s) for {source} vs: {absent}")
        unknown = given - set(self._slot)
        if unknown:
            raise KeyError(f"specified edges for nodes not found in matrix: {unknown}")
        n = self._data.shape[0]
        edge = np.empty(n + 1, dtype=np.float64)
        for name, k in self._slot.items():
            edge[k] = data[name]
        edge[n] = data[source]
        if not (edge.min() >= 0.0 and edge.max() <= 1.0):
            raise ValueError(f"weight {edge[(edge < 0.0) | (edge > 1.0) | (edge != edge)][0]} not in [0.0, 1.0]")
        edge = np.array([round(float(w), 6) for w in edge])     # Python's round: what set_weight stores (matrix.py:323)
        grown = np.empty((n + 1, n + 1), dtype=np.float64)
        grown[:n, :n] = self._data
        grown[n, :] = edge
        grown[:, n] = edge
        self._data = grown
        self._slot[source] = n
        self._nodes.append(source)

    def get_weight(self, source, target):
        if source not in self:
            raise KeyError(f"node '{source}' not in matrix")
        if target not in self:
            raise KeyError(f"node '{target}' not in matrix")
        value = self._data[self._slot[source], self._slot[target]]
        return None if value != value else float(value)

    def set_weight(self, source, target, weight):
        if self._locked:
            raise AttributeError("matrix is marked as read-only")
        if source not in self:
            raise KeyError(f"node '{source}' not in matrix")
        if target not in self:
            raise KeyError(f"node '{target}' not in matrix")
        if not 0 <= weight <= 1:
            raise ValueError(f"weight {weight} not in [0.0, 1.0]")
        i, j = self._slot[source], self._slot[target]
        self._data[i, j] = self._data[j, i] = round(weight, 6)

    def invert(self):
        """distance <-> similarity in place: every stored w becomes round(1 - w, 6)
        (matrix.py:236-247).  Stored values are 6-place decimals, so numpy's rounding and
        Python's agree here."""
        if self._locked:
            raise AttributeError("matrix is marked as read-only")
        self._data = np.round(1.0 - self._data, 6)
        self._is_distance = not self.is_distance
        return self

    def reorder(self, nodes=None):
        """New traversal order (storage stays put); without ``nodes``: leaf order of the single-linkage tree
        (matrix.py:249-263)."""
        order = list(nodes) if nodes else [self._nodes[leaf] for leaf in _get_tree_order(self)]
        if len(order) != len(self._nodes):
            raise ValueError(f"need {len(self)} nodes but got {len(order)}")
        stranger = next((name for name in order if name not in self), None)
        if stranger is not None:
            raise KeyError(f"node '{stranger}' not in matrix")
        self._nodes = order

    def nearest_neighbors(self, source, threshold):
        """Nodes within ``threshold`` of ``source`` (<= on distances, >= on similarities), closest first; equally
        close ones keep their traversal order (matrix.py:265-296: a stable sort, reversed for similarities)."""
        order = self._order()
        if order.size == 0 or (order.size == 1 and self._nodes[0] == source):
            return []
        if source not in self:
            raise KeyError(f"node '{source}' not in matrix")
        row = self._data[self._slot[source], order]
        others = order != self._slot[source]
        if np.isnan(row[others]).any():
            raise TypeError("cannot rank neighbours of a node with unset edges")
        near = np.flatnonzero(others & ((row <= threshold) if self.is_distance else (row >= threshold)))
        ranked = near[np.argsort(row[near] if self.is_distance else -row[near], kind="stable")]
        return [self._nodes[k] for k in ranked]

    def lock(self):
        self._locked = True

    def unlock(self):
        self._locked = False

    def is_locked(self):
        return self._locked

    def to_ndarray(self, condensed=False):
        full = self._ordered()
        if condensed:
            return squareform(full, force="tovector")
        return full.copy() if full is self._data else full

    # -- container protocol ------------------------------------------------------------
    def __contains__(self, item):
        if not isinstance(item, str):
            raise TypeError(f"type(item) should be 'str', not '{type(item)}'")
        return item in self._slot

    def __getitem__(self, item):
        """Row of ``item`` as the reference stores it: only targets that do not sort before
        ``item`` as strings (matrix.py:231-232, 320-321)."""
        if item not in self:
            raise KeyError(f"node '{item}' not in matrix")
        row = self._data[self._slot[item]]
        return {name: float(row[k]) for name, k in self._slot.items() if not name < item and row[k] == row[k]}

    def __iter__(self):
        m = self._ordered()
        for i, source in enumerate(self._nodes):
            row = m[i]
            for j in range(i, len(self._nodes)):
                value = row[j]
                yield source, self._nodes[j], (None if value != value else float(value))

    def iterrows(self):
        m = self._ordered()
        for i, source in enumerate(self._nodes):
            yield source, [None if v != v else v for v in m[i].tolist()]

    def __len__(self):
        return len(self._nodes)

    def __lt__(self, other):
        if not isinstance(other, SymMatrix):
            raise TypeError(f"cannot compare SymMatrix to {type(other)}")
        return len(self) < len(other)

    def __str__(self):
        lines = [f"{len(self)}\n"]
        for source, row in self.iterrows():
            lines.append(f"{source:<24}\t" + "\t".join([f"{x:.6f}" for x in row]) + "\n")
        return "".join(lines)


# ---------------------------------------------------------------------------------------
# matrix_de_novo
# ---------------------------------------------------------------------------------------
def _outside_in_index_iterator(num_indices):
    """Indices from both ends toward the middle: 5 -> 0, 4, 1, 3, 2 (matrix.py:409-423)."""
    lo, hi = 0, num_indices - 1
    while lo < hi:
        yield lo
        yield hi
        lo, hi = lo + 1, hi - 1
    if lo == hi:
        yield lo


def calculate_adjacency(source, target, func, distance=True):
    return source.name, target.name, func(source, target, as_distance=distance)


_CONTEXTS = {}


def default_device():
    """PHAMCLUST_DEVICE, else this rank's LOCAL_RANK (one process per GPU under torch.distributed.run), else 0."""
    if "PHAMCLUST_DEVICE" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from phamclust_amd import distributed
        return distributed.local_device()
    return int(os.environ.get("PHAMCLUST_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def get_context(device_id=None):
    """One cached HIP context per device for the life of the process."""
    from phamclust_amd import hip
    device_id = default_device() if device_id is None else device_id
    if device_id not in _CONTEXTS:
        _CONTEXTS[device_id] = hip.Context(device_id)
    return _CONTEXTS[device_id]


def in_process_devices():
    """Device ordinals of an in-process multi-GPU fill, or None: ``PHAMCLUST_GPUS=N`` (what ``phamclust --gpus N`` sets when it
    keeps the job in this process) -> devices 0..N-1; ``PHAMCLUST_GPU_IDS=0,0,1`` names them (an id may repeat: rehearsal on a box
    with fewer GPUs).  Ignored under a launcher (WORLD_SIZE > 1: one process per GPU is then the job's shape)."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return None
    ids = os.environ.get("PHAMCLUST_GPU_IDS")
    if ids:
        return [int(x) for x in ids.split(",")]
    n = int(os.environ.get("PHAMCLUST_GPUS", "1"))
    return list(range(n)) if n > 1 else None


def get_multi_context(device_ids):
    from phamclust_amd import hip
    key = ("multi",) + tuple(device_ids)
    if key not in _CONTEXTS:
        try:
            _CONTEXTS[key] = hip.MultiContext(device_ids)
        except hip.HipLibraryError as exc:
            raise hip.HipLibraryError(f"{exc} -- asked for devices {list(device_ids)} (--gpus / PHAMCLUST_GPUS / PHAMCLUST_GPU_IDS): "
                                      f"ask for the GPUs this node has, or name them with PHAMCLUST_GPU_IDS") from None
    return _CONTEXTS[key]


def _metric_name(func):
    from phamclust_amd import metrics
    return metrics.ACCELERATED.get(func)


LAST_FILL = {}          # what the last accelerated matrix_de_novo did: the pipeline's log line reads it


def _packed_of(genomes):
    """The packed form of ``genomes``: the loader's own arrays when the list is exactly what the C loader produced
    (pack.load_tsv_genomes: untouched lazy genomes, all of them, in order), else a fresh pack of the objects."""
    from phamclust_amd.pack import pack_genomes, packed_behind
    return packed_behind(genomes) or pack_genomes(genomes)


def matrix_de_novo(genomes, func, cpus, as_distance=True):
    """Fill an N x N ``SymMatrix`` with ``func`` over every genome pair (matrix.py:432-497).

    The six METRICS run on the GPU.  Where the reference spreads the pairs over ``cpus`` worker processes
    (matrix.py:471-472), this spreads them over the GPUs of the job it runs in: under
    ``python -m torch.distributed.run --nproc-per-node N`` (one process per GPU; ``phamclust --gpus N`` starts that
    for you) every rank calls this with the same genomes, fills its static shard of the pair list, and ONE gather
    brings the shards to rank 0.  Rank 0 returns the matrix; every other rank returns ``None``.  Without a launcher,
    ``PHAMCLUST_GPUS=N`` (set by ``phamclust --gpus N``) spreads the same shards over N GPUs from THIS process (``pc_multi_*``:
    a host thread per device inside the library, device-to-device copies for the exchange).  For those six, ``cpus`` is
    accepted for signature compatibility only; any OTHER callable is filled the reference's way -- per pair, in its batch
    order, over ``cpus`` joblib workers (in this process when ``cpus`` is 1).
    """
    if len(genomes) == 0:
        raise ValueError("need at least 1 genome to construct matrix de novo")
    names = [g.name for g in genomes]
    metric = _metric_name(func)
    if metric is not None:
        import time
        from phamclust_amd import distributed
        rank, world = distributed.ensure_process_group()          # before the first GPU call of this process
        t0 = time.perf_counter()
        packed = _packed_of(genomes)
        t1 = time.perf_counter()
        devices = in_process_devices()
        ctx = get_multi_context(devices) if devices else get_context()
        # gcs / jc / pocp / af never read a residue (metrics.py:26-157): their upload skips the residue stage
        ctx.upload(packed, residues=metric in ("aai", "peq"))
        t2 = time.perf_counter()
        if world > 1:
            condensed, stats = distributed.fill_condensed(ctx, metric, as_distance)
        else:                                              # one process: one GPU, or several through pc_multi_* (no launcher, no process group)
            condensed, stats = ctx.fill(metric, as_distance=as_distance, want_stats=True, borrow=True)
        t3 = time.perf_counter()
        LAST_FILL.clear()
        LAST_FILL.update(stats, metric=metric, n_genomes=len(genomes), genome_pairs=packed.n_pairs, n_gpus=len(devices) if devices else world, rank=rank,
                         pack_s=t1 - t0, upload_s=t2 - t1, fill_s=t3 - t2)
        logging.debug(f"{len(genomes)} genomes -> {packed.n_pairs} edges on {world} device(s): pack {t1 - t0:.3f} s, "
                      f"upload {t2 - t1:.3f} s, fill+gather+D2H {t3 - t2:.3f} s (kernels {stats['ms_total']:.3f} ms on this rank)")
        if condensed is None:
            return None
        return SymMatrix.from_condensed(names, condensed, is_distance=as_distance)

    # any other callable: the reference's per-pair semantics, its batch order, its worker pool (matrix.py:460-493)
    n = len(genomes)
    n_pairs = n * (n - 1) // 2 + n
    if n_pairs < cpus:                                  # (matrix.py:460-462)
        logging.info(f"small dataset - reducing # CPUs to {n_pairs}")
        cpus = n_pairs
    matrix = SymMatrix(nodes=names, is_distance=as_distance)
    for genome in genomes:
        matrix.set_weight(genome.name, genome.name, 1.0 - as_distance)
    order = list(_outside_in_index_iterator(n))
    if cpus is None or cpus <= 1:
        for i in order:
            source = genomes[i]
            for target in genomes[i + 1:]:
                s, t, w = calculate_adjacency(source, target, func, as_distance)
                matrix.set_weight(s, t, w)
        return matrix
    # Batches of `stride` source rows in outside-in order hold about 10,000 pairs per worker (matrix.py:474-478); inside a batch
    # one task is one SOURCE ROW cut into pieces of at most 2,500 targets -- the reference pickles both genomes of every pair
    # (matrix.py:484-488), this sends a source once per piece -- and results come back in whatever order the workers finish.
    import joblib
    stride = max(1, int(n // (n_pairs / (10000 * cpus))))
    runner = joblib.Parallel(n_jobs=cpus, return_as="generator_unordered", max_nbytes=None)
    for b0 in range(0, n, stride):
        rows = order[b0:b0 + stride]
        pieces = [(genomes[i], genomes[j0:min(j0 + 2500, n)]) for i in rows for j0 in range(i + 1, n, 2500)]
        for edges in runner(joblib.delayed(_row_adjacency)(source, targets, func, as_distance) for source, targets in pieces):
            for s, t, w in edges:
                matrix.set_weight(s, t, w)
        logging.debug(f"finished {', '.join(genomes[i].name for i in rows)}")
    del runner
    return matrix


def _row_adjacency(source, targets, func, distance):
    """One worker task of the generic path: ``source`` against a run of targets."""
    return [calculate_adjacency(source, target, func, distance) for target in targets]


# ---------------------------------------------------------------------------------------
# TSV I/O: adjacency, squareform, lower triangle (matrix.py:500-670); "%.6f" everywhere
# ---------------------------------------------------------------------------------------
def read_adjacency(filepath):
    with open(filepath, "r") as handle:
        for line in handle:
            fields = line.rstrip().split("\t")
            yield fields[0], fields[1], float(fields[2])


def _diagonal_kind(diagonal, count):
    if len(diagonal) > 1:
        raise ValueError("values on matrix diagonal should be identical")
    total = sum(diagonal)
    if total == 0.0:
        return True
    if total == count:
        return False
    raise ValueError("values on matrix diagonal can only be 0.0 or 1.0")


def matrix_from_adjacency(filepath):
    names, diagonal = dict(), set()
    for source, target, weight in read_adjacency(filepath):
        if source == target:
            diagonal.add(weight)
        if target in names:
            break
        names[target] = None
    matrix = SymMatrix(list(names), is_distance=_diagonal_kind(diagonal, 1.0))
    for source, target, weight in read_adjacency(filepath):
        matrix.set_weight(source, target, weight)
    return matrix


_TEXT_LIB = None


def _text_lib():
    """csrc/libpc_pack.so's row formatter / parser ("%.6f", byte for byte), or None when it is not built: the text
    I/O then runs in Python (same bytes, ~100x slower at N = 10,000)."""
    global _TEXT_LIB
    if _TEXT_LIB is None:
        import ctypes
        import os
        from phamclust_amd.build import native_path
        path = native_path("libpc_pack.so")
        try:
            lib = ctypes.CDLL(path)
            lib.pcp_format_row.restype = ctypes.c_int64
            lib.pcp_format_row.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
            lib.pcp_format_adjacency.restype = ctypes.c_int64
            lib.pcp_format_adjacency.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64]
            lib.pcp_parse_row.restype = ctypes.c_int64
            lib.pcp_parse_row.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
            _TEXT_LIB = lib
        except (OSError, AttributeError):
            _TEXT_LIB = False
    return _TEXT_LIB or None


def matrix_to_adjacency(matrix, filepath, skip_zero=False):
    lib = _text_lib() if isinstance(matrix, SymMatrix) else None
    data = matrix._ordered() if lib else None
    if lib is None or np.isnan(data).any():                 # unset cells: the reference's formatting error surfaces below
        with open(filepath, "w") as handle:
            for source, target, weight in matrix:
                if skip_zero and not weight:
                    continue
                handle.write(f"{source}\t{target}\t{weight:.6f}\n")
        return filepath
    import ctypes
    names = [name.encode() for name in matrix.nodes]
    blob = b"".join(names)
    offsets = np.zeros(len(names) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in names], out=offsets[1:])
    n = len(names)
    width = (max((len(x) for x in names), default=0)) * 2 + 32
    cap = n * width + 1024
    buf = ctypes.create_string_buffer(cap)
    data = np.ascontiguousarray(data, dtype=np.float64)
    with open(filepath, "wb") as handle:
        for i, source in enumerate(names):
            size = lib.pcp_format_adjacency(source, len(source), blob, offsets.ctypes.data, data[i].ctypes.data, i, n,
                                            1 if skip_zero else 0, buf, cap)
            if size < 0:
                raise ValueError(f"{filepath}: a row of '{source.decode()}' does not fit its buffer (weights outside [0, 1]?)")
            handle.write(buf.raw[:size] if size < 65536 else memoryview(buf)[:size])
    return filepath


def read_squareform(filepath):
    with open(filepath, "r") as handle:
        next(handle)
        for line in handle:
            fields = line.rstrip().split("\t")
            yield fields[0], [float(x) for x in fields[1:]]


def _read_rows(filepath):
    """(name, ndarray row) per line: the C parser when csrc/libpc_pack.so is there, else read_squareform."""
    lib = _text_lib()
    if lib is None:
        for name, row in read_squareform(filepath):
            yield name, np.asarray(row, dtype=np.float64)
        return
    with open(filepath, "rb") as handle:
        next(handle)
        for line in handle:
            line = line.rstrip()
            name, tab, rest = line.partition(b"\t")
            cap = rest.count(b"\t") + 1 if tab else 0
            row = np.empty(cap, dtype=np.float64)
            got = lib.pcp_parse_row(rest, len(rest), row.ctypes.data, cap) if cap else 0
            if got != cap:
                raise ValueError(f"{filepath}: could not parse the row of {name.decode()!r}")
            yield name.decode(), row


def matrix_from_squareform(filepath):
    names, rows, diagonal = [], [], set()
    for i, (target, row) in enumerate(_read_rows(filepath)):
        names.append(target)
        rows.append(row)
        diagonal.add(float(row[i]))
    matrix = SymMatrix(names, is_distance=_diagonal_kind(diagonal, len(diagonal)))
    n = len(names)
    data = matrix._data
    for i, row in enumerate(rows):
        values = np.asarray(row[:i + 1], dtype=np.float64)
        if values.size and not (values.min() >= 0.0 and values.max() <= 1.0):
            raise ValueError(f"weight not in [0.0, 1.0] on row {i}")
        values = np.round(values, 6)
        data[i, :i + 1] = values
        data[:i + 1, i] = values
    assert data.shape == (n, n)
    return matrix


def matrix_to_squareform(matrix, filepath, lower_triangle=False):
    lib = _text_lib() if isinstance(matrix, SymMatrix) else None
    data = matrix._ordered() if lib else None
    if lib is None or np.isnan(data).any():
        with open(filepath, "w") as handle:
            header = f"{len(matrix)}"
            if not lower_triangle:
                header += "\t" + "\t".join(matrix.nodes)
            handle.write(f"{header}\n")
            for i, (source, row) in enumerate(matrix.iterrows()):
                cells = row[:i + 1] if lower_triangle else row
                handle.write(f"{source}\t" + "\t".join([f"{x:.6f}" for x in cells]) + "\n")
        return filepath
    import ctypes
    n = len(matrix)
    data = np.ascontiguousarray(data, dtype=np.float64)
    cap = n * 24 + 1024
    buf = ctypes.create_string_buffer(cap)
    with open(filepath, "wb") as handle:
        header = f"{n}"
        if not lower_triangle:
            header += "\t" + "\t".join(matrix.nodes)
        handle.write(f"{header}\n".encode())
        for i, source in enumerate(matrix.nodes):
            size = lib.pcp_format_row(data[i].ctypes.data, i + 1 if lower_triangle else n, buf, cap)
            if size < 0:
                raise ValueError(f"{filepath}: the row of '{source}' does not fit its buffer (weights outside [0, 1]?)")
            handle.write(source.encode() + b"\t")
            handle.write(memoryview(buf)[:size])
    return filepath


def _get_tree_order(matrix):
    """Leaf order of a single-linkage dendrogram on distances (matrix.py:673-686)."""
    if not matrix.is_distance:
        matrix = matrix.extract_submatrix(matrix.nodes)
        matrix.invert()
    z = linkage(matrix.to_ndarray(condensed=True), method="single")
    return dendrogram(z, no_plot=True, get_leaves=True, count_sort="descending")["leaves"]
