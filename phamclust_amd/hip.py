"""ctypes binding of ``libphamclust_hip.so`` (C-ABI: ``include/phamclust_hip.h``).

This is the only way the package computes anything: there is no CPU fallback.  If the
library is missing or a call fails, the error is raised, never papered over.
"""

import ctypes
import os
import weakref

import numpy as np

# PHAMCLUST_NATIVE_VARIANT=hooks loads the twin compiled with -DPC_TEST_HOOKS (fault injection; one GPU test runs on it); any other
# name but "asan" (the host libraries' sanitized twins) loads csrc/libphamclust_hip_<name>.so -- A/B of two builds in one GPU call
_variant = os.environ.get("PHAMCLUST_NATIVE_VARIANT", "")
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                        f"libphamclust_hip_{_variant}.so" if _variant not in ("", "asan") else "libphamclust_hip.so")
METRIC_IDS = {"gcs": 0, "jc": 1, "pocp": 2, "af": 3, "aai": 4, "peq": 5, "aai_ppos": 6}

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)


class HipLibraryError(RuntimeError):
    """The HIP library is missing, or one of its entry points returned an error."""


class PcPacked(ctypes.Structure):
    _fields_ = [("n_genomes", ctypes.c_int32), ("n_phams", ctypes.c_int32),
                ("words_per_row", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("bitmap", _u64p), ("nph", _i32p), ("ngen", _i32p), ("tlen", _i64p),
                ("gene_off", _i64p), ("gene_pham", _i32p), ("seq_off", _i64p), ("residues", _u8p)]


class PcStats(ctypes.Structure):
    _fields_ = [("n_pairs", ctypes.c_int64), ("n_alignments", ctypes.c_int64), ("n_cells", ctypes.c_int64),
                ("n_tasks", ctypes.c_int64), ("n_residue_bytes", ctypes.c_int64),
                ("n_align_launches", ctypes.c_int32), ("n_chunks", ctypes.c_int32),
                ("ms_total", ctypes.c_float), ("ms_plan", ctypes.c_float), ("ms_align", ctypes.c_float),
                ("ms_reduce", ctypes.c_float), ("n_distinct_alignments", ctypes.c_int64), ("n_distinct_cells", ctypes.c_int64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


EXPORTS = ["pc_version", "pc_test_hooks", "pc_last_error", "pc_ctx_create", "pc_ctx_destroy", "pc_upload", "pc_set_shard", "pc_set_shard_balanced",
           "pc_shard_pairs", "pc_shard_stride", "pc_fill", "pc_fill_borrow", "pc_fill_dev", "pc_fill_shard_dev", "pc_assemble_dev",
           "pc_align_pairs", "pc_last_align_ms", "pc_round6_probe", "pc_set_tie_rule", "pc_get_tie_rule", "pc_shard_table", "pc_target_costs",
           "pc_plan_dev", "pc_align_slice_dev", "pc_reduce_dev", "pc_upload_sets", "pc_upload_residues", "pc_set_plan_budget", "pc_chunk_plan",
           "pc_variant_width", "pc_last_set_kernel", "pc_multi_create", "pc_multi_destroy", "pc_multi_devices", "pc_multi_peer_access", "pc_multi_upload",
           "pc_multi_upload_residues", "pc_multi_set_tie_rule", "pc_multi_fill_borrow"]
NEEDS_RESIDUES = ("aai", "peq", "aai_ppos")

_lib = None


def load():
    """Load the library (once).  Raises :class:`HipLibraryError` when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(f"{LIB_PATH} not found: build it with `python -m phamclust_amd.build` "
                              f"(there is no CPU fallback)")
    # torch ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both open
    # the GPU.  Importing torch first makes our library bind to the runtime torch uses, so
    # torch tensors, streams and torch.distributed (RCCL) share one device context with it.
    # PHAMCLUST_NO_TORCH=1 (the single-rank CLI sets it) skips that import -- ~1.5 s, a minute or two on a machine that pages torch
    # in for the first time -- for processes that will never import torch: the library then binds to the system's HIP runtime.
    import sys
    if os.environ.get("PHAMCLUST_NO_TORCH") != "1" or "torch" in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {exc}") from None
    vp = ctypes.c_void_p
    L.pc_version.restype = ctypes.c_int
    L.pc_test_hooks.restype = ctypes.c_int
    L.pc_last_error.restype = ctypes.c_char_p
    L.pc_ctx_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    L.pc_ctx_destroy.argtypes = [vp]
    L.pc_ctx_destroy.restype = None
    L.pc_upload.argtypes = [vp, ctypes.POINTER(PcPacked)]
    L.pc_upload_sets.argtypes = [vp, ctypes.POINTER(PcPacked)]
    L.pc_upload_residues.argtypes = [vp, ctypes.POINTER(PcPacked)]
    L.pc_set_plan_budget.argtypes = [vp, ctypes.c_int64]
    L.pc_chunk_plan.argtypes = [_u64p, ctypes.c_int, ctypes.c_uint64, _i32p, ctypes.c_int]
    L.pc_variant_width.argtypes = [ctypes.c_int]
    L.pc_set_shard.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.pc_set_shard_balanced.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.pc_shard_pairs.argtypes = [vp]
    L.pc_shard_pairs.restype = ctypes.c_int64
    L.pc_shard_stride.argtypes = [vp]
    L.pc_shard_stride.restype = ctypes.c_int64
    L.pc_fill.argtypes = [vp, ctypes.c_int, ctypes.c_int, _f64p, ctypes.POINTER(PcStats)]
    L.pc_fill_borrow.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_f64p), ctypes.POINTER(PcStats)]
    L.pc_fill_dev.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.POINTER(PcStats)]
    L.pc_fill_shard_dev.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.POINTER(PcStats)]
    L.pc_assemble_dev.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    L.pc_plan_dev.argtypes = [vp, ctypes.c_int, vp, ctypes.POINTER(PcStats)]
    L.pc_align_slice_dev.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.POINTER(PcStats)]
    L.pc_reduce_dev.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]
    L.pc_align_pairs.argtypes = [vp, _i32p, _i32p, ctypes.c_int64, ctypes.c_int, _i32p, _i32p]
    L.pc_round6_probe.argtypes = [vp, _f64p, _f64p, ctypes.c_int64]
    L.pc_last_align_ms.argtypes = [vp]
    L.pc_last_align_ms.restype = ctypes.c_float
    L.pc_shard_table.argtypes = [vp, _i32p, _i64p]
    L.pc_target_costs.argtypes = [vp, _u64p]
    L.pc_last_set_kernel.argtypes = [vp]
    L.pc_multi_create.argtypes = [ctypes.POINTER(vp), _i32p, ctypes.c_int]
    L.pc_multi_destroy.argtypes = [vp]
    L.pc_multi_destroy.restype = None
    L.pc_multi_devices.argtypes = [vp]
    L.pc_multi_peer_access.argtypes = [vp, _i32p]
    L.pc_multi_upload.argtypes = [vp, ctypes.POINTER(PcPacked), ctypes.c_int]
    L.pc_multi_upload_residues.argtypes = [vp, ctypes.POINTER(PcPacked)]
    L.pc_multi_set_tie_rule.argtypes = [vp, ctypes.c_int]
    L.pc_multi_fill_borrow.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_f64p), ctypes.POINTER(PcStats),
                                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.pc_set_tie_rule.argtypes = [vp, ctypes.c_int]
    L.pc_get_tie_rule.argtypes = [vp]
    _lib = L
    return L


def _ptr(arr, typ):
    return arr.ctypes.data_as(typ)


class BorrowedArray(np.lib.mixins.NDArrayOperatorsMixin):
    """Result of ``Context.fill(borrow=True)``: a read-only window on page-locked memory the context owns.  It keeps the
    context alive (so the memory is not freed behind it by garbage collection), and once the loan has ended -- the next fill or
    upload on the context, or its ``close()`` -- every way into the data raises instead of reading recycled memory.

    It is deliberately NOT an ``ndarray`` subclass: numpy hands an ndarray subclass's buffer to ``np.asarray`` /
    ``np.ascontiguousarray`` without asking it (r03's class was one, and those two read a dead loan silently).  Being a foreign
    object, everything numpy does with it starts at ``__array__`` / ``__array_ufunc__`` / ``__array_function__``, and all three
    check the loan.  While the loan lasts, those calls see an ordinary read-only view (zero copy); ``x.copy()`` gives an
    array that owns its memory.  What cannot be guarded is a plain view or pointer taken while the loan was alive and kept."""

    __slots__ = ("_view", "_owner", "_live", "_root", "__weakref__")

    def __init__(self, view, owner, root=None):
        self._view, self._owner, self._live = view, owner, True
        self._root = self if root is None else root         # slices of a loan share its fate through the root object

    def _end_loan(self):
        self._live, self._owner = False, None

    def _check_loan(self):
        if not self._root._live:
            raise HipLibraryError("this array was lent by Context.fill(borrow=True) and the loan has ended (a later fill / "
                                  "upload / close on the context recycled its memory); copy it while it is valid")

    shape = property(lambda self: self._view.shape)
    dtype = property(lambda self: self._view.dtype)
    size = property(lambda self: self._view.size)
    ndim = property(lambda self: self._view.ndim)
    nbytes = property(lambda self: self._view.nbytes)

    def __len__(self):
        return len(self._view)

    def __getitem__(self, item):
        self._check_loan()
        got = self._view[item]
        return BorrowedArray(got, None, self._root) if isinstance(got, np.ndarray) and got.base is not None else got

    def __setitem__(self, item, value):                       # lent memory is the context's: read-only, like the view under it
        raise ValueError("assignment destination is read-only (a borrowed result of Context.fill)")

    def __iter__(self):
        self._check_loan()
        return iter(self._view)

    def __array__(self, dtype=None, copy=None):
        self._check_loan()
        if dtype is not None and np.dtype(dtype) != self._view.dtype:
            return self._view.astype(dtype)
        return self._view.copy() if copy else self._view

    def copy(self, order="C"):
        self._check_loan()
        return np.array(self._view, order=order, copy=True)

    def __getattr__(self, name):                             # .max(), .sum(), .tolist(), .astype() ...: the view's, after the check
        if name.startswith("_"):
            raise AttributeError(name)
        self._check_loan()
        return getattr(self._view, name)

    @staticmethod
    def _plain(x):
        if isinstance(x, BorrowedArray):
            x._check_loan()
            return x._view
        if isinstance(x, (list, tuple)):
            return type(x)(BorrowedArray._plain(y) for y in x)
        if isinstance(x, dict):
            return {k: BorrowedArray._plain(v) for k, v in x.items()}
        return x

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return getattr(ufunc, method)(*self._plain(inputs), **self._plain(kwargs))

    def __array_function__(self, func, types, args, kwargs):
        return func(*self._plain(args), **self._plain(kwargs))

    def __repr__(self):
        return f"BorrowedArray({'live' if self._root._live else 'loan ended'}, shape={self.shape}, dtype={self.dtype})"


class Context:
    """One GPU.  ``upload`` once, then ``fill`` any of the six metrics."""

    def __init__(self, device_id=0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        self._packed = None
        self.device_id = int(device_id)
        self._check(self._lib.pc_ctx_create(ctypes.byref(self._h), int(device_id)))

    def _check(self, rc):
        if rc != 0:
            raise HipLibraryError(f"libphamclust_hip: status {rc}: {self._lib.pc_last_error().decode()}")

    def close(self):
        self._invalidate_loans()
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pc_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ------------------------------------------------------------------
    @staticmethod
    def _struct(packed):
        return PcPacked(packed.n_genomes, packed.n_phams, packed.words_per_row, 0,
                        _ptr(packed.bitmap, _u64p), _ptr(packed.nph, _i32p), _ptr(packed.ngen, _i32p),
                        _ptr(packed.tlen, _i64p), _ptr(packed.gene_off, _i64p), _ptr(packed.gene_pham, _i32p),
                        _ptr(packed.seq_off, _i64p), _ptr(packed.residues, _u8p))

    def upload(self, packed, residues=True):
        """Genomes to HBM.  ``residues=False`` uploads only what gcs / jc / pocp / af read (pham sets, gene counts,
        translation lengths: metrics.py:26-157) -- about a tenth of the time; the residues follow by themselves the first
        time an aai / peq fill (or ``align_pairs``) asks for them, from the packed object this context keeps alive."""
        packed.validate()
        s = self._struct(packed)
        self._invalidate_loans()
        self._check((self._lib.pc_upload if residues else self._lib.pc_upload_sets)(self._h, ctypes.byref(s)))
        self._packed = packed
        self._residues = bool(residues)
        self._shard = (0, 1, False)               # pc_upload resets the shard to "everything"
        return self

    def ensure_residues(self):
        if not getattr(self, "_residues", False):
            if self._packed is None:
                raise HipLibraryError("no genomes uploaded")
            s = self._struct(self._packed)
            self._check(self._lib.pc_upload_residues(self._h, ctypes.byref(s)))
            self._residues = True

    def set_plan_budget(self, n_bytes):
        """HBM one chunk of an aai / peq fill's plan may take (0 = automatic); fills above it run in chunks, same values."""
        self._check(self._lib.pc_set_plan_budget(self._h, int(n_bytes)))

    # a borrowed result (fill(borrow=True)) is a view of pinned memory the CONTEXT owns and recycles: every array lent out
    # is remembered weakly and told when its loan ends (BorrowedArray)
    def _invalidate_loans(self):
        for ref in getattr(self, "_loans", []):
            arr = ref()
            if arr is not None:
                arr._end_loan()
        self._loans = []

    @property
    def n_genomes(self):
        return self._packed.n_genomes

    @property
    def n_pairs(self):
        return self._packed.n_pairs

    def set_shard(self, rank, world, balanced=False):
        """Static shard of the pair list; ``balanced`` deals target genomes by measured alignment work instead of
        boustrophedon-wise (same on every rank of a job)."""
        key = (int(rank), int(world), bool(balanced))
        if getattr(self, "_shard", None) != key:
            call = self._lib.pc_set_shard_balanced if balanced else self._lib.pc_set_shard
            self._check(call(self._h, int(rank), int(world)))
            self._shard = key

    def shard_table(self):
        """(t_rank[N], t_lbase[N]) of the deal in force: pair (s, t) is element t_lbase[t] + s of rank t_rank[t]'s shard."""
        t_rank = np.zeros(self.n_genomes, dtype=np.int32)
        t_lbase = np.zeros(self.n_genomes, dtype=np.int64)
        self._check(self._lib.pc_shard_table(self._h, _ptr(t_rank, _i32p), _ptr(t_lbase, _i64p)))
        return t_rank, t_lbase

    def target_costs(self):
        """DP cells behind every target genome (available once a cost-balanced deal has been computed)."""
        cost = np.zeros(self.n_genomes, dtype=np.uint64)
        self._check(self._lib.pc_target_costs(self._h, _ptr(cost, _u64p)))
        return cost

    def shard_pairs(self):
        return int(self._lib.pc_shard_pairs(self._h))

    def shard_stride(self):
        return int(self._lib.pc_shard_stride(self._h))

    # -- fills -------------------------------------------------------------------
    def fill(self, metric, as_distance=True, want_stats=False, borrow=False):
        """Whole matrix -> host condensed f64 vector (scipy order).  ``borrow=True`` returns a READ-ONLY view of page-locked
        memory the context owns (no pageable staging: the D2H copy runs at PCIe speed); the view is only good until the
        next fill or upload on this context -- copy it, or consume it at once as matrix_de_novo does."""
        stats = PcStats()
        if metric in NEEDS_RESIDUES:
            self.ensure_residues()
        self._invalidate_loans()
        if borrow:
            ptr = _f64p()
            self._check(self._lib.pc_fill_borrow(self._h, METRIC_IDS[metric], int(bool(as_distance)), ctypes.byref(ptr), ctypes.byref(stats)))
            view = np.ctypeslib.as_array(ptr, shape=(max(self.n_pairs, 1),))[:self.n_pairs]
            view.flags.writeable = False
            out = BorrowedArray(view, self)
            self._loans.append(weakref.ref(out))
            return (out, stats.as_dict()) if want_stats else out
        out = np.empty(max(self.n_pairs, 0), dtype=np.float64)
        buf = out if out.size else np.zeros(1)
        self._check(self._lib.pc_fill(self._h, METRIC_IDS[metric], int(bool(as_distance)), _ptr(buf, _f64p),
                                      ctypes.byref(stats)))
        return (out, stats.as_dict()) if want_stats else out

    def fill_dev(self, metric, as_distance, out_ptr, stream=None, want_stats=True):
        stats = PcStats()
        if metric in NEEDS_RESIDUES:
            self.ensure_residues()
        self._check(self._lib.pc_fill_dev(self._h, METRIC_IDS[metric], int(bool(as_distance)), ctypes.c_void_p(out_ptr),
                                          ctypes.c_void_p(stream or 0), ctypes.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def fill_shard_dev(self, metric, as_distance, shard_ptr, stream=None, want_stats=True):
        stats = PcStats()
        if metric in NEEDS_RESIDUES:
            self.ensure_residues()
        self._check(self._lib.pc_fill_shard_dev(self._h, METRIC_IDS[metric], int(bool(as_distance)),
                                                ctypes.c_void_p(shard_ptr), ctypes.c_void_p(stream or 0),
                                                ctypes.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def assemble_dev(self, gathered_ptr, world, out_ptr, stream=None):
        self._check(self._lib.pc_assemble_dev(self._h, ctypes.c_void_p(gathered_ptr), int(world),
                                              ctypes.c_void_p(out_ptr), ctypes.c_void_p(stream or 0)))

    # -- alignment-sliced multi-GPU route (aai / peq): plan everywhere, align a slice, sum to the root, reduce there
    def plan_dev(self, metric, stream=None):
        """Plan the whole (unsharded) aai / peq fill; stats["n_distinct_alignments"] is the length of the result array."""
        stats = PcStats()
        self.ensure_residues()
        self._check(self._lib.pc_plan_dev(self._h, METRIC_IDS[metric], ctypes.c_void_p(stream or 0), ctypes.byref(stats)))
        return stats.as_dict()

    def align_slice_dev(self, slice_rank, slice_world, res_ptr, stream=None, want_stats=True):
        """Align every slice_world-th task of each launch class from slice_rank; 8 bytes per distinct alignment at res_ptr."""
        stats = PcStats()
        self._check(self._lib.pc_align_slice_dev(self._h, int(slice_rank), int(slice_world), ctypes.c_void_p(res_ptr),
                                                 ctypes.c_void_p(stream or 0), ctypes.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def reduce_dev(self, metric, as_distance, res_ptr, out_ptr, stream=None):
        self._check(self._lib.pc_reduce_dev(self._h, METRIC_IDS[metric], int(bool(as_distance)), ctypes.c_void_p(res_ptr),
                                            ctypes.c_void_p(out_ptr), ctypes.c_void_p(stream or 0)))

    # -- test hooks --------------------------------------------------------------
    def align_pairs(self, a_gene, b_gene, variant=0):
        self.ensure_residues()
        a = np.ascontiguousarray(a_gene, dtype=np.int32)
        b = np.ascontiguousarray(b_gene, dtype=np.int32)
        n = a.shape[0]
        ident, diag = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._check(self._lib.pc_align_pairs(self._h, _ptr(a, _i32p), _ptr(b, _i32p), n, int(variant),
                                             _ptr(ident, _i32p), _ptr(diag, _i32p)))
        return ident, diag

    def set_tie_rule(self, rule):
        """Row of the aligner's tie-rule table (include/phamclust_hip.h); 0 = SURVEY 8c as recalled."""
        self._check(self._lib.pc_set_tie_rule(self._h, int(rule)))

    def tie_rule(self):
        return int(self._lib.pc_get_tie_rule(self._h))

    @staticmethod
    def variant_width(lb):
        """Columns per lane of the systolic variant chosen for a column gene of ``lb`` residues (0: general kernel)."""
        return int(load().pc_variant_width(int(lb)))

    def last_align_ms(self):
        return float(self._lib.pc_last_align_ms(self._h))

    SET_KERNELS = {0: "popc", 1: "sparse", 2: "sparse64", 3: "walker", 4: "sparsecol", -1: None}

    def last_set_kernel(self):
        """Kernel family the selector gave the last gcs / jc / pocp / af fill (the names PC_SET_KERNEL takes)."""
        return self.SET_KERNELS[int(self._lib.pc_last_set_kernel(self._h))]

    @staticmethod
    def chunk_plan(counts, max_per_chunk):
        """The chunking rule of memory-bounded fills (host arithmetic in the library; no GPU): range starts + [n]."""
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        cuts = np.zeros(counts.shape[0] + 2, dtype=np.int32)
        n = load().pc_chunk_plan(_ptr(counts, _u64p), counts.shape[0], int(max_per_chunk), _ptr(cuts, _i32p), cuts.shape[0])
        if n < 0:
            raise HipLibraryError(load().pc_last_error().decode())
        return cuts[:n + 1]

    def round6(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        out = np.empty_like(v)
        self._check(self._lib.pc_round6_probe(self._h, _ptr(v, _f64p), _ptr(out, _f64p), v.shape[0]))
        return out


class MultiContext:
    """Several GPUs of this node driven from THIS process (``pc_multi_*``: a context and a host thread per device inside the
    library, the shard exchange as device-to-device copies) -- the multi-GPU route that costs no launcher, no interpreter and no
    process group per GPU.  ``device_ids[0]`` delivers the matrix; an id may repeat (several contexts on one GPU: rehearsal).
    Same surface as :class:`Context` where it matters to ``matrix_de_novo``: ``upload`` / ``fill``."""

    def __init__(self, device_ids):
        self._lib = load()
        self._h = ctypes.c_void_p()
        self._packed = None
        self.device_ids = [int(x) for x in device_ids]
        ids = np.ascontiguousarray(self.device_ids, dtype=np.int32)
        rc = self._lib.pc_multi_create(ctypes.byref(self._h), _ptr(ids, _i32p), int(ids.shape[0]))
        if rc != 0:
            # (the other multi-GPU route starts fresh CHILD processes, one per GPU; this process is never re-executed)
            raise HipLibraryError(f"libphamclust_hip: status {rc}: {self._lib.pc_last_error().decode()} -- devices {self.device_ids} could not "
                                  f"all be opened from one process; PHAMCLUST_MULTI=launcher runs the same shard as one process per GPU "
                                  f"(torch.distributed.run, one RCCL gather)")

    PEER_ACCESS = {2: "same-device", 1: "peer", 0: "staged (no peer access)", -1: "staged (peer access failed)"}

    def peer_access(self):
        """How each device's shard reaches ``device_ids[0]``: a list of ``{"device", "access", "code"}`` plus the runtime's reasons
        for every copy that is NOT device to device (``pc_multi_peer_access``).  Never silent: ``fill`` stats and the CLI log carry it."""
        codes = np.zeros(self.n_devices, dtype=np.int32)
        staged = self._lib.pc_multi_peer_access(self._h, _ptr(codes, _i32p))
        if staged < 0:
            self._check(staged)
        note = self._lib.pc_last_error().decode().strip() if staged else ""
        return {"devices": [{"device": d, "code": int(c), "access": self.PEER_ACCESS[int(c)]} for d, c in zip(self.device_ids, codes)],
                "staged_through_host": int(staged), "note": note}

    _check = Context._check
    _struct = staticmethod(Context._struct)
    _invalidate_loans = Context._invalidate_loans

    @property
    def n_devices(self):
        return len(self.device_ids)

    @property
    def n_pairs(self):
        return self._packed.n_pairs

    def close(self):
        self._invalidate_loans()
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pc_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:                                          # noqa: BLE001 -- interpreter shutdown
            pass

    def upload(self, packed, residues=True):
        packed.validate()
        s = self._struct(packed)
        self._invalidate_loans()
        self._check(self._lib.pc_multi_upload(self._h, ctypes.byref(s), int(bool(residues))))
        self._packed, self._residues = packed, bool(residues)
        return self

    def set_tie_rule(self, rule):
        self._check(self._lib.pc_multi_set_tie_rule(self._h, int(rule)))

    def fill(self, metric, as_distance=True, want_stats=False, borrow=True):
        """The whole matrix over all devices: condensed f64 vector (a loan of the root context's pinned memory, like
        ``Context.fill(borrow=True)``; ``borrow=False`` copies it).  Stats: the root device's fill plus ``per_device`` (every
        device's own), ``ms_exchange`` (slowest device-to-root copy), ``ms_assemble``."""
        if self._packed is None:
            raise HipLibraryError("no genomes uploaded")
        if metric in NEEDS_RESIDUES and not self._residues:
            s = self._struct(self._packed)
            self._check(self._lib.pc_multi_upload_residues(self._h, ctypes.byref(s)))
            self._residues = True
        self._invalidate_loans()
        stats = (PcStats * self.n_devices)()
        ptr, x_ms, a_ms = _f64p(), ctypes.c_float(0.0), ctypes.c_float(0.0)
        self._check(self._lib.pc_multi_fill_borrow(self._h, METRIC_IDS[metric], int(bool(as_distance)), ctypes.byref(ptr), stats,
                                                   ctypes.byref(x_ms), ctypes.byref(a_ms)))
        view = np.ctypeslib.as_array(ptr, shape=(max(self.n_pairs, 1),))[:self.n_pairs]
        view.flags.writeable = False
        out = BorrowedArray(view, self)
        self._loans.append(weakref.ref(out))
        if not borrow:
            out = out.copy()
        if not want_stats:
            return out
        per = [s.as_dict() for s in stats]
        merged = dict(per[0], per_device=per, ms_exchange=float(x_ms.value), ms_assemble=float(a_ms.value), n_devices=self.n_devices,
                      ms_total=max(p["ms_total"] for p in per), peer_access=self.peer_access())
        return out, merged
