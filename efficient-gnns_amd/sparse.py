"""``torch_sparse.SparseTensor`` stand-in backed by the gfx950 kernels.

Mirrors the subset of the torch-sparse interface the reference calls
(/root/reference/arxiv_pyg/gnn.py:240-241,248; /root/reference/mag_pyg/gnn.py:13,151,162):
``SparseTensor(row=, col=, value=, sparse_sizes=)``, ``.to_symmetric()``, ``.to(device)``, ``.coo()``,
``.matmul(x, reduce=)``, ``.storage.rowptr()/col()/value()``, ``.set_value()``, ``.sparse_sizes()``, ``.t()``.

Data layout in HBM: CSR, rows sorted, columns ascending within a row; the exported index arrays are
int64 (torch-sparse convention, bit-exact vs the oracle).  For the kernels an int32 copy of
``rowptr``/``col`` is cached when nnz and N fit (halves index traffic), together with the transposed
CSR (the ``csr2csc`` route of the backward pass) and the list of "long" rows the SpMM hands to whole
workgroups.  Structure building (sort / unique) is one-off integer preprocessing done with torch ops
on whatever device the indices live on; ``matmul`` and ``gcn_norm`` require the GPU.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

import os

SHORT_ROW_MAX = 64          # rows up to this length share a wavefront
LONG_ROW_THRESHOLD = 512    # rows above it get a 16-wave workgroup
SEG_MAX = 64                # entries per range of the segment schedule
BLK_ROWS = 32               # rows per workgroup of the row-block schedule (multiple of 32; 32 / 64 / 128: 346 / 352 / 357 us, r02_spmm_lab.md)
PLAN_CHUNK = 1              # >1: length-sort short rows inside chunks of this many rows (measured slower: locality wins)


def _ind2ptr(row: Tensor, n_rows: int) -> Tensor:
    rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=row.device)
    if row.is_cuda:
        row = row.contiguous()
        _lib.check(_lib.load().egnn_rowptr_from_sorted_rows_i64(_lib.ptr(row), row.numel(), n_rows, _lib.ptr(rowptr),
                                                                _lib.stream()), "egnn_rowptr_from_sorted_rows_i64")
        return rowptr
    rowptr[0] = 0
    torch.cumsum(torch.bincount(row, minlength=n_rows), 0, out=rowptr[1:])
    return rowptr


def _ptr2ind(rowptr: Tensor, nnz: int) -> Tensor:
    n = rowptr.numel() - 1
    return torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=rowptr.device),
                                   rowptr[1:] - rowptr[:-1], output_size=nnz)


class _Storage:
    def __init__(self, owner: "SparseTensor"):
        self._o = owner

    def rowptr(self):
        return self._o._rowptr

    def row(self):
        return self._o._row()

    def col(self):
        return self._o._col

    def value(self):
        return self._o._value

    def csr2csc(self):
        return self._o._transpose_meta()[1]

    def colptr(self):
        return self._o._transpose_meta()[0]

    def rowcount(self):
        return self._o._rowptr[1:] - self._o._rowptr[:-1]


class SparseTensor:
    def __init__(self, row: Tensor | None = None, rowptr: Tensor | None = None, col: Tensor | None = None,
                 value: Tensor | None = None, sparse_sizes=None, is_sorted: bool = False):
        if col is None or (row is None and rowptr is None):
            raise ValueError("SparseTensor needs col and one of row / rowptr")
        if sparse_sizes is None or sparse_sizes[0] is None or sparse_sizes[1] is None:
            m = (int(row.max()) + 1 if row.numel() else 0) if row is not None else rowptr.numel() - 1
            n = int(col.max()) + 1 if col.numel() else 0
            sparse_sizes = (m, n)
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        # columns ascending within every row?  True on the sorting paths below, None (unknown) for caller-provided CSR
        # arrays: gcn_norm's fill kernel needs it and checks / sorts once when it is not known
        self._cols_sorted = True if (row is not None) else None
        if row is not None and not is_sorted:
            perm = torch.argsort(row * self._sizes[1] + col, stable=True)
            row, col = row[perm], col[perm]
            value = None if value is None else value[perm]
        self._rowptr = rowptr if rowptr is not None else _ind2ptr(row, self._sizes[0])
        self._row_cache = row
        self._col = col.contiguous()
        self._value = None if value is None else value.contiguous()
        self._struct = {}  # caches that depend on structure only (shared by set_value copies)
        self.storage = _Storage(self)

    # ---- accessors ---------------------------------------------------------------------------
    @property
    def device(self):
        return self._col.device

    def _row(self) -> Tensor:
        if self._row_cache is None:
            self._row_cache = _ptr2ind(self._rowptr, self._col.numel())
        return self._row_cache

    def sparse_sizes(self):
        return self._sizes

    def sparse_size(self, dim: int) -> int:
        return self._sizes[dim]

    def size(self, dim: int) -> int:
        return self._sizes[dim]

    def sizes(self):
        return list(self._sizes)

    def nnz(self) -> int:
        return self._col.numel()

    def has_value(self) -> bool:
        return self._value is not None

    def coo(self):
        return self._row(), self._col, self._value

    def csr(self):
        return self._rowptr, self._col, self._value

    def set_value(self, value, layout=None) -> "SparseTensor":
        out = SparseTensor(rowptr=self._rowptr, col=self._col, value=value, sparse_sizes=self._sizes)
        out._row_cache = self._row_cache
        out._struct = self._struct
        out._cols_sorted = self._cols_sorted
        return out

    def fill_value(self, fill: float, dtype=torch.float32) -> "SparseTensor":
        return self.set_value(torch.full((self.nnz(),), fill, dtype=dtype, device=self.device))

    def to(self, device, *_, **__) -> "SparseTensor":
        device = torch.device(device)
        if device == self.device:
            return self
        out = SparseTensor(rowptr=self._rowptr.to(device), col=self._col.to(device),
                           value=None if self._value is None else self._value.to(device), sparse_sizes=self._sizes)
        out._cols_sorted = self._cols_sorted
        return out

    def cuda(self):
        return self.to("cuda")

    def is_cuda(self) -> bool:
        return self._col.is_cuda

    # ---- structure ---------------------------------------------------------------------------
    def to_symmetric(self) -> "SparseTensor":
        """Union of (r,c) and (c,r), sorted, deduplicated (torch-sparse ``to_symmetric``, value-less)."""
        if self._value is not None:
            raise NotImplementedError("to_symmetric with values is not used by the reference")
        n = max(self._sizes)
        row, col = self._row(), self._col
        if col.is_cuda and 2 * col.numel() < 2 ** 31 - 1:
            rowptr, col_s = csr_from_coo(row, col, n, symmetric=True)
            out = SparseTensor(rowptr=rowptr, col=col_s, sparse_sizes=(n, n))
            out._cols_sorted = True
            return out
        key = torch.unique(torch.cat([row * n + col, col * n + row]))
        return SparseTensor(row=torch.div(key, n, rounding_mode="floor"), col=key % n, sparse_sizes=(n, n), is_sorted=True)

    def _transpose_meta(self):
        """(colptr, csr2csc): stable sort of the entries by column (rows stay ascending per column)."""
        if "tmeta" not in self._struct:
            if self._col.is_cuda and self.nnz() < 2 ** 31 - 1:   # egnn_csr_transpose_i64: also yields the transposed col array
                lib, dev, nnz = _lib.load(), self.device, self.nnz()
                colptr = torch.empty(self._sizes[1] + 1, dtype=torch.int64, device=dev)
                t_col = torch.empty(nnz, dtype=torch.int64, device=dev)
                perm = torch.empty(nnz, dtype=torch.int64, device=dev)
                nws = lib.egnn_csr_transpose_ws_bytes(nnz, self._sizes[1])
                ws = torch.empty(nws, dtype=torch.uint8, device=dev)
                _lib.check(lib.egnn_csr_transpose_i64(_lib.ptr(self._rowptr), _lib.ptr(self._col), self._sizes[0], self._sizes[1], nnz,
                                                      _lib.ptr(colptr), _lib.ptr(t_col), _lib.ptr(perm), _lib.ptr(ws), nws, _lib.stream()),
                           "egnn_csr_transpose_i64")
                self._struct["t_col"] = t_col
            else:
                perm = torch.argsort(self._col, stable=True)
                colptr = _ind2ptr(self._col[perm], self._sizes[1])
            self._struct["tmeta"] = (colptr, perm)
        return self._struct["tmeta"]

    def t(self) -> "SparseTensor":
        """Transposed CSR (= the CSC view torch-sparse caches as ``csr2csc``); cached."""
        if self._value is None and "t_obj" in self._struct:
            return self._struct["t_obj"]
        if getattr(self, "_t_obj", None) is not None:
            return self._t_obj
        colptr, perm = self._transpose_meta()
        if "t_col" not in self._struct:
            self._struct["t_col"] = self._row()[perm].contiguous()
        out = SparseTensor(rowptr=colptr, col=self._struct["t_col"],
                           value=None if self._value is None else self._value[perm],
                           sparse_sizes=(self._sizes[1], self._sizes[0]))
        out._struct = self._struct.setdefault("t_struct", {})
        if self._value is None:
            self._struct["t_obj"] = out
        else:
            self._t_obj = out
        return out

    # ---- kernel-side views ----------------------------------------------------------------------
    def _index_arrays(self):
        """(rowptr, col, bits): int32 narrowing when every index fits, else the int64 originals."""
        if "idx" not in self._struct:
            fits = self.nnz() < 2 ** 31 and max(self._sizes) < 2 ** 31
            if fits and self._col.is_cuda:
                lib = _lib.load()
                rp32 = torch.empty(self._rowptr.numel(), dtype=torch.int32, device=self.device)
                c32 = torch.empty(self.nnz(), dtype=torch.int32, device=self.device)
                _lib.check(lib.egnn_narrow_i64_to_i32(_lib.ptr(self._rowptr), self._rowptr.numel(), _lib.ptr(rp32), None,
                                                      _lib.stream()), "egnn_narrow_i64_to_i32")
                _lib.check(lib.egnn_narrow_i64_to_i32(_lib.ptr(self._col), self.nnz(), _lib.ptr(c32), None,
                                                      _lib.stream()), "egnn_narrow_i64_to_i32")
                self._struct["idx"] = (rp32, c32, 32)
            else:
                self._struct["idx"] = (self._rowptr, self._col, 64)
        return self._struct["idx"]

    def _row_plan(self):
        """(short_rows, mid_rows, long_rows) int64 row-id lists for egnn_spmm_csr_f32 (cached per structure).

        Short rows keep their natural order at PLAN_CHUNK granularity (index arrays stay L2-local) but are
        sorted by length inside each chunk so the 64/G rows sharing a wavefront finish together."""
        if "plan" not in self._struct:
            cnt = self._rowptr[1:] - self._rowptr[:-1]
            dev = cnt.device
            short = torch.nonzero(cnt <= SHORT_ROW_MAX).view(-1)
            mid = torch.nonzero((cnt > SHORT_ROW_MAX) & (cnt <= LONG_ROW_THRESHOLD)).view(-1).contiguous()
            long_ = torch.nonzero(cnt > LONG_ROW_THRESHOLD).view(-1).contiguous()
            ns = short.numel()
            if ns > 0:
                pad = (-ns) % PLAN_CHUNK
                ids = torch.cat([short, torch.full((pad,), -1, dtype=torch.int64, device=dev)]) if pad else short
                deg = torch.where(ids >= 0, cnt[ids.clamp(min=0)], torch.full_like(ids, -1))
                order = torch.argsort(deg.view(-1, PLAN_CHUNK), dim=1, descending=True, stable=True)
                ids = torch.gather(ids.view(-1, PLAN_CHUNK), 1, order).view(-1)
                short = ids[ids >= 0].contiguous()
            self._struct["plan"] = (short, mid, long_)
        return self._struct["plan"]

    def _seg_plan(self):
        """Segment schedule for egnn_spmm_csr_seg_f32 (cached per structure): (seg [n_seg,3], comb_rows, comb_ptr, slots).

        Rows with at most SEG_MAX entries are one direct range each, in natural row order (index arrays stay
        L2-local).  Longer rows are cut into equal ranges of at most SEG_MAX entries that write partial slots; their
        ranges come first in the list so that the heaviest work is dispatched first."""
        if "segplan" not in self._struct:
            rp = self._rowptr
            dev = rp.device
            cnt = rp[1:] - rp[:-1]
            n = cnt.numel()
            multi = torch.nonzero(cnt > SEG_MAX).view(-1)
            single = torch.nonzero(cnt <= SEG_MAX).view(-1)
            direct = torch.stack([rp[single], rp[single + 1], single], dim=1)
            if multi.numel() > 0:
                c = cnt[multi]
                nseg = (c + SEG_MAX - 1) // SEG_MAX
                cptr = torch.zeros(multi.numel() + 1, dtype=torch.int64, device=dev)
                torch.cumsum(nseg, 0, out=cptr[1:])
                slots = int(cptr[-1])
                owner = torch.repeat_interleave(torch.arange(multi.numel(), device=dev), nseg)   # which multi row a slot belongs to
                k = torch.arange(slots, device=dev) - cptr[owner]                                  # segment number inside its row
                # equal split: segment k of a row with c entries and s segments covers [k*c//s, (k+1)*c//s)
                cs, ss, base = c[owner], nseg[owner], rp[multi][owner]
                parts = torch.stack([base + (k * cs) // ss, base + ((k + 1) * cs) // ss, n + torch.arange(slots, device=dev)], dim=1)
                seg = torch.cat([parts, direct], dim=0).contiguous()
            else:
                cptr = torch.zeros(1, dtype=torch.int64, device=dev)
                slots = 0
                seg = direct.contiguous()
            self._struct["segplan"] = (seg, multi.contiguous(), cptr.contiguous(), slots)
        return self._struct["segplan"]

    def _blk_plan(self):
        """Hub part of the row-block schedule (egnn_spmm_csr_blk_f32), cached per structure:
        (hub_seg int32 [n_hseg,4] = (first entry, end entry, partial slot, 0), hub_rows int64, comb_ptr int64, slots).
        Rows with more than SEG_MAX entries are cut into equal ranges of at most SEG_MAX entries (as in _seg_plan); every
        other row is written directly by the block kernel and needs no list."""
        if "blkplan" not in self._struct:
            rp = self._rowptr
            dev = rp.device
            cnt = rp[1:] - rp[:-1]
            multi = torch.nonzero(cnt > SEG_MAX).view(-1)
            if multi.numel() > 0:
                c = cnt[multi]
                nseg = (c + SEG_MAX - 1) // SEG_MAX
                cptr = torch.zeros(multi.numel() + 1, dtype=torch.int64, device=dev)
                torch.cumsum(nseg, 0, out=cptr[1:])
                slots = int(cptr[-1])
                owner = torch.repeat_interleave(torch.arange(multi.numel(), device=dev), nseg)
                k = torch.arange(slots, device=dev) - cptr[owner]
                cs, ss, base = c[owner], nseg[owner], rp[multi][owner]
                hseg = torch.stack([base + (k * cs) // ss, base + ((k + 1) * cs) // ss, torch.arange(slots, device=dev),
                                    torch.zeros(slots, dtype=torch.int64, device=dev)], dim=1).to(torch.int32).contiguous()
            else:
                cptr = torch.zeros(1, dtype=torch.int64, device=dev)
                slots = 0
                hseg = torch.zeros(0, 4, dtype=torch.int32, device=dev)
            self._struct["blkplan"] = (hseg, multi.contiguous(), cptr.contiguous(), slots)
        return self._struct["blkplan"]

    def stage_diagonal_blocks(self, rows_per_blk: int | None = 256) -> "SparseTensor":
        """Opt-in for graphs whose node order has locality (communities stored contiguously): the row-block aggregation
        stages every block's own X rows in LDS and serves the block's intra-block entries from there (csrc/spmm_blk.hip,
        north_star's "LDS staging of neighbour feature tiles").  ``rows_per_blk``: 128 / 256 / 384 / 512; ``None`` switches it
        off.  Measured on the synthetic community graph (41 % of the entries intra-block at 512 rows) it does NOT pay on
        MI355X -- 373 vs 295 us at K = 256 (profiles/r02_spmm_lab.md) -- so nothing enables it by default."""
        if rows_per_blk is None:
            self._struct.pop("locality", None)
            return self
        if rows_per_blk % 128 != 0 or not 128 <= rows_per_blk <= 512 or self._sizes[0] != self._sizes[1]:
            raise ValueError("stage_diagonal_blocks: square adjacency and rows_per_blk in {128, 256, 384, 512}")
        rowptr, col, bits = self._index_arrays()
        if bits != 32:
            raise ValueError("stage_diagonal_blocks needs int32-addressable structure")
        win = torch.empty(self._sizes[0], 2, dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().egnn_spmm_blk_window_i32(_lib.ptr(rowptr), _lib.ptr(col), self._sizes[0], rows_per_blk, None, 0, _lib.ptr(win),
                                                        _lib.stream()), "egnn_spmm_blk_window_i32")
        self._struct["locality"] = (rows_per_blk, win)
        return self

    def permute(self, perm: Tensor) -> "SparseTensor":
        """Symmetric relabelling of a square adjacency: new node i is old node ``perm[i]`` (rows AND columns), entries re-sorted
        by (row, col).  Values travel with their entries.  The result describes the same graph: aggregating features permuted
        the same way (``x[perm]``) gives the permuted result, bit-for-bit up to the summation order inside a row."""
        n = self._sizes[0]
        if self._sizes[1] != n or perm.numel() != n:
            raise ValueError("permute: square adjacency and a permutation of its nodes")
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n, dtype=perm.dtype, device=perm.device)
        row, col = inv[self._row()], inv[self._col]
        order = torch.argsort(row * n + col, stable=True)
        out = SparseTensor(row=row[order], col=col[order], value=None if self._value is None else self._value[order],
                           sparse_sizes=(n, n), is_sorted=True)
        return out

    def _scratch(self, key, shape, dtype=torch.float32) -> Tensor:
        """Per-structure scratch buffers of the aggregation (hub partial slots, statistics partials): allocated once per
        (purpose, shape, stream) instead of on every call; calls on one stream are ordered, so reuse is safe."""
        k = ("scratch", key, tuple(shape), dtype, torch.cuda.current_stream().cuda_stream)
        buf = self._struct.get(k)
        if buf is None:
            buf = self._struct[k] = torch.empty(*shape, dtype=dtype, device=self.device)
        return buf

    def _inv_rowcount(self) -> Tensor:
        if "invcnt" not in self._struct:
            cnt = (self._rowptr[1:] - self._rowptr[:-1]).clamp(min=1)
            self._struct["invcnt"] = (1.0 / cnt.to(torch.float32)).contiguous()
        return self._struct["invcnt"]

    # ---- matmul --------------------------------------------------------------------------------
    def matmul(self, x: Tensor, reduce: str = "sum") -> Tensor:
        from . import _lib
        from .ops import spmm
        return spmm(self, _lib.real(x), reduce)

    def __matmul__(self, x: Tensor) -> Tensor:
        return self.matmul(x, "sum")

    def spmm_algorithmic_bytes(self, K: int) -> int:
        _, _, bits = self._index_arrays()
        return int(_lib.load().egnn_spmm_algorithmic_bytes(self._sizes[0], self._sizes[1], K, self.nnz(), bits,
                                                           int(self._value is not None)))


def csr_from_coo(row: Tensor, col: Tensor, n: int, symmetric: bool):
    """(rowptr, col) of the CSR sorted by (row, col) on the GPU (egnn_csr_from_coo_i64): ``symmetric`` adds the transposed
    entries and merges duplicates (to_symmetric); otherwise duplicates are kept (ToSparseTensor)."""
    _lib.require_gpu(row, col)
    lib, dev, E = _lib.load(), col.device, col.numel()
    row, col = row.contiguous(), col.contiguous()
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    col_out = torch.empty((2 * E if symmetric else E), dtype=torch.int64, device=dev)
    nnz = torch.empty(1, dtype=torch.int64, device=dev)
    nws = lib.egnn_csr_from_coo_ws_bytes(E, n, int(symmetric))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    _lib.check(lib.egnn_csr_from_coo_i64(_lib.ptr(row), _lib.ptr(col), E, n, int(symmetric), _lib.ptr(rowptr), _lib.ptr(col_out),
                                         _lib.ptr(nnz), _lib.ptr(ws), nws, _lib.stream()), "egnn_csr_from_coo_i64")
    if symmetric:
        col_out = col_out[:int(nnz)].clone()   # one host read per constructed graph
    return rowptr, col_out


def gcn_norm(adj_t: SparseTensor) -> SparseTensor:
    """A^ = D^-1/2 (A + I) D^-1/2 with ``fill_diag(1)`` (PyG ``gcn_norm``, SparseTensor branch) on the GPU."""
    if adj_t.has_value():
        raise NotImplementedError("gcn_norm on a valued adjacency is not used by the reference")
    rowptr, col, _ = adj_t.csr()
    _lib.require_gpu(rowptr, col)
    n = adj_t.sparse_size(0)
    if adj_t.sparse_size(1) != n:
        raise ValueError("gcn_norm needs a square adjacency")
    if adj_t._cols_sorted is None:
        # caller-provided CSR arrays: the fill kernel places entries by (col < row | col > row) rank and needs ascending
        # columns per row (torch-sparse's own invariant).  One check per structure (one host read), one sort if violated.
        row = adj_t._row()
        key = row * n + col
        if key.numel() > 1 and bool((key[1:] < key[:-1]).any()):
            order = torch.argsort(key, stable=True)
            adj_t = SparseTensor(rowptr=rowptr, col=col[order].contiguous(), sparse_sizes=(n, n))
            rowptr, col, _ = adj_t.csr()
        adj_t._cols_sorted = True
    lib, st = _lib.load(), _lib.stream()
    dev = col.device
    counts = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.check(lib.egnn_gcn_norm_count_i64(_lib.ptr(rowptr), _lib.ptr(col), n, _lib.ptr(counts), st), "egnn_gcn_norm_count_i64")
    rowptr_out = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr_out[1:])
    nnz_out = adj_t.nnz() + n  # upper bound; exact count below (one host read, once per run)
    nnz_out = int(rowptr_out[-1])
    col_out = torch.empty(nnz_out, dtype=torch.int64, device=dev)
    dinv = torch.empty(n, dtype=torch.float32, device=dev)
    val = torch.empty(nnz_out, dtype=torch.float32, device=dev)
    _lib.check(lib.egnn_gcn_norm_fill_i64(_lib.ptr(rowptr), _lib.ptr(col), n, _lib.ptr(rowptr_out), _lib.ptr(col_out),
                                          _lib.ptr(dinv), st), "egnn_gcn_norm_fill_i64")
    _lib.check(lib.egnn_gcn_norm_values_i64(_lib.ptr(rowptr_out), _lib.ptr(col_out), n, _lib.ptr(dinv), _lib.ptr(val), st),
               "egnn_gcn_norm_values_i64")
    out = SparseTensor(rowptr=rowptr_out, col=col_out, value=val, sparse_sizes=(n, n))
    out._cols_sorted = True
    return out


def community_order(adj_t: SparseTensor, iters: int = 20, max_community: int = 4096, hub_degree: int = 64, seed: int = 0) -> Tensor:
    """Locality-aware node order for the aggregation (``perm``: new node i is old node perm[i]): size-capped label propagation
    -- every node repeatedly adopts the most frequent label among its non-hub neighbours (ties: the smallest label; half of
    the nodes per sweep, chosen by a hash, so that the sweeps do not oscillate; a label that already has ``max_community``
    members accepts no newcomers, and neighbours with more than ``hub_degree`` entries do not vote: without these two rules
    the hubs of a power-law graph spread one label over 80 % of the nodes) -- then nodes are sorted by (label, old id).
    Graphs with community structure (citation graphs: papers cite inside their sub-field) end up with most of a row's
    sources a few thousand rows away, which is what the per-XCD L2 (32 768 lines of one column slice) can hold: on the
    synthetic community graph with shuffled ids it recovers 90 % of the locality of the true community order
    (57 % vs 62 % of the entries within +-2048 rows; 2 % before).  A graph without such structure (the Chung-Lu headline
    workload) gains nothing (profiles/r01_l2_lru_model.txt).  Integer work on whatever device the structure lives on
    (sorts / segment maxima; once per graph)."""
    n = adj_t.sparse_size(0)
    rowptr, col, _ = adj_t.csr()
    dev = col.device
    row = adj_t._row()
    labels = torch.arange(n, dtype=torch.int64, device=dev)
    if col.numel() == 0:
        return labels
    cnt_all = rowptr[1:] - rowptr[:-1]
    bound = int(cnt_all.max()) + 1
    g = torch.Generator(device="cpu").manual_seed(seed)
    phase = torch.randint(0, 2, (n,), generator=g).to(dev)
    vote = cnt_all[col] <= hub_degree
    r2, c2 = row[vote], col[vote]
    for it in range(iters):
        sizes = torch.bincount(labels, minlength=n)
        lab_nb = labels[c2]
        ok = (sizes[lab_nb] < max_community) | (lab_nb == labels[r2])
        uniq, cnt = torch.unique((r2 * n + lab_nb)[ok], return_counts=True)          # sorted by (node, label)
        node, lab = torch.div(uniq, n, rounding_mode="floor"), uniq % n
        # per node: the label with the largest count, smallest label on ties = first entry after sorting by (node, -count, label)
        order = torch.argsort((node * bound + (bound - 1 - cnt)) * n + lab)
        node_s, lab_s = node[order], lab[order]
        first = torch.ones_like(node_s, dtype=torch.bool)
        first[1:] = node_s[1:] != node_s[:-1]
        best = labels.clone()
        best[node_s[first]] = lab_s[first]
        move = (phase == (it & 1)) & (best != labels)
        labels = torch.where(move, best, labels)
    return torch.argsort(labels * n + torch.arange(n, device=dev), stable=True)
