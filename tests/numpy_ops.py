"""Test double for katome_amd.dist's `ops`: the device primitives restated with Python ints / numpy so
that the multi-rank protocol (routing, split sizes, id resolution) can run under gloo on CPU.
TEST INFRASTRUCTURE: lives in tests/, never imported by the product."""
import ctypes as C

import numpy as np
import torch

from helpers import hostshim, int_to_words, words_to_int

INVALID = (1 << 64) - 1


def _rc(v, k):
    out = 0
    for i in range(k):
        out |= (3 - ((v >> (2 * i)) & 3)) << (2 * (k - 1 - i))
    return out


class NumpyOps:
    def __init__(self, k, rc, min_weight=0):
        self.k, self.rc, self.min_weight = k, rc, min_weight
        self.nw = 1 if 2 * k <= 62 else 2
        self.table = {}
        self.tiles = {}
        self.span = 1
        self.L = hostshim()

    # -- conversions -------------------------------------------------------------------------------
    def _to_tensor(self, ints, nw=None):
        nw = nw or self.nw
        a = np.array([w for v in ints for w in (int_to_words(v, nw) if v != INVALID else [INVALID] * nw)], dtype=np.uint64)
        return torch.from_numpy(a.view(np.int64).copy())

    def _to_ints(self, t, nw=None):
        nw = nw or self.nw
        a = t.numpy().view(np.uint64).reshape(-1, nw)
        return [INVALID if int(r[0]) == INVALID else words_to_int(r) for r in a]

    def empty(self, n, dtype=torch.int64):
        return torch.empty(n, dtype=dtype)

    # -- primitives --------------------------------------------------------------------------------
    def extract_fixed(self, packed, n_reads, read_len, skip, out, first_read):
        stride = (read_len + 3) // 4
        p = packed.numpy()
        recs = []
        for r in range(first_read, first_read + n_reads):
            row = p[r * stride:(r + 1) * stride]
            bases = [(int(b) >> s) & 3 for b in row for s in (6, 4, 2, 0)][:read_len]
            for w in range(read_len - self.k + 1):
                if skip is not None and int(skip[r]):
                    recs.append(INVALID)
                    continue
                v = 0
                for c in bases[w:w + self.k]:
                    v = (v << 2) | c
                recs.append(min(v, _rc(v, self.k)) if self.rc else v)
        return self._to_tensor(recs)

    def _owner(self, v, n_parts, nw, core=None):
        if core is not None:                  # kmer_bits.h core_owner: canonical core of `bases` bases, `shift` bits up
            shift, bases = core
            m = (v >> shift) & ((1 << (2 * bases)) - 1)
            v, nw = min(m, _rc(m, bases)), nw
        return self.L.hs_owner((C.c_uint64 * nw)(*int_to_words(v, nw)), nw, n_parts)

    def partition(self, records, n_parts, key_words=None, values=None, core=None):
        nw = key_words or self.nw
        ints = self._to_ints(records, nw)
        vals = values.tolist() if values is not None else [0] * len(ints)
        parts = [[] for _ in range(n_parts)]
        for v, x in zip(ints, vals):
            if v != INVALID:
                parts[self._owner(v, n_parts, nw, core)].append((v, x))
        flat = [v for p in parts for v, _ in p]
        counts = [len(p) for p in parts]
        out = self._to_tensor(flat, nw)
        if values is None:
            return out, counts
        return out, counts, torch.tensor([x for p in parts for _, x in p], dtype=values.dtype)

    def insert(self, records, weights=None):
        ints = self._to_ints(records)
        ws = [1] * len(ints) if weights is None else [int(np.uint32(w)) for w in weights.tolist()]
        for v, w in zip(ints, ws):
            if v != INVALID:
                self.table[v] = (self.table.get(v, 0) + w) & 0xFFFFFFFF

    # -- tiled counting ---------------------------------------------------------------------------
    def tile_span(self, read_len):
        from katome_amd import _lib
        return _lib.lib().katome_tile_span(self.k, read_len)          # pure host function of the library

    def tile_words(self, span):
        bits = 2 * (self.k + span - 1)
        return 1 if bits <= 62 else 2 if bits <= 126 else 3

    def tile_plan(self, read_len):
        import ctypes as C_
        from katome_amd import _lib
        sp, t, r = C_.c_uint32(), C_.c_uint32(), C_.c_uint32()
        _lib.lib().katome_tile_plan(self.k, read_len, C_.byref(sp), C_.byref(t), C_.byref(r))     # pure host function
        return sp.value, t.value, r.value

    def extract_remainder(self, packed, n_reads, read_len, span, skip, first_read):
        stride = (read_len + 3) // 4
        W = read_len - self.k + 1
        p = packed.numpy()
        recs = []
        for r in range(first_read, first_read + n_reads):
            row = p[r * stride:(r + 1) * stride]
            bases = [(int(b) >> s) & 3 for b in row for s in (6, 4, 2, 0)][:read_len]
            for w in range((W // span) * span, W):
                if skip is not None and int(skip[r]):
                    recs.append(INVALID)
                    continue
                v = 0
                for c in bases[w:w + self.k]:
                    v = (v << 2) | c
                recs.append(min(v, _rc(v, self.k)) if self.rc else v)
        return self._to_tensor(recs)

    def extract_tiles(self, packed, n_reads, read_len, span, skip, out, first_read):
        stride = (read_len + 3) // 4
        kk = self.k + span - 1
        p = packed.numpy()
        recs = []
        for r in range(first_read, first_read + n_reads):
            row = p[r * stride:(r + 1) * stride]
            bases = [(int(b) >> s) & 3 for b in row for s in (6, 4, 2, 0)][:read_len]
            for t in range((read_len - self.k + 1) // span):
                if skip is not None and int(skip[r]):
                    recs.append(INVALID)
                    continue
                v = 0
                for c in bases[t * span:t * span + kk]:
                    v = (v << 2) | c
                recs.append(min(v, _rc(v, kk)) if self.rc else v)
        return self._to_tensor(recs, self.tile_words(span))

    def insert_tiles(self, records, span):
        self.span = span
        for v in self._to_ints(records, self.tile_words(span)):
            if v != INVALID:
                self.tiles[v] = (self.tiles.get(v, 0) + 1) & 0xFFFFFFFF

    def expand_tiles(self):
        keys, ws = [], []
        mask = (1 << (2 * self.k)) - 1
        for t, n in self.tiles.items():
            for o in range(self.span):
                x = (t >> (2 * (self.span - 1 - o))) & mask
                keys.append(min(x, _rc(x, self.k)) if self.rc else x)
                ws.append(n)
        self.tiles = {}
        return self._to_tensor(keys), torch.tensor(ws, dtype=torch.int64).to(torch.int32)

    def edges(self):
        out = {}
        for v, c in self.table.items():
            if self.rc:
                r = _rc(v, self.k)
                if r == v:
                    out[v] = (2 * c) & 0xFFFFFFFF
                else:
                    out[v] = c
                    out[r] = c
            else:
                out[v] = c
        keys = sorted(v for v in out if out[v] >= self.min_weight)      # Clean::remove_weak_edges (pruner.rs:84-93)
        return (self._to_tensor(keys).reshape(-1, self.nw),
                torch.tensor([out[v] for v in keys], dtype=torch.int64).to(torch.int32))

    def source_ids(self, keys):
        ints = self._to_ints(keys.reshape(-1))
        srcs = sorted({v >> 2 for v in ints})
        pos = {v: i for i, v in enumerate(srcs)}
        return self._to_tensor(srcs), torch.tensor([pos[v >> 2] for v in ints], dtype=torch.int64)

    def target_keys(self, keys):
        mask = (1 << (2 * (self.k - 1))) - 1
        return self._to_tensor([v & mask for v in self._to_ints(keys.reshape(-1))])

    def node_ids(self, keys):
        ints = self._to_ints(keys.reshape(-1))
        mask = (1 << (2 * (self.k - 1))) - 1
        srcs = sorted({v >> 2 for v in ints})
        extra = sorted({v & mask for v in ints} - set(srcs))
        nodes = srcs + extra
        pos = {v: i for i, v in enumerate(nodes)}
        return (self._to_tensor(nodes), torch.tensor([pos[v >> 2] for v in ints], dtype=torch.int64),
                torch.tensor([pos[v & mask] for v in ints], dtype=torch.int64))

    def sort_unique(self, keys, bits):
        return self._to_tensor(sorted(set(self._to_ints(keys))))

    def rank(self, sorted_keys, queries, bits):
        pos = {v: i for i, v in enumerate(self._to_ints(sorted_keys))}
        return torch.tensor([pos.get(v, -1) for v in self._to_ints(queries)], dtype=torch.int64)

    def labels(self, keys):
        stride = 1 + (self.k + 3) // 4
        rows = []
        for v in self._to_ints(keys.reshape(-1)):
            buf = (C.c_uint8 * stride)()
            self.L.hs_label((C.c_uint64 * self.nw)(*int_to_words(v, self.nw)), self.k, buf)
            rows.append(bytes(buf))
        return torch.from_numpy(np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(-1, stride).copy())

    def close(self):
        pass
