"""GPU: each hand-written device primitive against numpy on the same seeded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _keys(rng, n, nw, bits):
    """random keys of `bits` bits as [n, nw] uint64 (most significant word first)"""
    a = np.zeros((n, nw), np.uint64)
    lo_bits = min(bits, 64)
    lo = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    if lo_bits < 64:
        lo &= np.uint64((1 << lo_bits) - 1)
    a[:, nw - 1] = lo
    if nw == 2 and bits > 64:
        hi = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64)
        hi &= np.uint64((1 << (bits - 64)) - 1)
        a[:, 0] = hi
    return a


def _to_dev(a):
    return torch.from_numpy(a.view(np.int64).copy()).cuda()


def _from_dev(t, nw):
    return t.cpu().numpy().view(np.uint64).reshape(-1, nw)


def _lexsort(a):
    return np.lexsort(tuple(a[:, j] for j in range(a.shape[1] - 1, -1, -1)))


@pytest.mark.parametrize("nw,bits", [(1, 62), (1, 60), (1, 13), (2, 126), (2, 80), (2, 64)])
@pytest.mark.parametrize("n", [0, 1, 2, 4095, 4096, 4097, 100003, 1 << 20])
def test_radix_sort_keys(nw, bits, n):
    from katome_amd import device as kd
    rng = np.random.default_rng(n * 7 + bits)
    a = _keys(rng, n, nw, bits)
    if n > 1000:                       # duplicates exercise stability-independent equality
        a[n // 2:n // 2 + 500] = a[:500]
    d = _to_dev(a)
    kd.sort_keys(d, bits, nw)
    torch.cuda.synchronize()
    got = _from_dev(d, nw)
    want = a[_lexsort(a)]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nw,bits", [(1, 62), (2, 126)])
def test_radix_sort_with_values_is_stable(nw, bits):
    from katome_amd import device as kd
    n = 300001
    rng = np.random.default_rng(3)
    a = _keys(rng, n, nw, bits)
    a[:, nw - 1] &= np.uint64(0xFFFF)          # many equal keys: stability is observable through the values
    if nw == 2:
        a[:, 0] &= np.uint64(0x3)
    v = np.arange(n, dtype=np.uint32)
    d, dv = _to_dev(a), torch.from_numpy(v.view(np.int32).copy()).cuda()
    kd.sort_keys(d, bits, nw, dv)
    torch.cuda.synchronize()
    order = np.lexsort((v,) + tuple(a[:, j] for j in range(nw - 1, -1, -1)))
    assert np.array_equal(_from_dev(d, nw), a[order])
    assert np.array_equal(dv.cpu().numpy().view(np.uint32), v[order])


@pytest.mark.parametrize("nw,bits,n,dup", [(1, 62, 3_000_001, False), (1, 62, 2_200_000, True), (2, 126, 1_500_000, True), (1, 24, 1_100_000, True)])
def test_radix_sort_of_millions_is_stable(nw, bits, n, dup):
    """millions of keys (hundreds of workgroup tiles per pass, several offset chunks): sorted, and stable -- the values of
    equal keys keep their order -- for spread-out keys (top passes + run sort) and for heavy duplication"""
    from katome_amd import device as kd
    rng = np.random.default_rng(n + bits)
    a = _keys(rng, n, nw, bits)
    if dup:
        a[:, nw - 1] &= np.uint64(0xFFFFF)
        if nw == 2:
            a[:, 0] &= np.uint64(0x7)
    v = np.arange(n, dtype=np.uint32)
    d, dv = _to_dev(a), torch.from_numpy(v.view(np.int32).copy()).cuda()
    kd.sort_keys(d, bits, nw, dv)
    torch.cuda.synchronize()
    order = np.lexsort((v,) + tuple(a[:, j] for j in range(nw - 1, -1, -1)))
    assert np.array_equal(_from_dev(d, nw), a[order])
    assert np.array_equal(dv.cpu().numpy().view(np.uint32), v[order])


@pytest.mark.parametrize("nw", [1, 2])
@pytest.mark.parametrize("n", [1, 2, 2047, 2048, 2049, 500000])
def test_unique(nw, n):
    from katome_amd import device as kd
    rng = np.random.default_rng(n + nw)
    a = _keys(rng, n, nw, 62 if nw == 1 else 100)
    a[:, nw - 1] %= np.uint64(max(n // 3, 1))
    if nw == 2:
        a[:, 0] %= np.uint64(2)
    a = a[_lexsort(a)]
    d = _to_dev(a)
    out = kd.unique_sorted(d, nw)
    torch.cuda.synchronize()
    want = np.unique(a, axis=0)
    assert np.array_equal(_from_dev(out, nw), want)


@pytest.mark.parametrize("nw,bits", [(1, 60), (1, 10), (2, 124)])
def test_rank_in_sorted(nw, bits):
    from katome_amd import device as kd
    rng = np.random.default_rng(11)
    a = np.unique(_keys(rng, 200000, nw, bits), axis=0)
    q_idx = rng.integers(0, len(a), size=50000)
    d, q = _to_dev(a), _to_dev(a[q_idx])
    r = kd.rank_in_sorted(d, q, bits, nw)
    torch.cuda.synchronize()
    assert np.array_equal(r.cpu().numpy(), q_idx)
    # absent keys -> all ones
    missing = a[:10].copy()
    missing[:, nw - 1] ^= np.uint64(1)
    present = {tuple(x) for x in a.tolist()}
    r = kd.rank_in_sorted(d, _to_dev(missing), bits, nw).cpu().numpy()
    for row, rr in zip(missing.tolist(), r):
        assert (rr == -1) == (tuple(row) not in present)


@pytest.mark.parametrize("k", [31, 40])
@pytest.mark.parametrize("n_parts", [1, 2, 8])
def test_partition_by_owner(k, n_parts):
    """records grouped by owner = mulhi(mix(hash(key) ^ c), n_parts) (kmer_bits.h whole_key_owner); invalid records dropped; multiset kept"""
    import ctypes as C
    from katome_amd import device as kd
    from helpers import hostshim
    nw = kd.record_words(k)
    rng = np.random.default_rng(k + n_parts)
    n = 70001
    a = _keys(rng, n, nw, 2 * k)
    inv = rng.random(n) < 0.01
    a[inv] = np.uint64(0xFFFFFFFFFFFFFFFF)
    b = kd.Builder(k, True)
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    out, counts, vout = b.partition(_to_dev(a).view(-1), n_parts, values=vals)
    out2, counts2 = b.partition(_to_dev(a).view(-1), n_parts)
    torch.cuda.synchronize()
    assert counts2 == counts and torch.equal(out2[:sum(counts) * nw], out[:sum(counts) * nw])
    assert np.array_equal(vout.cpu().numpy()[:sum(counts)] >= 0, np.ones(sum(counts), bool))
    # values follow their records
    src_rows = a[vout.cpu().numpy()[:sum(counts)]]
    got = _from_dev(out, nw)
    assert np.array_equal(got[:sum(counts)], src_rows)
    L = hostshim()
    owners = np.array([L.hs_owner((C.c_uint64 * nw)(*[int(x) for x in row]), nw, n_parts) for row in a[~inv]])
    assert counts == [int((owners == p).sum()) for p in range(n_parts)]
    assert sum(counts) == int((~inv).sum())
    pos = 0
    valid = a[~inv]
    for p in range(n_parts):
        seg = got[pos:pos + counts[p]]
        want = valid[owners == p]
        assert np.array_equal(seg, want)           # the pass is stable
        pos += counts[p]
    b.close()


@pytest.mark.parametrize("k,n_parts,node", [(31, 8, False), (31, 3, True), (40, 8, False), (63, 5, True), (5, 2, False), (3, 4, True)])
def test_partition_by_core_owner(k, n_parts, node):
    """katome_dev_partition_core against the host statement of the same owner function (kmer_bits.h core_owner):
    k-mers by their canonical middle, (k-1)-mer nodes by their canonical tail"""
    import ctypes as C
    from katome_amd import device as kd
    from helpers import hostshim
    nw = kd.record_words(k)
    rng = np.random.default_rng(7 * k + n_parts)
    n = 50021
    a = _keys(rng, n, nw, 2 * (k - 1) if node else 2 * k)
    core = (0, k - 2) if node else (2, k - 2)
    b = kd.Builder(k, True)
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    out, counts, vout = b.partition(_to_dev(a).view(-1), n_parts, values=vals, core=core)
    torch.cuda.synchronize()
    L = hostshim()
    owners = np.array([L.hs_core_owner((C.c_uint64 * nw)(*[int(x) for x in row]), nw, core[0], core[1], n_parts) for row in a])
    assert [kd.key_owner([int(x) for x in row], n_parts, core) for row in a[:200]] == owners[:200].tolist()
    assert counts == [int((owners == p).sum()) for p in range(n_parts)] and sum(counts) == n
    got, pos = _from_dev(out, nw), 0
    for p in range(n_parts):
        assert np.array_equal(got[pos:pos + counts[p]], a[owners == p])          # stable
        pos += counts[p]
    assert np.array_equal(a[vout.cpu().numpy()], got[:n])
    b.close()


@pytest.mark.parametrize("k", [5, 31, 40])
def test_source_ids(k):
    """first half of node numbering: distinct sources of sorted edges and every edge's position among them"""
    from katome_amd import device as kd
    from helpers import words_to_int
    nw = kd.record_words(k)
    rng = np.random.default_rng(k)
    a = _keys(rng, 30000 if k > 5 else 700, nw, 2 * k)
    ints = sorted({words_to_int(r) for r in a})
    edges = np.array([[(v >> 64) & (2**64 - 1), v & (2**64 - 1)][2 - nw:] for v in ints], dtype=np.uint64).reshape(-1, nw)
    nodes, src = kd.source_ids(_to_dev(edges).view(-1), k)
    torch.cuda.synchronize()
    want_nodes = sorted({v >> 2 for v in ints})
    got_nodes = [words_to_int(r) for r in _from_dev(nodes, nw)]
    assert got_nodes == want_nodes
    pos = {v: i for i, v in enumerate(want_nodes)}
    assert src.cpu().tolist() == [pos[v >> 2] for v in ints]


def _node_ids_reference(ints, k):
    """sources ascending, then the targets that are no source, ascending (radix.hip node_ids_t)"""
    mask = (1 << (2 * (k - 1))) - 1
    srcs = sorted({v >> 2 for v in ints})
    src_set = set(srcs)
    extra = sorted({v & mask for v in ints} - src_set)
    pos = {v: i for i, v in enumerate(srcs + extra)}
    return srcs + extra, [pos[v >> 2] for v in ints], [pos[v & mask] for v in ints]


@pytest.mark.parametrize("k,n,shape", [(5, 900, "random"), (5, 1024, "all"), (9, 60000, "random"), (31, 30000, "random"),
                                        (31, 200000, "chains"), (31, 50000, "sinks"), (31, 400000, "sinks"), (32, 30000, "chains"), (40, 30000, "random"),
                                        (40, 120000, "chains"), (63, 40000, "chains"), (31, 5000, "repeats")])
def test_node_ids(k, n, shape):
    """node numbering off the sorted edges: targets merged against the sources segment by segment (four quarters by first
    base), the ones without out-edges appended -- against a dictionary on the host; with KATOME_DST_RANK unset this is the
    merging look-up, whose segments (2048 sources) and staging buffer these sizes cross"""
    from katome_amd import device as kd
    from helpers import words_to_int
    nw = kd.record_words(k)
    rng = np.random.default_rng(1000 * k + n)
    full = (1 << (2 * k)) - 1
    if shape == "all":
        ints = list(range(4 ** k))
    elif shape == "chains":          # walks: most targets are sources too, chain ends are not
        ints = set()
        while len(ints) < n:
            v = int(rng.integers(0, 2 ** 62)) * int(rng.integers(1, 2 ** 62)) & full
            for _ in range(int(rng.integers(1, 200))):
                ints.add(v)
                v = ((v << 2) | int(rng.integers(0, 4))) & full
        ints = sorted(ints)
    elif shape == "sinks":           # all sources share their first bases: nearly every target is a node without out-edges
        ints = sorted({(int(rng.integers(0, 2 ** 62)) & (full >> 24)) | (0x2A5 << (2 * k - 12)) for _ in range(n)})
    else:
        a = _keys(rng, n, nw, 2 * k)
        ints = sorted({words_to_int(r) for r in a})
    if shape == "repeats":           # BFCounter lists may hold a k-mer more than once (parallel edges)
        ints = sorted(ints + ints[::3] + ints[::7])
    edges = np.array([[(v >> 64) & (2**64 - 1), v & (2**64 - 1)][2 - nw:] for v in ints], dtype=np.uint64).reshape(-1, nw)
    nodes, src, dst = kd.node_ids(_to_dev(edges).view(-1), k)
    torch.cuda.synchronize()
    want_nodes, want_src, want_dst = _node_ids_reference(ints, k)
    assert [words_to_int(r) for r in _from_dev(nodes, nw)] == want_nodes
    assert src.cpu().tolist() == want_src
    assert dst.cpu().tolist() == want_dst


@pytest.mark.parametrize("k", [3, 31, 32, 40, 63])
def test_endpoints_and_labels(oracle, k):
    from katome_amd import device as kd
    from helpers import int_to_kmer, words_to_int
    nw = kd.record_words(k)
    rng = np.random.default_rng(k)
    n = 1000
    a = _keys(rng, n, nw, 2 * k)
    d = _to_dev(a)
    src, dst = kd.endpoints(d.view(-1), k)
    lab = kd.labels(d.view(-1), k)
    torch.cuda.synchronize()
    src, dst, lab = _from_dev(src, nw), _from_dev(dst, nw), lab.cpu().numpy()
    oracle.set_k(k)
    for i in range(0, n, 7):
        v = words_to_int(a[i])
        s = int_to_kmer(v, k)
        assert words_to_int(src[i]) == v >> 2
        assert words_to_int(dst[i]) == v & ((1 << (2 * (k - 1))) - 1)
        assert bytes(lab[i]) == oracle.compress_edge(s.encode())


@pytest.mark.parametrize("L,npct", [(150, 0), (150, 1), (101, 5), (33, 0)])
def test_synth_reads_match_cpu_definition(oracle, L, npct):
    from katome_amd import device as kd
    from helpers import pack_reads_ascii
    n, G, e, first = 3000, 100000, 1e-2, 12345
    packed, skip = kd.synth_reads(first, n, L, G, e, npct)
    torch.cuda.synchronize()
    ascii_reads = oracle.synth_reads(first, n, L, G, e, npct)
    has_n = (ascii_reads == ord("N")).any(axis=1)
    assert np.array_equal(skip.cpu().numpy()[:n].astype(bool), has_n)
    clean = ascii_reads.copy()
    stride = (L + 3) // 4
    got = packed.cpu().numpy()[:n * stride].reshape(n, stride)
    ok = ~has_n
    assert np.array_equal(got[ok], pack_reads_ascii(clean[ok]))
    if npct:
        assert 0 < has_n.sum() < n


@pytest.mark.parametrize("nw,bits", [(1, 62), (2, 126), (2, 80), (2, 100)])
@pytest.mark.parametrize("shape", ["shared_top_bits", "short_runs", "runs_across_tiles", "runs_near_the_halo", "runs_past_the_halo"])
def test_sort_by_top_bits_and_run_sort(nw, bits, shape):
    """dev_sort only runs the passes over the top 8 * ceil(log2(n) / 8) bits and lets the run sort place every record
    inside the run that shares them (radix.hip): run_sort_wave_kernel finds a run of up to 15 records by wave shuffles, walks a
    longer one in global memory and gives up past 4096 records -- the whole sort then falls back to all passes (the staged
    kernel, KATOME_RUN_SORT=1, gives up past its halo of 512 records either side); equal keys keep their input order either way"""
    from katome_amd import device as kd
    rng = np.random.default_rng(nw * 100 + bits + len(shape))
    n = 300001
    low_bits = bits - 24                      # what the run sort is left with (2^16 <= n < 2^24: 3 top passes)
    a = np.zeros((n, nw), np.uint64)
    if shape == "shared_top_bits":            # one run of n records
        top = np.full(n, 0x2AAAAA, np.uint64)
    elif shape == "short_runs":               # runs of ~3 with duplicates inside
        top = rng.integers(0, n // 3, n).astype(np.uint64)
    elif shape == "runs_near_the_halo":       # runs of 300..512: followed to their ends through the halo, across tiles
        top = np.repeat(np.arange(n // 300 + 1, dtype=np.uint64), rng.integers(300, 513, n // 300 + 1))[:n]
        rng.shuffle(top)
    elif shape == "runs_past_the_halo":       # a few runs of 513..3000 among short ones: fall back
        top = rng.integers(0, n // 3, n).astype(np.uint64)
        top[1000:1513] = 7; top[90000:93000] = 11
        rng.shuffle(top)
    else:                                     # runs of ~40: many cross the 4096-position tiles
        top = rng.integers(0, n // 40, n).astype(np.uint64)
    low = rng.integers(0, 50, n).astype(np.uint64) if shape != "shared_top_bits" else rng.integers(0, 1 << 20, n).astype(np.uint64)
    if nw == 1:
        a[:, 0] = (top << np.uint64(low_bits)) | low
    else:
        hi_bits = bits - 64
        full_top = [int(t) << low_bits for t in top.tolist()]
        a[:, 0] = np.array([(v >> 64) & ((1 << hi_bits) - 1) for v in full_top], np.uint64)
        a[:, 1] = np.array([v & (2**64 - 1) for v in full_top], np.uint64) | low
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    keys = _to_dev(a).view(-1)
    kd.sort_keys(keys, bits, nw, vals)
    torch.cuda.synchronize()
    got, gv = _from_dev(keys, nw), vals.cpu().numpy()
    order = _lexsort(a)                       # numpy's lexsort is stable: the expected permutation exactly
    assert np.array_equal(gv, order.astype(np.int32))
    assert np.array_equal(got, a[order])


@pytest.mark.parametrize("m", [0, 1, 1000, 1024, 65536, 65537, 300001, (1 << 20) + 17])
def test_scan_counts(m):
    """dev_scan_counts: one workgroup for short arrays, chunk sums + chunk scans for long ones"""
    from katome_amd import device as kd
    rng = np.random.default_rng(m)
    counts = rng.integers(0, 5000, m).astype(np.int64)
    if m > 10:
        counts[rng.integers(0, m, 5)] = 0xFFFFFFF0            # sums pass 2^32
    d = torch.from_numpy(counts).to(torch.int32 if m == 0 else torch.int64).cuda()
    d = (d & 0xFFFFFFFF).to(torch.int64)
    d32 = torch.empty(m, dtype=torch.int32, device="cuda")
    if m:
        d32.copy_(torch.where(d >= (1 << 31), d - (1 << 32), d).to(torch.int32))
    offs = kd.scan_counts(d32).cpu().numpy()
    want = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.uint64)
    assert (offs.view(np.uint64) == want).all()
