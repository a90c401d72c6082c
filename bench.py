#!/usr/bin/env python3
"""bench.py -- katome `build` stage on MI355X: k-mers/s (whole job) + distinct-edges/s.

A "step" is one full build of the workload with the packed reads already resident in HBM (set-up before the W warm-up
steps: the reads are synthesised and a few builds prime the library's device-memory cache):
k-mer extraction -> k-mer table -> sorted distinct edges -> node numbering, endpoints, labels
(everything `Build::create` + the PtGraph::create post-pass do in the reference), result left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5] [--reads R]

N>1: one rank per GPU.  Started by a launcher (torch.distributed.run: WORLD_SIZE set) or, without one, by this very
script (katome_amd/launch.py: the parent starts the N ranks before anything touches the GPU and relays rank 0's line).
Reads shard by index; records are routed to their owner ranks by RCCL all-to-alls issued inside libkatome_gpu.so
(katome_amd/csrc/dist.hip, comm.cpp); torch.distributed only hands round the communicator id and brackets the timed region.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
# what the card reaches for the partition pass's MEMORY PATTERN with the ranking work taken out (4096-record tiles written as 128-byte
# stretches into 256 streams; tools/microbench_scatter.hip, profiles/r04_scatter_ceiling.txt) and for a plain copy of the same bytes
SCATTER_PATTERN_CEILING_GBS = 4870.0
COPY_CEILING_GBS = 5620.0
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_latest.json")   # written from tools/profile_round.sh output


def source_id():
    """identifies the kernels the library was built from: sha256 over the HIP/C++ sources and headers.  Stored in
    profiles/pmc_latest.json when the PMC passes are taken; traffic from passes taken on other sources is refused"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "katome_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(parts, config):
    """HBM bytes per launch of a phase from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, KiB per dispatch) -- only when they were taken on this very configuration.
    `parts` = [(kernel name, dispatches per phase launch, streaming)].  gfx950 correction from the MI355X guide:
    FETCH_SIZE reads half the bytes of a wide coalesced stream, so it is doubled for streaming kernels;
    random-access kernels are left uncorrected (uncalibrated there)."""
    try:
        pmc = json.load(open(PMC_FILE))
    except Exception:
        return None
    if any(pmc.get("config", {}).get(k) != config.get(k) for k in ("reads", "read_len", "k", "batch_reads", "tile_span")):
        return None
    if pmc.get("source_id") != source_id():          # counters of other kernels than the ones that run now: stale
        return None
    total, detail = 0.0, []
    for kernel_name, mult, streaming in parts:
        v = pmc.get("kernels", {}).get(kernel_name)
        if not v:
            return None
        total += mult * (v["fetch_kib"] * (2 if streaming else 1) + v["write_kib"]) * 1024.0
        detail.append({"kernel": kernel_name, "dispatches": mult, "fetch_kib": v["fetch_kib"], "write_kib": v["write_kib"],
                       "fetch_x2": bool(streaming)})
    return {"bytes_per_launch": total, "parts": detail, "source": pmc.get("source")}


# reads per extraction + insertion round (one launch each): 16 Mi reads make the extraction launch ~0.4 ms -- at 4 Mi
# (0.1 ms) its start and tail cost a tenth of the rate; the records of a round take 1 GiB
DEFAULT_BATCH_READS = 16 * 1024 * 1024


def lc_phases(reset):
    """experiment builds of the library (-DKATOME_LC_PHASES, KATOME_LIB=...) add up the shader clocks the LDS counting kernels spend per
    phase; the shipped library does not export this.  reset=None: is it there?  True: zero it.  False: per-phase shares"""
    import ctypes as C
    from katome_amd._lib import lib
    try:
        f = lib().katome_debug_lc_phases
    except AttributeError:
        return None
    if reset is None:
        return True
    out = (C.c_uint64 * 16)()
    f.argtypes = [C.POINTER(C.c_uint64)]
    f(out)
    if reset:
        return None
    names = ["clear", "insert", "read_out_scan", "write"]
    res = {}
    for base, kernel in ((0, "lds_count_kernel"), (4, "lds_count_wide_kernel"), (8, "lds_count_packed_kernel")):
        tot = sum(out[base:base + 4]) or 1
        res[kernel] = {n: round(out[base + i] / tot, 3) for i, n in enumerate(names)}
        res[kernel]["clocks"] = int(tot)
    return res


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--reads", type=int, default=0, help="override the workload's read count (same coverage)")
    ap.add_argument("--genome-len", type=int, default=0,
                    help="with --reads: keep this genome length instead of scaling it with the reads (a rank's share of a bigger job)")
    ap.add_argument("--batch-reads", type=int, default=DEFAULT_BATCH_READS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-c2", action="store_true", help="skip the second CPU line: BASELINE configs[1] (1 M reads) in full on one core and on the GPU")
    ap.add_argument("--first-seen-order", action="store_true",
                    help="number edges and nodes in the reference's first-seen (petgraph) order (single GPU)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU driver even with one rank (testing)")
    ap.add_argument("--min-weight", type=int, default=0,
                    help="Clean::remove_weak_edges(threshold) as the edges are read out (pruner.rs:84-93; not the BASELINE metric's configuration)")
    ap.add_argument("--table-factor", type=float, default=1.8, help="k-mer table slots per expected distinct canonical k-mer")
    ap.add_argument("--cpu-sample-reads", type=int, default=600_000,
                    help="reads of the workload the oracle builds on one host core (600 k of C3: ~55 s, 122 M edges; 1e6 = all of C2, ~85 s)")
    ap.add_argument("--prune", action="store_true",
                    help="BASELINE config 5: Prunable::remove_dead_paths after the build (reference order; N > 1: on the gathered graph)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline: skip the reference-order build, the unhinted build and the end-to-end region")
    ap.add_argument("--end-to-end-reads", type=int, default=20_000_000,
                    help="SURVEY 8(d) region (ii): host packed reads -> host arrays through katome_build_packed, on this many reads")
    ap.add_argument("--next-stages-reads", type=int, default=20_000_000,
                    help="also time first-seen-order build + remove_dead_paths + shrink on this many reads (0 = skip; N = 1 only)")
    return ap.parse_args()


class PhaseTimer:
    """accumulates the library's per-phase HIP-event timings (events are recorded by the library on
    the stream its kernels are launched on; katome_builder_profile_read)"""

    def __init__(self):
        self.acc = {}
        self.counts = {}

    def add(self, prof):
        for name, (ms, launches, work) in prof.items():
            a = self.acc.setdefault(name, [0.0, 0, 0])
            a[0] += ms
            a[1] += launches
            a[2] += work

    def collect(self):
        out = {n: {"launches": c, "total_ms": ms, "avg_ms": ms / c, "work": w} for n, (ms, c, w) in self.acc.items() if c}
        self.acc = {}
        return out


def one_build_single(wl, packed, skip, recbuf, batch_reads, timer, first_seen=False, min_weight=0, table_factor=2.2, prune=False):
    """one step on one GPU; returns (n_edges, n_nodes).  table_factor 0 = no hint: the tables start small and grow by
    re-hashing, as for a caller that knows nothing about its input (settings.table_slots_hint = 0)"""
    from katome_amd import device as kd
    hint = int(wl.expected_distinct_canonical() * table_factor)
    b = kd.Builder(wl.k, wl.reverse_complement, device=packed.device.index, table_slots_hint=hint,
                   first_seen_order=first_seen or prune)
    b.profile(True)
    if min_weight:
        b.remove_weak_edges(min_weight)
    span, tiles, rest = b.tile_plan(wl.read_len)
    try:
        for r0 in range(0, wl.reads, batch_reads):
            nr = min(batch_reads, wl.reads - r0)
            if span > 1:      # tiled counting: tiles of `span` windows, expanded before the edges are read out (+ left-over windows)
                # (extraction + the tile level's insertion in one call: the records are made where the builder keeps them;
                # KATOME_BENCH_TWO_CALLS=1: extract into a buffer of the caller's, then insert -- the earlier boundary)
                if os.environ.get("KATOME_BENCH_TWO_CALLS") == "1":
                    b.insert_tiles(b.extract_tiles(packed, nr, wl.read_len, span, skip, out=recbuf, first_read=r0), span)
                else:
                    b.count_tiles(packed, nr, wl.read_len, span, skip, first_read=r0)
                if rest:
                    b.insert(b.extract_remainder(packed, nr, wl.read_len, span, skip, out=recbuf, first_read=r0))
            else:
                rec = b.extract_fixed(packed, nr, wl.read_len, skip, out=recbuf, first_read=r0)
                b.insert(rec)
        dg = b.finalize()
        n_edges, n_nodes = dg.n_edges, dg.n_nodes
        if prune:                                   # BASELINE config 5's pruner pass (pruner.rs:36-82)
            dg, _ = b.remove_dead_paths()
            n_edges, n_nodes = dg.n_edges, dg.n_nodes
        timer.add(b.profile_read())
        timer.counts = b.counts()
        return n_edges, n_nodes
    finally:
        b.close()


def timed(step, steps, torch):
    """ms per step of `steps` steps, after set-up builds until the library's memory cache has settled"""
    for _ in range(3):
        free_before = torch.cuda.mem_get_info()[0]
        step()
        if torch.cuda.mem_get_info()[0] + (64 << 20) >= free_before:
            break
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps, out


def end_to_end(wl, reads):
    """SURVEY 8(d) region (ii): packed reads in HOST memory -> katome_build_packed (H2D, build, D2H) -> host arrays; the
    PCIe- and copy-inclusive figure a caller of the host ABI sees -- never `value`"""
    import numpy as np
    from katome_amd import device as kd
    from katome_amd.build import GpuGraph
    w = wl.scaled(min(reads, wl.reads))
    packed, skip = kd.synth_reads(0, w.reads, w.read_len, w.genome_len, w.err_rate, w.n_inject_percent, device=0)
    h_packed = packed[:w.reads * w.stride].cpu().numpy()
    h_skip = skip[:w.reads].cpu().numpy() if w.n_inject_percent else None
    del packed, skip
    hint = int(w.expected_distinct_canonical() * 2.2)
    best = None
    for _ in range(2):                              # (the first call also pays for the host pages of the result)
        t0 = time.perf_counter()
        g, rb = GpuGraph.create_from_packed(h_packed, w.reads, w.read_len, skip=h_skip, reverse_complement=w.reverse_complement,
                                            k=w.k, table_slots_hint=hint)
        dt = time.perf_counter() - t0
        d2h = g.n_edges * (8 + 8 + 4 + g.label_stride + 8 * g.key_words) + g.n_nodes * 8 * g.key_words
        best = {"reads": w.reads, "ms": dt * 1e3, "kmers_per_s": (rb // w.read_len) * w.windows_per_read / dt,
                "h2d_bytes": int(h_packed.nbytes), "d2h_bytes": int(d2h), "distinct_edges": int(g.n_edges),
                "what": "host packed reads -> katome_build_packed -> host arrays (H2D + build + D2H), second of two calls"}
        del g
    return best


def next_stages(wl, sample_reads):
    """Not part of the metric: the stages SURVEY 8(f) lists after the path, timed on a bounded prefix of the workload --
    build in the reference's numbering, remove_dead_paths (pruner.rs:36-82), shrink (shrinker.rs:165-209; of the pruned
    graph, which it leaves untouched), then the remaining stages of assemble_with_graph up to collapse"""
    import torch
    from katome_amd import device as kd
    w = wl.scaled(min(sample_reads, wl.reads))
    packed, skip = kd.synth_reads(0, w.reads, w.read_len, w.genome_len, w.err_rate, w.n_inject_percent, device=0)
    skip_arg = skip if w.n_inject_percent else None
    b = kd.Builder(w.k, w.reverse_complement, table_slots_hint=int(w.expected_distinct_canonical() * 2.2), first_seen_order=True)
    out = {"reads": w.reads}
    try:
        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3, r

        def build():
            step = 4 << 20
            for r0 in range(0, w.reads, step):
                b.count_reads(packed, min(step, w.reads - r0), w.read_len, skip_arg, first_read=r0)
            return b.finalize()
        ms, dg = timed(build)
        out["build_first_seen_order_ms"] = ms
        out["edges"], out["nodes"] = dg.n_edges, dg.n_nodes
        ms, (dg, st) = timed(b.remove_dead_paths)
        out["remove_dead_paths_ms"] = ms
        out["remove_dead_paths"] = {k: st[k] for k in ("passes", "removed_edges", "removed_nodes", "host_ms")}
        out["edges_after_pruning"] = dg.n_edges
        # shrink in its two forms: all on the device, traversal-free (this library's numbering), and the reference's own cuts and
        # numbering index for index (its traversal order is sequential: one host core, `host_ms` of the total)
        ms, dc = timed(lambda: b.shrink("fast"))
        out["shrink_ms"] = ms
        out["edges_after_shrink"] = dc.n_edges
        del dc
        ms, dc = timed(lambda: b.shrink("exact"))
        out["shrink_exact_ms"] = ms
        out["shrink_exact_host_ms"] = b.last_shrink_host_ms
        out["edges_after_shrink_exact"] = dc.n_edges
        # the rest of assemble_with_graph up to collapse (asm/basic_assembler.rs:63-72), threshold 2
        ms1, _ = timed(b.standardize_contigs)
        ms2, _ = timed(lambda: b.remove_weak_edges(2))
        ms3, _ = timed(b.standardize_contigs)
        ms4, _ = timed(lambda: b.standardize_edges(w.genome_len, 2))
        ms5, (dg, st2) = timed(b.remove_dead_paths)
        out["up_to_collapse_ms"] = {"standardize_contigs": ms1, "remove_weak_edges": ms2, "standardize_contigs_2": ms3,
                                    "standardize_edges": ms4, "remove_dead_paths_2": ms5}
        out["edges_before_collapse"] = dg.n_edges
    finally:
        b.close()
    return out


def _sort_passes(key_bits, n):
    """(radix passes, run-sort passes) dev_sort runs for n keys of key_bits bits (radix.hip sort_t)"""
    all_passes = (key_bits + 7) // 8
    top = 1
    while top < 8 and (int(n) >> (8 * top)):
        top += 1
    if top + 2 <= all_passes and n >= (1 << 16):
        return top, 1
    return all_passes, 0


def cpu_baseline(wl, sample_reads):
    """the oracle (C restatement of katome's PtGraph::create) on 1 host core, on a bounded sample"""
    from oracle import oracle as o
    n = min(sample_reads, wl.reads)
    reads = o.synth_reads(0, n, wl.read_len, wl.genome_len, wl.err_rate, wl.n_inject_percent)

    def status_kib(field):
        try:
            for line in open("/proc/self/status"):
                if line.startswith(field + ":"):
                    return int(line.split()[1])
        except OSError:
            pass
        return None
    try:                                   # restart the process's resident-set high-water mark (Linux: clear_refs 5)
        open("/proc/self/clear_refs", "w").write("5")
    except OSError:
        pass
    rss0 = status_kib("VmRSS")
    t0 = time.perf_counter()
    g = o.build_ascii(reads, wl.k, wl.reverse_complement)
    dt = time.perf_counter() - t0
    hwm = status_kib("VmHWM")
    accepted = g.read_bytes // wl.read_len
    return {"value": accepted * wl.windows_per_read / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of workload %s (k=%d, rc=%s): %.1f s on one core of %d; %d distinct edges"
                      % (n, wl.name, wl.k, wl.reverse_complement, dt, os.cpu_count() or 0, g.n_edges),
            "distinct_edges_per_s": g.n_edges / dt,
            # the graph the CPU build holds at its peak (resident-set high-water mark minus what the process held before)
            "peak_rss_mib": round((hwm - rss0) / 1024.0, 1) if hwm is not None and rss0 is not None else None}


def c2_in_full(torch):
    """BASELINE.json configs[1] as a whole -- 1 M synthetic 150-bp reads, k = 31, both strands -- on one host core (the oracle) and on the
    GPU (the same reads, generated on the device by the same generator), with the parity the config asks for: the same node and edge
    counts, the same sum of weights, the same multiset of (label, weight)"""
    import numpy as np
    from katome_amd import device as kd
    from katome_amd.workloads import WORKLOADS
    from oracle import oracle as o
    wl = WORKLOADS["c2"]
    reads = o.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, wl.n_inject_percent)
    t0 = time.perf_counter()
    g = o.build_ascii(reads, wl.k, wl.reverse_complement)
    cpu_s = time.perf_counter() - t0
    accepted = g.read_bytes // wl.read_len
    packed, skip = kd.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, wl.n_inject_percent)

    def build():
        b = kd.Builder(wl.k, wl.reverse_complement, device=packed.device.index)
        b.count_reads(packed, wl.reads, wl.read_len, skip, first_read=0)
        return b, b.finalize()
    b, dg = build()
    lab = dg.edge_label.cpu().numpy().reshape(dg.n_edges, -1)
    wts = dg.edge_weight.cpu().numpy().view(np.uint32)
    same = (dg.n_nodes, dg.n_edges) == (g.n_nodes, g.n_edges) and int(wts.astype(np.uint64).sum()) == int(g.edge_weight.astype(np.uint64).sum())
    if same:          # the multiset, as rows of (label bytes, weight) in sorted order
        mine = np.concatenate([lab, wts.view(np.uint8).reshape(-1, 4)], axis=1)
        ref = np.concatenate([g.edge_label.reshape(g.n_edges, -1), g.edge_weight.astype(np.uint32).view(np.uint8).reshape(-1, 4)], axis=1)
        order = lambda a: a[np.lexsort(a.T[::-1])]
        same = bool(np.array_equal(order(mine), order(ref)))
    b.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b, dg = build()
        b.close()
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) * 1e3 / 5
    kmers = accepted * wl.windows_per_read
    return {"what": "BASELINE configs[1] in full: %d reads of %d bp, k=%d, rc=%s" % (wl.reads, wl.read_len, wl.k, wl.reverse_complement),
            "cpu_s": cpu_s, "cpu_kmers_per_s": kmers / cpu_s, "cores": 1, "kind": "port", "gpu_ms": gpu_ms, "gpu_kmers_per_s": kmers / (gpu_ms * 1e-3),
            "edges": int(g.n_edges), "nodes": int(g.n_nodes), "same_multiset": same}


def run_as_parent(args):
    """`--gpus N` (N > 1) without a launcher: this process starts the N ranks itself -- BEFORE anything touches the GPU
    (no torch import here), so no process that has initialised HIP is ever re-executed -- relays rank 0's one JSON line
    and fails if a rank fails or if the line is not an N-GPU line."""
    from katome_amd.launch import launch_ranks, relay_one_json_line
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    rc, out = launch_ranks(args.gpus, argv)
    if rc != 0:
        raise SystemExit("bench.py --gpus %d: a rank failed (exit code %d); no result" % (args.gpus, rc))
    line = relay_one_json_line(out)
    if line is None:
        raise SystemExit("bench.py --gpus %d: rank 0 printed no JSON line" % args.gpus)
    if json.loads(line).get("n_gpus") != args.gpus:
        raise SystemExit("bench.py --gpus %d: the ranks report n_gpus=%r" % (args.gpus, json.loads(line).get("n_gpus")))
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return run_as_parent(args)
    # libraries (RCCL) print banners on fd 1; the contract is ONE JSON line on stdout, so the real stdout is
    # set aside for that line and fd 1 points at stderr for everything else
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from katome_amd.workloads import WORKLOADS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher must start exactly --gpus ranks" % (args.gpus, world))
    n_visible = torch.cuda.device_count()            # (counting devices does not initialise HIP)
    if n_visible < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        raise SystemExit("rank %d: --gpus %d but only %d GPU(s) visible on this node" % (rank, args.gpus, n_visible))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the build has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if world > 1:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29531"
        # torch.distributed: rendezvous, the barrier and the MAX over ranks around the timed region; the build's own
        # exchanges are RCCL calls inside libkatome_gpu.so (katome_amd/csrc/comm.cpp)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    if args.reads:
        wl = wl.scaled(args.reads)
        if args.genome_len:
            import dataclasses
            wl = dataclasses.replace(wl, genome_len=args.genome_len, name="%s/genome %d" % (wl.name, args.genome_len))

    from katome_amd import device as kd
    timer = PhaseTimer()
    W = wl.windows_per_read
    batch_reads = max(64, (args.batch_reads // 64) * 64)

    if not use_dist:
        packed, skip = kd.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, wl.n_inject_percent,
                                      device=local_rank)
        skip_arg = skip if wl.n_inject_percent else None
        accepted = wl.reads - (int(skip[:wl.reads].sum().item()) if wl.n_inject_percent else 0)
        # (a record buffer of the caller's is only needed for windows counted one by one: reads that are not whole tiles, the
        # two-call boundary, or no tiling at all -- 16 GB that a tiled build does not take from the card)
        from katome_amd._lib import lib as _kl
        import ctypes as _Ct
        _sp, _tl, _rs = _Ct.c_uint32(), _Ct.c_uint32(), _Ct.c_uint32()
        _kl().katome_tile_plan(wl.k, wl.read_len, _Ct.byref(_sp), _Ct.byref(_tl), _Ct.byref(_rs))
        per_read_buf = W if (_sp.value <= 1 or os.environ.get("KATOME_BENCH_TWO_CALLS") == "1") else _rs.value
        recbuf = torch.empty(max(min(batch_reads, wl.reads) * per_read_buf * kd.record_words(wl.k), 1), dtype=torch.int64,
                             device=packed.device)

        def step():
            return one_build_single(wl, packed, skip_arg, recbuf, batch_reads, timer, args.first_seen_order, args.min_weight,
                                    args.table_factor, args.prune)
    else:
        from katome_amd import shard as ks
        comm = ks.Comm.rccl(rank, world, local_rank)
        job = ks.DistBuild(wl, comm, batch_reads=0 if args.batch_reads == DEFAULT_BATCH_READS else batch_reads, timer=timer,
                           first_seen_order=args.first_seen_order or args.prune, min_weight=args.min_weight,
                           table_factor=args.table_factor, prune=args.prune)
        accepted = job.accepted_total

        def step():
            return job.build()

    # Set-up, like synthesising the reads: builds until the library's device-memory cache holds the working set.  The first
    # builds of a process ask the driver for memory (the later ones still re-cut what the first left; it settles within
    # three), and hipMalloc costs anything from nothing to seconds depending on what the device ran before (DESIGN.md
    # section 3); none of that is the build.  Then the W warm-up steps and the K timed steps of the contract, whole builds all.
    for _ in range(4):
        free_before = torch.cuda.mem_get_info()[0]
        step()
        settled = torch.cuda.mem_get_info()[0] + (64 << 20) >= free_before
        if world > 1:                     # (every rank takes part in every build: agree)
            flag = torch.tensor([1 if settled else 0], dtype=torch.int64, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            settled = bool(flag.item())
        if settled:
            break
    for _ in range(args.warmup):
        step()
    timer.collect()
    if use_dist:
        job.exchange = {}                  # (exchange accounting restarts with the timed steps, like the phase timers)
    if os.environ.get("KATOME_TRACE_ALLOC"):
        print("[bench] warm-up done", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lc_phases(True)
    t0 = time.perf_counter()
    n_edges = n_nodes = 0
    for _ in range(args.steps):
        n_edges, n_nodes = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    phases = timer.collect()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())             # (the ranks already report whole-graph edge and node counts)

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        kmers = accepted * W
        nw = kd.record_words(wl.k)
        # algorithmic bytes: extraction = ceil(L/4) B read + 8*NW*W B written per read;
        # insertion = 8*NW B record + 16*NW B slot per insertion
        from katome_amd._lib import lib as _katome_lib
        import ctypes as _C
        _sp, _t, _r = _C.c_uint32(), _C.c_uint32(), _C.c_uint32()          # both routes follow the library's plan
        _katome_lib().katome_tile_plan(wl.k, wl.read_len, _C.byref(_sp), _C.byref(_t), _C.byref(_r))
        span, tiles, rest = _sp.value, _t.value, _r.value
        nwt = _katome_lib().katome_tile_words(wl.k, span)
        # extraction writes one record per tile of `span` windows (span = 1: one per window);
        # an insertion moves a record (8*NW B) and touches a slot (16*NW B)
        cnt = timer.counts or {}
        steps = args.steps
        sort_passes = [((2 * wl.k + 7) // 8, 0)]
        per_read_extract = (wl.stride + 8 * nwt * tiles + ((wl.stride + 8 * nw * rest) if rest else 0)) if span > 1 else (wl.stride + 8 * nw * W)
        alg = {"extract": lambda launches, reads: reads * per_read_extract,
               "insert": lambda launches, reads: reads * (rest if span > 1 else W) * (8 * nw + 16 * nw),
               "insert_tiles": lambda launches, reads: reads * tiles * (8 * nwt + 16 * nwt)}
        sorted_last_level = False
        if not use_dist and cnt:
            # expansion: one scan of the tile table + a 16*NW-byte slot touch per (distinct tile, k-mer) pair;
            # edge sort: ceil(2k/8) passes, each reading and writing every (key, weight) pair once
            ms2 = cnt.get("mid_span", 0)
            if ms2:      # two levels: big tiles -> mid tiles (128-bit upserts) -> k-mers
                nwm = _katome_lib().katome_tile_words(wl.k, ms2)
                alg["expand_mid_tiles"] = lambda launches, reads: steps * (
                    cnt["tile_slots"] * 16 * nwt + cnt["distinct_tiles"] * (span // ms2) * 16 * nwm)
                alg["expand_tiles"] = lambda launches, reads: steps * (
                    cnt["mid_tile_slots"] * 16 * nwm + cnt["distinct_mid_tiles"] * ms2 * 16 * nw)
                if not cnt["mid_tile_slots"] and not cnt["tile_slots"]:
                    # all three levels by sorting: a batch's tile records copied behind those kept so far (read + write), then two
                    # partition passes, the group index and the counting pass over all of them, one list entry per distinct tile;
                    # the mid level the same, its records cut out of that list
                    pair_t, pair_m = 8 * nwt + 4, 8 * nwm + 4
                    # (the big tiles' records carry no counts: 8 * nwt bytes each through the passes)
                    # (... and are written where they are kept by the extraction: no copy, unless reads are skipped)
                    copy_b = 16 * nwt if (os.environ.get("KATOME_BENCH_TWO_CALLS") == "1" or skip_arg is not None) else 0
                    alg["insert_tiles"] = lambda launches, reads: reads * tiles * (copy_b + 2 * (8 * nwt + 2 * 8 * nwt) + 8 + 8 * nwt) + steps * cnt["distinct_tiles"] * pair_t
                    hist_m = 8 * nwm * (1 if os.environ.get("KATOME_FUSED_HIST", "1") != "0" else 2)       # (the first pass's digits are counted as the records are written)
                    alg["expand_mid_tiles"] = lambda launches, reads: steps * (
                        cnt["distinct_tiles"] * pair_t + cnt["distinct_tiles"] * (span // ms2) * (pair_m + hist_m + 2 * 2 * pair_m + 8 + pair_m)
                        + cnt["distinct_mid_tiles"] * pair_m)
                elif not cnt["mid_tile_slots"] and cnt["tile_slots"]:
                    # the mid tiles were counted by sorting (api.hip, KATOME_SORTED_TILES): one scan of the big-tile table, a record
                    # written per sub-tile, two partition passes (histogram reads the keys, scatter reads and writes the records),
                    # the group index, the counting pass, and one list entry written per distinct mid tile
                    pair_m = 8 * nwm + 4
                    alg["expand_mid_tiles"] = lambda launches, reads: steps * (
                        cnt["tile_slots"] * 16 * nwt + cnt["distinct_tiles"] * (span // ms2) * (pair_m + 2 * (8 * nwm + 2 * pair_m) + 8 + pair_m)
                        + cnt["distinct_mid_tiles"] * pair_m)
            else:
                alg["expand_tiles"] = lambda launches, reads: steps * (cnt["tile_slots"] * 16 * nwt + cnt["distinct_tiles"] * span * 16 * nw)
            sorted_last_level = bool(cnt.get("distinct_kmers")) and not cnt.get("kmer_slots")
            if sorted_last_level:
                # the last level was counted by sorting (table.hip, lds_count_kernel; no k-mer table): one scan of the last tile
                # table, a 12-byte record written per (tile, k-mer) pair, two partition passes (histogram read 8 B, scatter read
                # and write 12 B each), the group index (8 B), the counting pass (12 B read) and 12 B written per edge
                last_slots, last_tiles, last_span, last_nw = ((cnt["mid_tile_slots"], cnt["distinct_mid_tiles"], ms2, nwm) if ms2
                                                              else (cnt["tile_slots"], cnt["distinct_tiles"], span, nwt))
                n_rec = last_tiles * last_span
                last_read = last_slots * 16 * last_nw if last_slots else last_tiles * (8 * last_nw + 4)      # (a table scanned, or a compact list)
                # (records cut out of a list: the first pass's digits are counted as they are written -- one histogram read less)
                hist_reads = 1 if (not last_slots and not rest and os.environ.get("KATOME_FUSED_HIST", "1") != "0") else 2
                alg["expand_tiles"] = lambda launches, reads: steps * (last_read + n_rec * (12 + hist_reads * 8 + 2 * (12 + 12) + 8 + 12) + n_edges * 12)
            # dev_sort: passes over the top log2(n)+9 bits (all of them if that saves fewer than four), each reading and writing
            # every (key, weight) pair once, then one more read + write by the run sort
            sort_passes[0] = _sort_passes(2 * wl.k, n_edges)
            alg["sort_edges"] = lambda launches, reads: steps * n_edges * sum(sort_passes[0]) * 2 * (8 * nw + 4)
            alg["emit_edges"] = lambda launches, reads: steps * (cnt["kmer_slots"] * 16 * nw + n_edges * (8 * nw + 4))
            # node numbering off the sorted edges: keys read by the run-head count and write passes and by the target look-up,
            # one node key probed per target, source and target ids and the node keys written (N ~ E)
            alg["node_set"] = lambda launches, reads: steps * n_edges * (4 * 8 * nw + 8 + 8 + 8 * nw)
            alg["labels"] = lambda launches, reads: steps * n_edges * (8 * nw + 1 + (wl.k + 3) // 4)
        # the records of a level cut out of a list are written with the first partition pass's digits counted on the way
        # (table.hip list_to_records_hist_kernel): that pass has no histogram kernel of its own
        fused_hist = os.environ.get("KATOME_FUSED_HIST", "1") != "0"
        kernel_names = {"extract": "extract_fixed_kernel", "insert": "insert_kernel",
                        "insert_tiles": "insert_kernel", "expand_tiles": "expand_tiles_kernel",
                        "expand_mid_tiles": "expand_tiles_kernel (big tiles -> mid tiles)",
                        "region_order": "radix_hist_kernel+radix_scatter_kernel (HashDigit)",
                        "emit_edges": "emit_edges_kernel", "sort_edges": "radix sort (edges)",
                        "node_set": "src_count/src_write_kernel + dst_seg_kernel + dst_merge_kernel (+ missing_rank_kernel)", "rank": "bucket_index + rank_kernel",
                        "labels": "labels_kernel"}
        if sorted_last_level and args.first_seen_order:
            kernel_names["expand_tiles"] = "seen_records_kernel + 2 x radix pass (HashTaggedDigit) + hash_group_index_kernel + lds_count_seen_kernel (no k-mer table; the edges leave with their sequence numbers)"
        elif sorted_last_level:
            kernel_names["expand_tiles"] = "tiles_to_records_kernel + 2 x radix pass (HashDigit) + hash_group_index_kernel + LDS count (lds_count_packed_kernel: 8-byte slots, one visit; lds_count_kernel / _wide / _full for other shapes) (no k-mer table; the edges are written here)"
            if cnt.get("mid_span") and not cnt.get("mid_tile_slots"):
                kernel_names["expand_tiles"] = kernel_names["expand_tiles"].replace("tiles_to_records_kernel", "list_to_records_kernel")
                if cnt.get("tile_slots"):
                    kernel_names["expand_mid_tiles"] = "tiles_to_records_kernel (big tiles -> mid-tile records) + 2 x radix pass (HashDigit) + hash_group_index_kernel + LDS count (lds_count_full_kernel / lds_count_wide_kernel) (no mid-tile table; a compact list of (mid tile, count) is written)"
                else:
                    kernel_names["insert_tiles"] = "tile records kept aside per batch + 2 x radix pass (HashDigit) + hash_group_index_kernel + LDS count (lds_count_full_kernel: whole two-word keys in the slots; lds_count_wide_kernel behind it and for three-word keys) (no tile table)"
                    kernel_names["expand_mid_tiles"] = "list_to_records_hist_kernel (records + the first pass's digit counts) + 2 x radix pass (HashDigit) + hash_group_index_kernel + LDS count (lds_count_full_kernel / lds_count_wide_kernel) (no mid-tile table)"
        kernels = {}                      # phases of the build (one or several launches each)
        kernel_launches = {}              # single kernels timed launch by launch inside the phases (library: KernelScope)
        reads_per_rank_step = wl.reads / world
        for name, ph in phases.items():
            if name.startswith("k:"):
                continue
            entry = {"kernel": kernel_names.get(name, name), "launches_per_step": ph["launches"] / args.steps,
                     "avg_ms": ph["avg_ms"], "ms_per_step": ph["total_ms"] / args.steps}
            if name in alg:
                by = alg[name](ph["launches"], reads_per_rank_step * args.steps) / ph["launches"]
                entry["alg_bytes_per_launch"] = by
                entry["achieved_GBs"] = by / (ph["avg_ms"] * 1e-3) / 1e9
                entry["frac_of_hbm_peak"] = entry["achieved_GBs"] / HBM_PEAK_GBS
            kernels[name] = entry
        cfg_now = {"reads": wl.reads, "read_len": wl.read_len, "k": wl.k, "batch_reads": batch_reads, "tile_span": span}
        rcs = "true" if wl.reverse_complement else "false"
        even_s = "true" if (wl.reverse_complement and wl.k % 2 == 0) else "false"       # (lds_count_kernel<RC, PER, EVEN_K>)
        exact = {"extract": "void extract_fixed_kernel<%d, %s, %d>" % (nwt, rcs, 256 if wl.stride <= 64 else 64), "insert": "void insert_kernel<%d>" % nw,
                 "insert_tiles": "void insert_kernel<%d>" % nwt,
                 "expand_tiles": "void expand_tiles_kernel<%d, %d, %s, true>" % (
                     _katome_lib().katome_tile_words(wl.k, cnt.get("mid_span") or span), nw, rcs),
                 "expand_mid_tiles": "void expand_tiles_kernel<%d, %d, %s, true>" % (
                     nwt, _katome_lib().katome_tile_words(wl.k, cnt.get("mid_span") or 1), rcs),
                 "sort_edges": "void radix_scatter_kernel<%d, true, RadixDigit<%d>, true>" % (nw, nw),
                 "emit_edges": "void emit_edges_kernel<%d, %s, %d>" % (nw, rcs, 4 if (args.first_seen_order or args.prune or nw > 1) else 8)}
        # per-kernel algorithmic bytes of ONE launch (the default build on one GPU only: other routes run the same kernels on
        # other record counts), and the name rocprofv3 lists the kernel under
        kalg, kexact = {}, {}       # kalg: algorithmic bytes per ELEMENT the kernel processes (keys of a pass, slots of a scan)
        if not use_dist and cnt and not (args.first_seen_order or args.prune):
            pair = 8 * nw + 4
            kalg.update({"radix_scatter_kernel<RadixDigit>": 2 * pair, "radix_hist_kernel<RadixDigit>": 8 * nw, "run_sort": 2 * pair,       # (the library times both run sorts under this name; `kernel` below says which one ran)
                         "src_count+src_write": (2 * 8 * nw + 8 + 8 * nw * n_nodes / max(n_edges, 1)) / 2.0,       # two launches, each reads the keys
                         "dst_merge_kernel": 8 * nw + 8 + 8 * nw * n_nodes / max(n_edges, 1)})
            kexact.update({"radix_scatter_kernel<RadixDigit>": exact["sort_edges"],
                           "radix_hist_kernel<RadixDigit>": "void radix_hist_kernel<%d, RadixDigit<%d> >" % (nw, nw),
                           "run_sort": ("void run_sort_kernel<%d, true>" if os.environ.get("KATOME_RUN_SORT") == "1" else "void run_sort_wave_kernel<%d, true>") % nw,
                           "dst_merge_kernel": "void dst_merge_kernel<%d, false>" % nw})
            if sorted_last_level:
                ms2 = cnt.get("mid_span", 0)
                last_slots, last_tiles, last_span, last_nw = ((cnt["mid_tile_slots"], cnt["distinct_mid_tiles"], ms2, _katome_lib().katome_tile_words(wl.k, ms2)) if ms2
                                                              else (cnt["tile_slots"], cnt["distinct_tiles"], span, nwt))
                n_rec = last_tiles * last_span
                sorted_tiles = not last_slots            # the tile levels were counted by sorting too: no tile tables, the records are cut out of compact lists
                # (the group index is 2^16 + 1 binary searches of ~31 reads each, not a pass over the records)
                per = 8 if (n_rec >> 16) <= 5800 else 13           # table.hip, records_to_edges_sorted: the smaller LDS table when a group fits it
                kalg.update({"radix_scatter_kernel<HashDigit>": 2 * pair, "radix_hist_kernel<HashDigit>": 8 * nw,
                             "tiles_to_records_kernel": (16 * last_nw + float(pair) * n_rec / last_slots) if last_slots else (8 * last_nw + 4 + pair * last_span),
                             "hash_group_index_kernel": 65537.0 * (31 * 8 * nw + 8) / max(n_rec, 1),
                             "lds_count_kernel": pair + float(pair) * n_edges / n_rec})
                kexact.update({"radix_scatter_kernel<HashDigit>": "void radix_scatter_kernel<%d, true, HashDigit<%d>, true>" % (nw, nw),
                               "radix_hist_kernel<HashDigit>": "void radix_hist_kernel<%d, HashDigit<%d> >" % (nw, nw),
                               "tiles_to_records_kernel": (("void list_to_records_hist_kernel<%d, %d, %s>" if (fused_hist and not rest) else "void list_to_records_kernel<%d, %d, %s>")
                                                           if sorted_tiles else "void tiles_to_records_kernel<%d, %d, %s>") % (last_nw, nw, rcs),
                               "hash_group_index_kernel": "void hash_group_index_kernel<%d, %d>" % (nw, nw),
                               "lds_count_kernel": ("void lds_count_kernel<%s, %d, %s>" % (rcs, per, even_s)) if nw == 1 else ("void lds_count_wide_kernel<%s, %d, %d, %s>" % (rcs, per, nw, even_s))})
                # (table.hip, records_to_edges_sorted: one-word k-mers whose groups would take two visits in 12-byte slots are counted in 8-byte
                # slots, one visit -- unless k is even and both strands are counted)
                avg_rec = n_rec >> 16
                fill_rec = 5800 if per == 8 else 9425
                r_try = max(1, math.ceil(avg_rec * float(os.environ.get("KATOME_LC_OPTIMISM", "0.75")) / fill_rec))
                r_packed = max(1, math.ceil(avg_rec * 0.56 / (19456 * 0.66)))
                if nw == 1 and even_s == "false" and os.environ.get("KATOME_LC_PACKED", "1") != "0" and r_try > 1 and r_packed < r_try:
                    kexact["lds_count_kernel"] = "void lds_count_packed_kernel<%s>" % rcs
                # (two-word k-mers whose distinct keys per group fit the smaller table of whole keys: lds_count_full_kernel -- the library
                # decides by counting 256 groups; here: by the distinct keys it reported)
                lf_fill = 1024 * 7 // 20 * 11
                lc_full_on = os.environ.get("KATOME_LC_FULL", "-1") != "0"
                if nw == 2 and lc_full_on and 0 < avg_rec <= 8 * lf_fill and (cnt.get("distinct_kmers", 0) >> 16) <= lf_fill:
                    kexact["lds_count_kernel"] = "void lds_count_full_kernel<%s, %s, %d>" % (rcs, even_s, 4 if (cnt.get("distinct_kmers", 0) >> 16) <= 4096 // 20 * 7 else 7)
            if sorted_last_level and cnt.get("mid_span") and not cnt.get("mid_tile_slots") and nwt == _katome_lib().katome_tile_words(wl.k, cnt["mid_span"]):
                # the tile levels counted by sorting: the same kernels on tile records (two-word keys, 20 bytes) -- the mid tiles'
                # (cut out of the big-tile table, or out of the list of big tiles) and, without a tile table, the big tiles' as well
                ms2 = cnt["mid_span"]
                nwm = _katome_lib().katome_tile_words(wl.k, ms2)
                pair_m = 8 * nwm + 4
                n_mid = cnt["distinct_tiles"] * (span // ms2)
                per_m = 8 if (n_mid >> 16) <= 5800 else 13
                n_big = 0 if cnt["tile_slots"] else reads_per_rank_step * tiles          # (tile records kept aside: counted here too)
                n_all, d_all = n_mid + n_big, cnt["distinct_mid_tiles"] + (cnt["distinct_tiles"] if n_big else 0)
                # (the big tiles' records carry no counts -- one each --, so their passes move the 16-byte keys only)
                rec_big = 8 * nwt
                kalg.update({"radix_scatter_kernel<HashDigit> (tile records)": 2.0 * (pair_m * n_mid + rec_big * n_big) / max(n_all, 1), "radix_hist_kernel<HashDigit> (tile records)": 8 * nwm,
                             "tiles_to_records_kernel (tile records)": (16 * nwt + float(pair_m) * n_mid / cnt["tile_slots"]) if cnt["tile_slots"]
                             else pair_m + pair_m * (span // ms2),
                             "lds_count_kernel (tile records)": (float(pair_m) * n_mid + rec_big * n_big + float(pair_m) * d_all) / max(n_all, 1)})
                kexact.update({"radix_scatter_kernel<HashDigit> (tile records)": "void radix_scatter_kernel<%d, true, HashDigit<%d>, true>" % (nwm, nwm),
                               "radix_hist_kernel<HashDigit> (tile records)": "void radix_hist_kernel<%d, HashDigit<%d> >" % (nwm, nwm),
                               "tiles_to_records_kernel (tile records)": ("void tiles_to_records_kernel<%d, %d, %s>" if cnt["tile_slots"] else
                                                                          "void list_to_records_hist_kernel<%d, %d, %s>" if fused_hist else "void list_to_records_kernel<%d, %d, %s>") % (nwt, nwm, rcs),
                               "lds_count_kernel (tile records)": ("void lds_count_kernel<false, %d, false>" % per_m) if nwm == 1 else ("void lds_count_wide_kernel<false, %d, %d, false>" % (per_m, nwm))})
                # (two-word tiles of few distinct keys per group are counted with their whole keys in the slots: lds_count_full_kernel)
                if nwm == 2 and os.environ.get("KATOME_LC_FULL", "-1") != "0" and max(cnt.get("distinct_tiles", 0), cnt.get("distinct_mid_tiles", 0)) >> 16 <= 1024 * 7 // 20 * 11:
                    kexact["lds_count_kernel (tile records)"] = "void lds_count_full_kernel<false, false, 7>"       # (4 for a level of at most 1433 distinct tiles per group)
        for name, ph in phases.items():
            if not name.startswith("k:"):
                continue
            kn = name[2:]
            entry = {"launches_per_step": ph["launches"] / args.steps, "avg_ms": ph["avg_ms"], "ms_per_step": ph["total_ms"] / args.steps,
                     "elements_per_step": ph["work"] / args.steps}
            if kalg.get(kn) and ph["work"]:
                # launches of one kernel may differ in size (the edge sort's passes and the small sort of the nodes without out-edges):
                # the rate is all their bytes over all their time; "per launch" figures are the averages
                total_bytes = kalg[kn] * ph["work"]
                entry["alg_bytes_per_element"] = kalg[kn]
                entry["alg_bytes_per_launch"] = total_bytes / ph["launches"]
                entry["achieved_GBs"] = total_bytes / (ph["total_ms"] * 1e-3) / 1e9
                entry["frac_of_hbm_peak"] = entry["achieved_GBs"] / HBM_PEAK_GBS
            kernel_launches[kn] = entry

        def roof_phase(name):
            # a PHASE made of several launches (the passes of the edge sort, the five kernels of the sorted last level): bytes of
            # the phase / time of the phase.  Not `roofline` (that is one kernel, below); kept under `roofline_phase`
            passes = sort_passes[0][0]
            parts = {"sort_edges": [(exact["sort_edges"], passes, True),
                                    ("void radix_hist_kernel<%d, RadixDigit<%d> >" % (nw, nw), passes, True),
                                    ("radix_chunk_kernel", passes, True)]}
            label = kernel_names.get(name, name)
            if sorted_last_level and "tiles_to_records_kernel" in kexact:     # (the phase is five kernels: records, two partition passes with their histograms, index, counting)
                n_hist = 1 if "list_to_records_hist_kernel" in kexact["tiles_to_records_kernel"] else 2
                parts["expand_tiles"] = [(kexact["tiles_to_records_kernel"], 1, True), (kexact["radix_scatter_kernel<HashDigit>"], 2, True),
                                         (kexact["radix_hist_kernel<HashDigit>"], n_hist, True), (kexact["hash_group_index_kernel"], 1, True),
                                         (kexact["lds_count_kernel"], 1, False)]
            if "lds_count_kernel (tile records)" in kexact:          # (the mid tiles counted by sorting: the same five kernels)
                t_ = " (tile records)"
                n_hist_m = 1 if "list_to_records_hist_kernel" in kexact["tiles_to_records_kernel" + t_] else 2
                parts["expand_mid_tiles"] = [(kexact["tiles_to_records_kernel" + t_], 1, True), (kexact["radix_scatter_kernel<HashDigit>" + t_], 2, True),
                                             (kexact["radix_hist_kernel<HashDigit>" + t_], n_hist_m, True), (kexact["hash_group_index_kernel"].replace("<1, 1>", "<2, 2>"), 1, True),
                                             (kexact["lds_count_kernel" + t_], 1, False)]
                if not cnt.get("tile_slots"):      # (... and the big tiles: their records copied batch by batch, keys-only passes)
                    n_batches = -(-int(reads_per_rank_step) // int(batch_reads))
                    two_calls = os.environ.get("KATOME_BENCH_TWO_CALLS") == "1" or skip_arg is not None
                    parts["insert_tiles"] = ([("void keep_rest_kernel<%d, false>" % nwt, n_batches, True)] if two_calls else []) + [
                                             (kexact["radix_scatter_kernel<HashDigit>" + t_].replace(", true,", ", false,"), 2, True),
                                             (kexact["radix_hist_kernel<HashDigit>" + t_], 2, True), (kexact["hash_group_index_kernel"].replace("<1, 1>", "<2, 2>"), 1, True),
                                             (kexact["lds_count_kernel" + t_], 1, False)]
            single = name not in parts
            parts = parts.get(name, [(exact.get(name, name), 1, name == "extract")])
            t = pmc_traffic(parts, cfg_now) if not use_dist else None
            lps = kernels[name]["launches_per_step"]
            # (a phase whose scope is opened several times a step -- once per batch and once at the end for the big tiles, twice for
            # the k-mer level -- has its kernels listed per STEP above: its traffic per "launch" is the step's over the step's launches,
            # as its algorithmic bytes are)
            per_step = name in multi            # (a phase of several kernels lists its kernels per STEP, however many times its scope was opened)
            traffic = (t["bytes_per_launch"] / (lps if per_step else 1)) if t else None
            return {"kernel": exact[name].replace("void ", "") if (single and name in exact) else None, "kernels": label, "phase": name, "bound": "hbm",
                    "achieved": kernels[name]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kernels[name]["frac_of_hbm_peak"], "traffic": traffic,
                    "traffic_detail": t, "alg_bytes_per_launch": kernels[name]["alg_bytes_per_launch"],
                    "launches_per_step": lps, "alg_bytes_per_step": kernels[name]["alg_bytes_per_launch"] * lps,
                    "traffic_per_step": traffic * lps if traffic is not None else None,
                    "avg_launch_ms": kernels[name]["avg_ms"], "ms_per_step": kernels[name]["ms_per_step"]}

        def roof_kernel(kn):
            e = kernel_launches[kn]
            streaming = not kn.startswith("lds_count_kernel")
            t = pmc_traffic([(kexact[kn], 1, streaming)], cfg_now) if kn in kexact and not use_dist else None
            scatter = kn.startswith("radix_scatter_kernel")
            return {"kernel": kexact.get(kn, kn).replace("void ", ""), "bound": "hbm", "achieved": e["achieved_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": e["frac_of_hbm_peak"],
                    # (measured on this card type, not live: the same bytes moved in the kernel's memory pattern with no other work)
                    "ceiling": SCATTER_PATTERN_CEILING_GBS if scatter else COPY_CEILING_GBS,
                    "ceiling_what": ("4096-record tiles scattered as 128-byte stretches into 256 streams, no ranking (tools/microbench_scatter.hip)"
                                     if scatter else "plain copy of the same bytes (tools/microbench_scatter.hip)"),
                    "frac_of_ceiling": e["achieved_GBs"] / (SCATTER_PATTERN_CEILING_GBS if scatter else COPY_CEILING_GBS),
                    "traffic": t["bytes_per_launch"] if t else None, "traffic_detail": t,
                    "alg_bytes_per_launch": e["alg_bytes_per_launch"], "avg_launch_ms": e["avg_ms"],
                    "launches_per_step": e["launches_per_step"], "ms_per_step": e["ms_per_step"],
                    "timed": "HIP events around every launch of this kernel on the build's stream (library KernelScope)"}
        # `roofline` = the ONE kernel the step spends most time in: a kernel timed launch by launch, or a phase that is one kernel
        multi = {"sort_edges", "node_set"} | ({"expand_tiles"} if sorted_last_level else set())
        mid_sorted = "lds_count_kernel (tile records)" in kexact
        if mid_sorted:
            multi.add("expand_mid_tiles")
        big_sorted = bool(cnt) and span > 1 and sorted_last_level and not cnt.get("tile_slots")
        if big_sorted:
            multi.add("insert_tiles")
        cands = [("k", kn, e["ms_per_step"]) for kn, e in kernel_launches.items() if "alg_bytes_per_launch" in e]
        cands += [("p", n, e["ms_per_step"]) for n, e in kernels.items() if e.get("alg_bytes_per_launch", 0) > 0 and n not in multi]
        kind, dom_name, _ = max(cands, key=lambda c: c[2])
        roofline = roof_kernel(dom_name) if kind == "k" else roof_phase(dom_name)
        # the two table levels are bound by device-scope atomics, not by bytes (1.8-2.7e10 random atomics/s
        # whatever the footprint): say so next to the byte figures the contract asks for
        upserts = {"insert_tiles": reads_per_rank_step * (tiles if span > 1 else 0),
                   "expand_mid_tiles": cnt.get("distinct_tiles", 0) * ((span // cnt["mid_span"]) if cnt.get("mid_span") else 0)} if cnt else {}
        if mid_sorted:
            upserts.pop("expand_mid_tiles", None)
        if big_sorted:
            upserts.pop("insert_tiles", None)
        if kind == "p" and upserts.get(dom_name):
            rate = upserts[dom_name] / (kernels[dom_name]["ms_per_step"] * 1e-3)
            roofline.update({"limiter": "device-scope atomics (one upsert = a compare-and-swap or an add on a random slot)", "upserts_per_step": upserts[dom_name],
                             "upserts_per_s": rate, "atomic_wall_per_s": 2.5e10, "frac_of_atomic_wall": rate / 2.5e10})
        # the largest kernel that IS bound by bytes, for comparison with earlier rounds (round 2's `roofline` was this kernel)
        stream_cands = [c for c in cands if c[0] == "k" and not c[1].startswith("lds_count_kernel")]
        roofline_streaming = roof_kernel(max(stream_cands, key=lambda c: c[2])[1]) if stream_cands else None
        dom = max((n for n in kernels if kernels[n].get("alg_bytes_per_launch", 0) > 0), key=lambda n: kernels[n]["ms_per_step"])
        roof = roof_phase
        line = {
            "metric": "k-mers/s", "value": kmers / (ms_per_step * 1e-3), "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64" if nw == 1 else "u128",
            "data": "synthetic",
            "config": {"workload": "%s: %d synthetic %d bp reads, k=%d, reverse_complement=%s, genome %d, err %.0e"
                                   % (wl.name, wl.reads, wl.read_len, wl.k, wl.reverse_complement, wl.genome_len,
                                      wl.err_rate),
                       "reads": wl.reads, "read_len": wl.read_len, "k": wl.k, "batch_reads": batch_reads,
                       "tile_span": span, "order": "first-seen (petgraph)" if (args.first_seen_order or args.prune) else "by packed key",
                       "min_weight": args.min_weight,
                       "parallelism": "reads sharded by index over %d GPU(s); %s" % (
                           world, {"local": "every rank counts its reads, distinct k-mers routed by hash (one all-to-all)",
                                   "supermers": "reads cut into supermers (runs of windows with one minimizer), routed once by a hash of the minimizer BEFORE any counting (one all-to-all of 16-byte records); every rank counts what it receives",
                                   "tiles": "tiles, mid tiles and k-mer records routed by hash (three all-to-alls)"}.get(getattr(job, "route", ""), getattr(job, "route", "?")))
                       if use_dist else "1 GPU"},
            "distinct_edges": n_edges, "nodes": n_nodes, "distinct_edges_per_s": n_edges / (ms_per_step * 1e-3),
            "roofline": roofline, "roofline_streaming": roofline_streaming, "roofline_phase": roof_phase(dom), "roofline_extract": roof_phase("extract"),
            # the three counting levels (big tiles, mid tiles, k-mers), each a phase of several kernels: bytes of the phase / its time,
            # and the HBM traffic of its kernels from the committed PMC passes
            "roofline_levels": [roof_phase(n) for n in ("insert_tiles", "expand_mid_tiles", "expand_tiles") if kernels.get(n, {}).get("alg_bytes_per_launch", 0) > 0],
            "kernels": kernels, "kernel_launches": kernel_launches, "counts": cnt, "source_id": source_id(),
            **({"lc_phases": lc_phases(False)} if lc_phases(None) else {}),
        }
        if args.prune:
            line["config"]["pruner"] = "remove_dead_paths after the build (reference order%s)" % (", on the graph gathered to rank 0" if use_dist else "")
        if use_dist:
            # per exchange phase: bytes that left rank 0 per step, time inside the exchange (HIP events on the stream the RCCL
            # calls are enqueued on), and the per-link rate that implies on the full xGMI mesh (bytes / (N-1) links) against
            # ~153 GB/s per link and direction
            ex = {}
            for name, x in job.exchange.items():
                per_step_bytes, per_step_ms = x["bytes_out"] / (args.steps + 0.0), x["ms"] / args.steps
                links = max(world - 1, 1)
                ex[name] = {"calls_per_step": x["calls"] / args.steps, "bytes_out_per_step": per_step_bytes, "ms_per_step": per_step_ms,
                            "largest_message_bytes": x["max_message_bytes"],
                            "per_link_GBs": (per_step_bytes / links / (per_step_ms * 1e-3) / 1e9) if per_step_ms > 0 and world > 1 else None,
                            "link_peak_GBs": 153.0}
            line["exchange"] = ex
            line["config"]["transport"] = comm.kind
            line["config"]["route"] = getattr(job, "route", None)
            line["config"]["comm_ranks"] = comm.world          # (the communicator's own rank count: RCCL saw this many ranks)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample_reads)
            if not args.no_cpu_c2:
                try:
                    line["cpu_baseline_c2"] = c2_in_full(torch)
                except Exception as e:   # noqa: BLE001  (a side line: never the reason the bench line is missing)
                    line["cpu_baseline_c2"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and not use_dist and not args.no_extras:
            # Beside the headline (not `value`): the build in the reference's own numbering -- what INTEGRATION.md's GpuGIR binds
            # (flags: 1) --, the build of a caller that passes no table hint, and the end-to-end region incl. H2D/D2H
            try:
                if not args.first_seen_order and not args.prune:
                    t2 = PhaseTimer()
                    ms, _ = timed(lambda: one_build_single(wl, packed, skip_arg, recbuf, batch_reads, t2, True, 0, args.table_factor), 2, torch)
                    line["reference_order"] = {"ms_per_step": ms, "kmers_per_s": kmers / (ms * 1e-3), "order": "first-seen (petgraph)",
                                               "steps": 2, "what": "same workload, KATOME_FLAG_FIRST_SEEN_ORDER (INTEGRATION.md's binding)"}
                t3 = PhaseTimer()
                ms, _ = timed(lambda: one_build_single(wl, packed, skip_arg, recbuf, batch_reads, t3, args.first_seen_order, 0, 0.0), 2, torch)
                line["unhinted_table"] = {"ms_per_step": ms, "kmers_per_s": kmers / (ms * 1e-3), "steps": 2,
                                          "what": "same workload, table_slots_hint = 0 (the k-mer table is sized from the distinct tiles; the tile table grows by re-hashing)"}
            except Exception as e:     # noqa: BLE001
                line["extras_error"] = "%s: %s" % (type(e).__name__, e)
            try:
                del packed, recbuf
                kd.release_cache()
                if args.end_to_end_reads > 0:
                    line["end_to_end"] = end_to_end(wl, args.end_to_end_reads)
            except Exception as e:     # noqa: BLE001
                line["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not use_dist and args.next_stages_reads > 0 and not args.no_extras:
            try:                       # not part of the metric: whatever goes wrong here must not cost the line above
                kd.release_cache()
                line["next_stages"] = next_stages(wl, args.next_stages_reads)
            except Exception as e:     # noqa: BLE001
                line["next_stages"] = {"error": "%s: %s" % (type(e).__name__, e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
