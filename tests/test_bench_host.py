"""bench.py's host-side bookkeeping (no GPU): the PMC traffic behind `roofline.traffic` is only used when it was taken on
the very kernels that run now and on the same configuration."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_stale_pmc_counters_are_refused(tmp_path, monkeypatch):
    import bench
    cfg = {"reads": 1000, "read_len": 150, "k": 31, "batch_reads": 64, "tile_span": 30}
    sid = bench.source_id()
    assert len(sid) == 16 and sid == bench.source_id()                 # a function of the sources alone
    kernels = {"void some_kernel<1>": {"dispatches": 3, "fetch_kib": 10.0, "write_kib": 20.0}}
    f = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "PMC_FILE", str(f))
    parts = [("void some_kernel<1>", 2, True)]
    f.write_text(json.dumps({"config": cfg, "source_id": sid, "kernels": kernels}))
    t = bench.pmc_traffic(parts, cfg)
    assert t and t["bytes_per_launch"] == 2 * (10.0 * 2 + 20.0) * 1024            # FETCH_SIZE doubled for a streaming kernel
    f.write_text(json.dumps({"config": cfg, "source_id": "0" * 16, "kernels": kernels}))
    assert bench.pmc_traffic(parts, cfg) is None                                   # counters of other kernels
    f.write_text(json.dumps({"config": cfg, "kernels": kernels}))
    assert bench.pmc_traffic(parts, cfg) is None                                   # no id at all (round 2's file)
    f.write_text(json.dumps({"config": dict(cfg, reads=2000), "source_id": sid, "kernels": kernels}))
    assert bench.pmc_traffic(parts, cfg) is None                                   # another workload
    f.write_text(json.dumps({"config": cfg, "source_id": sid, "kernels": {}}))
    assert bench.pmc_traffic(parts, cfg) is None                                   # the kernel was not profiled


def test_committed_pmc_file_names_its_sources():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert len(d.get("source_id", "")) == 16 and d["config"]["reads"] == 200_000_000
