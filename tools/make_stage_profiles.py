"""Writes profiles/<tag>_pipeline.md and profiles/<tag>_prune.md from the outputs of
  python tools/bench_pipeline.py --workload c3 --reads 100000000 > gpurun_out/pipeline100.json
  python tools/bench_pipeline.py --workload c3                    > gpurun_out/pipeline_c3.json
  KATOME_TRACE_PRUNE=1 python tools/bench_prune.py --workload c3 --reads 20000000 > gpurun_out/prune20.json 2> gpurun_out/prune20.err
  KATOME_TRACE_PRUNE=1 python tools/bench_prune.py --workload c3                  > gpurun_out/prune_c3.json 2> gpurun_out/prune_c3.err
usage: python tools/make_stage_profiles.py r01i"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = lambda f: os.path.join(ROOT, "gpurun_out", f)
keep = lambda d: {k: d[k] for k in ("workload", "reads", "k", "rc", "build_ms", "prune_ms", "before", "after", "stats", "shrink")}

out = ["# %s — every stage before collapse on one MI355X (tools/bench_pipeline.py)\n" % tag,
       "First-seen-order build, remove_dead_paths, standardize_contigs, remove_weak_edges(2), standardize_contigs, standardize_edges(G, k, 2), "
       "remove_dead_paths — `assemble_with_graph` (asm/basic_assembler.rs:58-72) up to `collapse`, each stage index for index the reference's graph "
       "(tests/test_gpu_prune.py::test_every_stage_up_to_collapse, tools/soak_pipeline.py).\n",
       "One process per run, so `build` includes the first allocations, whose cost is the driver's: near zero on a fresh box, 10-35 GB/s when "
       "another process has just used the memory (DESIGN.md section 3). The warm build is 1.02 s (`bench.py --first-seen-order`).\n"]
for f, cmd in (("pipeline100.json", "--workload c3 --reads 100000000"), ("pipeline_c3.json", "--workload c3")):
    d = json.load(open(G(f)))
    out += ["## `python tools/bench_pipeline.py %s`\n" % cmd, "| stage | ms | nodes after | edges after |\n|---|---|---|---|"]
    out += ["| %s | %.1f | %d | %d |" % (s["stage"], s["ms"], s["nodes"], s["edges"]) for s in d["stages"]]
    out.append("| **total** | **%.1f** | | |\n" % d["total_ms"])
out.append("r01g (first version of the stages, 100 M reads): 7649 ms in all — build 3929 (cold allocations), remove_dead_paths 2925 (host replays), "
           "standardize_contigs 262, remove_weak_edges 101, standardize_contigs 22, standardize_edges 3, remove_dead_paths 406.\n")
open(os.path.join(ROOT, "profiles", "%s_pipeline.md" % tag), "w").write("\n".join(out))

d20, dc3 = json.load(open(G("prune20.json"))), json.load(open(G("prune_c3.json")))
lines = [l for l in open(G("prune_c3.err")).read().splitlines() if l.startswith("[prune]")]


def block(n):
    idx = [i for i, l in enumerate(lines) if l.startswith("[prune] pass %d:" % n)][0]
    j = idx - 1
    while j >= 0 and not lines[j].startswith("[prune] pass"):
        j -= 1
    return lines[j + 1:idx + 1]


out = ["# %s — remove_dead_paths with both swap_remove replays on the device, and shrink (tools/bench_prune.py)\n" % tag,
       "Same workloads as `r01e_prune.md` (first version: walks and moves on the device, the two index replays on one host thread). Every count in "
       "`stats` is unchanged; `host_ms` is 0 because no pass fell back to the sequential replay.\n",
       "## C3 in full (200 M reads, 1.61 G edges): `python tools/bench_prune.py --workload c3`\n",
       "```json\n" + json.dumps(keep(dc3)) + "\n```\n",
       "`build_ms` is the first build of the process, allocations included (see the pipeline note); the warm build is 1.02 s.\n",
       "Per-stage times (`KATOME_TRACE_PRUNE=1`), pass 1 and a closing pass:\n",
       "```\n" + "\n".join(block(1)) + "\n...\n" + "\n".join(block(30)) + "\n```\n",
       "All passes:\n\n```\n" + "\n".join(l for l in lines if l.startswith("[prune] pass")) + "\n```\n",
       "## C3 / 20 M reads\n", "```json\n" + json.dumps(keep(d20)) + "\n```\n",
       """## Before / after

| | first version (r01e) | now |
|---|---|---|
| C3 / 20 M reads, 38 passes | 593–618 ms (311–343 ms in the host replays) | %.0f ms |
| C3 in full, 45 passes | 6.24 s (2.66 s host) | %.2f s |
| pass 1 of C3 in full (165 M edge removals, 132 M node removals) | 2317 ms: marks to host 170, edge replay 725, deaths to host 244, node replay 958, apply 160 | ~112 ms: walks 17, edge replay 13, degree updates + node replay 55, apply 27 (adjacency and edge slots, once: 91 ms) |
| a closing pass of C3 in full (~2 400 removals) | 43 ms | ~2.5 ms: walks 0.2, marks 0.4, nodes 0.8, apply + next pass's walk set 1.0 |
| shrink, C3 in full (1.26 G edges -> 100 M merged edges, 1.2 GB of labels) | 435–490 ms | %.0f ms |
| shrink, C3 / 20 M reads | 44 ms | %.0f ms |

What changed, in order: walks from a compacted list of in-degree-0 vertices (full lanes); `remove_edge` replay as a scan of
n -> max(n - c, d) functions + pointer jumping; `remove_node` replay as holes of the vacated tail positions settled in rounds;
one 8-byte look-up per walk step (degrees + first successor in one word per node; the same for shrink's and
standardize_contigs' walks); mark compaction skips untouched 2048-edge groups; endpoint re-labelling fused into the
first-out rebuild; prefix sums of long count arrays over many workgroups; per-vertex edge slots (out-edges by last base,
in-edges by first base) so that a pass touches only what it changes; the list of Input vertices carried from pass to pass;
only walks within reach of a change repeated (r01i intermediate states: 0.94 s -> 0.79 -> 0.73 -> 0.52 s at C3).
""" % (d20["prune_ms"], dc3["prune_ms"] / 1e3, dc3["shrink"]["ms"], d20["shrink"]["ms"])]
open(os.path.join(ROOT, "profiles", "%s_prune.md" % tag), "w").write("\n".join(out))
print("pipeline c3 total %.0f ms; prune c3 %.0f ms, 20M %.0f ms; shrink c3 %.0f ms" %
      (json.load(open(G("pipeline_c3.json")))["total_ms"], dc3["prune_ms"], d20["prune_ms"], dc3["shrink"]["ms"]))
