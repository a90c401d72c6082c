"""Test double for katome_amd/csrc/prune.hip: the same decomposition of Prunable::remove_dead_paths (walks over a
degree / first_edge summary, marked indices consumed from the top, the two swap_remove replays, moves applied at the
end of each pass) in plain Python, so that the decomposition itself can be checked against the oracle's literal
petgraph restatement without a GPU.  Not product code."""


def remove_dead_paths(src, dst, k):
    """src/dst: petgraph endpoints per edge index -> (src, dst, first-seen index of every surviving edge,
    first-seen id of every surviving node, stats)"""
    src, dst = [int(x) for x in src], [int(x) for x in dst]
    n_nodes = max(src + dst) + 1 if src else 0
    orig, node_orig = list(range(len(src))), list(range(n_nodes))
    stats = dict(passes=0, marked=0, removed_by_duplicates=0, self_loops=0)
    while True:
        stats["passes"] += 1
        n_e, n_n = len(src), len(node_orig)
        # device: degree_kernel
        first_out, best, indeg, outdeg = [-1] * n_n, [-1] * n_n, [0] * n_n, [0] * n_n
        for e in range(n_e):
            a, b = src[e], dst[e]
            outdeg[a] += 1
            indeg[b] += 1
            if orig[e] > best[a]:
                best[a], first_out[a] = orig[e], e
        # device: walk_kernel (check_dead_path from every vertex without incoming edges)
        mult = {}
        for v in range(n_n):
            if indeg[v] != 0:
                continue
            cur, cnt, path = v, 0, []
            while True:
                cnt += 1
                if cnt >= 2 * k:
                    path = []
                    break
                e = first_out[cur]
                if e < 0:
                    break
                path.append(e)
                cur = dst[e]
                if indeg[cur] >= 3:
                    break
            for e in path:
                mult[e] = mult.get(e, 0) + 1
        if not mult:
            break
        stats["marked"] += sum(mult.values())
        pos = sorted(mult)
        # host: replay_edges
        occ, size, q, victims = list(pos), n_e, len(pos) - 1, []
        for j in range(len(pos) - 1, -1, -1):
            d = pos[j]
            for r in range(mult[d]):
                if d >= size:
                    break
                last = size - 1
                while q >= 0 and pos[q] > last:
                    q -= 1
                mover = occ[q] if q >= 0 and pos[q] == last else last
                victims.append(occ[j])
                stats["removed_by_duplicates"] += 1 if r else 0
                if d != last:
                    occ[j] = mover
                size -= 1
        e_new = size
        edge_moves = [(pos[j], occ[j]) for j in range(len(pos)) if pos[j] < e_new and occ[j] != pos[j]]
        # device: death_count_kernel / death_emit_kernel
        deg = [indeg[v] + outdeg[v] for v in range(n_n)]
        last_touch = [0] * n_n
        for t, e in enumerate(victims):
            for v in (src[e], dst[e]):
                deg[v] -= 1
                last_touch[v] = t + 1
        die = []
        for t, e in enumerate(victims):
            a, b = src[e], dst[e]
            stats["self_loops"] += a == b
            die.append((a if deg[a] == 0 and last_touch[a] == t + 1 else None,
                        b if b != a and deg[b] == 0 and last_touch[b] == t + 1 else None))
        # host: replay_nodes
        n_die = sum((a is not None) + (b is not None) for a, b in die)
        base = n_n - n_die
        tail_pos = list(range(base, n_n))
        tail_occ = list(range(base, n_n))
        dead = [False] * n_die
        size = n_n

        def pos_of(v):
            return v if v < base else tail_pos[v - base]

        def remove(v):
            nonlocal size
            p, top = pos_of(v), size - 1
            y = tail_occ[top - base]
            if v >= base:
                dead[v - base] = True
            if p != top:
                if p >= base:
                    tail_occ[p - base] = y
                tail_pos[y - base] = p
            size -= 1

        for a, b in die:
            if a is not None and b is not None:
                for v in ((b, a) if pos_of(a) < pos_of(b) else (a, b)):
                    remove(v)
            elif a is not None:
                remove(a)
            elif b is not None:
                remove(b)
        n_new = size
        node_moves = [(tail_pos[i], base + i) for i in range(n_die) if not dead[i]]
        # device: move_edges_kernel, move_nodes_kernel, remap_kernel
        for d, s in edge_moves:
            src[d], dst[d], orig[d] = src[s], dst[s], orig[s]
        del src[e_new:], dst[e_new:], orig[e_new:]
        tail_map = {}
        for d, s in node_moves:
            node_orig[d] = node_orig[s]
            tail_map[s] = d
        del node_orig[n_new:]
        src = [tail_map[a] if a >= n_new else a for a in src]
        dst = [tail_map[b] if b >= n_new else b for b in dst]
    return src, dst, orig, node_orig, stats
