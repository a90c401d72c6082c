/* selftest.c -- drives the oracle from a plain C main so that the whole restatement (petgraph lists, swap_remove,
 * label merging) can run under AddressSanitizer / UBSan on the CPU (tests/test_sanitizers.py).  TEST INFRASTRUCTURE.
 * usage: selftest <k> <rc> <stages> <weak-threshold> <fastq>...   -> prints "nodes edges read_bytes sum_weights seq_bytes" */
#include <stdio.h>
#include <stdlib.h>

#include "katome_oracle.h"

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s k rc stages weak_threshold files...\n", argv[0]); return 2; }
    const size_t k = (size_t)atoi(argv[1]);
    const int rc = atoi(argv[2]);
    ko_set_post_build(argv[3][0] == '-' ? "" : argv[3], (uint32_t)atoi(argv[4]));
    ko_graph *g = NULL;
    int st = ko_build_files((const char *const *)(argv + 5), (size_t)(argc - 5), 1, rc, k, 1, &g);
    if (st) { fprintf(stderr, "oracle error %d: %s\n", st, ko_last_error()); return 1; }
    unsigned long long sum = 0;
    for (uint64_t e = 0; e < g->n_edges; ++e) sum += g->edge_weight[e];
    printf("%llu %llu %llu %llu %llu\n", (unsigned long long)g->n_nodes, (unsigned long long)g->n_edges,
           (unsigned long long)g->read_bytes, sum, (unsigned long long)(g->edge_seq_off ? g->edge_seq_off[g->n_edges] : 0));
    ko_graph_free(g);
    return 0;
}
