// Host-only driver of katome_amd/csrc/ingest.cpp (file checks, FASTQ/FASTA/BFCounter scan, ACGT filter, 2-bit packing,
// the threaded scan) for sanitizer runs (tests/test_sanitizers.py).
// usage: ingest_selftest <k> <file_type> <files...>  -> "status n_records n_reads read_bytes packed_bytes total_windows fixed_len checksum"
#include <stdio.h>
#include <stdlib.h>

#include "../../katome_amd/csrc/common.h"

using namespace katome;

// the caching device allocator is not part of this build: ingest never touches the device
namespace katome {
int dev_malloc(void**, size_t, hipStream_t) { return KATOME_E_DEVICE; }
void dev_free(void*, hipStream_t) {}
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s k file_type files...\n", argv[0]); return 2; }
    katome_settings s{};
    s.k = (uint32_t)atoi(argv[1]);
    s.file_type = (uint8_t)atoi(argv[2]);
    HostReads hr;
    const int st = ingest_files(&s, (const char* const*)(argv + 3), (size_t)(argc - 3), hr);
    unsigned long long sum = 0;
    for (uint64_t i = 0; i < hr.packed_bytes; ++i) sum = sum * 1099511628211ull + hr.packed[i];
    printf("%d %llu %llu %llu %llu %llu %u %llu\n", st, (unsigned long long)hr.n_records, (unsigned long long)hr.n_reads,
           (unsigned long long)hr.read_bytes, (unsigned long long)hr.packed_bytes, (unsigned long long)hr.total_windows, hr.fixed_len, sum);
    if (st) fprintf(stderr, "%s\n", get_error());
    return 0;
}
