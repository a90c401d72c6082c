// ingest.cpp -- host ingest in front of the GPU build: file checks, FASTQ/FASTA record scan,
// the ACGT-only read filter and 2-bit packing.
//
// Restates reference src/katome/algorithms/builder.rs: check_files (57-77), create_fastq (142-165),
// create_fasta (118-140); packing follows compress_node's bit order (compress.rs:55-73) with the
// symbol code of encode_fasta_symbol (compress.rs:347-378).  Record framing follows bio 0.10.0's
// io::fastq / io::fasta readers as published (crate not vendored in the reference): FASTQ = 4
// lines per record, header starts with '@', sequence = the 2nd line right-trimmed; FASTA =
// '>' header, sequence = following lines up to the next '>' right-trimmed and joined.
#include <fcntl.h>
#include <limits.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace katome {

static thread_local char g_error[1024];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_error; }

HostReads::~HostReads() { free(packed); free(byte_off); free(len); free(weight); }

namespace {

struct Mapped {
    const uint8_t* p = nullptr; size_t n = 0; int fd = -1;
    ~Mapped() { if (p && n) munmap((void*)p, n); if (fd >= 0) close(fd); }
};

// symbol -> 2-bit code, 0xFF for anything builder.rs:155 rejects (only upper-case ACGT pass)
struct CodeTable {
    uint8_t t[256];
    CodeTable() { memset(t, 0xFF, sizeof t); t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3; }
};
const CodeTable CODE;

inline size_t rtrim(const uint8_t* s, size_t n) {        // str::trim_right on ASCII
    while (n && (s[n - 1] == ' ' || (s[n - 1] >= 9 && s[n - 1] <= 13))) --n;
    return n;
}

struct LineReader {                                      // BufRead::read_line: includes the '\n'
    const uint8_t* p; size_t n, pos = 0;
    LineReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    bool next(const uint8_t*& s, size_t& len) {
        if (pos >= n) { s = p + n; len = 0; return false; }
        const uint8_t* nl = (const uint8_t*)memchr(p + pos, '\n', n - pos);
        size_t end = nl ? (size_t)(nl - p) + 1 : n;
        s = p + pos; len = end - pos; pos = end;
        return true;
    }
};

int reserve_reads(HostReads& r, uint64_t extra_bytes) {
    if (r.n_reads + 2 > r.cap_reads) {
        uint64_t nc = r.cap_reads ? r.cap_reads * 2 : 4096;
        uint64_t* bo = (uint64_t*)realloc(r.byte_off, nc * sizeof(uint64_t));
        if (!bo) { set_error("out of host memory"); return KATOME_E_OOM; }
        r.byte_off = bo;
        uint32_t* ln = (uint32_t*)realloc(r.len, nc * sizeof(uint32_t));
        if (!ln) { set_error("out of host memory"); return KATOME_E_OOM; }
        r.len = ln;
        r.cap_reads = nc;
    }
    if (r.packed_bytes + extra_bytes + 32 > r.packed_cap) {
        uint64_t nc = (r.packed_bytes + extra_bytes + 32) * 2;
        uint8_t* pk = (uint8_t*)realloc(r.packed, nc);
        if (!pk) { set_error("out of host memory"); return KATOME_E_OOM; }
        r.packed = pk; r.packed_cap = nc;
    }
    return KATOME_OK;
}

// builder.rs:152-160 loop body up to the hand-off to add_read_fastaq
int accept_read(HostReads& r, const uint8_t* seq, size_t n, uint32_t k) {
    ++r.n_records;
    for (size_t i = 0; i < n; ++i) if (CODE.t[seq[i]] == 0xFF) return KATOME_OK;     // `continue` (155-157)
    r.read_bytes += n;                                                             // total += seq.len() (158)
    if (n < k) { set_error("Read is too short!"); return KATOME_E_SHORT_READ; }     // pt_graph.rs:278
    if (n > 0xFFFFFFFFull) { set_error("read longer than 2^32 bases"); return KATOME_E_UNSUPPORTED; }
    const size_t nb = (n + 3) / 4;
    KCHECK(reserve_reads(r, nb));
    uint8_t* out = r.packed + r.packed_bytes;
    size_t i = 0, o = 0;
    for (; i + 4 <= n; i += 4)
        out[o++] = (uint8_t)((CODE.t[seq[i]] << 6) | (CODE.t[seq[i + 1]] << 4) | (CODE.t[seq[i + 2]] << 2) | CODE.t[seq[i + 3]]);
    if (i < n) {
        uint8_t c = 0; size_t rem = n - i;
        for (size_t j = 0; j < rem; ++j) c = (uint8_t)((c << 2) | CODE.t[seq[i + j]]);
        out[o++] = (uint8_t)(c << (2 * (4 - rem)));       // last byte left-aligned (compress.rs:64-72)
    }
    if (r.n_reads == 0) r.fixed_len = (uint32_t)n;
    else if (r.fixed_len != (uint32_t)n) r.all_fixed = false;
    r.byte_off[r.n_reads] = r.packed_bytes;
    r.len[r.n_reads] = (uint32_t)n;
    r.packed_bytes += nb;
    r.byte_off[++r.n_reads] = r.packed_bytes;
    r.total_windows += n - k + 1;
    return KATOME_OK;
}

int scan_fastq(const uint8_t* p, size_t n, HostReads& r, uint32_t k) {
    LineReader lr(p, n);
    const uint8_t *h, *s, *sep, *q; size_t hn, sn, sepn, qn;
    while (lr.next(h, hn)) {
        if (h[0] != '@') { set_error("Expected @ at record start."); return KATOME_E_PARSE; }
        lr.next(s, sn); lr.next(sep, sepn); lr.next(q, qn);
        if (qn == 0) { set_error("Incomplete record. Each FastQ record has to consist of 4 lines: header, sequence, separator and qualities."); return KATOME_E_PARSE; }
        KCHECK(accept_read(r, s, rtrim(s, sn), k));
    }
    return KATOME_OK;
}

int scan_fasta(const uint8_t* p, size_t n, HostReads& r, uint32_t k) {
    LineReader lr(p, n);
    const uint8_t* line; size_t ln;
    std::vector<uint8_t> seq;
    bool have = lr.next(line, ln);
    while (have) {
        if (line[0] != '>') { set_error("Expected > at record start."); return KATOME_E_PARSE; }
        seq.clear();
        while ((have = lr.next(line, ln)) && line[0] != '>') seq.insert(seq.end(), line, line + rtrim(line, ln));
        KCHECK(accept_read(r, seq.data(), seq.size(), k));
    }
    return KATOME_OK;
}

// create_bfc (builder.rs:79-115): one `kmer<TAB>weight` per line; lines with weight < minimal_weight_threshold are
// dropped (106-108); every kept line is one k-mer with that weight (add_read_bfc, pt_graph.rs:317-330).
// `BufRead::lines()` strips "\n" and "\r\n".  The reference encodes whatever bytes it is given
// (encode_fasta_symbol is undefined outside ACGT, compress.rs:319-321); here a non-ACGT byte is a parse error.
int scan_bfc(const uint8_t* p, size_t n, HostReads& r, uint32_t k, uint32_t min_weight) {
    LineReader lr(p, n);
    const uint8_t* line; size_t ln;
    while (lr.next(line, ln)) {
        if (ln && line[ln - 1] == '\n') --ln;
        if (ln && line[ln - 1] == '\r') --ln;
        const uint8_t* tab = (const uint8_t*)memchr(line, '\t', ln);
        if (!tab) { set_error("called `Option::unwrap()` on a `None` value (BFCounter line without a tab)"); return KATOME_E_PARSE; }
        const size_t klen = (size_t)(tab - line);
        const uint8_t* w = tab + 1; size_t wn = ln - klen - 1;
        if (const uint8_t* tab2 = (const uint8_t*)memchr(w, '\t', wn)) wn = (size_t)(tab2 - w);
        uint64_t weight = 0;
        size_t i = 0;
        if (i < wn && w[i] == '+') ++i;                            // u32::from_str accepts a leading '+'
        if (i == wn) { set_error("Parse int error (if the kind is overflow user should change type of EdgeWeight in prelude.rs): cannot parse integer from empty string"); return KATOME_E_PARSE; }
        for (; i < wn; ++i) {
            if (w[i] < '0' || w[i] > '9') { set_error("Parse int error (if the kind is overflow user should change type of EdgeWeight in prelude.rs): invalid digit found in string"); return KATOME_E_PARSE; }
            weight = weight * 10 + (w[i] - '0');
            if (weight > 0xFFFFFFFFull) { set_error("Parse int error (if the kind is overflow user should change type of EdgeWeight in prelude.rs): number too large to fit in target type"); return KATOME_E_PARSE; }
        }
        if (weight < min_weight) { ++r.n_records; continue; }       // builder.rs:106-108
        for (size_t j = 0; j < klen; ++j) if (CODE.t[line[j]] == 0xFF) { set_error("BFCounter k-mer with a non-ACGT byte"); return KATOME_E_PARSE; }
        if (klen > k) { set_error("BFCounter line holds %zu bases, k is %u", klen, k); return KATOME_E_ARG; }
        KCHECK(accept_read(r, line, klen, k));                      // total += edge.len() (109); "Read is too short!" (pt_graph.rs:318)
        uint32_t* nw = (uint32_t*)realloc(r.weight, (r.n_reads + 1) * sizeof(uint32_t));
        if (!nw) { set_error("out of host memory"); return KATOME_E_OOM; }
        r.weight = nw;
        r.weight[r.n_reads - 1] = (uint32_t)weight;
    }
    return KATOME_OK;
}

int scan_range(const katome_settings* s, const uint8_t* p, size_t n, HostReads& r) {
    if (s->file_type == 2) return scan_bfc(p, n, r, s->k, s->min_weight);
    return s->file_type == 1 ? scan_fastq(p, n, r, s->k) : scan_fasta(p, n, r, s->k);
}

// One file, scanned by several host threads.  The file is cut at RECORD starts so that every thread sees whole
// records in file order: FASTQ records are exactly four lines (bio 0.10.0), so a record starts at every line whose
// index is a multiple of 4 (line indices come from a parallel newline count); FASTA records start at lines that
// begin with '>'; BFCounter input is one record per line.  Results are appended in file order, and the error of
// the first failing piece (in file order) is the error a sequential scan would have hit first.
int scan_parallel(const katome_settings* s, const uint8_t* p, size_t n, HostReads& out) {
    unsigned T = std::thread::hardware_concurrency();
    if (const char* e = getenv("KATOME_INGEST_THREADS")) T = (unsigned)atoi(e);
    size_t min_chunk = 4u << 20;
    if (const char* e = getenv("KATOME_INGEST_MIN_CHUNK")) min_chunk = std::max<size_t>(1, (size_t)atoll(e));
    T = std::max(1u, std::min({T, 32u, (unsigned)(n / min_chunk) + 1u}));
    if (T == 1) return scan_range(s, p, n, out);
    std::vector<size_t> raw(T + 1), cut(T + 1);
    for (unsigned c = 0; c <= T; ++c) raw[c] = (size_t)((unsigned __int128)n * c / T);
    std::vector<uint64_t> newlines(T, 0);
    if (s->file_type == 1) {
        std::vector<std::thread> th;
        for (unsigned c = 0; c < T; ++c)
            th.emplace_back([&, c] {
                uint64_t cnt = 0;
                const uint8_t* q = p + raw[c]; const uint8_t* end = p + raw[c + 1];
                while (q < end && (q = (const uint8_t*)memchr(q, '\n', (size_t)(end - q)))) { ++cnt; ++q; }
                newlines[c] = cnt;
            });
        for (auto& t : th) t.join();
    }
    cut[0] = 0; cut[T] = n;
    uint64_t lines_before = 0;                                  // newlines in [0, raw[c])
    for (unsigned c = 1; c < T; ++c) {
        lines_before += newlines[c - 1];
        size_t pos = raw[c];
        uint64_t line = lines_before;                          // index of the line that contains byte `pos`
        if (pos > 0 && p[pos - 1] != '\n') {                   // mid-line: go to the next line start
            const uint8_t* nl = (const uint8_t*)memchr(p + pos, '\n', n - pos);
            pos = nl ? (size_t)(nl - p) + 1 : n;
            ++line;
        }
        if (s->file_type == 1) {
            while (pos < n && line % 4 != 0) {
                const uint8_t* nl = (const uint8_t*)memchr(p + pos, '\n', n - pos);
                pos = nl ? (size_t)(nl - p) + 1 : n;
                ++line;
            }
        } else if (s->file_type == 0) {
            while (pos < n && p[pos] != '>') {
                const uint8_t* nl = (const uint8_t*)memchr(p + pos, '\n', n - pos);
                pos = nl ? (size_t)(nl - p) + 1 : n;
            }
        }
        cut[c] = std::max(pos, cut[c - 1]);
    }
    struct Piece { HostReads r; int rc = KATOME_OK; std::string err; };
    std::vector<Piece> pieces(T);
    {
        std::vector<std::thread> th;
        for (unsigned c = 0; c < T; ++c)
            th.emplace_back([&, c] {
                Piece& pc = pieces[c];
                if (cut[c + 1] <= cut[c]) return;
                pc.rc = reserve_reads(pc.r, 0);
                if (!pc.rc) { pc.r.byte_off[0] = 0; pc.rc = scan_range(s, p + cut[c], cut[c + 1] - cut[c], pc.r); }
                if (pc.rc) pc.err = get_error();
            });
        for (auto& t : th) t.join();
    }
    for (unsigned c = 0; c < T; ++c) {
        // everything before the first failing piece was accepted by the sequential scan too (byte totals included)
        if (pieces[c].rc) {
            for (unsigned d = 0; d <= c; ++d) { out.n_records += pieces[d].r.n_records; out.read_bytes += pieces[d].r.read_bytes; }
            set_error("%s", pieces[c].err.c_str());
            return pieces[c].rc;
        }
    }
    // all pieces are good: size the output once and let every thread copy its own piece into place
    std::vector<uint64_t> n0(T + 1, out.n_reads), b0(T + 1, out.packed_bytes);
    for (unsigned c = 0; c < T; ++c) { n0[c + 1] = n0[c] + pieces[c].r.n_reads; b0[c + 1] = b0[c] + pieces[c].r.packed_bytes; }
    const bool weighted = s->file_type == 2;
    if (n0[T] + 2 > out.cap_reads) {
        const uint64_t nc = n0[T] + 2;
        uint64_t* bo = (uint64_t*)realloc(out.byte_off, nc * sizeof(uint64_t));
        if (!bo) { set_error("out of host memory"); return KATOME_E_OOM; }
        out.byte_off = bo;
        uint32_t* ln = (uint32_t*)realloc(out.len, nc * sizeof(uint32_t));
        if (!ln) { set_error("out of host memory"); return KATOME_E_OOM; }
        out.len = ln;
        out.cap_reads = nc;
    }
    KCHECK(reserve_reads(out, b0[T] - out.packed_bytes));
    if (weighted) {
        uint32_t* nw = (uint32_t*)realloc(out.weight, (n0[T] + 1) * sizeof(uint32_t));
        if (!nw) { set_error("out of host memory"); return KATOME_E_OOM; }
        out.weight = nw;
    }
    {
        std::vector<std::thread> th;
        for (unsigned c = 0; c < T; ++c)
            th.emplace_back([&, c] {
                const HostReads& part = pieces[c].r;
                if (part.packed_bytes) memcpy(out.packed + b0[c], part.packed, part.packed_bytes);
                for (uint64_t i = 0; i < part.n_reads; ++i) { out.byte_off[n0[c] + i] = b0[c] + part.byte_off[i]; out.len[n0[c] + i] = part.len[i]; }
                if (weighted && part.n_reads) memcpy(out.weight + n0[c], part.weight, part.n_reads * sizeof(uint32_t));
            });
        for (auto& t : th) t.join();
    }
    for (unsigned c = 0; c < T; ++c) {
        const HostReads& part = pieces[c].r;
        if (part.n_reads) {
            if (out.n_reads == 0) { out.fixed_len = part.fixed_len; out.all_fixed = part.all_fixed; }
            else if (!part.all_fixed || part.fixed_len != out.fixed_len) out.all_fixed = false;
        }
        out.n_reads += part.n_reads;
        out.n_records += part.n_records; out.read_bytes += part.read_bytes; out.total_windows += part.total_windows;
    }
    out.packed_bytes = b0[T];
    out.byte_off[out.n_reads] = out.packed_bytes;
    return KATOME_OK;
}

}  // namespace

int ingest_files(const katome_settings* s, const char* const* paths, size_t n_paths, HostReads& out) {
    KCHECK(check_k(s->k));
    if (s->file_type > 2) { set_error("unknown input file type %u", s->file_type); return KATOME_E_ARG; }
    // check_files (builder.rs:57-77): every path is vetted before any file is opened
    std::vector<std::string> files;
    for (size_t i = 0; i < n_paths; ++i) {
        char resolved[PATH_MAX];
        if (!realpath(paths[i], resolved)) { set_error("Coulndt resolve path: %s", paths[i]); return KATOME_E_PATH; }
        struct stat st;
        if (stat(resolved, &st) != 0) { set_error("%s does not exist", resolved); return KATOME_E_NOT_EXIST; }
        if (S_ISDIR(st.st_mode)) { set_error("%s is a directory", resolved); return KATOME_E_IS_DIR; }
        files.push_back(resolved);
    }
    KCHECK(reserve_reads(out, 0));
    out.byte_off[0] = 0;
    std::vector<Mapped> maps(files.size());
    for (size_t i = 0; i < files.size(); ++i) {            // Reader::from_file for all inputs first (builder.rs:146-149)
        Mapped& m = maps[i];
        m.fd = open(files[i].c_str(), O_RDONLY);
        struct stat st;
        if (m.fd < 0 || fstat(m.fd, &st) != 0) { set_error("Couldn't open all files: %s", files[i].c_str()); return KATOME_E_OPEN; }
        m.n = (size_t)st.st_size;
        if (m.n) {
            void* p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
            if (p == MAP_FAILED) { m.n = 0; set_error("Couldn't open all files: %s", files[i].c_str()); return KATOME_E_OPEN; }
            m.p = (const uint8_t*)p;
            madvise(p, m.n, MADV_SEQUENTIAL);
        }
    }
    for (size_t i = 0; i < files.size(); ++i) KCHECK(scan_parallel(s, maps[i].p, maps[i].n, out));
    if (!out.all_fixed || out.n_reads == 0) out.fixed_len = 0;
    memset(out.packed + out.packed_bytes, 0, 32);          // slack for vector loads
    return KATOME_OK;
}

}  // namespace katome
