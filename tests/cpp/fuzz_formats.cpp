// fuzz_formats.cpp -- sanitizer + mutation harness for everything behind `ModelCmd::Load` that parses untrusted bytes on
// the host (SURVEY section 5 "race detection / sanitizers": the build's equivalent of the reference's clippy-only hygiene;
// infur/src/predict_onnx.rs:288-309: a load error is a `Result`, never fatal):
//   * onnx_reader.cpp / onnx_qreader.cpp -- the protobuf wire-format reader and the two graph walkers (float models,
//                         QOperator int8 models) behind infur_onnx_to_blob
//   * blob_dir.h       -- header + directory of the INFURW01 and INFURQ01 blobs (what model_load_dev / model_load_q_dev
//                         check before touching the GPU)
// Built by `make -C infur_amd/csrc asan` with g++ -fsanitize=address,undefined -fno-sanitize-recover; no HIP, no GPU.
//
//   fuzz_formats <onnx|blob> <file> <mutations> <seed>
// applies <mutations> seeded single-byte / multi-byte length-field / truncation mutations IN PLACE (each undone before the
// next), parses, and requires: no crash, no sanitizer report, and either a format error or an accepted file whose produced
// blob passes the blob checks.  Prints "accepted=<n> rejected=<m>" and exits 0.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sanitizer/asan_interface.h>

#include "../../infur_amd/csrc/blob_dir.h"
#include "../../infur_amd/csrc/onnx_reader.h"

using namespace infur;

static uint64_t rng_state;
static uint64_t rnd() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static bool check_blob(const uint8_t* b, size_t len, std::string* err) {
    BlobHeader h;
    std::vector<ConvSpec> g;
    if (!blob_parse_header(b, len, &h, &g, err)) return false;
    std::vector<BlobEntry> ents;
    if (!blob_parse_directory(b + kBlobHdr, len, g, &ents, err)) return false;
    // every tensor the runtime would copy lies inside the blob: touch first and last byte (ASan sees an overrun)
    volatile uint8_t sink = 0;
    for (size_t i = 0; i < g.size(); i++) {
        const size_t wn = (size_t)g[i].cout * g[i].cin * g[i].k * g[i].k * 4, bn = (size_t)g[i].cout * 4;
        sink ^= b[ents[i].w_off];
        sink ^= b[ents[i].w_off + wn - 1];
        sink ^= b[ents[i].b_off];
        sink ^= b[ents[i].b_off + bn - 1];
    }
    (void)sink;
    return true;
}

// INFURQ01 (quantised models): the same, with the three tensors per conv the loader copies
static bool check_qblob(const uint8_t* b, size_t len, std::string* err) {
    BlobHeader h;
    std::vector<ConvSpec> g;
    uint32_t n_adds = 0;
    if (!qblob_parse_header(b, len, &h, &n_adds, &g, err)) return false;
    std::vector<QBlobConv> qc;
    std::vector<QBlobAdd> qa;
    if (!qblob_parse_directory(b + kBlobHdr, len, g, n_adds, &qc, &qa, err)) return false;
    volatile uint8_t sink = 0;
    for (size_t i = 0; i < g.size(); i++) {
        const size_t wn = (size_t)g[i].cout * g[i].cin * g[i].k * g[i].k, cn = (size_t)g[i].cout * 4;
        sink ^= b[qc[i].w_off];
        sink ^= b[qc[i].w_off + wn - 1];
        sink ^= b[qc[i].ws_off];
        sink ^= b[qc[i].ws_off + cn - 1];
        sink ^= b[qc[i].b_off];
        sink ^= b[qc[i].b_off + cn - 1];
    }
    (void)sink;
    return true;
}

// the dispatch of infur_model_load_blob: by magic
static bool check_any_blob(const uint8_t* b, size_t len, std::string* err) {
    return len >= 8 && memcmp(b, "INFURQ01", 8) == 0 ? check_qblob(b, len, err) : check_blob(b, len, err);
}

static bool parse(bool onnx, const uint8_t* d, size_t len, std::string* err) {
    if (!onnx) return check_any_blob(d, len, err);
    static std::vector<uint8_t> blob;  // (reused: a fresh 141 MB allocation per accepted file is all page faults under ASan)
    OnnxInfo info;
    if (onnx_to_blob(d, len, blob, info, *err) != 0) return false;
    std::string e2;
    if (!check_any_blob(blob.data(), blob.size(), &e2)) {
        fprintf(stderr, "the reader accepted a file but produced a blob that fails its own checks: %s\n", e2.c_str());
        abort();
    }
    return true;
}

// ---- where the structure of an ONNX file lives: everything except the interior of the big length-delimited payloads ----
struct Range { size_t lo, hi; };
// tag bytes of the `input` (field 1) / `output` (field 2) strings of every NodeProto: turning one into the tag of an unknown
// length-delimited field (15) makes the node LOSE that input / output while the file stays well-formed protobuf -- the shape
// behind ADVICE r3's two crashes (a MaxPool / QuantizeLinear without outputs), which random byte noise almost never produces
static std::vector<size_t> node_io_tags;
static bool rd_varint(const uint8_t* d, size_t len, size_t* p, uint64_t* v) {
    *v = 0;
    for (int sh = 0; sh < 64 && *p < len; sh += 7) {
        const uint8_t b = d[(*p)++];
        *v |= (uint64_t)(b & 0x7f) << sh;
        if (!(b & 0x80)) return true;
    }
    return false;
}
// walks one message; descends into length-delimited fields listed in `into` (field numbers per depth), records every byte
// range that is NOT the interior of a payload longer than 256 bytes
static void walk(const uint8_t* d, size_t lo, size_t hi, int depth, std::vector<Range>* out, bool in_node = false) {
    size_t p = lo, run = lo;
    while (p < hi) {
        uint64_t key, v;
        const size_t tag_pos = p;
        if (!rd_varint(d, hi, &p, &key)) break;
        const int wt = (int)(key & 7), field = (int)(key >> 3);
        if (wt == 0) { if (!rd_varint(d, hi, &p, &v)) break; }
        else if (wt == 1) p += 8;
        else if (wt == 5) p += 4;
        else if (wt == 2) {
            if (!rd_varint(d, hi, &p, &v) || v > hi - p) break;
            const bool sub = (depth == 0 && field == 7) || (depth == 1 && (field == 1 || field == 5 || field == 11 || field == 12));  // graph; node, initializer, input, output
            if (in_node && (field == 1 || field == 2) && p - tag_pos == 2) node_io_tags.push_back(tag_pos);  // (one-byte tag, one-byte length)
            if (sub) {
                out->push_back({run, p});
                walk(d, p, p + (size_t)v, depth + 1, out, depth == 1 && field == 1);
                run = p + (size_t)v;
            } else if (v > 256) {  // raw_data and friends: keep 4 bytes at either end, skip the interior
                out->push_back({run, p + 4});
                run = p + (size_t)v - 4;
            }
            p += (size_t)v;
        } else break;
    }
    if (run < hi) out->push_back({run, hi});
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <onnx|blob> <file> <mutations> <seed>\n", argv[0]);
        return 2;
    }
    const bool onnx = strcmp(argv[1], "onnx") == 0;
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    // exact-size heap buffer: a read past the end of the (possibly truncated) input is an ASan report
    std::vector<uint8_t> full((size_t)sz);
    if (fread(full.data(), 1, (size_t)sz, f) != (size_t)sz) return 2;
    fclose(f);
    const long n_mut = atol(argv[3]);
    rng_state = strtoull(argv[4], nullptr, 0);
    std::string err;
    if (!parse(onnx, full.data(), full.size(), &err)) {
        fprintf(stderr, "the unmutated file is rejected: %s\n", err.c_str());
        return 3;
    }
    // Structural bytes sit at the front of a blob (header + directory) and all over an ONNX file (node / tensor headers between
    // the raw_data payloads); half of the mutations go to the first 64 KB, half anywhere (an ONNX exporter writes the
    // graph's nodes before the initializers, or after them: both ends are covered by "anywhere" over many seeds).
    std::vector<Range> hot;
    if (onnx) walk(full.data(), 0, full.size(), 0, &hot);
    size_t hot_bytes = 0;
    for (const Range& r : hot) hot_bytes += r.hi - r.lo;
    auto hot_pos = [&](uint64_t q) {
        size_t k = (size_t)(q % (hot_bytes ? hot_bytes : 1));
        for (const Range& r : hot) {
            if (k < r.hi - r.lo) return r.lo + k;
            k -= r.hi - r.lo;
        }
        return (size_t)0;
    };
    long accepted = 0, rejected = 0;
    const uint8_t extremes[] = {0x00, 0x01, 0x7f, 0x80, 0xff, 0xfe, 0x0a, 0x12};
    for (long it = 0; it < n_mut; it++) {
        const uint64_t r = rnd();
        int kind = (int)(r % 11);
        if (kind == 10 && (!onnx || node_io_tags.empty())) kind = 3;
        const size_t span = (r >> 8) & 1 ? full.size() : (full.size() < 65536 ? full.size() : 65536);
        // ONNX: 15 of 16 mutations inside the structural byte ranges (node / tensor / value-info headers and the edges of the
        // payloads), the rest anywhere; blob: half in the first 64 KB (header + directory), half anywhere
        const size_t pos = onnx && hot_bytes && ((r >> 12) & 15) ? hot_pos(rnd()) : (size_t)(rnd() % span);
        err.clear();
        bool ok;
        if (kind == 10) {  // structural: one or two nodes lose an input / output (tag -> unknown field 15, same length)
            size_t at[2];
            uint8_t sv[2];
            const int n = 1 + (int)(rnd() & 1);
            for (int k = 0; k < n; k++) {
                at[k] = node_io_tags[(size_t)(rnd() % node_io_tags.size())];
                sv[k] = full[at[k]];
                full[at[k]] = 0x7a;
            }
            ok = parse(onnx, full.data(), full.size(), &err);
            for (int k = n - 1; k >= 0; k--) full[at[k]] = sv[k];
        } else if (kind == 0) {  // truncation: the cut-off tail is poisoned, so a read past the new end is an ASan report
            const size_t cut = (r >> 9) & 1 ? pos : full.size() - 1 - (size_t)(rnd() % (full.size() < 4096 ? full.size() : 4096));
            const size_t pl = (cut + 7) & ~(size_t)7;  // (poisoning is 8-byte granular: the first partial granule stays readable)
            if (pl < full.size()) ASAN_POISON_MEMORY_REGION(full.data() + pl, full.size() - pl);
            ok = parse(onnx, full.data(), cut, &err);
            if (pl < full.size()) ASAN_UNPOISON_MEMORY_REGION(full.data() + pl, full.size() - pl);
        } else {
            uint8_t saved[8];
            const size_t nb = kind <= 5 ? 1 : (kind <= 7 ? 2 : (kind == 8 ? 4 : 8));
            const size_t n = pos + nb <= full.size() ? nb : full.size() - pos;
            memcpy(saved, &full[pos], n);
            for (size_t k = 0; k < n; k++) {
                const uint64_t q = rnd();
                full[pos + k] = kind <= 2 ? (uint8_t)(saved[k] ^ (1u << (q & 7)))       // bit flip
                                          : (kind <= 5 ? (uint8_t)q                       // random byte
                                                       : extremes[q % sizeof extremes]);  // length / varint extremes
            }
            ok = parse(onnx, full.data(), full.size(), &err);
            memcpy(&full[pos], saved, n);
        }
        if (ok) {
            accepted++;
        } else {
            rejected++;
            if (err.empty()) {
                fprintf(stderr, "mutation %ld rejected without a message\n", it);
                abort();
            }
        }
    }
    printf("accepted=%ld rejected=%ld\n", accepted, rejected);
    return 0;
}
