// blob_dir.h -- the INFURW01 weight blob's header and conv directory: parsing and every check that can be made without
// the device.  Host-only and header-only on purpose: the runtime (infur_capi.cpp: model_load_dev) and the sanitizer /
// mutation harness (tests/cpp/fuzz_formats.cpp, `make asan`) compile the SAME code, so what the harness hammers is what
// `ModelCmd::Load` runs on untrusted bytes (infur/src/predict_onnx.rs:288-309: load errors are `Result`s, never fatal).
//
//   header (32 bytes): "INFURW01", u32 depth, u32 num_classes, u32 has_aux, u32 n_convs, u32 input_kind, u32 reserved
//   directory: n_convs entries of 80 bytes: char name[40], u32 shape[4] (O, I, KH, KW), u64 weight offset, u64 bias offset
//   then the f32 tensors (OIHW weights, biases), at the offsets the directory names
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace infur {

constexpr size_t kBlobHdr = 32, kBlobEntry = 80;

// one convolution of torchvision's fcn_resnet{50,101}, output stride 8 (the graph the blob must describe)
struct ConvSpec {
    std::string name;
    int cout = 0, cin = 0, k = 0, stride = 1, pad = 0, dil = 1;
    bool relu = false;
    char role = 0;  // s stem, 1 2 3 block convs, d downsample, h head3x3, c classifier
};

inline bool layer_blocks(int depth, int lb[4]) {
    if (depth == 50) { lb[0] = 3; lb[1] = 4; lb[2] = 6; lb[3] = 3; return true; }
    if (depth == 101) { lb[0] = 3; lb[1] = 4; lb[2] = 23; lb[3] = 3; return true; }
    return false;
}

inline std::vector<ConvSpec> graph_spec(int depth, int ncls, bool aux) {
    std::vector<ConvSpec> g;
    int lb[4];
    if (!layer_blocks(depth, lb)) return g;
    auto add = [&](const std::string& n, int cout, int cin, int k, int s, int p, int d, bool relu, char role) {
        ConvSpec c;
        c.name = n; c.cout = cout; c.cin = cin; c.k = k; c.stride = s; c.pad = p; c.dil = d;
        c.relu = relu; c.role = role;
        g.push_back(c);
    };
    add("backbone.conv1", 64, 3, 7, 2, 3, 1, true, 's');
    int inplanes = 64, dilation = 1;
    for (int L = 0; L < 4; L++) {
        const int planes = 64 << L;
        int stride = L == 0 ? 1 : 2;
        const int prev = dilation;
        if (L >= 2) {  // replace_stride_with_dilation = [False, True, True]
            dilation *= stride;
            stride = 1;
        }
        for (int b = 0; b < lb[L]; b++) {
            const int bs = b == 0 ? stride : 1, bd = b == 0 ? prev : dilation;
            const std::string p = "backbone.layer" + std::to_string(L + 1) + "." + std::to_string(b);
            add(p + ".conv1", planes, inplanes, 1, 1, 0, 1, true, '1');
            add(p + ".conv2", planes, planes, 3, bs, bd, bd, true, '2');
            add(p + ".conv3", planes * 4, planes, 1, 1, 0, 1, true, '3');
            if (b == 0) add(p + ".downsample.0", planes * 4, inplanes, 1, bs, 0, 1, false, 'd');
            inplanes = planes * 4;
        }
    }
    add("classifier.0", 512, 2048, 3, 1, 1, 1, true, 'h');
    add("classifier.4", ncls, 512, 1, 1, 0, 1, false, 'c');
    if (aux) {
        add("aux_classifier.0", 256, 1024, 3, 1, 1, 1, true, 'h');
        add("aux_classifier.4", ncls, 256, 1, 1, 0, 1, false, 'c');
    }
    return g;
}

struct BlobHeader {
    int depth = 0, num_classes = 0;
    bool aux = false, input_u8 = false;
    uint32_t n_convs = 0;
    bool resize_u8 = false;  // INFURQ01 flag bit 0: the file resizes the u8 logits before it dequantises them
};

struct BlobEntry {
    uint64_t w_off = 0, b_off = 0;
};

inline std::string blob_msg(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return buf;
}

// hdr: the first kBlobHdr bytes (the caller checked len >= kBlobHdr); len = size of the whole blob.
// On success *graph is the conv list the directory must match and the directory is (*graph).size() * kBlobEntry bytes.
inline bool blob_parse_header(const uint8_t* hdr, size_t len, BlobHeader* h, std::vector<ConvSpec>* graph, std::string* err) {
    if (len < kBlobHdr) { *err = blob_msg("weight blob too short (%zu bytes)", len); return false; }
    if (memcmp(hdr, "INFURW01", 8) != 0) { *err = "bad magic: not an INFURW01 weight blob"; return false; }
    uint32_t h32[6];
    memcpy(h32, hdr + 8, 24);
    // the fields are untrusted: compare as unsigned before anything is narrowed to int
    if (h32[0] != 50 && h32[0] != 101) { *err = blob_msg("unsupported backbone depth %u (50 or 101)", h32[0]); return false; }
    if (h32[1] == 0 || h32[1] > 256) { *err = blob_msg("unsupported class count %u", h32[1]); return false; }
    // input kind: 0 = Float image input (RGB planes normalised with the torchvision constants), 1 = Uint8 (the bytes
    // themselves, BGR kept) -- the two ColorRange arms of ImageSession::forward (predict_onnx.rs:114-139)
    if (h32[4] > 1) { *err = blob_msg("unknown input kind %u in the weight blob (0 = Float, 1 = Uint8)", h32[4]); return false; }
    h->depth = (int)h32[0];
    h->num_classes = (int)h32[1];
    h->aux = h32[2] != 0;
    h->n_convs = h32[3];
    h->input_u8 = h32[4] == 1;
    *graph = graph_spec(h->depth, h->num_classes, h->aux);
    if (h->n_convs != graph->size()) { *err = blob_msg("blob has %u convs, graph needs %zu", h->n_convs, graph->size()); return false; }
    if ((len - kBlobHdr) / kBlobEntry < h->n_convs) { *err = "truncated conv table"; return false; }
    return true;
}

// table: graph.size() * kBlobEntry bytes following the header.  Checks names, shapes, alignment and that every tensor lies
// inside the blob (offsets compared without ever forming off + n, which could wrap).
inline bool blob_parse_directory(const uint8_t* table, size_t len, const std::vector<ConvSpec>& graph, std::vector<BlobEntry>* ents,
                                 std::string* err) {
    ents->assign(graph.size(), BlobEntry());
    auto in_range = [len](uint64_t off, size_t n) { return off <= len && n <= len - off; };
    for (size_t i = 0; i < graph.size(); i++) {
        const uint8_t* e = table + i * kBlobEntry;
        char name[41];
        memcpy(name, e, 40);
        name[40] = 0;
        uint32_t d[4];
        memcpy(d, e + 40, 16);
        BlobEntry& en = (*ents)[i];
        memcpy(&en.w_off, e + 56, 8);
        memcpy(&en.b_off, e + 64, 8);
        const ConvSpec& L = graph[i];
        if (L.name != name) {
            for (char* p = name; *p; p++)
                if ((unsigned char)*p < 0x20 || (unsigned char)*p > 0x7e) *p = '?';  // the name is untrusted bytes: keep the message printable
            *err = blob_msg("conv %zu is '%s', expected '%s'", i, name, L.name.c_str());
            return false;
        }
        if (d[0] != (uint32_t)L.cout || d[1] != (uint32_t)L.cin || d[2] != (uint32_t)L.k || d[3] != (uint32_t)L.k) {
            *err = blob_msg("conv '%s' has shape [%u,%u,%u,%u], expected [%d,%d,%d,%d]", L.name.c_str(), d[0], d[1], d[2], d[3], L.cout, L.cin, L.k, L.k);
            return false;
        }
        const size_t wn = (size_t)L.cout * L.cin * L.k * L.k * 4, bn = (size_t)L.cout * 4;
        if (en.w_off % 4 || en.b_off % 4 || !in_range(en.w_off, wn) || !in_range(en.b_off, bn)) {
            *err = blob_msg("conv '%s' data out of range", L.name.c_str());
            return false;
        }
    }
    return true;
}

// ---- INFURQ01: the quantised (QOperator) form of the same network (infur_amd/weights.py documents the layout) ----
constexpr size_t kQEntry = 96, kQAdd = 24;

struct QBlobConv {
    float x_scale = 0.f, y_scale = 0.f;
    int32_t x_zp = 0, y_zp = 0;
    uint64_t w_off = 0, ws_off = 0, b_off = 0;
};
struct QBlobAdd {
    float a_scale = 0.f, b_scale = 0.f, c_scale = 0.f;
    int32_t a_zp = 0, b_zp = 0, c_zp = 0;
};

inline int graph_adds(int depth) {
    int lb[4];
    return layer_blocks(depth, lb) ? lb[0] + lb[1] + lb[2] + lb[3] : 0;
}

inline bool qscale_ok(float s) { return s > 0.0f && s < 1e30f; }  // (NaN fails both comparisons)

// hdr: first kBlobHdr bytes; len: whole blob.  *n_adds = residual sums the directory lists after the convs.
inline bool qblob_parse_header(const uint8_t* hdr, size_t len, BlobHeader* h, uint32_t* n_adds, std::vector<ConvSpec>* graph, std::string* err) {
    if (len < kBlobHdr) { *err = blob_msg("weight blob too short (%zu bytes)", len); return false; }
    if (memcmp(hdr, "INFURQ01", 8) != 0) { *err = "bad magic: not an INFURQ01 quantised weight blob"; return false; }
    uint32_t h32[6];
    memcpy(h32, hdr + 8, 24);
    if (h32[0] != 50 && h32[0] != 101) { *err = blob_msg("unsupported backbone depth %u (50 or 101)", h32[0]); return false; }
    if (h32[1] == 0 || h32[1] > 256) { *err = blob_msg("unsupported class count %u", h32[1]); return false; }
    h->depth = (int)h32[0];
    h->num_classes = (int)h32[1];
    h->aux = h32[2] != 0;
    h->n_convs = h32[3];
    h->input_u8 = false;
    // flags (offset 28): bit 0 = Resize runs on the u8 logits, DequantizeLinear after it (QLinearConv -> Resize -> DequantizeLinear)
    if (h32[5] > 1) { *err = blob_msg("unknown flags %#x in the quantised weight blob", h32[5]); return false; }
    h->resize_u8 = (h32[5] & 1) != 0;
    *n_adds = h32[4];
    *graph = graph_spec(h->depth, h->num_classes, h->aux);
    if (h->n_convs != graph->size()) { *err = blob_msg("blob has %u convs, graph needs %zu", h->n_convs, graph->size()); return false; }
    if (*n_adds != (uint32_t)graph_adds(h->depth)) { *err = blob_msg("blob has %u residual sums, graph needs %d", *n_adds, graph_adds(h->depth)); return false; }
    if ((len - kBlobHdr) / kQEntry < h->n_convs || (len - kBlobHdr - h->n_convs * kQEntry) / kQAdd < *n_adds) { *err = "truncated directory"; return false; }
    return true;
}

// table: n_convs * kQEntry + n_adds * kQAdd bytes after the header
inline bool qblob_parse_directory(const uint8_t* table, size_t len, const std::vector<ConvSpec>& graph, uint32_t n_adds,
                                  std::vector<QBlobConv>* convs, std::vector<QBlobAdd>* adds, std::string* err) {
    convs->assign(graph.size(), QBlobConv());
    adds->assign(n_adds, QBlobAdd());
    auto in_range = [len](uint64_t off, size_t n) { return off <= len && n <= len - off; };
    auto zp_ok = [](int32_t z) { return z >= 0 && z <= 255; };
    for (size_t i = 0; i < graph.size(); i++) {
        const uint8_t* e = table + i * kQEntry;
        char name[41];
        memcpy(name, e, 40);
        name[40] = 0;
        uint32_t d[4];
        memcpy(d, e + 40, 16);
        QBlobConv& q = (*convs)[i];
        memcpy(&q.x_scale, e + 56, 4);
        memcpy(&q.x_zp, e + 60, 4);
        memcpy(&q.y_scale, e + 64, 4);
        memcpy(&q.y_zp, e + 68, 4);
        memcpy(&q.w_off, e + 72, 8);
        memcpy(&q.ws_off, e + 80, 8);
        memcpy(&q.b_off, e + 88, 8);
        const ConvSpec& L = graph[i];
        if (L.name != name) {
            for (char* p = name; *p; p++)
                if ((unsigned char)*p < 0x20 || (unsigned char)*p > 0x7e) *p = '?';
            *err = blob_msg("conv %zu is '%s', expected '%s'", i, name, L.name.c_str());
            return false;
        }
        if (d[0] != (uint32_t)L.cout || d[1] != (uint32_t)L.cin || d[2] != (uint32_t)L.k || d[3] != (uint32_t)L.k) {
            *err = blob_msg("conv '%s' has shape [%u,%u,%u,%u], expected [%d,%d,%d,%d]", L.name.c_str(), d[0], d[1], d[2], d[3], L.cout, L.cin, L.k, L.k);
            return false;
        }
        if (!qscale_ok(q.x_scale) || !qscale_ok(q.y_scale) || !zp_ok(q.x_zp) || !zp_ok(q.y_zp)) {
            *err = blob_msg("conv '%s': scales must be positive and finite, zero points in 0..255 (u8 activations)", L.name.c_str());
            return false;
        }
        // QLinearConv pads with the zero point; this path pads with the byte 0 (hardware bounds check): the same thing exactly
        // when the input's zero point is 0 -- true behind every ReLU, i.e. for every padded convolution but the stem, which
        // has its own kernel
        if (L.pad != 0 && L.role != 's' && q.x_zp != 0) {
            *err = blob_msg("conv '%s' pads an input whose zero point is %d: only 0 is supported there", L.name.c_str(), q.x_zp);
            return false;
        }
        const size_t wn = (size_t)L.cout * L.cin * L.k * L.k, cn = (size_t)L.cout * 4;
        if (q.ws_off % 4 || q.b_off % 4 || !in_range(q.w_off, wn) || !in_range(q.ws_off, cn) || !in_range(q.b_off, cn)) {
            *err = blob_msg("conv '%s' data out of range", L.name.c_str());
            return false;
        }
    }
    for (uint32_t i = 0; i < n_adds; i++) {
        const uint8_t* e = table + graph.size() * kQEntry + (size_t)i * kQAdd;
        QBlobAdd& a = (*adds)[i];
        memcpy(&a.a_scale, e, 4); memcpy(&a.a_zp, e + 4, 4);
        memcpy(&a.b_scale, e + 8, 4); memcpy(&a.b_zp, e + 12, 4);
        memcpy(&a.c_scale, e + 16, 4); memcpy(&a.c_zp, e + 20, 4);
        if (!qscale_ok(a.a_scale) || !qscale_ok(a.b_scale) || !qscale_ok(a.c_scale) || !zp_ok(a.a_zp) || !zp_ok(a.b_zp) || !zp_ok(a.c_zp)) {
            *err = blob_msg("residual sum %u: scales must be positive and finite, zero points in 0..255", i);
            return false;
        }
    }
    return true;
}

}  // namespace infur
