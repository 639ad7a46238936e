// onnx_reader.cpp -- ModelCmd::Load("...fcn-resnet50-12.onnx") without ONNX Runtime.
//
// The reference hands the model file to ONNX Runtime (infur/src/predict_onnx.rs:288-293) and
// then inspects input 0 to infer the image layout (infer_img_pre_proc, :223-265).  This file is
// the minimum needed to accept the same file here: a hand-written protobuf wire-format reader
// (ModelProto -> GraphProto -> node / initializer / input / output) that
//   * applies the reference's input checks with the reference's messages (:228-262),
//   * walks the Conv nodes in graph order -- torchvision's FCN traces as stem, then per
//     bottleneck conv1, conv2, conv3, downsample, then classifier and aux_classifier, which is
//     exactly the INFURW01 order -- folding a trailing BatchNormalization when the exporter
//     did not, and checks every shape / stride / dilation against the expected graph,
//   * emits the INFURW01 blob that infur_model_load_blob consumes.
// No onnx / protobuf library is used (none exists in the build image).  Host only.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "onnx_reader.h"

namespace infur {
namespace {

struct PB {  // protobuf wire reader over [p, end)
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    PB(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    bool done() const { return p >= end || !ok; }
    uint64_t varint() {
        uint64_t v = 0;
        int shift = 0;
        while (p < end && shift < 64) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
        }
        ok = false;
        return 0;
    }
    // reads one field header + payload; for length-delimited fields sub = [data, len)
    bool next(uint32_t& field, uint32_t& wt, uint64_t& val, const uint8_t*& data, size_t& len) {
        if (done()) return false;
        const uint64_t key = varint();
        if (!ok) return false;
        field = (uint32_t)(key >> 3);
        wt = (uint32_t)(key & 7);
        data = nullptr;
        len = 0;
        val = 0;
        switch (wt) {
            case 0: val = varint(); break;
            case 1:
                if (end - p < 8) { ok = false; return false; }
                memcpy(&val, p, 8); p += 8; break;
            case 2: {
                const uint64_t n = varint();
                if (!ok || (uint64_t)(end - p) < n) { ok = false; return false; }
                data = p; len = (size_t)n; p += n; break;
            }
            case 5: {
                if (end - p < 4) { ok = false; return false; }
                uint32_t v32; memcpy(&v32, p, 4); val = v32; p += 4; break;
            }
            default: ok = false; return false;
        }
        return ok;
    }
};

struct Tensor {
    std::vector<int64_t> dims;
    int dtype = 0;
    const uint8_t* raw = nullptr;
    size_t raw_len = 0;
    std::vector<float> fdata;  // float_data (field 4)
    size_t count() const {
        size_t n = 1;
        for (auto d : dims) n *= (size_t)d;
        return n;
    }
    bool floats(std::vector<float>& out) const {
        const size_t n = count();
        if (dtype != 1) return false;  // FLOAT
        if (raw && raw_len == n * 4) {
            out.resize(n);
            memcpy(out.data(), raw, n * 4);
            return true;
        }
        if (fdata.size() == n) {
            out = fdata;
            return true;
        }
        return false;
    }
};

struct Node {
    std::string op;
    std::vector<std::string> in, out;
    std::map<std::string, std::vector<int64_t>> ints;
    std::map<std::string, float> f;
};

void read_packed_i64(const uint8_t* d, size_t n, std::vector<int64_t>& v) {
    PB r(d, n);
    while (!r.done()) v.push_back((int64_t)r.varint());
}

bool parse_tensor(const uint8_t* d, size_t n, std::string& name, Tensor& t) {
    PB r(d, n);
    uint32_t f, wt; uint64_t v; const uint8_t* s; size_t l;
    while (r.next(f, wt, v, s, l)) {
        if (f == 1) { if (wt == 2) read_packed_i64(s, l, t.dims); else t.dims.push_back((int64_t)v); }
        else if (f == 2) t.dtype = (int)v;
        else if (f == 4) {
            if (wt == 2) { t.fdata.resize(l / 4); memcpy(t.fdata.data(), s, l / 4 * 4); }
            else { float x; uint32_t u = (uint32_t)v; memcpy(&x, &u, 4); t.fdata.push_back(x); }
        }
        else if (f == 8 && wt == 2) name.assign((const char*)s, l);
        else if (f == 9 && wt == 2) { t.raw = s; t.raw_len = l; }
        else if (f == 14 && v != 0) return false;  // external data is not supported
    }
    return r.ok;
}

bool parse_attr(const uint8_t* d, size_t n, Node& node) {
    PB r(d, n);
    uint32_t f, wt; uint64_t v; const uint8_t* s; size_t l;
    std::string name;
    std::vector<int64_t> ints;
    bool has_i = false, has_f = false;
    int64_t iv = 0;
    float fv = 0;
    while (r.next(f, wt, v, s, l)) {
        if (f == 1 && wt == 2) name.assign((const char*)s, l);
        else if (f == 2 && wt == 5) { uint32_t u = (uint32_t)v; memcpy(&fv, &u, 4); has_f = true; }
        else if (f == 3 && wt == 0) { iv = (int64_t)v; has_i = true; }
        else if (f == 8) { if (wt == 2) read_packed_i64(s, l, ints); else ints.push_back((int64_t)v); }
    }
    if (!ints.empty()) node.ints[name] = ints;
    else if (has_i) node.ints[name] = {iv};
    if (has_f) node.f[name] = fv;
    return r.ok;
}

bool parse_node(const uint8_t* d, size_t n, Node& node) {
    PB r(d, n);
    uint32_t f, wt; uint64_t v; const uint8_t* s; size_t l;
    while (r.next(f, wt, v, s, l)) {
        if (wt != 2) continue;
        if (f == 1) node.in.emplace_back((const char*)s, l);
        else if (f == 2) node.out.emplace_back((const char*)s, l);
        else if (f == 4) node.op.assign((const char*)s, l);
        else if (f == 5 && !parse_attr(s, l, node)) return false;
    }
    return r.ok;
}

struct ValueInfo {
    std::string name;
    int elem_type = 0;
    std::vector<int64_t> dims;  // -1 = symbolic
    bool has_shape = false;
};

bool parse_value_info(const uint8_t* d, size_t n, ValueInfo& vi) {
    PB r(d, n);
    uint32_t f, wt; uint64_t v; const uint8_t* s; size_t l;
    while (r.next(f, wt, v, s, l)) {
        if (f == 1 && wt == 2) vi.name.assign((const char*)s, l);
        else if (f == 2 && wt == 2) {  // TypeProto
            PB t(s, l);
            uint32_t f2, w2; uint64_t v2; const uint8_t* s2; size_t l2;
            while (t.next(f2, w2, v2, s2, l2)) {
                if (f2 != 1 || w2 != 2) continue;  // tensor_type
                PB tt(s2, l2);
                uint32_t f3, w3; uint64_t v3; const uint8_t* s3; size_t l3;
                while (tt.next(f3, w3, v3, s3, l3)) {
                    if (f3 == 1 && w3 == 0) vi.elem_type = (int)v3;
                    else if (f3 == 2 && w3 == 2) {  // TensorShapeProto
                        vi.has_shape = true;
                        PB sh(s3, l3);
                        uint32_t f4, w4; uint64_t v4; const uint8_t* s4; size_t l4;
                        while (sh.next(f4, w4, v4, s4, l4)) {
                            if (f4 != 1 || w4 != 2) continue;  // Dimension
                            int64_t dv = -1;
                            PB dm(s4, l4);
                            uint32_t f5, w5; uint64_t v5; const uint8_t* s5; size_t l5;
                            while (dm.next(f5, w5, v5, s5, l5))
                                if (f5 == 1 && w5 == 0) dv = (int64_t)v5;
                            vi.dims.push_back(dv);
                        }
                    }
                }
            }
        }
    }
    return r.ok;
}

struct ExpConv { std::string name; int cout, cin, k, stride, pad, dil; };

std::vector<ExpConv> expected_graph(int depth, int ncls, bool aux) {
    std::vector<ExpConv> g;
    const int lb50[4] = {3, 4, 6, 3}, lb101[4] = {3, 4, 23, 3};
    const int* lb = depth == 50 ? lb50 : lb101;
    g.push_back({"backbone.conv1", 64, 3, 7, 2, 3, 1});
    int inplanes = 64, dilation = 1;
    for (int L = 0; L < 4; L++) {
        const int planes = 64 << L;
        int stride = L == 0 ? 1 : 2;
        const int prev = dilation;
        if (L >= 2) { dilation *= stride; stride = 1; }
        for (int b = 0; b < lb[L]; b++) {
            const int bs = b == 0 ? stride : 1, bd = b == 0 ? prev : dilation;
            const std::string p = "backbone.layer" + std::to_string(L + 1) + "." + std::to_string(b);
            g.push_back({p + ".conv1", planes, inplanes, 1, 1, 0, 1});
            g.push_back({p + ".conv2", planes, planes, 3, bs, bd, bd});
            g.push_back({p + ".conv3", planes * 4, planes, 1, 1, 0, 1});
            if (b == 0) g.push_back({p + ".downsample.0", planes * 4, inplanes, 1, bs, 0, 1});
            inplanes = planes * 4;
        }
    }
    g.push_back({"classifier.0", 512, 2048, 3, 1, 1, 1});
    g.push_back({"classifier.4", ncls, 512, 1, 1, 0, 1});
    if (aux) {
        g.push_back({"aux_classifier.0", 256, 1024, 3, 1, 1, 1});
        g.push_back({"aux_classifier.4", ncls, 256, 1, 1, 0, 1});
    }
    return g;
}

void put_u32(std::vector<uint8_t>& b, size_t off, uint32_t v) { memcpy(b.data() + off, &v, 4); }
void put_u64(std::vector<uint8_t>& b, size_t off, uint64_t v) { memcpy(b.data() + off, &v, 8); }

}  // namespace

bool looks_like_onnx(const uint8_t* d, size_t n) {
    // ModelProto starts with field 1 (ir_version, varint: key 0x08) in every exporter's output
    return n > 4 && d[0] == 0x08;
}

int onnx_to_blob(const uint8_t* data, size_t len, std::vector<uint8_t>& blob, OnnxInfo& info, std::string& err) {
    PB m(data, len);
    uint32_t f, wt; uint64_t v; const uint8_t* s; size_t l;
    const uint8_t* g = nullptr;
    size_t gl = 0;
    while (m.next(f, wt, v, s, l))
        if (f == 7 && wt == 2) { g = s; gl = l; }
    if (!m.ok || !g) { err = "not an ONNX ModelProto (no graph)"; return 1; }

    std::map<std::string, Tensor> inits;
    std::vector<Node> nodes;
    std::vector<ValueInfo> inputs, outputs;
    PB gr(g, gl);
    while (gr.next(f, wt, v, s, l)) {
        if (wt != 2) continue;
        if (f == 1) { Node n; if (!parse_node(s, l, n)) { err = "malformed NodeProto"; return 1; } nodes.push_back(std::move(n)); }
        else if (f == 5) { std::string name; Tensor t; if (!parse_tensor(s, l, name, t)) { err = "malformed or external-data TensorProto"; return 1; } inits[name] = std::move(t); }
        else if (f == 11) { ValueInfo vi; if (!parse_value_info(s, l, vi)) { err = "malformed graph input"; return 1; } inputs.push_back(std::move(vi)); }
        else if (f == 12) { ValueInfo vi; if (!parse_value_info(s, l, vi)) { err = "malformed graph output"; return 1; } outputs.push_back(std::move(vi)); }
    }
    if (!gr.ok) { err = "malformed GraphProto"; return 1; }

    // ---- input 0: the reference's infer_img_pre_proc (predict_onnx.rs:223-265) ----
    const ValueInfo* in0 = nullptr;
    for (auto& vi : inputs)
        if (!inits.count(vi.name)) { in0 = &vi; break; }  // old exporters list initializers as inputs too
    if (!in0) { err = "model has no image input"; return 1; }
    int col = -1;
    for (size_t i = 0; i < in0->dims.size(); i++)
        if (in0->dims[i] == 3) { col = (int)i; break; }
    if (col < 0) { err = "couldn't locate model's color input by dimension length 3"; return 2; }
    if (in0->dims.size() != 4) { err = "only 4 dimensions supported got " + std::to_string(in0->dims.size()); return 2; }
    if (col != 1 && col != 3) { err = "color dimension only at NCHW or NHWC but not in position " + std::to_string(col) + " supported"; return 2; }
    if (in0->elem_type != 1 && in0->elem_type != 2) { err = "only Float (f32) and Uint8 (u8) input supported, got elem_type " + std::to_string(in0->elem_type); return 2; }
    if (col != 1 || in0->elem_type != 1) { err = "this build runs NCHW Float segmentation models only (fcn-resnet50/101); got a different input layout/dtype"; return 2; }
    info.input_name = in0->name;
    info.input_dtype = "Float";
    for (auto& o : outputs) info.output_names.push_back(o.name);

    // ---- Conv nodes in graph order (+ BatchNormalization folding) ----
    std::map<std::string, const Node*> consumer_bn;
    for (auto& n : nodes)
        if (n.op == "BatchNormalization" && !n.in.empty()) consumer_bn[n.in[0]] = &n;
    struct Folded { std::vector<float> w, b; int cout, cin, kh, kw, stride, pad, dil; };
    std::vector<Folded> convs;
    for (auto& n : nodes) {
        if (n.op == "QLinearConv" || n.op == "ConvInteger") { err = "quantised model (" + n.op + "): only float Conv models are supported"; return 2; }
        if (n.op != "Conv") continue;
        if (n.in.size() < 2 || !inits.count(n.in[1])) { err = "Conv without an initializer weight"; return 1; }
        const Tensor& W = inits[n.in[1]];
        if (W.dims.size() != 4) { err = "Conv weight is not 4-D"; return 1; }
        Folded c;
        c.cout = (int)W.dims[0]; c.cin = (int)W.dims[1]; c.kh = (int)W.dims[2]; c.kw = (int)W.dims[3];
        if (!W.floats(c.w)) { err = "Conv weight '" + n.in[1] + "' is not inline float data"; return 1; }
        if (n.in.size() > 2 && !n.in[2].empty()) {
            if (!inits.count(n.in[2]) || !inits[n.in[2]].floats(c.b) || (int)c.b.size() != c.cout) { err = "bad Conv bias"; return 1; }
        } else {
            c.b.assign(c.cout, 0.0f);
        }
        auto geti = [&](const char* k, size_t i, int64_t dflt) { auto it = n.ints.find(k); return it != n.ints.end() && it->second.size() > i ? it->second[i] : dflt; };
        if (geti("group", 0, 1) != 1) { err = "grouped Conv is not part of FCN-ResNet"; return 1; }
        c.stride = (int)geti("strides", 0, 1);
        c.dil = (int)geti("dilations", 0, 1);
        c.pad = (int)geti("pads", 0, 0);
        if (geti("strides", 1, c.stride) != c.stride || geti("dilations", 1, c.dil) != c.dil || geti("pads", 1, c.pad) != c.pad ||
            geti("pads", 2, c.pad) != c.pad || geti("pads", 3, c.pad) != c.pad) { err = "anisotropic Conv attributes are not part of FCN-ResNet"; return 1; }
        auto bn = n.out.empty() ? consumer_bn.end() : consumer_bn.find(n.out[0]);
        if (bn != consumer_bn.end()) {  // W' = W*g/sqrt(v+eps), b' = (b-mean)*g/sqrt(v+eps) + beta
            const Node& B = *bn->second;
            std::vector<float> ga, be, mu, va;
            if (B.in.size() < 5 || !inits.count(B.in[1]) || !inits.count(B.in[2]) || !inits.count(B.in[3]) || !inits.count(B.in[4]) ||
                !inits[B.in[1]].floats(ga) || !inits[B.in[2]].floats(be) || !inits[B.in[3]].floats(mu) || !inits[B.in[4]].floats(va) ||
                (int)ga.size() != c.cout) { err = "bad BatchNormalization parameters"; return 1; }
            auto e = B.f.find("epsilon");
            const double eps = e != B.f.end() ? e->second : 1e-5;
            const size_t per = (size_t)c.cin * c.kh * c.kw;
            for (int o = 0; o < c.cout; o++) {
                const double sc = (double)ga[o] / std::sqrt((double)va[o] + eps);
                for (size_t i = 0; i < per; i++) c.w[o * per + i] = (float)((double)c.w[o * per + i] * sc);
                c.b[o] = (float)(((double)c.b[o] - mu[o]) * sc + be[o]);
            }
        }
        convs.push_back(std::move(c));
    }

    int depth = 0;
    bool aux = false;
    switch (convs.size()) {
        case 57: depth = 50; aux = true; break;
        case 55: depth = 50; break;
        case 108: depth = 101; aux = true; break;
        case 106: depth = 101; break;
        default: err = "model has " + std::to_string(convs.size()) + " Conv nodes; fcn_resnet50 has 57 (55 without aux), fcn_resnet101 108 (106)"; return 2;
    }
    const int ncls = convs[aux ? convs.size() - 3 : convs.size() - 1].cout;
    const std::vector<ExpConv> exp = expected_graph(depth, ncls, aux);
    for (size_t i = 0; i < exp.size(); i++) {
        const Folded& c = convs[i];
        const ExpConv& e = exp[i];
        if (c.cout != e.cout || c.cin != e.cin || c.kh != e.k || c.kw != e.k || c.stride != e.stride || c.pad != e.pad || c.dil != e.dil) {
            char buf[256];
            snprintf(buf, sizeof buf, "Conv #%zu (%s) is [%d,%d,%d,%d] s%d p%d d%d, expected [%d,%d,%d,%d] s%d p%d d%d", i, e.name.c_str(),
                     c.cout, c.cin, c.kh, c.kw, c.stride, c.pad, c.dil, e.cout, e.cin, e.k, e.k, e.stride, e.pad, e.dil);
            err = buf;
            return 2;
        }
    }

    // ---- INFURW01 blob ----
    const size_t n = exp.size();
    size_t off = (32 + n * 80 + 63) & ~(size_t)63;
    std::vector<std::pair<size_t, size_t>> offs(n);
    for (size_t i = 0; i < n; i++) {
        offs[i].first = off;
        off = (off + convs[i].w.size() * 4 + 63) & ~(size_t)63;
        offs[i].second = off;
        off = (off + convs[i].b.size() * 4 + 63) & ~(size_t)63;
    }
    blob.assign(off, 0);
    memcpy(blob.data(), "INFURW01", 8);
    put_u32(blob, 8, (uint32_t)depth);
    put_u32(blob, 12, (uint32_t)ncls);
    put_u32(blob, 16, aux ? 1u : 0u);
    put_u32(blob, 20, (uint32_t)n);
    for (size_t i = 0; i < n; i++) {
        const size_t e = 32 + i * 80;
        memcpy(blob.data() + e, exp[i].name.c_str(), exp[i].name.size() < 39 ? exp[i].name.size() : 39);
        put_u32(blob, e + 40, (uint32_t)convs[i].cout);
        put_u32(blob, e + 44, (uint32_t)convs[i].cin);
        put_u32(blob, e + 48, (uint32_t)convs[i].kh);
        put_u32(blob, e + 52, (uint32_t)convs[i].kw);
        put_u64(blob, e + 56, offs[i].first);
        put_u64(blob, e + 64, offs[i].second);
        memcpy(blob.data() + offs[i].first, convs[i].w.data(), convs[i].w.size() * 4);
        memcpy(blob.data() + offs[i].second, convs[i].b.data(), convs[i].b.size() * 4);
    }
    info.depth = depth;
    info.num_classes = ncls;
    info.aux = aux;
    return 0;
}

}  // namespace infur
