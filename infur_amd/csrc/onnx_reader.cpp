// onnx_reader.cpp -- ModelCmd::Load("...fcn-resnet50-12.onnx") without ONNX Runtime.
//
// The reference hands the model file to ONNX Runtime (infur/src/predict_onnx.rs:288-293) and
// then inspects input 0 to infer the image layout (infer_img_pre_proc, :223-265).  This file is
// the minimum needed to accept the same file here: a hand-written protobuf wire-format reader
// (ModelProto -> GraphProto -> node / initializer / input / output) that
//   * applies the reference's input checks with the reference's messages (:228-262),
//   * follows the graph's EDGES from the image input -- stem Conv -> Relu -> MaxPool, then per bottleneck
//     conv1 -> Relu -> conv2 -> Relu -> conv3 -> Add(identity | downsample Conv) -> Relu, then the classifier
//     (Conv -> Relu -> [Dropout] -> Conv -> Resize -> output 0) and the aux head off layer3 (-> output 1) --
//     so every weight tensor is assigned to its INFURW01 slot by its place in the topology, never by its
//     position in the serialized node list (conv3 and downsample.0 of layer1.0 have identical shapes);
//     a trailing BatchNormalization is folded when the exporter did not; every shape / stride / dilation,
//     the MaxPool, the Relu placement and the final Resize (linear, half-pixel coordinates) are checked
//     against torchvision's fcn_resnet50/101 and anything else is a format error,
//   * emits the INFURW01 blob that infur_model_load_blob consumes.
// No onnx / protobuf library is used (none exists in the build image).  Host only.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "onnx_pb.h"
#include "onnx_reader.h"

namespace infur {
namespace {

using namespace pb;

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
    // (DimSeq, ColorRange) exactly as the reference infers them: NCHW / NHWC by the position of the 3, Float32 / Uint8
    // by the element type.  What they mean for the path (predict_onnx.rs:103-138,296-301): a Float input gets RGB planes
    // normalised with the torchvision constants; a Uint8 input gets the raw bytes in BGR order; NHWC inputs get the
    // frame's own layout.  The graph side of a non-default input is checked below ("front").
    const bool in_nhwc = col == 3, in_u8 = in0->elem_type == 2;
    info.input_name = in0->name;
    info.input_dtype = in_u8 ? "Uint8" : "Float";
    info.input_u8 = in_u8;
    info.input_nhwc = in_nhwc;
    for (auto& o : outputs) info.output_names.push_back(o.name);

    // ---- opset: Resize's coordinate rule is only spelled out from opset 11 on ----
    int64_t opset = 0;
    {
        PB mm(data, len);
        while (mm.next(f, wt, v, s, l)) {
            if (f != 8 || wt != 2) continue;  // opset_import
            PB os(s, l);
            uint32_t f2, w2; uint64_t v2; const uint8_t* s2; size_t l2;
            std::string domain;
            int64_t ver = 0;
            while (os.next(f2, w2, v2, s2, l2)) {
                if (f2 == 1 && w2 == 2) domain.assign((const char*)s2, l2);
                else if (f2 == 2 && w2 == 0) ver = (int64_t)v2;
            }
            if (domain.empty() || domain == "ai.onnx") opset = ver;
        }
    }

    // ---- structure: every NodeProto names at least one output, every operator but Constant at least one input (the walkers
    // below index out[0] / in[0] of the nodes they reach; a file that drops them is malformed, never a crash -- ADVICE r3) ----
    for (auto& n : nodes) {
        if (n.out.empty()) { err = "malformed graph: a " + (n.op.empty() ? std::string("node") : n.op) + " node has no outputs"; return 1; }
        if (n.in.empty() && n.op != "Constant") { err = "malformed graph: a " + (n.op.empty() ? std::string("node") : n.op) + " node has no inputs"; return 1; }
    }

    // ---- edges ----
    // Constant nodes are initializers in all but name (old exporters write weights that way)
    for (auto& n : nodes)
        if (n.op == "Constant" && n.value_t && !n.out.empty()) {
            std::string nm; Tensor t;
            if (!parse_tensor(n.value_t, n.value_len, nm, t)) { err = "malformed Constant tensor"; return 1; }
            inits[n.out[0]] = std::move(t);
        }
    std::map<std::string, std::vector<int>> consumers;  // tensor -> nodes reading it as a DATA input
    std::map<std::string, int> producer;
    for (size_t i = 0; i < nodes.size(); i++) {
        for (auto& o : nodes[i].out) producer[o] = (int)i;
        for (auto& in : nodes[i].in)
            if (!in.empty() && !inits.count(in)) consumers[in].push_back((int)i);
    }
    // value-preserving nodes an exporter may leave between two layers (inference semantics)
    auto transparent = [&](const Node& n) { return n.op == "Identity" || n.op == "Dropout"; };
    // the nodes that compute on tensor t, looking through Identity / Dropout; Shape readers (the Resize size
    // arithmetic) are not data consumers
    // (depth-limited: a hostile file may chain Identity nodes into a cycle, which must end in a format error, not in
    // unbounded recursion behind the C ABI)
    std::function<void(const std::string&, std::vector<int>&, int)> users_d = [&](const std::string& t, std::vector<int>& out, int depth) {
        auto it = consumers.find(t);
        if (it == consumers.end() || depth > 64) return;
        for (int i : it->second) {
            const Node& n = nodes[i];
            if (n.op == "Shape") continue;
            if (transparent(n)) { if (!n.out.empty() && n.in[0] == t) users_d(n.out[0], out, depth + 1); }
            else out.push_back(i);
        }
    };
    auto users = [&](const std::string& t, std::vector<int>& out) { users_d(t, out, 0); };
    // origin of a tensor, looking back through Identity / Dropout
    auto origin = [&](std::string t) {
        for (int guard = 0; guard < 64; guard++) {
            auto it = producer.find(t);
            if (it == producer.end() || !transparent(nodes[it->second]) || nodes[it->second].in.empty()) break;
            t = nodes[it->second].in[0];
        }
        return t;
    };
    auto sole_user = [&](const std::string& t, const char* op, const char* what, int& idx) {
        std::vector<int> u;
        users(t, u);
        if (u.size() != 1 || nodes[u[0]].op != op) {
            err = std::string("expected exactly one ") + op + " after " + what + ", found " + std::to_string(u.size()) +
                  (u.empty() ? "" : " (" + nodes[u[0]].op + ")");
            return false;
        }
        idx = u[0];
        return true;
    };

    {   // QDQ format: float Conv nodes whose input is a DequantizeLinear output (onnx_qreader.cpp fuses the groups)
        std::map<std::string, char> dq_out;
        for (auto& n : nodes)
            if (n.op == "DequantizeLinear" && !n.out.empty()) dq_out[n.out[0]] = 1;
        for (auto& n : nodes)
            if (n.op == "Conv" && !n.in.empty() && dq_out.count(n.in[0])) return onnx_q_to_blob(data, len, blob, info = OnnxInfo(), err);
    }
    for (auto& n : nodes) {
        // the QOperator int8 form (fcn-resnet50-12-int8.onnx, the file the reference's tests load) has its own reader and blob
        if (n.op == "QLinearConv") return onnx_q_to_blob(data, len, blob, info = OnnxInfo(), err);
        if (n.op == "ConvInteger" || n.op == "DynamicQuantizeLinear") { err = "dynamically quantised model (" + n.op + "): only float Conv and static QOperator (QLinearConv) models are supported"; return 2; }
    }
    size_t n_conv_nodes = 0;
    for (auto& n : nodes) n_conv_nodes += n.op == "Conv";
    int depth = 0;
    bool aux = false;
    switch (n_conv_nodes) {
        case 57: depth = 50; aux = true; break;
        case 55: depth = 50; break;
        case 108: depth = 101; aux = true; break;
        case 106: depth = 101; break;
        default: err = "model has " + std::to_string(n_conv_nodes) + " Conv nodes; fcn_resnet50 has 57 (55 without aux), fcn_resnet101 108 (106)"; return 2;
    }

    struct Folded { std::vector<float> w, b; int cout = 0, cin = 0, kh = 0, kw = 0, stride = 1, pad = 0, dil = 1; std::string out; };
    std::vector<char> conv_used(nodes.size(), 0);
    // Conv node -> folded weights; `out` = the tensor carrying its result (after a BatchNormalization it feeds alone)
    auto fold = [&](int idx, Folded& c) -> int {
        const Node& n = nodes[idx];
        if (conv_used[idx]) { err = "a Conv node is reached twice while walking the graph"; return 2; }
        conv_used[idx] = 1;
        if (n.in.size() < 2 || !inits.count(n.in[1])) { err = "Conv without an initializer weight"; return 1; }
        const Tensor& W = inits[n.in[1]];
        if (W.external) { err = "tensor '" + n.in[1] + "' uses external_data: only single-file models are supported"; return 1; }
        if (W.dims.size() != 4) { err = "Conv weight is not 4-D"; return 1; }
        if (W.dtype != 1) { err = "Conv weight '" + n.in[1] + "' has elem_type " + std::to_string(W.dtype) + ": only float (1) models are supported"; return 2; }
        size_t cnt;
        if (!W.count(cnt)) { err = "Conv weight '" + n.in[1] + "' has non-positive or absurd dimensions"; return 1; }
        c.cout = (int)W.dims[0]; c.cin = (int)W.dims[1]; c.kh = (int)W.dims[2]; c.kw = (int)W.dims[3];
        if (!W.floats(c.w)) { err = "Conv weight '" + n.in[1] + "' is not inline float data"; return 1; }
        if (n.in.size() > 2 && !n.in[2].empty()) {
            if (!inits.count(n.in[2]) || !inits[n.in[2]].floats(c.b) || (int)c.b.size() != c.cout) { err = "bad Conv bias"; return 1; }
        } else {
            c.b.assign(c.cout, 0.0f);
        }
        auto geti = [&](const char* k, size_t i, int64_t dflt) { auto it = n.ints.find(k); return it != n.ints.end() && it->second.size() > i ? it->second[i] : dflt; };
        if (geti("group", 0, 1) != 1) { err = "grouped Conv is not part of FCN-ResNet"; return 1; }
        auto ap = n.strs.find("auto_pad");
        if (ap != n.strs.end() && ap->second != "NOTSET") { err = "Conv auto_pad " + ap->second + " is not part of FCN-ResNet"; return 1; }
        if (geti("kernel_shape", 0, c.kh) != c.kh || geti("kernel_shape", 1, c.kw) != c.kw) { err = "Conv kernel_shape disagrees with its weight"; return 1; }
        c.stride = (int)geti("strides", 0, 1);
        c.dil = (int)geti("dilations", 0, 1);
        c.pad = (int)geti("pads", 0, 0);
        if (geti("strides", 1, c.stride) != c.stride || geti("dilations", 1, c.dil) != c.dil || geti("pads", 1, c.pad) != c.pad ||
            geti("pads", 2, c.pad) != c.pad || geti("pads", 3, c.pad) != c.pad) { err = "anisotropic Conv attributes are not part of FCN-ResNet"; return 1; }
        if (n.out.empty()) { err = "Conv without an output"; return 1; }
        c.out = n.out[0];
        std::vector<int> u;
        users(c.out, u);
        if (u.size() == 1 && nodes[u[0]].op == "BatchNormalization" && !nodes[u[0]].in.empty() && origin(nodes[u[0]].in[0]) == c.out) {
            // W' = W*g/sqrt(v+eps), b' = (b-mean)*g/sqrt(v+eps) + beta
            const Node& B = nodes[u[0]];
            std::vector<float> ga, be, mu, va;
            if (B.in.size() < 5 || !inits.count(B.in[1]) || !inits.count(B.in[2]) || !inits.count(B.in[3]) || !inits.count(B.in[4]) ||
                !inits[B.in[1]].floats(ga) || !inits[B.in[2]].floats(be) || !inits[B.in[3]].floats(mu) || !inits[B.in[4]].floats(va) ||
                (int)ga.size() != c.cout || (int)be.size() != c.cout || (int)mu.size() != c.cout || (int)va.size() != c.cout || B.out.empty()) {
                err = "bad BatchNormalization parameters"; return 1;
            }
            auto e = B.f.find("epsilon");
            const double eps = e != B.f.end() ? e->second : 1e-5;
            const size_t per = (size_t)c.cin * c.kh * c.kw;
            for (int o = 0; o < c.cout; o++) {
                const double sc = (double)ga[o] / std::sqrt((double)va[o] + eps);
                for (size_t i = 0; i < per; i++) c.w[o * per + i] = (float)((double)c.w[o * per + i] * sc);
                c.b[o] = (float)(((double)c.b[o] - mu[o]) * sc + be[o]);
            }
            c.out = B.out[0];
        }
        return 0;
    };

    // ---- walk ----
    std::vector<ExpConv> exp = expected_graph(depth, /*ncls: patched below*/ 0, aux);
    std::vector<Folded> convs(exp.size());
    size_t slot = 0;
    int idx = -1, rc = 0;
    auto take = [&](int node_idx) -> int {  // fold nodes[node_idx] into the next slot
        if (slot >= convs.size()) { err = "more convolutions on the path than fcn_resnet has"; return 2; }
        return fold(node_idx, convs[slot++]);
    };
    auto relu_after = [&](const std::string& t, const char* what, std::string& out) {
        int r;
        if (!sole_user(t, "Relu", what, r)) return false;
        out = nodes[r].out.empty() ? std::string() : nodes[r].out[0];
        return true;
    };

    std::string t = in0->name, t3;
    // ---- front: what a Uint8 and / or NHWC image input must pass on its way to the stem ----
    // The session is handed bytes (Uint8) and / or the frame's own H x W x 3 layout (NHWC); the convolution needs NCHW
    // floats, so such a file carries exactly one Cast(to = FLOAT) and / or one Transpose(perm = 0,3,1,2) in front of the
    // stem, in either order.  Anything else there (in-graph arithmetic on the pixels) is not a model this path runs.
    {
        bool seen_cast = false, seen_tr = false;
        for (int hop = 0; hop < 2; hop++) {
            std::vector<int> u;
            users(t, u);
            if (u.size() != 1 || nodes[u[0]].out.empty()) break;
            const Node& n = nodes[u[0]];
            if (n.op == "Cast" && in_u8 && !seen_cast) {
                auto it = n.ints.find("to");
                if (it == n.ints.end() || it->second.empty() || it->second[0] != 1) { err = "the Cast after the Uint8 image input must produce FLOAT"; return 2; }
                seen_cast = true;
            } else if (n.op == "Transpose" && in_nhwc && !seen_tr) {
                auto it = n.ints.find("perm");
                const std::vector<int64_t> want = {0, 3, 1, 2};
                if (it == n.ints.end() || it->second != want) { err = "the Transpose after the NHWC image input must be perm = [0,3,1,2]"; return 2; }
                seen_tr = true;
            } else {
                break;
            }
            t = n.out[0];
        }
        if (in_u8 && !seen_cast) { err = "a Uint8 image input must reach the stem convolution through one Cast to FLOAT"; return 2; }
        if (in_nhwc && !seen_tr) { err = "an NHWC image input must reach the stem convolution through one Transpose(perm = [0,3,1,2])"; return 2; }
    }
    if (!sole_user(t, "Conv", "the image input", idx)) return 2;
    if ((rc = take(idx))) return rc;
    if (!relu_after(convs[slot - 1].out, "the stem convolution", t)) return 2;
    if (!sole_user(t, "MaxPool", "the stem's Relu", idx)) return 2;
    {
        const Node& mp = nodes[idx];
        auto geti = [&](const char* k, size_t i, int64_t dflt) { auto it = mp.ints.find(k); return it != mp.ints.end() && it->second.size() > i ? it->second[i] : dflt; };
        bool ok = geti("kernel_shape", 0, 0) == 3 && geti("kernel_shape", 1, 0) == 3 && geti("strides", 0, 1) == 2 && geti("strides", 1, 1) == 2 &&
                  geti("ceil_mode", 0, 0) == 0 && geti("dilations", 0, 1) == 1 && geti("dilations", 1, 1) == 1;
        for (size_t i = 0; i < 4; i++) ok = ok && geti("pads", i, 0) == 1;
        if (!ok || mp.out.empty()) { err = "the stem's MaxPool is not 3x3 / stride 2 / pad 1"; return 2; }
        t = mp.out[0];
    }
    const int lb50[4] = {3, 4, 6, 3}, lb101[4] = {3, 4, 23, 3};
    const int* lb = depth == 50 ? lb50 : lb101;
    int aux_conv = -1;  // the aux head's first conv: the Conv reader of layer3's output that belongs to no block
    for (int L = 0; L < 4; L++)
        for (int b = 0; b < lb[L]; b++) {
            const std::string where = "backbone.layer" + std::to_string(L + 1) + "." + std::to_string(b);
            std::vector<int> u;
            users(t, u);
            // conv1 = the Conv reader of the block input that starts conv -> Relu -> conv -> Relu -> conv -> Add
            int c1 = -1, c2 = -1, c3 = -1, add = -1;
            for (int cand : u) {
                if (nodes[cand].op != "Conv" || nodes[cand].out.empty()) continue;
                auto step = [&](const std::string& from, const char* op) {  // sole user of `from` (through a folded BN) with that op
                    std::string cur = from;
                    for (int hop = 0; hop < 2; hop++) {
                        std::vector<int> uu;
                        users(cur, uu);
                        if (uu.size() != 1) return -1;
                        if (nodes[uu[0]].op == op) return uu[0];
                        if (nodes[uu[0]].op != "BatchNormalization" || nodes[uu[0]].out.empty()) return -1;
                        cur = nodes[uu[0]].out[0];
                    }
                    return -1;
                };
                const int r1 = step(nodes[cand].out[0], "Relu");
                const int k2 = r1 >= 0 && !nodes[r1].out.empty() ? step(nodes[r1].out[0], "Conv") : -1;
                const int r2 = k2 >= 0 && !nodes[k2].out.empty() ? step(nodes[k2].out[0], "Relu") : -1;
                const int k3 = r2 >= 0 && !nodes[r2].out.empty() ? step(nodes[r2].out[0], "Conv") : -1;
                const int ad = k3 >= 0 && !nodes[k3].out.empty() ? step(nodes[k3].out[0], "Add") : -1;
                if (ad < 0) continue;
                if (c1 >= 0) { err = where + ": two Conv chains leave the block input"; return 2; }
                c1 = cand; c2 = k2; c3 = k3; add = ad;
            }
            if (c1 < 0) { err = where + ": no conv1 -> Relu -> conv2 -> Relu -> conv3 -> Add chain leaves the block input"; return 2; }
            if ((rc = take(c1)) || (rc = take(c2)) || (rc = take(c3))) return rc;
            const Folded& f3 = convs[slot - 1];
            const Node& A = nodes[add];
            if (A.in.size() != 2 || A.out.empty()) { err = where + ": malformed Add"; return 2; }
            const std::string a0 = origin(A.in[0]), a1 = origin(A.in[1]);
            if (a0 != f3.out && a1 != f3.out) { err = where + ": the residual Add does not read conv3"; return 2; }
            const std::string other = a0 == f3.out ? a1 : a0;
            const bool want_ds = b == 0;
            if (!want_ds) {
                if (other != origin(t)) { err = where + ": the residual Add must read the block input (identity shortcut)"; return 2; }
            } else {
                // the shortcut is a Conv (+BN) that reads the block input
                auto pit = producer.find(other);
                int dn = pit == producer.end() ? -1 : pit->second;
                if (dn >= 0 && nodes[dn].op == "BatchNormalization" && !nodes[dn].in.empty()) {
                    auto p2 = producer.find(origin(nodes[dn].in[0]));
                    dn = p2 == producer.end() ? -1 : p2->second;
                }
                if (dn < 0 || nodes[dn].op != "Conv" || nodes[dn].in.empty() || origin(nodes[dn].in[0]) != origin(t)) {
                    err = where + ": the shortcut of a stage's first block must be a downsample Conv reading the block input"; return 2;
                }
                if ((rc = take(dn))) return rc;
                if (convs[slot - 1].out != other) { err = where + ": the downsample branch does not feed the residual Add directly"; return 2; }
            }
            // every other reader of the block input must be accounted for: the Add (identity), or -- only for
            // layer3's output, seen as layer4.0's input -- the aux head's conv
            for (int cand : u) {
                if (cand == c1 || cand == add || conv_used[cand]) continue;
                if (nodes[cand].op == "Conv" && L == 3 && b == 0 && aux && aux_conv < 0) { aux_conv = cand; continue; }
                err = where + ": unexpected " + nodes[cand].op + " reads the block input"; return 2;
            }
            if (L == 3 && b == 0) t3 = t;
            if (!relu_after(A.out[0], (where + "'s residual Add").c_str(), t)) return 2;
        }
    // ---- heads: Conv 3x3 -> Relu -> [Dropout] -> Conv 1x1 -> Resize -> graph output ----
    auto head = [&](const std::string& feat, int first_conv, const char* what, size_t out_index) -> int {
        int h0 = first_conv;
        if (h0 < 0) {
            std::vector<int> u;
            users(feat, u);
            for (int cand : u)
                if (nodes[cand].op == "Conv" && !conv_used[cand]) { if (h0 >= 0) { err = std::string(what) + ": two candidate head convolutions"; return 2; } h0 = cand; }
                else if (!conv_used[cand]) { err = std::string(what) + ": unexpected " + nodes[cand].op + " reads the backbone output"; return 2; }
            if (h0 < 0) { err = std::string(what) + ": no head convolution reads the backbone output"; return 2; }
        }
        int r2;
        if ((r2 = take(h0))) return r2;
        std::string th;
        if (!relu_after(convs[slot - 1].out, what, th)) return 2;
        int h1;
        if (!sole_user(th, "Conv", what, h1)) return 2;
        if ((r2 = take(h1))) return r2;
        std::vector<int> u;
        users(convs[slot - 1].out, u);
        if (u.size() != 1 || (nodes[u[0]].op != "Resize" && nodes[u[0]].op != "Upsample")) { err = std::string(what) + ": the logits must feed exactly one Resize"; return 2; }
        const Node& R = nodes[u[0]];
        if (R.op == "Upsample" || opset < 11) { err = std::string(what) + ": " + R.op + " under opset " + std::to_string(opset) + " has no half-pixel coordinate rule (need Resize, opset >= 11)"; return 2; }
        auto m = R.strs.find("mode");
        if (m == R.strs.end() || m->second != "linear") { err = std::string(what) + ": Resize mode must be linear (bilinear up-sampling), got '" + (m == R.strs.end() ? "nearest" : m->second) + "'"; return 2; }
        auto cm = R.strs.find("coordinate_transformation_mode");
        const std::string cmode = cm == R.strs.end() ? "half_pixel" : cm->second;
        if (cmode != "pytorch_half_pixel" && cmode != "half_pixel") { err = std::string(what) + ": Resize coordinate_transformation_mode '" + cmode + "' is not align_corners=False bilinear"; return 2; }
        if (origin(R.in[0]) != convs[slot - 1].out || R.out.empty()) { err = std::string(what) + ": Resize does not read the logits"; return 2; }
        // ... and that Resize is graph output #out_index (the reference decodes outputs[0], app.rs:116)
        if (out_index >= outputs.size() || origin(outputs[out_index].name) != R.out[0]) {
            err = std::string(what) + ": its up-sampled logits are not graph output #" + std::to_string(out_index); return 2;
        }
        return 0;
    };
    if ((rc = head(t, -1, "classifier", 0))) return rc;
    if (aux) {
        if (aux_conv < 0) { err = "aux_classifier: no head convolution reads layer3's output"; return 2; }
        if ((rc = head(t3, aux_conv, "aux_classifier", 1))) return rc;
    } else if (aux_conv >= 0) {
        err = "a convolution besides layer4 reads layer3's output but the model has no aux head"; return 2;
    }
    if (slot != convs.size()) { err = "the walk assigned " + std::to_string(slot) + " of " + std::to_string(convs.size()) + " convolutions"; return 2; }
    for (size_t i = 0; i < nodes.size(); i++)
        if (nodes[i].op == "Conv" && !conv_used[i]) { err = "a Conv node is not part of the FCN-ResNet topology"; return 2; }

    const int ncls = convs[aux ? convs.size() - 3 : convs.size() - 1].cout;
    if (aux && convs.back().cout != ncls) { err = "out and aux heads disagree on the class count"; return 2; }
    exp = expected_graph(depth, ncls, aux);
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
    put_u32(blob, 24, in_u8 ? 1u : 0u);  // input kind: 0 = Float (normalised RGB planes), 1 = Uint8 (raw BGR bytes)
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
