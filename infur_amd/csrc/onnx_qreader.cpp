// onnx_qreader.cpp -- ModelCmd::Load of the QUANTISED model file: the QOperator int8 form of FCN-ResNet the reference's own
// tests load (`fcn-resnet50-12-int8.onnx`: infur-test-gen/build.rs:88-93, infur/src/predict_onnx.rs:357-381) -> INFURQ01.
//
// The zoo file itself is not available in this environment (no network); what is accepted here is the graph ONNX Runtime's /
// Neural Compressor's static quantisation writes for that network in QOperator format, followed by its EDGES from the image
// input exactly as onnx_reader.cpp follows the float graph:
//   input (Float, NCHW) -> QuantizeLinear -> QLinearConv 7x7/2 -> MaxPool 3x3/2 (on u8)
//   per bottleneck: QLinearConv 1x1 -> QLinearConv 3x3 -> QLinearConv 1x1 -> QLinearAdd(com.microsoft) with the identity or
//                   the downsample QLinearConv of the block's input; ReLUs are folded into the clamps (a Relu node left on a
//                   tensor whose zero point is 0 is looked through)
//   heads: QLinearConv 3x3 -> QLinearConv 1x1 -> DequantizeLinear -> Resize(linear, [pytorch_]half_pixel) -> output 0 / 1,
//          or ... -> Resize (on the u8 codes) -> DequantizeLinear -> output (blob flag bit 0)
// Requirements (format errors otherwise): u8 activations (zero points UINT8), INT8 or UINT8 weights whose distance from their zero
// point (scalar or per output channel) fits 8 bits -- stored re-centred as s8 -- with one scale per tensor or per output channel, int32 bias, group 1, every shape / stride / pad / dilation as torchvision's fcn_resnet50/101.
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "blob_dir.h"
#include "onnx_pb.h"
#include "onnx_reader.h"

namespace infur {
namespace {

using namespace pb;

struct QGraph {
    std::map<std::string, Tensor> inits;
    std::vector<Node> nodes;
    std::map<std::string, std::vector<int>> consumers;
    std::map<std::string, int> producer;
    std::map<std::string, int> graph_outputs;
};

bool scalar_f32(const QGraph& g, const std::string& name, float* v) {
    auto it = g.inits.find(name);
    std::vector<float> f;
    if (it == g.inits.end() || it->second.dims.size() > 1 || !it->second.floats(f) || f.size() != 1) return false;
    *v = f[0];
    return true;
}

// bytes of an integer tensor (UINT8 = 2, INT8 = 3, INT32 = 6) stored as raw_data or as int32_data (field 5 is not parsed by
// onnx_pb.h: exporters write quantised initializers as raw_data)
bool int_bytes(const Tensor& t, int dtype, size_t elem, size_t* n, const uint8_t** p) {
    size_t cnt = 1;
    if (!t.dims.empty() && !t.count(cnt)) return false;
    if (t.dtype != dtype || t.external || !t.raw || t.raw_len != cnt * elem) return false;
    *n = cnt;
    *p = t.raw;
    return true;
}

bool scalar_u8(const QGraph& g, const std::string& name, int32_t* v) {
    auto it = g.inits.find(name);
    if (it == g.inits.end()) return false;
    size_t n;
    const uint8_t* p;
    if (!int_bytes(it->second, 2, 1, &n, &p) || n != 1) return false;
    *v = p[0];
    return true;
}

struct QC {  // one QLinearConv as the blob wants it
    int node = -1;
    int cout = 0, cin = 0, kh = 0, kw = 0, stride = 1, pad = 0, dil = 1;
    float x_scale = 0, y_scale = 0;
    int32_t x_zp = 0, y_zp = 0;
    const uint8_t* w = nullptr;   // s8 OIHW: the initializer's own bytes, or w_own when the file's weights had to be re-centred
    std::vector<uint8_t> w_own;
    std::vector<float> w_scale;
    std::vector<int32_t> bias;
    std::string out;
};

bool read_qconv(const QGraph& g, int ni, QC* q, std::string* err) {
    const Node& n = g.nodes[ni];
    auto bad = [&](const std::string& m) { *err = "QLinearConv '" + (n.out.empty() ? std::string("?") : n.out[0]) + "': " + m; return false; };
    if (n.in.size() < 8 || n.out.empty()) return bad("needs 8 or 9 inputs and an output");
    if (!scalar_f32(g, n.in[1], &q->x_scale) || !scalar_u8(g, n.in[2], &q->x_zp)) return bad("x_scale / x_zero_point must be a float / UINT8 scalar initializer");
    if (!scalar_f32(g, n.in[6], &q->y_scale) || !scalar_u8(g, n.in[7], &q->y_zp)) return bad("y_scale / y_zero_point must be a float / UINT8 scalar initializer");
    if (!qscale_ok(q->x_scale) || !qscale_ok(q->y_scale)) return bad("scales must be positive and finite");
    auto wi = g.inits.find(n.in[3]);
    if (wi == g.inits.end() || wi->second.dims.size() != 4) return bad("the weight must be a 4-D initializer");
    const Tensor& wt = wi->second;
    for (auto d : wt.dims)
        if (d <= 0 || d > 65536) return bad("absurd weight shape");
    q->cout = (int)wt.dims[0]; q->cin = (int)wt.dims[1]; q->kh = (int)wt.dims[2]; q->kw = (int)wt.dims[3];
    // weights: INT8 or UINT8 with a zero point of the same type (scalar or one per output channel).  The blob and the kernels carry
    // s8 weights with zero point 0, so w - w_zp is formed here; it must fit s8 (UINT8 weights around 128 do, the usual case)
    size_t wn;
    const int wdt = wt.dtype;
    const uint8_t* wraw;
    if ((wdt != 2 && wdt != 3) || !int_bytes(wt, wdt, 1, &wn, &wraw)) return bad("the weight must be INT8 or UINT8 raw_data");
    {
        auto zi = g.inits.find(n.in[5]);
        size_t zn;
        const uint8_t* zp;
        if (zi == g.inits.end() || !int_bytes(zi->second, wdt, 1, &zn, &zp) || (zn != 1 && zn != (size_t)q->cout))
            return bad("w_zero_point must be a scalar or [cout] initializer of the weight's type");
        bool all_zero = wdt == 3;
        for (size_t i = 0; i < zn && all_zero; i++) all_zero = zp[i] == 0;
        if (all_zero) {
            q->w = wraw;
        } else {
            q->w_own.resize(wn);
            const size_t per = wn / (size_t)q->cout;
            for (int o = 0; o < q->cout; o++) {
                const int z = wdt == 3 ? (int)(int8_t)zp[zn == 1 ? 0 : o] : (int)zp[zn == 1 ? 0 : o];
                for (size_t k = 0; k < per; k++) {
                    const size_t i = (size_t)o * per + k;
                    const int v = (wdt == 3 ? (int)(int8_t)wraw[i] : (int)wraw[i]) - z;
                    if (v < -128 || v > 127) return bad("w - w_zero_point does not fit 8 bits (weights this far from their zero point are not supported)");
                    q->w_own[i] = (uint8_t)(int8_t)v;
                }
            }
            q->w = nullptr;  // (set after the QC has reached its final place: w_own moves with it)
        }
    }
    {
        auto si = g.inits.find(n.in[4]);
        std::vector<float> ws;
        if (si == g.inits.end() || si->second.dtype != 1) return bad("w_scale must be a float initializer");
        if (si->second.dims.size() > 1 || !si->second.floats(ws)) return bad("w_scale must be a float scalar or [cout] initializer");
        if (ws.size() == 1) ws.assign((size_t)q->cout, ws[0]);
        if (ws.size() != (size_t)q->cout) return bad("w_scale must be a scalar or have one value per output channel");
        for (float s : ws)
            if (!qscale_ok(s)) return bad("weight scales must be positive and finite");
        q->w_scale = ws;
    }
    q->bias.assign((size_t)q->cout, 0);
    if (n.in.size() > 8 && !n.in[8].empty()) {
        auto bi = g.inits.find(n.in[8]);
        size_t bn;
        const uint8_t* bp;
        if (bi == g.inits.end() || !int_bytes(bi->second, 6, 4, &bn, &bp) || bn != (size_t)q->cout) return bad("the bias must be an INT32 [cout] initializer");
        memcpy(q->bias.data(), bp, bn * 4);
    }
    auto attr1 = [&](const char* name, int64_t dflt, size_t count, int64_t* v) {
        auto it = n.ints.find(name);
        if (it == n.ints.end()) { *v = dflt; return true; }
        if (it->second.size() != count) return false;
        for (auto x : it->second)
            if (x != it->second[0]) return false;
        *v = it->second[0];
        return true;
    };
    int64_t st, pd, dl, grp;
    if (!attr1("strides", 1, 2, &st) || !attr1("pads", 0, 4, &pd) || !attr1("dilations", 1, 2, &dl) || !attr1("group", 1, 1, &grp)) return bad("strides / pads / dilations must be uniform");
    if (grp != 1 || st < 1 || st > 2 || pd < 0 || pd > 8 || dl < 1 || dl > 8) return bad("unsupported group / stride / pad / dilation");
    auto ap = n.strs.find("auto_pad");
    if (ap != n.strs.end() && ap->second != "NOTSET") return bad("auto_pad is not supported");
    q->stride = (int)st; q->pad = (int)pd; q->dil = (int)dl;
    q->node = ni;
    q->out = n.out[0];
    return true;
}

// ---- QDQ format -> the QOperator node list the walker below understands ----
// onnxruntime >= 1.11 writes statically quantised models in QDQ format by default: every tensor is DequantizeLinear(QuantizeLinear(.)),
// the operators stay float -- and ONNX Runtime fuses each DQ -> op -> [Relu ->] Q group back into the QLinear operator when the
// session is created (the reference asks for ORT_ENABLE_EXTENDED, predict_onnx.rs:291), so the arithmetic that runs is the QOperator
// arithmetic.  This pass does the same fusion on the parsed graph, in place:
//   DQ(x), DQ(w) [, DQ(b)] -> Conv -> [Relu ->] Q        =>  QLinearConv(x_q, ..., w_q, ..., y_scale, y_zp, b_q) -> y_q
//   DQ(a), DQ(b) -> Add -> [Relu ->] Q                    =>  QLinearAdd(a_q, ..., b_q, ..., c_scale, c_zp) -> c_q
//   DQ(x) -> MaxPool -> Q  (same parameters)              =>  MaxPool(x_q) -> y_q
//   DQ(x) -> Resize -> Q (same parameters) -> DQ -> out   =>  Resize(x_q) -> DQ -> out          (else DQ -> Resize stays as it is)
// A Relu between the operator and its Q is the clamp only when Q's zero point is 0: anything else is a format error.
// DequantizeLinear nodes nobody reads afterwards are dropped.  Returns false with *err set on a graph it cannot fuse.
bool qdq_to_qoperator(QGraph& g, std::string* err) {
    struct DQ { std::string q, scale, zp; };
    std::map<std::string, DQ> dq_of;  // float tensor -> what it dequantises
    for (const Node& n : g.nodes)
        if (n.op == "DequantizeLinear" && n.in.size() >= 2 && !n.out.empty()) dq_of[n.out[0]] = DQ{n.in[0], n.in[1], n.in.size() > 2 ? n.in[2] : std::string()};
    std::map<std::string, std::vector<int>> cons;
    for (size_t i = 0; i < g.nodes.size(); i++)
        for (auto& in : g.nodes[i].in)
            if (!in.empty()) cons[in].push_back((int)i);
    std::vector<char> dead(g.nodes.size(), 0);
    // the QuantizeLinear that ends an operator's group: op -> [Relu ->] Q, every link the only DATA consumer of the one before
    // (Shape readers -- the Resize size arithmetic -- do not count)
    auto data_cons = [&](const std::string& t) {
        std::vector<int> u;
        auto it = cons.find(t);
        if (it != cons.end())
            for (int k : it->second)
                if (g.nodes[k].op != "Shape") u.push_back(k);
        return u;
    };
    auto closing_q = [&](const Node& op, int* relu, int* qn) {
        *relu = -1;
        *qn = -1;
        if (op.out.empty()) return false;
        std::vector<int> u = data_cons(op.out[0]);
        if (u.size() == 1 && g.nodes[u[0]].op == "Relu" && !g.nodes[u[0]].out.empty()) {
            *relu = u[0];
            u = data_cons(g.nodes[u[0]].out[0]);
        }
        if (u.size() != 1 || g.nodes[u[0]].op != "QuantizeLinear" || g.nodes[u[0]].in.size() < 3 || g.nodes[u[0]].out.empty()) return false;
        *qn = u[0];
        return true;
    };
    auto same_scalar = [&](const std::string& a, const std::string& b, bool is_float) {
        if (a == b) return true;
        if (is_float) {
            float x, y;
            return scalar_f32(g, a, &x) && scalar_f32(g, b, &y) && x == y;
        }
        int32_t x, y;
        return scalar_u8(g, a, &x) && scalar_u8(g, b, &y) && x == y;
    };
    for (size_t i = 0; i < g.nodes.size(); i++) {
        Node& n = g.nodes[i];
        if (dead[i]) continue;
        int relu, qn;
        if (n.op == "Conv") {
            if (n.in.size() < 2 || !dq_of.count(n.in[0]) || !dq_of.count(n.in[1])) { *err = "QDQ model: a Conv whose input or weight is not a DequantizeLinear output"; return false; }
            if (!closing_q(n, &relu, &qn)) { *err = "QDQ model: a Conv (+ Relu) that does not end in exactly one QuantizeLinear"; return false; }
            const DQ x = dq_of[n.in[0]], w = dq_of[n.in[1]];
            if (!g.inits.count(w.q)) { *err = "QDQ model: a Conv weight that is not a quantised initializer"; return false; }
            std::string bq;
            if (n.in.size() > 2 && !n.in[2].empty()) {
                if (!dq_of.count(n.in[2]) || !g.inits.count(dq_of[n.in[2]].q)) { *err = "QDQ model: a Conv bias that is not DequantizeLinear of an INT32 initializer"; return false; }
                bq = dq_of[n.in[2]].q;
                // the int32 bias of QLinearConv is in units of x_scale * w_scale[o] with zero point 0 (ONNX QLinearConv-10): a bias
                // DequantizeLinear that says anything else describes a different function than the fused operator computes --
                // a format error here, not silently different logits
                const DQ bd = dq_of[n.in[2]];
                float xs;
                std::vector<float> wsv, bsv;
                auto wsi = g.inits.find(w.scale), bsi = g.inits.find(bd.scale);
                if (!scalar_f32(g, x.scale, &xs) || wsi == g.inits.end() || bsi == g.inits.end() || wsi->second.dims.size() > 1 || bsi->second.dims.size() > 1 ||
                    !wsi->second.floats(wsv) || !bsi->second.floats(bsv) || wsv.empty() || bsv.empty()) {
                    *err = "QDQ model: a Conv's input / weight / bias scales must be float initializers"; return false;
                }
                const size_t nb = std::max(wsv.size(), bsv.size());
                if ((wsv.size() != 1 && wsv.size() != nb) || (bsv.size() != 1 && bsv.size() != nb)) { *err = "QDQ model: a Conv's weight and bias scales have different lengths"; return false; }
                for (size_t o = 0; o < nb; o++) {
                    const float want = xs * wsv[wsv.size() == 1 ? 0 : o], have = bsv[bsv.size() == 1 ? 0 : o];
                    if (!(std::fabs(have - want) <= 1e-6f * std::fabs(want))) { *err = "QDQ model: a Conv bias whose DequantizeLinear scale is not x_scale * w_scale"; return false; }
                }
                if (!bd.zp.empty()) {
                    auto zi = g.inits.find(bd.zp);
                    size_t zn;
                    const uint8_t* zp;
                    bool zero = zi != g.inits.end() && int_bytes(zi->second, 6, 4, &zn, &zp);
                    for (size_t k = 0; zero && k < zn * 4; k++) zero = zp[k] == 0;
                    if (!zero) { *err = "QDQ model: a Conv bias whose DequantizeLinear zero point is not an all-zero INT32 initializer"; return false; }
                }
            }
            const Node& Q = g.nodes[qn];
            if (relu >= 0) {
                int32_t z;
                if (!scalar_u8(g, Q.in[2], &z) || z != 0) { *err = "QDQ model: a Relu in front of a QuantizeLinear whose zero point is not 0"; return false; }
                dead[relu] = 1;
            }
            if (w.zp.empty()) { *err = "QDQ model: a weight DequantizeLinear without a zero point input"; return false; }
            Node f;
            f.op = "QLinearConv";
            f.in = {x.q, x.scale, x.zp, w.q, w.scale, w.zp, Q.in[1], Q.in[2]};
            if (!bq.empty()) f.in.push_back(bq);
            f.out = {Q.out[0]};
            f.ints = n.ints;
            f.strs = n.strs;
            dead[qn] = 1;
            n = std::move(f);
        } else if (n.op == "Add") {
            if (n.in.size() == 2 && !dq_of.count(n.in[0]) && !dq_of.count(n.in[1])) continue;  // (integer arithmetic of a Resize size subgraph)
            if (n.in.size() != 2 || !dq_of.count(n.in[0]) || !dq_of.count(n.in[1])) { *err = "QDQ model: an Add with one DequantizeLinear input and one that is not"; return false; }
            if (!closing_q(n, &relu, &qn)) { *err = "QDQ model: an Add (+ Relu) that does not end in exactly one QuantizeLinear"; return false; }
            const DQ a = dq_of[n.in[0]], b = dq_of[n.in[1]];
            const Node& Q = g.nodes[qn];
            if (relu >= 0) {
                int32_t z;
                if (!scalar_u8(g, Q.in[2], &z) || z != 0) { *err = "QDQ model: a Relu in front of a QuantizeLinear whose zero point is not 0"; return false; }
                dead[relu] = 1;
            }
            Node f;
            f.op = "QLinearAdd";
            f.in = {a.q, a.scale, a.zp, b.q, b.scale, b.zp, Q.in[1], Q.in[2]};
            f.out = {Q.out[0]};
            dead[qn] = 1;
            n = std::move(f);
        } else if (n.op == "MaxPool") {
            if (n.in.empty() || !dq_of.count(n.in[0])) continue;  // (a float MaxPool is a format error of the walker)
            if (!closing_q(n, &relu, &qn) || relu >= 0) { *err = "QDQ model: a MaxPool that does not end in exactly one QuantizeLinear"; return false; }
            const DQ x = dq_of[n.in[0]];
            const Node& Q = g.nodes[qn];
            if (!same_scalar(x.scale, Q.in[1], true) || !same_scalar(x.zp, Q.in[2], false)) { *err = "QDQ model: MaxPool is requantised with different parameters"; return false; }
            n.in[0] = x.q;
            n.out[0] = Q.out[0];
            dead[qn] = 1;
        } else if (n.op == "Resize") {
            if (n.in.empty() || !dq_of.count(n.in[0]) || n.out.empty()) continue;
            const std::vector<int> u = data_cons(n.out[0]);
            if (u.size() == 1 && g.nodes[u[0]].op == "QuantizeLinear" && g.nodes[u[0]].in.size() >= 3 && !g.nodes[u[0]].out.empty()) {
                const DQ x = dq_of[n.in[0]];
                const Node& Q = g.nodes[u[0]];
                if (!same_scalar(x.scale, Q.in[1], true) || !same_scalar(x.zp, Q.in[2], false)) { *err = "QDQ model: Resize is requantised with different parameters"; return false; }
                n.in[0] = x.q;  // Resize on the u8 tensor; the DequantizeLinear behind Q stays and now follows the Resize
                n.out[0] = Q.out[0];
                dead[u[0]] = 1;
            }
        }
    }
    // drop the fused nodes, then every DequantizeLinear whose output nobody reads any more
    std::vector<Node> kept;
    for (size_t i = 0; i < g.nodes.size(); i++)
        if (!dead[i]) kept.push_back(std::move(g.nodes[i]));
    // (a DequantizeLinear that only Shape nodes still read -- the Resize size arithmetic -- goes too: the quantised tensor has the
    //  same shape, the Shape nodes are re-pointed at it)
    for (int pass = 0; pass < 2; pass++) {
        std::map<std::string, int> uses;
        for (const Node& n : kept)
            if (n.op != "Shape")
                for (auto& in : n.in) uses[in]++;
        std::map<std::string, std::string> gone;  // float tensor -> the quantised tensor behind it
        std::vector<Node> k2;
        for (Node& n : kept) {
            if (n.op == "DequantizeLinear" && !n.out.empty() && !n.in.empty() && !uses.count(n.out[0]) && !g.graph_outputs.count(n.out[0])) {
                gone[n.out[0]] = n.in[0];
                continue;
            }
            k2.push_back(std::move(n));
        }
        for (Node& n : k2)
            if (n.op == "Shape" && !n.in.empty() && gone.count(n.in[0])) n.in[0] = gone[n.in[0]];
        kept.swap(k2);
    }
    g.nodes.swap(kept);
    return true;
}

}  // namespace

// 0 ok; 1 malformed; 2 parsed but not a model this path can run.  `data` is a ModelProto whose graph contains QLinearConv, or a
// QDQ-format graph (float Conv nodes fed by DequantizeLinear).
int onnx_q_to_blob(const uint8_t* data, size_t len, std::vector<uint8_t>& blob, OnnxInfo& info, std::string& err) {
    PB m(data, len);
    uint32_t f, wt; uint64_t v; const uint8_t* s; size_t l;
    const uint8_t* gp = nullptr;
    size_t gl = 0;
    while (m.next(f, wt, v, s, l))
        if (f == 7 && wt == 2) { gp = s; gl = l; }
    if (!m.ok || !gp) { err = "not an ONNX ModelProto (no graph)"; return 1; }
    QGraph g;
    std::vector<ValueInfo> inputs, outputs;
    PB gr(gp, gl);
    while (gr.next(f, wt, v, s, l)) {
        if (wt != 2) continue;
        if (f == 1) { Node n; if (!parse_node(s, l, n)) { err = "malformed NodeProto"; return 1; } g.nodes.push_back(std::move(n)); }
        else if (f == 5) { std::string name; Tensor t; if (!parse_tensor(s, l, name, t)) { err = "malformed or external-data TensorProto"; return 1; } g.inits[name] = std::move(t); }
        else if (f == 11) { ValueInfo vi; if (!parse_value_info(s, l, vi)) { err = "malformed graph input"; return 1; } inputs.push_back(std::move(vi)); }
        else if (f == 12) { ValueInfo vi; if (!parse_value_info(s, l, vi)) { err = "malformed graph output"; return 1; } outputs.push_back(std::move(vi)); }
    }
    if (!gr.ok) { err = "malformed GraphProto"; return 1; }
    // every NodeProto names at least one output, every operator but Constant at least one input: the walkers index out[0] / in[0]
    // of the nodes they reach (a MaxPool or QuantizeLinear without outputs used to be a SIGSEGV behind the C ABI -- ADVICE r3)
    for (const Node& n : g.nodes) {
        if (n.out.empty()) { err = "malformed graph: a " + (n.op.empty() ? std::string("node") : n.op) + " node has no outputs"; return 1; }
        if (n.in.empty() && n.op != "Constant") { err = "malformed graph: a " + (n.op.empty() ? std::string("node") : n.op) + " node has no inputs"; return 1; }
    }
    // input 0: the reference's infer_img_pre_proc (predict_onnx.rs:223-265); the int8 zoo file keeps a Float NCHW input
    const ValueInfo* in0 = nullptr;
    for (auto& vi : inputs)
        if (!g.inits.count(vi.name)) { in0 = &vi; break; }
    if (!in0) { err = "model has no image input"; return 1; }
    int col = -1;
    for (size_t i = 0; i < in0->dims.size(); i++)
        if (in0->dims[i] == 3) { col = (int)i; break; }
    if (col < 0) { err = "couldn't locate model's color input by dimension length 3"; return 2; }
    if (in0->dims.size() != 4) { err = "only 4 dimensions supported got " + std::to_string(in0->dims.size()); return 2; }
    if (col != 1) { err = "quantised model: only an NCHW image input is supported"; return 2; }
    if (in0->elem_type != 1) { err = "quantised model: only a Float (f32) image input (QuantizeLinear is the first node) is supported"; return 2; }
    info.input_name = in0->name;
    info.input_dtype = "Float";
    info.input_u8 = false;
    info.input_nhwc = false;
    for (auto& o : outputs) info.output_names.push_back(o.name);
    for (auto& o : outputs) g.graph_outputs[o.name] = 1;
    {
        bool qdq = false, qop = false;
        for (const Node& n : g.nodes) {
            qop |= n.op == "QLinearConv";
            qdq |= n.op == "Conv";
        }
        if (qdq && qop) { err = "the model mixes float Conv and QLinearConv nodes"; return 2; }
        // (also for QOperator files: quantisers older than QLinearAdd leave the residual sums as DQ -> Add -> [Relu ->] Q between
        //  QLinearConv nodes, which ONNX Runtime fuses the same way; without such groups the pass changes nothing)
        if (!qdq_to_qoperator(g, &err)) return 2;
    }

    for (size_t i = 0; i < g.nodes.size(); i++) {
        for (auto& o : g.nodes[i].out) g.producer[o] = (int)i;
        for (auto& in : g.nodes[i].in)
            if (!in.empty() && !g.inits.count(in)) g.consumers[in].push_back((int)i);
    }
    // value-preserving nodes: Identity / Dropout always; a Relu on a u8 tensor whose zero point is 0 (checked by the caller)
    auto transparent = [&](const Node& n) { return n.op == "Identity" || n.op == "Dropout" || n.op == "Relu"; };
    std::function<void(const std::string&, std::vector<int>&, int)> users_d = [&](const std::string& t, std::vector<int>& out, int depth) {
        auto it = g.consumers.find(t);
        if (it == g.consumers.end() || depth > 64) return;
        for (int i : it->second) {
            const Node& n = g.nodes[i];
            if (n.op == "Shape") continue;
            if (transparent(n)) { if (!n.out.empty() && n.in[0] == t) users_d(n.out[0], out, depth + 1); }
            else out.push_back(i);
        }
    };
    auto users = [&](const std::string& t) { std::vector<int> u; users_d(t, u, 0); return u; };
    auto origin = [&](std::string t) {
        for (int guard = 0; guard < 64; guard++) {
            auto it = g.producer.find(t);
            if (it == g.producer.end() || !transparent(g.nodes[it->second]) || g.nodes[it->second].in.empty()) break;
            t = g.nodes[it->second].in[0];
        }
        return t;
    };

    size_t nq = 0;
    for (auto& n : g.nodes) nq += n.op == "QLinearConv";
    int depth = 0;
    bool aux = false;
    switch (nq) {
        case 57: depth = 50; aux = true; break;
        case 55: depth = 50; break;
        case 108: depth = 101; aux = true; break;
        case 106: depth = 101; break;
        default: err = "model has " + std::to_string(nq) + " QLinearConv nodes; fcn_resnet50 has 57 (55 without aux), fcn_resnet101 108 (106)"; return 2;
    }
    int lb[4];
    layer_blocks(depth, lb);

    std::vector<QC> convs;
    std::vector<QBlobAdd> adds;
    std::map<std::string, int32_t> zp_of;  // every u8 tensor the walk produced -> its zero point
    std::vector<char> used(g.nodes.size(), 0);
    auto take = [&](int ni, QC* q) {
        if (used[ni]) { err = "a QLinearConv is reached twice by the walk"; return false; }
        used[ni] = 1;
        if (!read_qconv(g, ni, q, &err)) return false;
        zp_of[q->out] = q->y_zp;
        return true;
    };
    auto sole = [&](const std::string& t, const char* op, const char* what, int* idx) {
        const std::vector<int> u = users(t);
        if (u.size() != 1 || g.nodes[u[0]].op != op) {
            err = std::string("expected exactly one ") + op + " after " + what + ", found " + std::to_string(u.size()) + (u.empty() ? "" : " (" + g.nodes[u[0]].op + ")");
            return false;
        }
        *idx = u[0];
        return true;
    };
    // ---- front: QuantizeLinear of the image ----
    int ni;
    if (!sole(in0->name, "QuantizeLinear", "the image input", &ni)) return 2;
    float in_scale;
    int32_t in_zp;
    if (g.nodes[ni].in.size() < 3 || !scalar_f32(g, g.nodes[ni].in[1], &in_scale) || !scalar_u8(g, g.nodes[ni].in[2], &in_zp) || !qscale_ok(in_scale)) {
        err = "the image's QuantizeLinear needs a float scale and a UINT8 zero point initializer"; return 2;
    }
    std::string t = g.nodes[ni].out[0];
    // ---- stem + max-pool ----
    QC stem;
    if (!sole(t, "QLinearConv", "QuantizeLinear", &ni) || !take(ni, &stem)) return 2;
    if (stem.x_scale != in_scale || stem.x_zp != in_zp) { err = "the stem's x_scale / x_zero_point differ from the image's QuantizeLinear"; return 2; }
    convs.push_back(stem);
    if (!sole(stem.out, "MaxPool", "the stem", &ni)) return 2;
    {
        const Node& mp = g.nodes[ni];
        auto chk = [&](const char* nm, std::vector<int64_t> want) { auto it = mp.ints.find(nm); return it != mp.ints.end() && it->second == want; };
        if (!chk("kernel_shape", {3, 3}) || !chk("strides", {2, 2}) || !chk("pads", {1, 1, 1, 1})) { err = "the stem's MaxPool must be 3x3 / 2 / pad 1"; return 2; }
        auto cm = mp.ints.find("ceil_mode");
        if (cm != mp.ints.end() && cm->second[0] != 0) { err = "MaxPool ceil_mode is not supported"; return 2; }
        t = mp.out[0];
        zp_of[t] = stem.y_zp;
    }
    // ---- bottlenecks ----
    std::string l3;
    for (int L = 0; L < 4; L++) {
        const int planes = 64 << L;
        for (int b = 0; b < lb[L]; b++) {
            const bool has_ds = b == 0;
            const std::vector<int> u = users(t);
            int i1 = -1, ids = -1, iadd = -1;
            size_t n_block_users = 0;
            for (int k : u) {
                const Node& n = g.nodes[k];
                n_block_users++;
                if (n.op == "QLinearAdd") iadd = k;
                else if (n.op == "QLinearConv") {
                    auto wi = g.inits.find(n.in.size() > 3 ? n.in[3] : std::string());
                    const bool w4 = wi != g.inits.end() && wi->second.dims.size() == 4;
                    const int64_t co = w4 ? wi->second.dims[0] : -1;
                    if (w4 && wi->second.dims[2] == 3) n_block_users--;  // the aux head reads layer3's output too: taken by head()
                    else if (co == planes && i1 < 0) i1 = k;
                    else if (co == 4 * planes && ids < 0) ids = k;
                    else { err = "unexpected QLinearConv at the input of a bottleneck"; return 2; }
                } else { err = "unexpected " + n.op + " at the input of a bottleneck"; return 2; }
            }
            if (i1 < 0 || (has_ds ? (ids < 0 || iadd >= 0) : (ids >= 0 || iadd < 0)) || n_block_users != 2) {
                err = "bottleneck " + std::to_string(L + 1) + "." + std::to_string(b) + ": its input must feed conv1 and " + (has_ds ? "the downsample QLinearConv" : "the block's QLinearAdd");
                return 2;
            }
            QC c1, c2, c3, ds;
            int k2, k3;
            if (!take(i1, &c1)) return 2;
            if (!sole(c1.out, "QLinearConv", "conv1", &k2) || !take(k2, &c2)) return 2;
            if (!sole(c2.out, "QLinearConv", "conv2", &k3) || !take(k3, &c3)) return 2;
            std::string idt = t;
            if (has_ds) {
                if (!take(ids, &ds)) return 2;
                idt = ds.out;
            }
            int ka;
            if (!sole(c3.out, "QLinearAdd", "conv3", &ka)) return 2;
            if (!has_ds && ka != iadd) { err = "conv3 and the identity do not meet in the same QLinearAdd"; return 2; }
            const Node& an = g.nodes[ka];
            if (an.in.size() < 8 || an.out.empty()) { err = "QLinearAdd needs 8 inputs"; return 2; }
            const bool a_first = origin(an.in[0]) == c3.out;
            const int ia = a_first ? 0 : 3, ib = a_first ? 3 : 0;
            if (origin(an.in[ia]) != c3.out || origin(an.in[ib]) != origin(idt)) { err = "QLinearAdd does not add conv3 and the block's identity / downsample branch"; return 2; }
            QBlobAdd qa;
            if (!scalar_f32(g, an.in[ia + 1], &qa.a_scale) || !scalar_u8(g, an.in[ia + 2], &qa.a_zp) || !scalar_f32(g, an.in[ib + 1], &qa.b_scale) ||
                !scalar_u8(g, an.in[ib + 2], &qa.b_zp) || !scalar_f32(g, an.in[6], &qa.c_scale) || !scalar_u8(g, an.in[7], &qa.c_zp) || !qscale_ok(qa.a_scale) ||
                !qscale_ok(qa.b_scale) || !qscale_ok(qa.c_scale)) {
                err = "QLinearAdd scales / zero points must be float / UINT8 scalar initializers"; return 2;
            }
            // the sum's A input IS conv3's output tensor: its (scale, zero point) must be conv3's y parameters (the fused conv3 + add
            // epilogue and the blob's directory check rely on it; said here, where the file is read -- ADVICE r3)
            if (qa.a_scale != c3.y_scale || qa.a_zp != c3.y_zp) { err = "QLinearAdd: the parameters of its conv3 input differ from conv3's y_scale / y_zero_point"; return 2; }
            convs.push_back(c1); convs.push_back(c2); convs.push_back(c3);
            if (has_ds) convs.push_back(ds);
            adds.push_back(qa);
            t = an.out[0];
            zp_of[t] = qa.c_zp;
            if (L == 2 && b == lb[2] - 1) l3 = t;
        }
    }
    // ---- heads: QLinearConv 3x3 -> QLinearConv 1x1 -> DequantizeLinear -> Resize -> graph output #k ----
    bool resize_u8 = false;
    int n_heads_seen = 0;
    auto head = [&](const std::string& feat, size_t out_index, const char* what) {
        const std::vector<int> u = users(feat);
        int ih = -1;
        for (int k : u)
            if (g.nodes[k].op == "QLinearConv" && !used[k]) {
                auto wi = g.inits.find(g.nodes[k].in.size() > 3 ? g.nodes[k].in[3] : std::string());
                if (wi != g.inits.end() && wi->second.dims.size() == 4 && wi->second.dims[2] == 3) ih = k;
            }
        if (ih < 0) { err = std::string(what) + ": no 3x3 head QLinearConv reads the feature map"; return 2; }
        QC h0, h1;
        int k1, kd, kr;
        if (!take(ih, &h0)) return 2;
        if (!sole(h0.out, "QLinearConv", what, &k1) || !take(k1, &h1)) return 2;
        // two forms: DequantizeLinear -> Resize (the float logits are resized), or Resize -> DequantizeLinear (onnxruntime's
        // QOperator quantiser keeps Resize on the u8 tensor: the codes are resized, then dequantised)
        const std::vector<int> uh = users(h1.out);
        if (uh.size() != 1 || (g.nodes[uh[0]].op != "DequantizeLinear" && g.nodes[uh[0]].op != "Resize")) {
            err = std::string(what) + ": the logit QLinearConv must feed exactly one DequantizeLinear or Resize"; return 2;
        }
        const bool resize_first = g.nodes[uh[0]].op == "Resize";
        std::string last;
        if (resize_first) {
            kr = uh[0];
            if (!sole(g.nodes[kr].out.empty() ? std::string() : g.nodes[kr].out[0], "DequantizeLinear", what, &kd)) return 2;
            last = g.nodes[kd].out.empty() ? std::string() : g.nodes[kd].out[0];
        } else {
            kd = uh[0];
            const std::vector<int> ur = users(g.nodes[kd].out.empty() ? std::string() : g.nodes[kd].out[0]);
            if (ur.size() != 1 || g.nodes[ur[0]].op != "Resize") { err = std::string(what) + ": the dequantised logits must feed exactly one Resize"; return 2; }
            kr = ur[0];
            last = g.nodes[kr].out.empty() ? std::string() : g.nodes[kr].out[0];
        }
        if (n_heads_seen++ == 0) resize_u8 = resize_first;
        else if (resize_u8 != resize_first) { err = "the two heads order Resize and DequantizeLinear differently"; return 2; }
        const Node& dq = g.nodes[kd];
        float ds_;
        int32_t dz;
        if (dq.in.size() < 3 || !scalar_f32(g, dq.in[1], &ds_) || !scalar_u8(g, dq.in[2], &dz) || ds_ != h1.y_scale || dz != h1.y_zp) {
            err = std::string(what) + ": DequantizeLinear must use the logit conv's y_scale / y_zero_point"; return 2;
        }
        const Node& R = g.nodes[kr];
        auto md = R.strs.find("mode");
        if (md == R.strs.end() || md->second != "linear") { err = std::string(what) + ": Resize mode must be linear"; return 2; }
        auto cm = R.strs.find("coordinate_transformation_mode");
        const std::string cmode = cm == R.strs.end() ? "half_pixel" : cm->second;
        if (cmode != "pytorch_half_pixel" && cmode != "half_pixel") { err = std::string(what) + ": Resize coordinate_transformation_mode '" + cmode + "' is not align_corners=False bilinear"; return 2; }
        if (out_index >= outputs.size() || last.empty() || origin(outputs[out_index].name) != last) { err = std::string(what) + ": its up-sampled logits are not graph output #" + std::to_string(out_index); return 2; }
        convs.push_back(h0); convs.push_back(h1);
        return 0;
    };
    int rc;
    if ((rc = head(t, 0, "classifier"))) return rc;
    if (aux && (rc = head(l3, 1, "aux_classifier"))) return rc;
    const int ncls = convs[aux ? convs.size() - 3 : convs.size() - 1].cout;
    if (ncls <= 0 || ncls > 256 || (aux && convs.back().cout != ncls)) { err = "out and aux heads disagree on the class count"; return 2; }
    const std::vector<ConvSpec> spec = graph_spec(depth, ncls, aux);
    if (spec.size() != convs.size()) { err = "the walk assigned " + std::to_string(convs.size()) + " of " + std::to_string(spec.size()) + " convolutions"; return 2; }
    for (size_t i = 0; i < g.nodes.size(); i++)
        if (g.nodes[i].op == "QLinearConv" && !used[i]) { err = "a QLinearConv node is not part of the FCN-ResNet topology"; return 2; }
    for (size_t i = 0; i < spec.size(); i++) {
        const QC& c = convs[i];
        const ConvSpec& e = spec[i];
        if (c.cout != e.cout || c.cin != e.cin || c.kh != e.k || c.kw != e.k || c.stride != e.stride || c.pad != e.pad || c.dil != e.dil) {
            char buf[256];
            snprintf(buf, sizeof buf, "QLinearConv #%zu (%s) is [%d,%d,%d,%d] s%d p%d d%d, expected [%d,%d,%d,%d] s%d p%d d%d", i, e.name.c_str(), c.cout, c.cin, c.kh,
                     c.kw, c.stride, c.pad, c.dil, e.cout, e.cin, e.k, e.k, e.stride, e.pad, e.dil);
            err = buf;
            return 2;
        }
        if (e.pad != 0 && e.role != 's' && c.x_zp != 0) { err = "QLinearConv (" + e.name + ") pads an input whose zero point is not 0: unsupported"; return 2; }
    }
    // a Relu that was looked through must sit on a u8 tensor whose zero point is 0 (then the clamp at 0 already is the ReLU)
    for (auto& n : g.nodes)
        if (n.op == "Relu") {
            auto z = zp_of.find(origin(n.in.empty() ? std::string() : n.in[0]));
            if (z == zp_of.end() || z->second != 0) { err = "a Relu follows a tensor that is not a quantised tensor with zero point 0"; return 2; }
        }

    // ---- INFURQ01 ----
    const size_t n = spec.size(), na = adds.size();
    size_t off = (kBlobHdr + n * kQEntry + na * kQAdd + 63) & ~(size_t)63;
    struct Offs { size_t w, ws, b; };
    std::vector<Offs> offs(n);
    for (size_t i = 0; i < n; i++) {
        const QC& c = convs[i];
        offs[i].w = off;
        off = (off + (size_t)c.cout * c.cin * c.kh * c.kw + 63) & ~(size_t)63;
        offs[i].ws = off;
        off = (off + (size_t)c.cout * 4 + 63) & ~(size_t)63;
        offs[i].b = off;
        off = (off + (size_t)c.cout * 4 + 63) & ~(size_t)63;
    }
    blob.assign(off, 0);
    memcpy(blob.data(), "INFURQ01", 8);
    auto put32 = [&](size_t o, uint32_t x) { memcpy(blob.data() + o, &x, 4); };
    auto putf = [&](size_t o, float x) { memcpy(blob.data() + o, &x, 4); };
    auto put64 = [&](size_t o, uint64_t x) { memcpy(blob.data() + o, &x, 8); };
    put32(8, (uint32_t)depth); put32(12, (uint32_t)ncls); put32(16, aux ? 1u : 0u); put32(20, (uint32_t)n); put32(24, (uint32_t)na);
    put32(28, resize_u8 ? 1u : 0u);  // flags
    for (size_t i = 0; i < n; i++) {
        const QC& c = convs[i];
        const size_t e = kBlobHdr + i * kQEntry;
        memcpy(blob.data() + e, spec[i].name.c_str(), spec[i].name.size() < 39 ? spec[i].name.size() : 39);
        put32(e + 40, (uint32_t)c.cout); put32(e + 44, (uint32_t)c.cin); put32(e + 48, (uint32_t)c.kh); put32(e + 52, (uint32_t)c.kw);
        putf(e + 56, c.x_scale); put32(e + 60, (uint32_t)c.x_zp); putf(e + 64, c.y_scale); put32(e + 68, (uint32_t)c.y_zp);
        put64(e + 72, offs[i].w); put64(e + 80, offs[i].ws); put64(e + 88, offs[i].b);
        memcpy(blob.data() + offs[i].w, c.w ? c.w : c.w_own.data(), (size_t)c.cout * c.cin * c.kh * c.kw);
        memcpy(blob.data() + offs[i].ws, c.w_scale.data(), (size_t)c.cout * 4);
        memcpy(blob.data() + offs[i].b, c.bias.data(), (size_t)c.cout * 4);
    }
    for (size_t i = 0; i < na; i++) {
        const size_t e = kBlobHdr + n * kQEntry + i * kQAdd;
        putf(e, adds[i].a_scale); put32(e + 4, (uint32_t)adds[i].a_zp); putf(e + 8, adds[i].b_scale); put32(e + 12, (uint32_t)adds[i].b_zp);
        putf(e + 16, adds[i].c_scale); put32(e + 20, (uint32_t)adds[i].c_zp);
    }
    info.depth = depth;
    info.num_classes = ncls;
    info.aux = aux;
    return 0;
}

}  // namespace infur
