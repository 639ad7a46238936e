// onnx_pb.h -- the protobuf wire-format pieces of the ONNX readers (onnx_reader.cpp: float models, onnx_qreader.cpp: the
// QOperator int8 form): a bounds-checked field walker and the few ONNX messages the readers need (TensorProto, NodeProto
// with its attributes, ValueInfoProto).  Internal, host only; no onnx / protobuf library exists in the build image.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace infur {
namespace pb {

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
    bool external = false;     // data_location = EXTERNAL / external_data entries present
    // element count; false for non-positive dims or more than 2^31 elements (no tensor of this model family
    // comes near; an absurd count must not reach a resize())
    bool count(size_t& n) const {
        n = 1;
        for (auto d : dims) {
            if (d <= 0 || d > (int64_t)1 << 31) return false;
            n *= (size_t)d;
            if (n > (size_t)1 << 31) return false;
        }
        return true;
    }
    bool floats(std::vector<float>& out) const {
        size_t n;
        if (dtype != 1 || external || !count(n)) return false;  // FLOAT, inline
        if (raw && raw_len / 4 == n && raw_len % 4 == 0) {
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
    std::map<std::string, std::string> strs;
    const uint8_t* value_t = nullptr;  // Constant: the `value` TensorProto
    size_t value_len = 0;
};

inline void read_packed_i64(const uint8_t* d, size_t n, std::vector<int64_t>& v) {
    PB r(d, n);
    while (!r.done()) v.push_back((int64_t)r.varint());
}

inline bool parse_tensor(const uint8_t* d, size_t n, std::string& name, Tensor& t) {
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
        else if (f == 13 && wt == 2) t.external = true;   // external_data entry
        else if (f == 14 && v != 0) t.external = true;    // data_location = EXTERNAL
    }
    return r.ok;
}

inline bool parse_attr(const uint8_t* d, size_t n, Node& node) {
    PB r(d, n);
    uint32_t f, wt; uint64_t v; const uint8_t* s; size_t l;
    std::string name, sv;
    std::vector<int64_t> ints;
    bool has_i = false, has_f = false, has_s = false;
    int64_t iv = 0;
    float fv = 0;
    const uint8_t* tv = nullptr;
    size_t tl = 0;
    while (r.next(f, wt, v, s, l)) {
        if (f == 1 && wt == 2) name.assign((const char*)s, l);
        else if (f == 2 && wt == 5) { uint32_t u = (uint32_t)v; memcpy(&fv, &u, 4); has_f = true; }
        else if (f == 3 && wt == 0) { iv = (int64_t)v; has_i = true; }
        else if (f == 4 && wt == 2) { sv.assign((const char*)s, l); has_s = true; }
        else if (f == 5 && wt == 2) { tv = s; tl = l; }
        else if (f == 8) { if (wt == 2) read_packed_i64(s, l, ints); else ints.push_back((int64_t)v); }
    }
    if (!ints.empty()) node.ints[name] = ints;
    else if (has_i) node.ints[name] = {iv};
    if (has_f) node.f[name] = fv;
    if (has_s) node.strs[name] = sv;
    if (tv && name == "value") { node.value_t = tv; node.value_len = tl; }
    return r.ok;
}

inline bool parse_node(const uint8_t* d, size_t n, Node& node) {
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

inline bool parse_value_info(const uint8_t* d, size_t n, ValueInfo& vi) {
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

}  // namespace pb
}  // namespace infur
