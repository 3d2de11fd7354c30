// onnx_graph.hpp -- the ONNX wire format as far as the classifier's model files need it: ModelProto.graph -> nodes (with their
// attributes), initializers (float32 / float64 / int64 / int32 payloads; raw_data or the typed repeated fields), Constant nodes
// as initializers by another spelling, graph inputs / outputs with their static shapes.  Hand-written protobuf walk on
// onnx_dense.hpp's Reader (no protobuf / onnx dependency).  Untrusted input: every length is checked against the buffer, every
// dimension product against the tensor's payload.  Used by onnx_conv.hpp (the conv-stack walk) and onnx_frontend.hpp (the
// front-end evaluator); reference: the file ClassifierBuilder::model_path() names (src/inference/classifier.rs:269-283).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "model.hpp"
#include "onnx_dense.hpp"

namespace bh {
namespace onnxc {

using onnxd::Reader;
using onnxd::Span;
using onnxd::str;

// activation / output-activation codes of the container (kernels.hpp Act, modelfile.py)
enum : uint32_t { A_NONE = 0, A_RELU = 1, A_RELU6 = 2, A_SWISH = 3, A_GELU_ERF = 4, A_GELU_TANH = 5, A_SIGMOID = 6 };
enum : uint32_t { O_NONE = 0, O_SIGMOID = 1, O_SOFTMAX = 2 };

struct Attr {
    bool has_f = false, has_i = false;
    float f = 0.f;
    int64_t i = 0;
    std::string s;
    std::vector<int64_t> ints;
    std::vector<float> floats;
};
struct Node {
    std::string op, name;
    std::vector<std::string> in, out;
    std::map<std::string, Attr> a;
    int64_t geti(const char *k, int64_t dflt) const { auto it = a.find(k); return it != a.end() && it->second.has_i ? it->second.i : dflt; }
    float getf(const char *k, float dflt) const { auto it = a.find(k); return it != a.end() && it->second.has_f ? it->second.f : dflt; }
    const std::vector<int64_t> *ints(const char *k) const { auto it = a.find(k); return it != a.end() && !it->second.ints.empty() ? &it->second.ints : nullptr; }
    std::string gets(const char *k, const char *dflt) const { auto it = a.find(k); return it != a.end() && !it->second.s.empty() ? it->second.s : std::string(dflt); }
};
// an initializer: float32 data stays in the file buffer (raw_data) or in `fl` (float_data); int64 / int32 values in `il`
struct Tensor {
    std::vector<int64_t> dims;
    int64_t dtype = 1;
    Span raw{};
    std::vector<float> fl;
    std::vector<int64_t> il;
    std::vector<double> dl;       // float64 constants (a front-end's eps, exponents, ...): only onnx_frontend.hpp reads them
    uint64_t count = 0;
    bool has_raw = false;
    bool is_f32() const { return dtype == 1; }
    // (parse_tensor leaves exactly one payload of `count` elements: raw_data when present -- float_data beside it is dropped -- else float_data)
    float at(uint64_t i) const {
        if (!has_raw) return fl[i];
        float v; memcpy(&v, raw.p + 4 * i, 4); return v;
    }
};
struct ValueInfo { std::string name; std::vector<int64_t> dims; };   // symbolic dimension: -1
struct Graph {
    std::vector<Node> nodes;
    std::map<std::string, Tensor> init;
    std::vector<ValueInfo> inputs, outputs;
};

inline void packed_ints(uint32_t wt, uint64_t v, Span sp, std::vector<int64_t> &out, bool &ok) {
    if (wt == 2) { Reader q(sp); while (q.more()) { const uint64_t d = q.varint(); if (q.ok) out.push_back((int64_t)d); } if (!q.ok) ok = false; }
    else if (wt == 0) out.push_back((int64_t)v);
}

inline bool parse_tensor(Span s, std::string &name, Tensor &t, std::string &err) {
    Reader r(s);
    uint32_t no, wt; uint64_t v; Span sp;
    bool has_raw = false;
    while (r.more()) {
        if (!r.field(no, wt, v, sp)) break;
        if (no == 1) packed_ints(wt, v, sp, t.dims, r.ok);
        else if (no == 2) t.dtype = (int64_t)v;
        else if (no == 4) {                  // float_data
            if (wt == 2) { if (sp.n % 4) { r.ok = false; break; } const size_t k = t.fl.size(); t.fl.resize(k + sp.n / 4); if (sp.n) memcpy(t.fl.data() + k, sp.p, sp.n); }
            else if (wt == 5) { float f; memcpy(&f, sp.p, 4); t.fl.push_back(f); }
        } else if (no == 5 || no == 7) packed_ints(wt, v, sp, t.il, r.ok);   // int32_data / int64_data
        else if (no == 10) {                 // double_data
            if (wt == 2) { if (sp.n % 8) { r.ok = false; break; } const size_t k = t.dl.size(); t.dl.resize(k + sp.n / 8); if (sp.n) memcpy(t.dl.data() + k, sp.p, sp.n); }
            else if (wt == 1) { double d; memcpy(&d, sp.p, 8); t.dl.push_back(d); }
        }
        else if (no == 8) name = str(sp);
        else if (no == 9) { t.raw = sp; has_raw = true; }
        else if (no == 13 || no == 14) { err = "tensor '" + name + "' keeps its data in an external file: not supported"; return false; }
    }
    if (!r.ok) { err = "malformed TensorProto"; return false; }
    t.count = 1;
    for (int64_t d : t.dims) {
        if (d < 0 || d > (1ll << 28)) { err = "tensor '" + name + "': bad dimension"; return false; }
        t.count *= (uint64_t)d;
        if (t.count > (1ull << 31)) { err = "tensor '" + name + "': too large"; return false; }
    }
    t.has_raw = has_raw;
    if (t.dtype == 1) {
        // a tensor carrying BOTH raw_data and a shorter float_data was read through float_data beyond its end.
        // raw_data wins, as in onnx's own helpers, and whatever float_data came with it is dropped.
        if (has_raw) { if (t.raw.n != t.count * 4) { err = "tensor '" + name + "': raw_data size does not match its dims"; return false; } t.fl.clear(); }
        else if (t.fl.size() != t.count) { err = "tensor '" + name + "': float_data size does not match its dims"; return false; }
    } else if (t.dtype == 7 || t.dtype == 6) {   // int64 / int32 (axes, shapes, Slice bounds)
        if (has_raw) {
            const size_t w = t.dtype == 7 ? 8 : 4;
            if (t.raw.n != t.count * w) { err = "tensor '" + name + "': raw_data size does not match its dims"; return false; }
            t.il.resize(t.count);
            for (uint64_t i = 0; i < t.count; i++) {
                if (w == 8) { int64_t x; memcpy(&x, t.raw.p + 8 * i, 8); t.il[i] = x; }
                else { int32_t x; memcpy(&x, t.raw.p + 4 * i, 4); t.il[i] = x; }
            }
        } else if (t.il.size() != t.count) { err = "tensor '" + name + "': integer data size does not match its dims"; return false; }
    } else if (t.dtype == 11) {                  // float64
        if (has_raw) {
            if (t.raw.n != t.count * 8) { err = "tensor '" + name + "': raw_data size does not match its dims"; return false; }
            t.dl.resize(t.count);
            if (t.count) memcpy(t.dl.data(), t.raw.p, t.raw.n);
        } else if (t.dl.size() != t.count) { err = "tensor '" + name + "': double_data size does not match its dims"; return false; }
    } else if (t.dtype == 9) {                   // bool: one byte per element in raw_data, else int32_data
        if (has_raw) {
            if (t.raw.n != t.count) { err = "tensor '" + name + "': raw_data size does not match its dims"; return false; }
            t.il.resize(t.count);
            for (uint64_t i = 0; i < t.count; i++) t.il[i] = t.raw.p[i] != 0;
        } else if (t.il.size() != t.count) { err = "tensor '" + name + "': integer data size does not match its dims"; return false; }
    }   // (other element types are carried without data: nothing reads them, and the front-end evaluator refuses them by name)
    return true;
}

inline bool parse_node(Span s, Node &n, std::map<std::string, Tensor> *const_out, std::string &err) {
    Reader r(s);
    uint32_t no, wt; uint64_t v; Span sp;
    // A Constant node's value in whichever spelling it arrives (`value` tensor, `value_float` / `value_int` / `value_floats` /
    // `value_ints`): kept aside until the whole node is read -- `op_type` may follow the attributes on the wire, and only a node
    // that IS a Constant turns into an initializer (a ConstantOfShape's `value` tensor was hoisted in its output's place).
    Span const_tensor{};
    std::string const_kind;
    Attr const_attr;
    while (r.more()) {
        if (!r.field(no, wt, v, sp)) break;
        if (no == 1 && wt == 2) n.in.push_back(str(sp));
        else if (no == 2 && wt == 2) n.out.push_back(str(sp));
        else if (no == 3 && wt == 2) n.name = str(sp);
        else if (no == 4 && wt == 2) n.op = str(sp);
        else if (no == 5 && wt == 2) {
            Reader a(sp);
            std::string an;
            Attr at;
            Span tensor{};
            uint32_t no2, wt2; uint64_t v2; Span sp2;
            while (a.more()) {
                if (!a.field(no2, wt2, v2, sp2)) break;
                if (no2 == 1 && wt2 == 2) an = str(sp2);
                else if (no2 == 2 && wt2 == 5) { memcpy(&at.f, sp2.p, 4); at.has_f = true; }
                else if (no2 == 3 && wt2 == 0) { at.i = (int64_t)v2; at.has_i = true; }
                else if (no2 == 4 && wt2 == 2) at.s = str(sp2);
                else if (no2 == 5 && wt2 == 2) tensor = sp2;
                else if (no2 == 7) {             // floats: packed or repeated fixed32
                    if (wt2 == 2) { if (sp2.n % 4) { a.ok = false; break; } const size_t k = at.floats.size(); at.floats.resize(k + sp2.n / 4); if (sp2.n) memcpy(at.floats.data() + k, sp2.p, sp2.n); }
                    else if (wt2 == 5) { float f; memcpy(&f, sp2.p, 4); at.floats.push_back(f); }
                }
                else if (no2 == 8) packed_ints(wt2, v2, sp2, at.ints, a.ok);
            }
            if (!a.ok) { err = "malformed AttributeProto"; return false; }
            if (an == "value" && tensor.p) { const_tensor = tensor; const_kind = an; }
            else if (an == "value_float" || an == "value_int" || an == "value_floats" || an == "value_ints") { const_kind = an; const_attr = at; }
            n.a[an] = std::move(at);
        }
    }
    if (!r.ok) { err = "malformed NodeProto"; return false; }
    if (const_out && n.op == "Constant") {
        // its value is an initializer by another spelling -- registered only with the payload its spelling names (a `value_float`
        // attribute that carries an integer, a `value_int` without one: "malformed", not a one-element tensor without data)
        if (n.out.empty() || const_kind.empty()) { err = "Constant node '" + n.name + "' without an output or without a value this reader knows (value, value_float, value_int, value_floats, value_ints)"; return false; }
        Tensor t;
        if (const_kind == "value") {
            std::string tn;
            if (!parse_tensor(const_tensor, tn, t, err)) return false;
        } else if (const_kind == "value_float" && const_attr.has_f) { t.dtype = 1; t.fl = {const_attr.f}; t.count = 1; }
        else if (const_kind == "value_int" && const_attr.has_i) { t.dtype = 7; t.il = {const_attr.i}; t.count = 1; }
        else if (const_kind == "value_floats") { t.dtype = 1; t.fl = const_attr.floats; t.dims = {(int64_t)const_attr.floats.size()}; t.count = t.fl.size(); }
        else if (const_kind == "value_ints") { t.dtype = 7; t.il = const_attr.ints; t.dims = {(int64_t)const_attr.ints.size()}; t.count = t.il.size(); }
        else { err = "malformed Constant node '" + n.name + "': attribute " + const_kind + " without its payload"; return false; }
        (*const_out)[n.out[0]] = std::move(t);
    }
    return true;
}

// ValueInfoProto: name (1), type (2) -> TypeProto.tensor_type (1) -> shape (2) -> dim (1) -> dim_value (1) | dim_param (2)
inline ValueInfo parse_value_info(Span s) {
    ValueInfo vi;
    Reader r(s);
    uint32_t no, wt; uint64_t v; Span sp;
    while (r.more()) {
        if (!r.field(no, wt, v, sp)) break;
        if (no == 1 && wt == 2) vi.name = str(sp);
        else if (no == 2 && wt == 2) {
            Reader ty(sp);
            while (ty.more()) {
                if (!ty.field(no, wt, v, sp)) break;
                if (no != 1 || wt != 2) continue;
                Reader tt(sp);
                while (tt.more()) {
                    if (!tt.field(no, wt, v, sp)) break;
                    if (no != 2 || wt != 2) continue;
                    Reader sh(sp);
                    while (sh.more()) {
                        if (!sh.field(no, wt, v, sp)) break;
                        if (no != 1 || wt != 2) continue;
                        Reader dm(sp);
                        int64_t val = -1;
                        uint32_t n3, w3; uint64_t v3; Span s3;
                        while (dm.more()) {
                            if (!dm.field(n3, w3, v3, s3)) break;
                            if (n3 == 1 && w3 == 0) val = (int64_t)v3;
                        }
                        vi.dims.push_back(val);
                    }
                }
            }
        }
    }
    return vi;
}

inline bool parse_graph(Span file, Graph &g, std::string &err) {
    Reader top(file);
    Span graph{};
    uint32_t no, wt; uint64_t v; Span sp;
    while (top.more()) {
        if (!top.field(no, wt, v, sp)) break;
        if (no == 7 && wt == 2) graph = sp;
    }
    if (!top.ok || !graph.p) { err = "not an ONNX ModelProto (no graph)"; return false; }
    Reader r(graph);
    while (r.more()) {
        if (!r.field(no, wt, v, sp)) break;
        if (no == 1 && wt == 2) {
            Node n;
            if (!parse_node(sp, n, &g.init, err)) return false;
            if (n.op != "Constant") g.nodes.push_back(std::move(n));
        } else if (no == 5 && wt == 2) {
            std::string name; Tensor t;
            if (!parse_tensor(sp, name, t, err)) return false;
            g.init[name] = std::move(t);
        } else if (no == 11 && wt == 2) g.inputs.push_back(parse_value_info(sp));
        else if (no == 12 && wt == 2) g.outputs.push_back(parse_value_info(sp));
        if (g.nodes.size() > 65536) { err = "more than 65 536 nodes"; return false; }
    }
    if (!r.ok) { err = "malformed GraphProto"; return false; }
    return true;
}

}  // namespace onnxc
}  // namespace bh
