// onnx_dense.hpp -- reads an ONNX file that is a stack of dense layers (the BirdNET geomodel's shape: Gemm / MatMul + Add,
// Relu, a final Sigmoid or Softmax) straight into the BHC1 in-memory form (model.hpp CustomModel), so that
// bh_range_filter_create takes the very file birdnet_onnx::RangeFilter::builder().model_path() takes
// (reference src/inference/range_filter.rs:19-37; fixture tests/fixtures/fixture-geomodel.onnx = Gemm + Sigmoid).
//
// Hand-written protobuf wire-format walk (no protobuf / onnx dependency): ModelProto.graph (7) -> GraphProto.node (1),
// .initializer (5), .input (11), .output (12); NodeProto input (1) output (2) op_type (4) attribute (5); AttributeProto name (1)
// f (2) i (3); TensorProto dims (1) data_type (2) float_data (4) name (8) raw_data (9).  Untrusted input: every length is
// checked against the buffer, every dimension product against the tensor's payload.  Anything that is not a dense stack is
// refused with the operator's name (the conv models go through birda_amd/convert.py -> BHM1).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "model.hpp"

namespace bh {
namespace onnxd {

struct Span { const uint8_t *p = nullptr; size_t n = 0; };

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    explicit Reader(Span s) : p(s.p), end(s.p + s.n) {}
    bool more() const { return ok && p < end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) { ok = false; return 0; }
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    // one field: number, wire type, and either the varint / fixed value or the length-delimited span
    bool field(uint32_t &no, uint32_t &wt, uint64_t &val, Span &sp) {
        const uint64_t key = varint();
        if (!ok) return false;
        no = (uint32_t)(key >> 3);
        wt = (uint32_t)(key & 7);
        sp = Span{};
        val = 0;
        switch (wt) {
        case 0: val = varint(); return ok;
        case 1: if ((size_t)(end - p) < 8) return ok = false; memcpy(&val, p, 8); sp = Span{p, 8}; p += 8; return true;
        case 5: if ((size_t)(end - p) < 4) return ok = false; { uint32_t v32; memcpy(&v32, p, 4); val = v32; } sp = Span{p, 4}; p += 4; return true;
        case 2: {
            const uint64_t len = varint();
            if (!ok || len > (uint64_t)(end - p)) return ok = false;
            sp = Span{p, (size_t)len};
            p += len;
            return true;
        }
        default: return ok = false;   // groups (3, 4) do not occur in ONNX files
        }
    }
};

inline std::string str(Span s) { return std::string(reinterpret_cast<const char *>(s.p), s.n); }

struct Tensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
    bool is_float = false;
};

struct Node {
    std::string op;
    std::vector<std::string> in, out;
    std::map<std::string, float> f;
    std::map<std::string, int64_t> i;
};

inline bool parse_tensor(Span s, std::string &name, Tensor &t, std::string &err) {
    Reader r(s);
    uint32_t no, wt; uint64_t v; Span sp;
    int64_t dtype = 1;
    Span raw{};
    bool has_raw = false;
    std::vector<float> floats;
    while (r.more()) {
        if (!r.field(no, wt, v, sp)) break;
        if (no == 1) {                       // dims: packed or repeated varints
            if (wt == 2) { Reader q(sp); while (q.more()) { const uint64_t d = q.varint(); if (q.ok) t.dims.push_back((int64_t)d); } if (!q.ok) r.ok = false; }
            else t.dims.push_back((int64_t)v);
        } else if (no == 2) dtype = (int64_t)v;
        else if (no == 4) {                  // float_data: packed or repeated fixed32
            if (wt == 2) { if (sp.n % 4) { r.ok = false; break; } const size_t k = floats.size(); floats.resize(k + sp.n / 4); memcpy(floats.data() + k, sp.p, sp.n); }
            else if (wt == 5) { float f; memcpy(&f, sp.p, 4); floats.push_back(f); }
        } else if (no == 8) name = str(sp);
        else if (no == 9) { raw = sp; has_raw = true; }
        else if (no == 13 || no == 14) { err = "tensor '" + name + "' keeps its data in an external file: not supported"; return false; }
    }
    if (!r.ok) { err = "malformed TensorProto"; return false; }
    t.is_float = dtype == 1;
    if (!t.is_float) return true;            // (shape constants of other types are never read by a dense stack)
    uint64_t count = 1;
    for (int64_t d : t.dims) {
        if (d < 0 || d > (1 << 26)) { err = "tensor '" + name + "': bad dimension"; return false; }
        count *= (uint64_t)d;
        if (count > (1ull << 31)) { err = "tensor '" + name + "': too large"; return false; }
    }
    if (has_raw) {
        if (raw.n != count * 4) { err = "tensor '" + name + "': raw_data size does not match its dims"; return false; }
        t.data.resize(count);
        if (count) memcpy(t.data.data(), raw.p, raw.n);
    } else {
        if (floats.size() != count) { err = "tensor '" + name + "': float_data size does not match its dims"; return false; }
        t.data = std::move(floats);
    }
    return true;
}

inline bool parse_node(Span s, Node &n) {
    Reader r(s);
    uint32_t no, wt; uint64_t v; Span sp;
    while (r.more()) {
        if (!r.field(no, wt, v, sp)) break;
        if (no == 1 && wt == 2) n.in.push_back(str(sp));
        else if (no == 2 && wt == 2) n.out.push_back(str(sp));
        else if (no == 4 && wt == 2) n.op = str(sp);
        else if (no == 5 && wt == 2) {
            Reader a(sp);
            std::string an; bool hf = false, hi = false; float fv = 0; int64_t iv = 0;
            uint32_t no2, wt2; uint64_t v2; Span sp2;
            while (a.more()) {
                if (!a.field(no2, wt2, v2, sp2)) break;
                if (no2 == 1 && wt2 == 2) an = str(sp2);
                else if (no2 == 2 && wt2 == 5) { memcpy(&fv, sp2.p, 4); hf = true; }
                else if (no2 == 3 && wt2 == 0) { iv = (int64_t)v2; hi = true; }
            }
            if (!a.ok) return false;
            if (hf) n.f[an] = fv;
            if (hi) n.i[an] = iv;
        }
    }
    return r.ok;
}

inline std::string value_info_name(Span s) {
    Reader r(s);
    uint32_t no, wt; uint64_t v; Span sp;
    while (r.more()) {
        if (!r.field(no, wt, v, sp)) break;
        if (no == 1 && wt == 2) return str(sp);
    }
    return "";
}

// The whole file -> CustomModel.  Activation codes are kernels.hpp's / modelfile.py's (ACT_NONE 0, ACT_RELU 1, ACT_SIGMOID 6);
// output_activation 0 none, 1 sigmoid, 2 softmax.  A final Sigmoid is folded into the last layer (act = ACT_SIGMOID, output
// activation none): the scores of ALL classes then leave the GEMM's epilogue activated, which is what a range filter returns.
inline bool load_dense_onnx(const char *path, CustomModel &m, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ONNX file ") + path; return false; }
    std::vector<uint8_t> buf;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long sz = ftell(f);
        if (sz > 0 && sz < (1l << 30)) { buf.resize((size_t)sz); rewind(f); if (fread(buf.data(), 1, buf.size(), f) != buf.size()) buf.clear(); }
    }
    fclose(f);
    if (buf.empty()) { err = std::string(path) + ": empty, unreadable or larger than 1 GiB"; return false; }
    Reader top(Span{buf.data(), buf.size()});
    Span graph{};
    uint32_t no, wt; uint64_t v; Span sp;
    while (top.more()) {
        if (!top.field(no, wt, v, sp)) break;
        if (no == 7 && wt == 2) graph = sp;
    }
    if (!top.ok || !graph.p) { err = std::string(path) + ": not an ONNX ModelProto (no graph)"; return false; }
    std::vector<Node> nodes;
    std::map<std::string, Tensor> init;
    std::vector<std::string> inputs, outputs;
    Reader g(graph);
    while (g.more()) {
        if (!g.field(no, wt, v, sp)) break;
        if (no == 1 && wt == 2) { Node n; if (!parse_node(sp, n)) { err = "malformed NodeProto"; return false; } nodes.push_back(std::move(n)); }
        else if (no == 5 && wt == 2) { std::string name; Tensor t; if (!parse_tensor(sp, name, t, err)) return false; init[name] = std::move(t); }
        else if (no == 11 && wt == 2) inputs.push_back(value_info_name(sp));
        else if (no == 12 && wt == 2) outputs.push_back(value_info_name(sp));
        if (nodes.size() > 4096) { err = "too many nodes for a dense stack"; return false; }
    }
    if (!g.ok) { err = std::string(path) + ": malformed GraphProto"; return false; }
    std::string cur;
    for (const auto &nm : inputs) if (!init.count(nm)) { if (!cur.empty()) { err = "a dense stack has one data input"; return false; } cur = nm; }
    if (cur.empty() || outputs.size() != 1) { err = "a dense stack has one data input and one output"; return false; }

    struct Dense { std::vector<float> w, b; uint32_t in = 0, out = 0, act = 0; };
    std::vector<Dense> layers;
    uint32_t out_act = 0;
    auto weight = [&](const std::string &name, const Tensor *&t) { auto it = init.find(name); if (it == init.end() || !it->second.is_float) return false; t = &it->second; return true; };
    // walk the chain: every node must consume `cur` as its first (data) input
    std::vector<bool> used(nodes.size(), false);
    for (size_t guard = 0; guard <= nodes.size(); guard++) {
        if (cur == outputs[0]) break;
        size_t k = nodes.size();
        for (size_t i = 0; i < nodes.size(); i++) if (!used[i] && !nodes[i].in.empty() && nodes[i].in[0] == cur) { k = i; break; }
        if (k == nodes.size()) { err = "graph is not a chain from its input to its output (at '" + cur + "')"; return false; }
        used[k] = true;
        const Node &n = nodes[k];
        if (n.out.empty()) { err = "node without an output"; return false; }
        if (out_act != 0) { err = "operator '" + n.op + "' after the output activation"; return false; }
        if (n.op == "Gemm" || n.op == "MatMul") {
            const Tensor *W = nullptr;
            if (n.in.size() < 2 || !weight(n.in[1], W) || W->dims.size() != 2) { err = n.op + ": weight must be a 2-D float initializer"; return false; }
            const bool gemm = n.op == "Gemm";
            const float alpha = gemm && n.f.count("alpha") ? n.f.at("alpha") : 1.0f, beta = gemm && n.f.count("beta") ? n.f.at("beta") : 1.0f;
            if (gemm && n.i.count("transA") && n.i.at("transA")) { err = "Gemm: transA is not a dense layer"; return false; }
            const bool tb = gemm && n.i.count("transB") && n.i.at("transB");
            Dense L;
            L.in = (uint32_t)(tb ? W->dims[1] : W->dims[0]);
            L.out = (uint32_t)(tb ? W->dims[0] : W->dims[1]);
            if (L.in == 0 || L.out == 0 || L.in > (1u << 20) || L.out > (1u << 24)) { err = n.op + ": bad weight shape"; return false; }
            L.w.resize((size_t)L.in * L.out);
            for (uint32_t i = 0; i < L.in; i++)
                for (uint32_t o = 0; o < L.out; o++) L.w[(size_t)i * L.out + o] = alpha * (tb ? W->data[(size_t)o * L.in + i] : W->data[(size_t)i * L.out + o]);
            L.b.assign(L.out, 0.0f);
            if (gemm && n.in.size() >= 3 && !n.in[2].empty()) {
                const Tensor *B = nullptr;
                if (!weight(n.in[2], B) || (B->data.size() != L.out && B->data.size() != 1)) { err = "Gemm: bias must be a float initializer of the output width"; return false; }
                for (uint32_t o = 0; o < L.out; o++) L.b[o] = beta * B->data[B->data.size() == 1 ? 0 : o];
            }
            if (!layers.empty() && layers.back().out != L.in) { err = n.op + ": layer widths do not chain"; return false; }
            layers.push_back(std::move(L));
        } else if (n.op == "Add") {
            const Tensor *B = nullptr;
            if (layers.empty() || layers.back().act != 0 || n.in.size() != 2 || !weight(n.in[1], B) || B->data.size() != layers.back().out) {
                err = "Add: only a bias (float initializer of the layer's width) right after MatMul / Gemm is a dense layer"; return false;
            }
            for (uint32_t o = 0; o < layers.back().out; o++) layers.back().b[o] += B->data[o];
        } else if (n.op == "Relu") {
            if (layers.empty() || layers.back().act != 0) { err = "Relu without a dense layer in front of it"; return false; }
            layers.back().act = 1;   // ACT_RELU
        } else if (n.op == "Sigmoid") {
            if (layers.empty() || layers.back().act != 0) { err = "Sigmoid without a dense layer in front of it"; return false; }
            layers.back().act = 6;   // ACT_SIGMOID; (a hidden sigmoid layer is expressed the same way)
        } else if (n.op == "Softmax") {
            if (layers.empty() || layers.back().act != 0) { err = "Softmax without a dense layer in front of it"; return false; }
            out_act = 2;
        } else if (n.op == "Identity" || n.op == "Flatten" || n.op == "Dropout") {
            // shape-preserving on [batch, width] rows
        } else {
            err = "operator '" + n.op + "' is not part of a dense stack (Gemm, MatMul + Add, Relu, Sigmoid, Softmax)"; return false;
        }
        cur = n.out[0];
    }
    if (cur != outputs[0] || layers.empty()) { err = "graph output is not reached from its input through dense layers"; return false; }
    if (layers.size() > 64) { err = "more than 64 dense layers"; return false; }
    // assemble the BHC1 in-memory form (16-float aligned tensors, as modelfile.py writes them)
    m = CustomModel{};
    memcpy(m.h.magic, "BHC1", 4);
    m.h.version = 1;
    m.h.input_dim = layers.front().in;
    m.h.n_layers = (uint32_t)layers.size();
    m.h.n_classes = layers.back().out;
    m.h.output_activation = out_act;
    for (const auto &L : layers) {
        CustomLayerRec rec{};
        rec.in_dim = L.in; rec.out_dim = L.out; rec.act = L.act;
        m.blob.resize((m.blob.size() + 15) / 16 * 16, 0.0f);
        rec.w_off = m.blob.size();
        m.blob.insert(m.blob.end(), L.w.begin(), L.w.end());
        m.blob.resize((m.blob.size() + 15) / 16 * 16, 0.0f);
        rec.b_off = m.blob.size();
        m.blob.insert(m.blob.end(), L.b.begin(), L.b.end());
        m.layers.push_back(rec);
    }
    m.h.blob_floats = m.blob.size();
    return true;
}

}  // namespace onnxd
}  // namespace bh
