// BHM1 model container reader (format: birda_amd/modelfile.py).  Host only.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace bh {

enum Op : uint32_t { OP_CONV = 1, OP_DWCONV = 2, OP_PWCONV = 3, OP_GAP = 4, OP_DENSE = 5, OP_SCALE = 6 };   // OP_SCALE: x * gate[n][c] (squeeze-excite; gate = res_tensor)
constexpr uint32_t NO_TENSOR = 0xFFFFFFFFu;

#pragma pack(push, 1)
struct HeaderRec {
    char magic[4];
    uint32_t version, family, sample_rate, sample_count;
    float segment_duration;
    uint32_t n_classes, embedding_dim, n_branches, n_layers, output_activation, embedding_tensor;
    uint64_t blob_offset, blob_floats;
    uint32_t spec_h, spec_w;
    float norm_eps;
};
struct BranchRec {
    uint32_t frame_length, frame_step, fft_length, n_bins, n_mels, n_frames;
    float fmin, fmax, mag_scale, out_scale, out_shift;
    uint32_t flags;
    uint64_t mel_w_off;
};
struct LayerRec {
    uint32_t op, act, in_tensor, res_tensor, cin, cout, kh, kw, sh, sw, pad_t, pad_l;
    uint32_t in_h, in_w, out_h, out_w, in_layout, reserved;
    uint64_t w_off, b_off;
};
#pragma pack(pop)

struct Model {
    HeaderRec h{};
    std::vector<BranchRec> branches;
    std::vector<LayerRec> layers;
    std::vector<float> blob;
    std::vector<uint64_t> tensor_floats;  // per-segment floats of tensor i (0 = spectrogram)

    uint64_t macs_per_segment() const {
        uint64_t t = 0;
        for (const auto &L : layers) {
            const uint64_t px = (uint64_t)L.out_h * L.out_w;
            if (L.op == OP_CONV) t += px * L.kh * L.kw * L.cin * L.cout;
            else if (L.op == OP_DWCONV) t += px * L.kh * L.kw * L.cout;
            else if (L.op == OP_PWCONV || L.op == OP_DENSE) t += px * L.cin * L.cout;
        }
        return t;
    }
};

inline bool load_model(const char *path, Model &m, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open model file ") + path; return false; }
    auto bad = [&](const char *why) { err = std::string(path) + ": " + why; fclose(f); return false; };
    unsigned char hdr[256];
    if (fread(hdr, 1, 256, f) != 256) return bad("truncated header");
    memcpy(&m.h, hdr, sizeof m.h);
    if (memcmp(m.h.magic, "BHM1", 4) != 0 || m.h.version != 1) return bad("not a BHM1 v1 model");
    if (m.h.n_branches == 0 || m.h.n_branches > 16 || m.h.n_layers == 0 || m.h.n_layers > 4096) return bad("bad counts");
    m.branches.resize(m.h.n_branches);
    m.layers.resize(m.h.n_layers);
    for (auto &b : m.branches) {
        unsigned char rec[64];
        if (fread(rec, 1, 64, f) != 64) return bad("truncated branch table");
        memcpy(&b, rec, sizeof b);
    }
    for (auto &L : m.layers) {
        unsigned char rec[128];
        if (fread(rec, 1, 128, f) != 128) return bad("truncated layer table");
        memcpy(&L, rec, sizeof L);
    }
    m.blob.resize(m.h.blob_floats);
    if (fseek(f, (long)m.h.blob_offset, SEEK_SET) != 0) return bad("bad blob offset");
    if (fread(m.blob.data(), sizeof(float), m.h.blob_floats, f) != m.h.blob_floats) return bad("truncated weights");
    fclose(f);
    m.tensor_floats.assign(m.h.n_layers + 1, 0);
    m.tensor_floats[0] = (uint64_t)m.h.n_branches * m.h.spec_h * m.h.spec_w;
    for (uint32_t i = 0; i < m.h.n_layers; i++) {
        const auto &L = m.layers[i];
        m.tensor_floats[i + 1] = (uint64_t)L.out_h * L.out_w * L.cout;
        if (L.in_tensor > i || (L.res_tensor != NO_TENSOR && L.res_tensor > i)) { err = "layer reads a later tensor"; return false; }
        const uint64_t wn = L.op == OP_CONV ? (uint64_t)L.kh * L.kw * L.cin * L.cout
                          : L.op == OP_DWCONV ? (uint64_t)L.kh * L.kw * L.cout
                          : (L.op == OP_PWCONV || L.op == OP_DENSE) ? (uint64_t)L.cin * L.cout : 0;
        if (L.op == OP_SCALE && (L.res_tensor == NO_TENSOR || m.tensor_floats[L.res_tensor] != L.cout || L.cin != L.cout)) { err = "scale layer without a [C] gate"; return false; }
        if (L.op != OP_GAP && L.op != OP_SCALE && (L.w_off + wn > m.h.blob_floats || L.b_off + L.cout > m.h.blob_floats)) {
            err = "layer weights outside blob"; return false;
        }
    }
    if (m.h.embedding_tensor > m.h.n_layers) { err = "bad embedding tensor"; return false; }
    return true;
}

// BHC1: a custom classifier on embeddings (birda_amd/modelfile.py write_custom_classifier): dense layers + output activation
#pragma pack(push, 1)
struct CustomHeaderRec {
    char magic[4];
    uint32_t version, input_dim, n_layers, n_classes, output_activation;
    uint64_t blob_offset, blob_floats;
};
struct CustomLayerRec {
    uint32_t in_dim, out_dim, act, reserved;
    uint64_t w_off, b_off;   // W [in][out] row-major, b [out]
};
#pragma pack(pop)

struct CustomModel {
    CustomHeaderRec h{};
    std::vector<CustomLayerRec> layers;
    std::vector<float> blob;
};

inline bool load_custom_model(const char *path, CustomModel &m, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open custom classifier file ") + path; return false; }
    auto bad = [&](const char *why) { err = std::string(path) + ": " + why; fclose(f); return false; };
    unsigned char hdr[64];
    if (fread(hdr, 1, 64, f) != 64) return bad("truncated header");
    memcpy(&m.h, hdr, sizeof m.h);
    if (memcmp(m.h.magic, "BHC1", 4) != 0 || m.h.version != 1) return bad("not a BHC1 v1 custom classifier");
    if (m.h.n_layers == 0 || m.h.n_layers > 64 || m.h.input_dim == 0 || m.h.n_classes == 0) return bad("bad counts");
    m.layers.resize(m.h.n_layers);
    for (auto &L : m.layers) {
        unsigned char rec[32];
        if (fread(rec, 1, 32, f) != 32) return bad("truncated layer table");
        memcpy(&L, rec, sizeof L);
    }
    if (m.h.blob_floats > (1ull << 32)) return bad("weights too large");
    m.blob.resize(m.h.blob_floats);
    if (fseek(f, (long)m.h.blob_offset, SEEK_SET) != 0) return bad("bad blob offset");
    if (fread(m.blob.data(), sizeof(float), m.h.blob_floats, f) != m.h.blob_floats) return bad("truncated weights");
    fclose(f);
    uint32_t dim = m.h.input_dim;
    for (const auto &L : m.layers) {
        if (L.in_dim != dim || L.out_dim == 0) { err = "custom classifier: layer widths do not chain"; return false; }
        if (L.w_off + (uint64_t)L.in_dim * L.out_dim > m.h.blob_floats || L.b_off + L.out_dim > m.h.blob_floats) { err = "custom classifier: weights outside blob"; return false; }
        dim = L.out_dim;
    }
    if (dim != m.h.n_classes) { err = "custom classifier: last layer width != n_classes"; return false; }
    return true;
}

}  // namespace bh
