// BHM1 model container reader (format: birda_amd/modelfile.py).  Host only.
#pragma once
#include <cmath>
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

// The checks every model goes through, whichever file it came from (a BHM1 container, or an ONNX graph read by onnx_conv.hpp):
// every dimension is bounded (so that no size product below can wrap), every layer must read exactly the tensor its producer
// wrote, and every weight must lie inside the blob.  Fills tensor_floats.
inline bool validate_model(Model &m, std::string &err) {
    if (m.h.n_branches == 0 || m.h.n_branches > 16 || m.h.n_layers == 0 || m.h.n_layers > 4096 || m.branches.size() != m.h.n_branches ||
        m.layers.size() != m.h.n_layers || m.blob.size() != m.h.blob_floats) { err = "bad counts"; return false; }
    if (m.h.sample_count == 0 || m.h.sample_count > (1u << 24) || m.h.sample_rate == 0 || m.h.n_classes == 0 || m.h.n_classes > (1u << 24) ||
        m.h.spec_h == 0 || m.h.spec_w == 0 || m.h.spec_h > (1u << 14) || m.h.spec_w > (1u << 16)) { err = "bad model dimensions"; return false; }
    // the front-end's branches (round 6, tools/fuzz_create.py: a mutated mel_w_off -- 4.5e18 -- passed every check here and the operator
    // build read the blob there: SIGBUS.  None of these fields was checked against the blob, the segment or the spectrogram before.)
    for (const auto &b : m.branches) {
        if (b.frame_length < 4 || b.frame_length > (1u << 16) || (b.frame_length & 1) || b.fft_length != b.frame_length || b.n_bins != b.frame_length / 2 + 1 ||
            b.frame_step == 0 || b.frame_step > (1u << 16) || b.n_mels == 0 || b.n_mels != m.h.spec_h || b.n_frames == 0 || b.n_frames != m.h.spec_w) {
            err = "front-end branch geometry out of range (frame length / step / bins / mels / frames against the spectrogram)"; return false;
        }
        if ((uint64_t)(b.n_frames - 1) * b.frame_step + b.frame_length > m.h.sample_count) { err = "front-end frames run past the segment"; return false; }
        if (b.mel_w_off > m.h.blob_floats || (uint64_t)b.n_bins * b.n_mels > m.h.blob_floats - b.mel_w_off) { err = "mel matrix outside blob"; return false; }
        if (!std::isfinite(b.mag_scale) || !std::isfinite(b.out_scale) || !std::isfinite(b.out_shift)) { err = "front-end constants are not finite"; return false; }
    }
    if (!std::isfinite(m.h.norm_eps) || m.h.norm_eps < 0) { err = "normalisation epsilon out of range"; return false; }
    m.tensor_floats.assign(m.h.n_layers + 1, 0);
    m.tensor_floats[0] = (uint64_t)m.h.n_branches * m.h.spec_h * m.h.spec_w;
    for (uint32_t i = 0; i < m.h.n_layers; i++) {
        const auto &L = m.layers[i];
        // (a global pool's "kernel" is its whole input image -- the squeeze of a squeeze-excite gate in an early block pools 64 x 249)
        const uint32_t kmax = L.op == OP_GAP ? (1u << 16) : 64;
        if (L.cin == 0 || L.cout == 0 || L.cin > (1u << 16) || L.cout > (1u << 24) || L.kh == 0 || L.kw == 0 || L.kh > kmax || L.kw > kmax ||
            L.sh == 0 || L.sw == 0 || L.sh > 16 || L.sw > 16 || L.pad_t > 64 || L.pad_l > 64 || L.in_h == 0 || L.in_w == 0 || L.out_h == 0 ||
            L.out_w == 0 || L.in_h > (1u << 16) || L.in_w > (1u << 16) || L.out_h > (1u << 16) || L.out_w > (1u << 16) ||
            (uint64_t)L.out_h * L.out_w * L.cout > (1ull << 31) || (uint64_t)L.in_h * L.in_w * L.cin > (1ull << 31)) {
            err = "layer dimensions out of range"; return false;
        }
        m.tensor_floats[i + 1] = (uint64_t)L.out_h * L.out_w * L.cout;
        if (L.in_tensor > i || (L.res_tensor != NO_TENSOR && L.res_tensor > i)) { err = "layer reads a later tensor"; return false; }
        // what the layer reads must be what its input tensor holds (a pool reads in_h x in_w x cout, a dense layer cin values)
        const uint64_t in_floats = L.op == OP_GAP || L.op == OP_DWCONV || L.op == OP_SCALE ? (uint64_t)L.in_h * L.in_w * L.cout
                                 : L.op == OP_DENSE ? (uint64_t)L.cin : (uint64_t)L.in_h * L.in_w * L.cin;
        if (in_floats != m.tensor_floats[L.in_tensor]) { err = "layer input shape does not match its tensor"; return false; }
        if (L.res_tensor != NO_TENSOR && L.op != OP_SCALE && m.tensor_floats[L.res_tensor] != m.tensor_floats[i + 1]) {
            err = "residual shape does not match the layer output"; return false;
        }
        if (L.op < OP_CONV || L.op > OP_SCALE) { err = "unknown layer op"; return false; }
        const uint64_t wn = L.op == OP_CONV ? (uint64_t)L.kh * L.kw * L.cin * L.cout
                          : L.op == OP_DWCONV ? (uint64_t)L.kh * L.kw * L.cout
                          : (L.op == OP_PWCONV || L.op == OP_DENSE) ? (uint64_t)L.cin * L.cout : 0;
        if (L.op == OP_SCALE && (L.res_tensor == NO_TENSOR || m.tensor_floats[L.res_tensor] != L.cout || L.cin != L.cout)) { err = "scale layer without a [C] gate"; return false; }
        if (L.op != OP_GAP && L.op != OP_SCALE && (L.w_off > m.h.blob_floats || L.b_off > m.h.blob_floats || L.w_off + wn > m.h.blob_floats ||
                                                   L.b_off + L.cout > m.h.blob_floats)) {
            err = "layer weights outside blob"; return false;
        }
    }
    if (m.h.embedding_tensor > m.h.n_layers) { err = "bad embedding tensor"; return false; }
    return true;
}

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
    // the file is untrusted input: sizes are checked against the file before anything is allocated from them, every dimension is
    // bounded (so that no product below can wrap), and every layer must read exactly the tensor its producer wrote
    if (fseek(f, 0, SEEK_END) != 0) return bad("cannot seek");
    const long fsize = ftell(f);
    if (fsize < 0 || m.h.blob_floats > (1ull << 31) || m.h.blob_offset > (uint64_t)fsize ||
        m.h.blob_floats * sizeof(float) > (uint64_t)fsize - m.h.blob_offset)
        return bad("weights outside the file");
    if (m.h.sample_count == 0 || m.h.sample_count > (1u << 24) || m.h.sample_rate == 0 || m.h.n_classes == 0 || m.h.n_classes > (1u << 24) ||
        m.h.spec_h == 0 || m.h.spec_w == 0 || m.h.spec_h > (1u << 14) || m.h.spec_w > (1u << 16))
        return bad("bad model dimensions");
    m.blob.resize(m.h.blob_floats);
    if (fseek(f, (long)m.h.blob_offset, SEEK_SET) != 0) return bad("bad blob offset");
    if (fread(m.blob.data(), sizeof(float), m.h.blob_floats, f) != m.h.blob_floats) return bad("truncated weights");
    fclose(f);
    return validate_model(m, err);
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
    if (fseek(f, 0, SEEK_END) != 0) return bad("cannot seek");
    const long fsize = ftell(f);
    if (fsize < 0 || m.h.blob_floats > (1ull << 31) || m.h.blob_offset > (uint64_t)fsize ||
        m.h.blob_floats * sizeof(float) > (uint64_t)fsize - m.h.blob_offset)
        return bad("weights outside the file");
    if (m.h.input_dim > (1u << 20) || m.h.n_classes > (1u << 24)) return bad("bad dimensions");
    m.blob.resize(m.h.blob_floats);
    if (fseek(f, (long)m.h.blob_offset, SEEK_SET) != 0) return bad("bad blob offset");
    if (fread(m.blob.data(), sizeof(float), m.h.blob_floats, f) != m.h.blob_floats) return bad("truncated weights");
    fclose(f);
    uint32_t dim = m.h.input_dim;
    for (const auto &L : m.layers) {
        if (L.in_dim != dim || L.out_dim == 0 || L.out_dim > (1u << 24)) { err = "custom classifier: layer widths do not chain"; return false; }
        if (L.w_off > m.h.blob_floats || L.b_off > m.h.blob_floats || L.w_off + (uint64_t)L.in_dim * L.out_dim > m.h.blob_floats || L.b_off + L.out_dim > m.h.blob_floats) { err = "custom classifier: weights outside blob"; return false; }
        dim = L.out_dim;
    }
    if (dim != m.h.n_classes) { err = "custom classifier: last layer width != n_classes"; return false; }
    return true;
}

}  // namespace bh
