// Internal launcher interface between the host executor (api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

enum Act : int { ACT_NONE = 0, ACT_RELU, ACT_RELU6, ACT_SWISH, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SIGMOID };

constexpr int MAX_BRANCHES = 4;

// One STFT/mel branch of the front-end, with the Hann window, the
// real-part DFT and the mel projection folded into one operator Gf[K = L/2][n_mels_pad].
struct BranchParams {
    const float *gf;  // device, MFMA-fragment-major gfF[K/16][nm_pad/16][64 lanes][4]: element
                      // (g, mt, lane, c) = Gf[k = 16g + 4(lane>>4) + c][mel = 16mt + (lane&15)],
                      // Gf row k <-> sample offset n = k + 1 (last row halved)
    int L, H, K;      // frame length, hop, folded depth L/2
    int n_mels, nm_pad, n_frames;
    float expo;       // 1 / (1 + exp(mag_scale))
    float out_scale, out_shift;
    int flip;
};
struct FrontendParams {
    BranchParams br[MAX_BRANCHES];
    int n_branches;
    int sample_count;
    float norm_eps;
};

void launch_minmax(const float *x, float *minmax, int n_seg, int sample_count, hipStream_t s);
void launch_mel(const float *x, const float *minmax, float *spec, const FrontendParams &p,
                const FrontendParams *d_p, int n_seg, hipStream_t s);

struct ConvParams {
    int in_h, in_w, out_h, out_w, cin, cout, kh, kw, sh, sw, pad_t, pad_l, in_layout, act;
};
// direct conv for the small-Cin stem; w [kh][kw][cin][cout]
void launch_conv_direct(const float *in, const float *w, const float *b, float *out, const ConvParams &p,
                        int n_seg, hipStream_t s);
// depthwise conv NHWC; w [kh][kw][c]
void launch_dwconv(const float *in, const float *w, const float *b, float *out, const ConvParams &p,
                   int n_seg, hipStream_t s);
// C[M][N] = act(A[M][K] . W[K][ldw] + bias) (+ R); W rows padded to ldw (multiple of 4)
void launch_pw_gemm(const float *A, const float *W, const float *bias, const float *R, float *C, int M, int K,
                    int N, int ldw, int act, hipStream_t s);
// global average pool [n][P][C] -> [n][C]
void launch_gap(const float *in, float *out, int n_seg, int P, int C, hipStream_t s);
// activation + top-k over logits [n][n_classes] -> idx/conf [n][top_k]
void launch_topk(const float *logits, int n_seg, int n_classes, int out_act, int top_k, float min_conf,
                 int32_t *idx, float *conf, hipStream_t s);

}  // namespace bh
