// Stacked recurrent layers (`layers = [h] * (L + 1)` with use_rnn: RNNNetwork builds nn.GRU(num_layers = L), marlbase/utils/models.py:74-90).
// A stack runs the one-layer kernels of gru.h / gru_bwd.h L times: layer l is a one-layer network whose parameter block starts
// l * GruShape::LAYER floats into the agent's (GruShape, gru.h), its input sequence is the hidden-state sequence of layer l - 1 - read from
// that layer's activation record, which is already in the B-operand layout the gate products want - and on the way back its dL/dh[t] is
// the dx1[t] the layer above left in ITS backward record.  L = 1 issues exactly the launches the one-layer path always did.
//   forward : l = 0 .. L-1   gru_seq_fwd(2)_kernel(x_in = rec[l-1], q_out only for l = L-1)
//   backward: l = L-1 .. 0   gru_seq_bwd(2)_kernel(dh_above = rec2[l+1], no ReLU mask for l >= 1)
//   weights : every l        gru_wgrad_kernel into the block's partial record at l * LAYER (dW1 from l = 0, dW3 / loss from l = L-1)
// Records are layer-major: rec [L][P][steps][nblk][REC], rec2 [L][P][steps][nblk][REC2]; packs [L][P][NFWD] / [L][P][NBWD];
// hidden states handed across calls (acting) [L][P][B][H] = nn.GRU's (num_layers, batch, hidden) per agent.
#pragma once
#include "gru_bwd.h"

namespace marl {

constexpr int GRU_MAX_LAYERS = 4;

// marlhip_net_shape.n_hidden = len(layers) (0 = 2): the number of GRU layers is one less (utils/models.py:77)
inline int gru_depth(const marlhip_net_shape* s) { return (s->n_hidden > 0 ? s->n_hidden : 2) - 1; }

template <class S>
int64_t gru_layer_rec(int P, int steps, int B) { return (int64_t)P * steps * ((B + 15) / 16) * S::REC; }
template <class S>
int64_t gru_layer_rec2(int P, int steps, int B) { return (int64_t)P * steps * ((B + 15) / 16) * GruBwd<S>::REC2; }

template <class S>
void gru_pack_fwd_layers(int P, int L, const float* params, const AgentMap& am, float* packs, hipStream_t st) {
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL((gru_pack_kernel<S>), dim3((S::NFWD + 255) / 256, P), dim3(256), 0, st, params, am, packs + (size_t)l * P * S::NFWD, S::nparam(L),
                           l * S::LAYER);
}

template <class S>
void gru_pack_bwd_layers(int P, int L, const float* params, const AgentMap& am, float* packs, hipStream_t st) {
    using Bk = GruBwd<S>;
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL((gru_bwd_pack_kernel<S>), dim3((Bk::NBWD + 255) / 256, P), dim3(256), 0, st, params, am, packs + (size_t)l * P * Bk::NBWD,
                           S::nparam(L), l * S::LAYER);
}

// one family's sequence forward.  rec: L records when top_rec (the caller wants the top layer's too: BPTT), else L - 1 (the chain only;
// may be NULL for L == 1); h_in / h_out: [L][P][B][H] or NULL
template <class S>
void gru_fwd_layers(int P, int L, const float* packs, const float* obs, size_t as, size_t rs, int steps, int B, const float* h_in, float* h_out,
                    float* out, float* rec, bool top_rec, hipStream_t st) {
    const int64_t recl = gru_layer_rec<S>(P, steps, B), hl = (int64_t)P * B * S::H;
    for (int l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        hipLaunchKernelGGL((gru_seq_fwd_kernel<S>), dim3((B + 63) / 64, P), dim3(256), S::LDS_FLOATS * sizeof(float), st, packs + (size_t)l * P * S::NFWD, obs, as, rs,
                           steps, B, h_in != nullptr ? h_in + l * hl : (const float*)nullptr, h_out != nullptr ? h_out + l * hl : (float*)nullptr,
                           last ? out : (float*)nullptr, (rec != nullptr && (!last || top_rec)) ? rec + l * recl : (float*)nullptr,
                           l > 0 ? (const float*)(rec + (l - 1) * recl) : (const float*)nullptr);
    }
}

// two networks of one shape over the same observations, one launch per layer (gru_seq_fwd2_kernel): the first keeps its L records (BPTT),
// the second the L - 1 its own chain needs (rec_second: [L-1] records at steps2; unused for L == 1)
template <class S>
void gru_fwd2_layers(int P, int L, const float* packs, const float* packs2, const float* obs, size_t as, size_t rs, int steps, int steps2, int B,
                     float* out, float* out2, float* rec, float* rec_second, hipStream_t st) {
    const int64_t recl = gru_layer_rec<S>(P, steps, B), recl2 = gru_layer_rec<S>(P, steps2, B);
    for (int l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        hipLaunchKernelGGL((gru_seq_fwd2_kernel<S>), dim3((B + 63) / 64, P, 2), dim3(256), S::LDS_FLOATS * sizeof(float), st, packs + (size_t)l * P * S::NFWD,
                           packs2 + (size_t)l * P * S::NFWD, obs, as, rs, steps, steps2, B, last ? out : (float*)nullptr, last ? out2 : (float*)nullptr,
                           rec != nullptr ? rec + l * recl : (float*)nullptr, last ? (float*)nullptr : rec_second + l * recl2,
                           l > 0 ? (const float*)(rec + (l - 1) * recl) : (const float*)nullptr,
                           l > 0 ? (const float*)(rec_second + (l - 1) * recl2) : (const float*)nullptr);
    }
}

template <class S>
void gru_bwd_layers(int P, int L, const float* packB, int steps, int B, const float* rec, const float* dq, float* rec2, hipStream_t st, bool alone) {
    const int64_t recl = gru_layer_rec<S>(P, steps, B), rec2l = gru_layer_rec2<S>(P, steps, B);
    for (int l = L - 1; l >= 0; --l)
        gru_launch_seq_bwd<S>(P, B, packB + (size_t)l * P * GruBwd<S>::NBWD, steps, rec + l * recl, dq, rec2 + l * rec2l, st, alone,
                              l < L - 1 ? (const float*)(rec2 + (l + 1) * rec2l) : (const float*)nullptr, l > 0 ? 1 : 0);
}

// partials: [P][nwg][nparam(L) + 2]
template <class S>
void gru_wgrad_layers(int P, int L, int nwg, int steps, int B, const float* obs, size_t as, size_t rs, const float* rec, const float* rec2, const float* dq,
                      const float* lrow, const float* filled, int loss_steps, float* partials, hipStream_t st) {
    const int64_t recl = gru_layer_rec<S>(P, steps, B), rec2l = gru_layer_rec2<S>(P, steps, B);
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL((gru_wgrad_kernel<S>), dim3(nwg, P, gru_wgrad_roles<S>()), dim3(256), gru_wgrad_lds_bytes<S>(), st, steps, B, obs, as, rs, rec + l * recl,
                           rec2 + l * rec2l, dq, lrow, filled, loss_steps, partials, S::nparam(L) + 2, l * S::LAYER, (l == 0 ? 1 : 0) | (l == L - 1 ? 2 : 0));
}

}  // namespace marl
