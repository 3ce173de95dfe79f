// QMixer of ANY configured shape (marlbase/dqn/model.py:272-331: `hypernet_layers` 1 or 2, any `embed_dim` / `hypernet_embed`, any
// (agents, observation) pair) - the mixer stage between the agent networks' forward and backward passes for everything csrc/qmix.h's
// fused kernels are not compiled for (those carry configs/algorithm/qmix.yaml's {embed_dim 64, hypernet_layers 2, hypernet_embed 32},
// narrower two-layer mixers zero-padded into them).  Slow path by construction, chosen by shape only: the state rows are materialised
// once, every hypernet layer is an f32 MFMA GEMM over all rows (wide_mlp.h) with its activations in HBM, the mixing network itself
// (|.|, elu, the agent-weighted sum, the TD error and its backward) is one wave per row, weight gradients are split-K GEMMs folded in
// fixed order.  Same interface as qmix_launch_mix / qmix_launch_reduce (QmixCtx, QmixIo).  Implemented in qmix_gen.hip.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

struct marlhip_batch;

namespace marl {

struct QmixCtx;
struct QmixIo;
struct ReplaySrc;

struct QmixGenDims {
    int P, D, E, HE, L;  // agents, observation width, embed_dim, hypernet_embed, hypernet_layers (1 | 2)
};

// 0 when the dimensions are ones the generic stage runs (sets the library's error text otherwise)
int qmix_gen_check(const QmixGenDims& d);
// parameters of the QMixer in mixer.parameters() order:
//   L == 2: hyper_w_1.0 [HE][SD] + [HE] | hyper_w_1.2 [E P][HE] + [E P] | hyper_w_final.0 [HE][SD] + [HE] | hyper_w_final.2 [E][HE] + [E]
//   L == 1: hyper_w_1 [E P][SD] + [E P] | hyper_w_final [E][SD] + [E]
//   then  : hyper_b_1 [E][SD] + [E] | V.0 [E][SD] + [E] | V.2 [1][E] + [1]
int64_t qmix_gen_nparams(const QmixGenDims& d);
int64_t qmix_gen_ws_bytes(const QmixGenDims& d, int T, int B);
// mixers forward, TD error, backward down to dL/dchosen_p (io.dq) and the per-row loss (io.lrow); rs == nullptr: rows from bt->obss
int qmix_gen_mix(const QmixCtx& qx, const QmixGenDims& d, const marlhip_batch* bt, const ReplaySrc* rs, const QmixIo& io, float gamma, hipStream_t st);
// after the agent networks' reduce has left loss[1] = sum(filled): the mixer's weight gradients into qx.mgrad
int qmix_gen_reduce(const QmixCtx& qx, const QmixGenDims& d, int T, int B, const float* loss, hipStream_t st);

}  // namespace marl
