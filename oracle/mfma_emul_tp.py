"""Lane-level numpy emulation of the hidden-dimension tensor-parallel learner kernels
(codebase_amd/csrc/dqn_update_tp.hip) used for hidden = 128.  TEST INFRASTRUCTURE ONLY.

A workgroup of W waves processes 16-row blocks; wave w owns TPW hidden tiles (16 units each) of every
layer.  Weights live in registers (distributed over the waves), activations are exchanged through LDS:
  pass F (forward, critic + target): h1 tile dumps -> barrier -> h2 tiles -> split-K partial Q -> barrier -> Q
  mixer (IDQN per agent / VDN sum)  : dq = dL/dchosen, per-row loss
  pass B (critic forward again + backward with dq): gradient slices accumulate in the owning wave
Checked here against torch autograd (tests/test_oracle_learner.py) before anything runs on a GPU."""
import numpy as np

from oracle.mfma_emul import G, J, LANE, Shape, argmax_rows, gather_rows, mfma, tile_read, tile_write


class TP:
    def __init__(self, S, W, TPW):
        assert S.H == 16 * W * TPW
        self.S, self.W, self.TPW, self.NT = S, W, TPW, W * TPW

    def weights(self, w_flat, wave):
        """register images of wave `wave` (lists indexed by owned tile u)"""
        S = self.S
        W1 = w_flat[S.oW1:S.ob1].reshape(S.H, S.D)
        b1 = w_flat[S.ob1:S.ob1 + S.H]
        W2 = w_flat[S.oW2:S.ob2].reshape(S.H, S.H)
        b2 = w_flat[S.ob2:S.ob2 + S.H]
        W3 = w_flat[S.oW3:S.ob3].reshape(S.A, S.H)
        b3 = w_flat[S.ob3:S.ob3 + S.A]
        out = []
        for u in range(self.TPW):
            tau = wave * self.TPW + u
            a1 = []
            for ks in range(S.KS1):
                k = 4 * ks + G
                v = np.zeros(64, np.float32)
                ok = k < S.D
                v[ok] = W1[16 * tau + J[ok], k[ok]]
                a1.append(v)
            a2 = [[W2[16 * tau + J, 16 * kap + 4 * G + r] for r in range(4)] for kap in range(self.NT)]
            a3, t3 = [], []
            for r in range(4):
                v = np.zeros(64, np.float32)
                ok = J < S.A
                v[ok] = W3[J[ok], 16 * tau + 4 * G[ok] + r]
                a3.append(v)
                a = 4 * G + r
                v = np.zeros(64, np.float32)
                ok = a < S.A
                v[ok] = W3[a[ok], 16 * tau + J[ok]]
                t3.append(v)
            t2 = [[W2[16 * kap + 4 * G + r, 16 * tau + J] for r in range(4)] for kap in range(self.NT)]
            b1s = np.stack([b1[16 * tau + 4 * G + r] for r in range(4)], 1)
            b2s = np.stack([b2[16 * tau + 4 * G + r] for r in range(4)], 1)
            out.append(dict(a1=a1, a2=a2, a3=a3, t3=t3, t2=t2, b1s=b1s, b2s=b2s))
        b3v = np.zeros((64, 4), np.float32)
        for r in range(4):
            a = 4 * G + r
            ok = a < S.A
            b3v[ok, r] = b3[a[ok]]
        return out, b3v

    def forward_block(self, regs, b3v, x, need_q=True):
        """all W waves on one row block: returns per-wave h1[u], h2[u] and (if need_q) the full q (C layout)"""
        S = self.S
        h1 = [[None] * self.TPW for _ in range(self.W)]
        dump = [None] * self.NT
        for w in range(self.W):
            for u in range(self.TPW):
                acc = regs[w][u]["b1s"].copy()
                for ks in range(S.KS1):
                    acc = mfma(regs[w][u]["a1"][ks], x[ks], acc)
                h1[w][u] = np.maximum(acc, 0)
                dump[w * self.TPW + u] = h1[w][u]  # LDS Hc[tau][lane] (C-layout dump)
        # ---- barrier
        h2 = [[None] * self.TPW for _ in range(self.W)]
        qp = []
        for w in range(self.W):
            for u in range(self.TPW):
                acc = regs[w][u]["b2s"].copy()
                for kap in range(self.NT):
                    hk = dump[kap]
                    for r in range(4):
                        acc = mfma(regs[w][u]["a2"][kap][r], hk[:, r], acc)
                h2[w][u] = np.maximum(acc, 0)
            if need_q:
                q = b3v.copy() if w == 0 else np.zeros((64, 4), np.float32)
                for u in range(self.TPW):
                    for r in range(4):
                        q = mfma(regs[w][u]["a3"][r], h2[w][u][:, r], q)
                qp.append(q)
        q = None
        if need_q:  # ---- barrier, then every wave sums the W partials in wave order
            q = np.zeros((64, 4), np.float32)
            for w in range(self.W):
                q = (q + qp[w]).astype(np.float32)
        return h1, h2, q


def load_x(S, obss_p, t, b0, B):
    x = []
    for ks in range(S.KS1):
        d = 4 * ks + G
        ok = (d < S.D) & (b0 + J < B)
        v = np.zeros(64, np.float32)
        v[ok] = obss_p[t, (b0 + J)[ok], d[ok]]
        x.append(v)
    return x


def pass_f(tp, params_p, tparams_p, obss_p, actions_p, b0, double_q):
    """-> chosen[T][16], tqsel[T][16] for the 16 episodes b0.. (rows beyond B undefined)"""
    S = tp.S
    T1, B = obss_p.shape[0], obss_p.shape[1]
    T = T1 - 1
    cw = [tp.weights(params_p, w) for w in range(tp.W)]
    tw = [tp.weights(tparams_p, w) for w in range(tp.W)]
    cregs, cb3 = [c[0] for c in cw], cw[0][1]
    tregs, tb3 = [c[0] for c in tw], tw[0][1]
    chosen = np.zeros((T, 16), np.float32)
    tqsel = np.zeros((T, 16), np.float32)
    bj = np.minimum(b0 + J, B - 1)
    for t in range(T, -1, -1):
        x = load_x(S, obss_p, t, b0, B)
        _, _, q = tp.forward_block(cregs, cb3, x)
        if t < T:
            chosen[t] = gather_rows(q, actions_p[t, bj].astype(np.int64))[:16]
        if t > 0:
            _, _, tq = tp.forward_block(tregs, tb3, x)
            a_p = argmax_rows(S, q) if double_q else argmax_rows(S, tq)
            tqsel[t - 1] = gather_rows(tq, a_p)[:16]
    return chosen, tqsel


def pass_b(tp, params_p, obss_p, actions_p, dq, b0, grads):
    """backward of the 16 episodes b0.. with external dq[T][16]; accumulates into grads (canonical flat block)"""
    S = tp.S
    T1, B = obss_p.shape[0], obss_p.shape[1]
    T = T1 - 1
    cw = [tp.weights(params_p, w) for w in range(tp.W)]
    regs = [c[0] for c in cw]
    bj = np.minimum(b0 + J, B - 1)
    rowok = (b0 + J) < B
    z = lambda: np.zeros((64, 4), np.float32)
    NT1 = S.DP // 16
    acc = [[dict(dW2=[z() for _ in range(tp.NT)], dW1=[z() for _ in range(NT1)], dW3=z(), db1=z(), db2=z())
            for _ in range(tp.TPW)] for _ in range(tp.W)]
    db3 = z()
    for t in range(T - 1, -1, -1):
        x = load_x(S, obss_p, t, b0, B)
        h1, h2, _ = tp.forward_block(regs, None, x, need_q=False)
        a_sel = actions_p[t, bj].astype(np.int64)
        dqs = np.where(rowok, dq[t][J], 0.0).astype(np.float32)
        dQ = np.stack([np.where(4 * G + r == a_sel, dqs, 0.0) for r in range(4)], 1).astype(np.float32)
        db3 += dQ
        hcT = tile_write([h1[w][u] for w in range(tp.W) for u in range(tp.TPW)])  # shared [H][16] (written in F1)
        pq = tile_write([dQ])
        g2dump = [None] * tp.NT
        dh2 = [[None] * tp.TPW for _ in range(tp.W)]
        for w in range(tp.W):
            for u in range(tp.TPW):
                a_ = z()
                for r in range(4):
                    a_ = mfma(regs[w][u]["t3"][r], dQ[:, r], a_)
                dh2[w][u] = np.where(h2[w][u] > 0, a_, 0).astype(np.float32)
                acc[w][u]["db2"] += dh2[w][u]
                g2dump[w * tp.TPW + u] = dh2[w][u]
                ph2 = tile_write([h2[w][u]])  # wave-private transposes
                aop = tile_read(pq, 0)
                bop = tile_read(ph2, 0)
                for ks in range(4):
                    acc[w][u]["dW3"] = mfma(aop[ks], bop[ks], acc[w][u]["dW3"])
        # ---- barrier
        for w in range(tp.W):
            for u in range(tp.TPW):
                a_ = z()
                for kap in range(tp.NT):
                    for r in range(4):
                        a_ = mfma(regs[w][u]["t2"][kap][r], g2dump[kap][:, r], a_)
                dh1 = np.where(h1[w][u] > 0, a_, 0).astype(np.float32)
                acc[w][u]["db1"] += dh1
                p2 = tile_write([dh2[w][u]])
                ag2 = tile_read(p2, 0)
                for nu in range(tp.NT):
                    bh1 = tile_read(hcT, nu)
                    for ks in range(4):
                        acc[w][u]["dW2"][nu] = mfma(ag2[ks], bh1[ks], acc[w][u]["dW2"][nu])
                p1 = tile_write([dh1])
                ag1 = tile_read(p1, 0)
                for nt in range(NT1):
                    for ks in range(4):
                        row = b0 + 4 * G + ks
                        d = 16 * nt + J
                        ok = (row < B) & (d < S.D)
                        bx = np.zeros(64, np.float32)
                        bx[ok] = obss_p[t, row[ok], d[ok]]
                        acc[w][u]["dW1"][nt] = mfma(ag1[ks], bx, acc[w][u]["dW1"][nt])
    # write-out: every wave owns disjoint slices of the gradient
    for w in range(tp.W):
        for u in range(tp.TPW):
            tau = w * tp.TPW + u
            for r in range(4):
                o = 16 * tau + 4 * G + r
                for nt in range(NT1):
                    d = 16 * nt + J
                    ok = d < S.D
                    grads[S.oW1 + o[ok] * S.D + d[ok]] += acc[w][u]["dW1"][nt][ok, r]
                for nu in range(tp.NT):
                    grads[S.oW2 + o * S.H + 16 * nu + J] += acc[w][u]["dW2"][nu][:, r]
                a = 4 * G + r
                ok = a < S.A
                grads[S.oW3 + a[ok] * S.H + 16 * tau + J[ok]] += acc[w][u]["dW3"][ok, r]
                for gg in range(4):
                    grads[S.ob1 + 16 * tau + 4 * gg + r] += acc[w][u]["db1"][G == gg, r].sum()
                    grads[S.ob2 + 16 * tau + 4 * gg + r] += acc[w][u]["db2"][G == gg, r].sum()
    for r in range(4):
        for gg in range(4):
            a = 4 * gg + r
            if a < S.A:
                grads[S.ob3 + a] += db3[G == gg, r].sum()


def idqn_loss_grad_tp(S, W, TPW, params, tparams, obss, actions, rewards, dones, filled, gamma, double_q, mode="idqn"):
    tp = TP(S, W, TPW)
    P, T1, B, _ = obss.shape
    T = T1 - 1
    chosen = np.zeros((P, T, B), np.float32)
    tqsel = np.zeros((P, T, B), np.float32)
    for p in range(P):
        for b0 in range(0, B, 16):
            c, q = pass_f(tp, params[p], tparams[p], obss[p], actions[p], b0, double_q)
            n = min(16, B - b0)
            chosen[p, :, b0:b0 + n] = c[:, :n]
            tqsel[p, :, b0:b0 + n] = q[:, :n]
    d1 = dones[1:]
    if mode == "idqn":
        delta = chosen - (rewards + gamma * tqsel * (1 - d1)[None])
        dq = 2 * filled[None] * delta
        loss = float((filled[None] * delta * delta).sum())
    else:
        delta = chosen.sum(0) - (rewards[0] + gamma * tqsel.sum(0) * (1 - d1))
        dq = np.repeat((2 * filled * delta)[None], P, 0)
        loss = float((filled * delta * delta).sum())
    grads = np.zeros((P, S.NPARAM), np.float32)
    for p in range(P):
        for b0 in range(0, B, 16):
            n = min(16, B - b0)
            dqb = np.zeros((T, 16), np.float32)
            dqb[:, :n] = dq[p][:, b0:b0 + n]
            pass_b(tp, params[p], obss[p], actions[p], dqb, b0, grads[p])
    nf = float(filled.sum())
    return loss / nf, grads / nf
