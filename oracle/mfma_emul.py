"""Lane-level numpy emulation of the fused IDQN loss/gradient kernel
(codebase_amd/csrc/dqn_update.hip).  TEST INFRASTRUCTURE ONLY.

Every "register" is an array of 64 lane values; `mfma` implements the documented
v_mfma_f32_16x16x4_f32 operand maps (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
D[row=(l>>4)*4+r][col=l&15]).  The code below is the kernel's dataflow statement by
statement (weight packs, transposed-activation chaining, reversed-time TD pipeline,
LDS transposes for the weight-gradient GEMMs, gradient write-out), so the index math
can be checked against torch autograd on a machine with no GPU.
"""
import numpy as np

LANE = np.arange(64)
G = LANE >> 4
J = LANE & 15


def mfma(a, b, c):
    A = np.zeros((16, 4), np.float64)
    B = np.zeros((4, 16), np.float64)
    A[LANE & 15, LANE >> 4] = a
    B[LANE >> 4, LANE & 15] = b
    Dm = A @ B
    out = c.astype(np.float64).copy()
    for r in range(4):
        out[:, r] += Dm[G * 4 + r, J]
    return out.astype(np.float32)


class Shape:
    def __init__(self, D, H, A):
        self.D, self.H, self.A = D, H, A
        self.DP = (D + 15) // 16 * 16
        self.KS1 = self.DP // 4
        self.MT = H // 16
        self.oW1 = 0
        self.ob1 = H * D
        self.oW2 = self.ob1 + H
        self.ob2 = self.oW2 + H * H
        self.oW3 = self.ob2 + H
        self.ob3 = self.oW3 + A * H
        self.NPARAM = self.ob3 + A


def fwd_pack(S, w):
    """A1[mt][ks4][lane][e], A2[mt2][mt1][lane][r], A3[mt1][lane][r], b1, b2, b3(16)"""
    W1 = w[S.oW1:S.ob1].reshape(S.H, S.D)
    W2 = w[S.oW2:S.ob2].reshape(S.H, S.H)
    W3 = w[S.oW3:S.ob3].reshape(S.A, S.H)
    A1 = np.zeros((S.MT, S.KS1 // 4, 64, 4), np.float32)
    for mt in range(S.MT):
        for ks4 in range(S.KS1 // 4):
            for e in range(4):
                k = 4 * (4 * ks4 + e) + G
                ok = k < S.D
                A1[mt, ks4, ok, e] = W1[16 * mt + J[ok], k[ok]]
    A2 = np.zeros((S.MT, S.MT, 64, 4), np.float32)
    for mt2 in range(S.MT):
        for mt1 in range(S.MT):
            for r in range(4):
                A2[mt2, mt1, :, r] = W2[16 * mt2 + J, 16 * mt1 + 4 * G + r]
    A3 = np.zeros((S.MT, 64, 4), np.float32)
    for mt1 in range(S.MT):
        for r in range(4):
            ok = J < S.A
            A3[mt1, ok, r] = W3[J[ok], 16 * mt1 + 4 * G[ok] + r]
    b3 = np.zeros(16, np.float32)
    b3[:S.A] = w[S.ob3:S.ob3 + S.A]
    return A1, A2, A3, w[S.ob1:S.ob1 + S.H], w[S.ob2:S.ob2 + S.H], b3


def bwd_pack(S, w):
    W2 = w[S.oW2:S.ob2].reshape(S.H, S.H)
    W3 = w[S.oW3:S.ob3].reshape(S.A, S.H)
    T3 = np.zeros((S.MT, 64, 4), np.float32)
    for mt in range(S.MT):
        for r in range(4):
            a = 4 * G + r
            ok = a < S.A
            T3[mt, ok, r] = W3[a[ok], 16 * mt + J[ok]]
    T2 = np.zeros((S.MT, S.MT, 64, 4), np.float32)
    for mt1 in range(S.MT):
        for mt2 in range(S.MT):
            for r in range(4):
                T2[mt1, mt2, :, r] = W2[16 * mt2 + 4 * G + r, 16 * mt1 + J]
    return T3, T2


def forward(S, pack, x):
    A1, A2, A3, b1, b2, b3 = pack
    acc = [np.stack([b1[16 * mt + 4 * G + r] for r in range(4)], 1) for mt in range(S.MT)]
    for ks4 in range(S.KS1 // 4):
        for e in range(4):
            for mt in range(S.MT):
                acc[mt] = mfma(A1[mt, ks4, :, e], x[4 * ks4 + e], acc[mt])
    h1 = [np.maximum(a, 0) for a in acc]
    acc = [np.stack([b2[16 * mt + 4 * G + r] for r in range(4)], 1) for mt in range(S.MT)]
    for k1 in range(S.MT):
        for r in range(4):
            for mt in range(S.MT):
                acc[mt] = mfma(A2[mt, k1, :, r], h1[k1][:, r], acc[mt])
    h2 = [np.maximum(a, 0) for a in acc]
    o = np.stack([b3[4 * G + r] for r in range(4)], 1)
    for k1 in range(S.MT):
        for r in range(4):
            o = mfma(A3[k1, :, r], h2[k1][:, r], o)
    return h1, h2, o


def argmax_rows(S, q):
    """first index of the max over a < A, per batch row j (same value in all 4 lanes of j)"""
    Q = np.full((16, 16), -np.inf, np.float32)
    for r in range(4):
        Q[4 * G + r, J] = q[:, r]
    best = np.argmax(Q[:S.A], axis=0)
    return best[J]


def gather_rows(q, a_sel):
    Q = np.zeros((16, 16), np.float32)
    for r in range(4):
        Q[4 * G + r, J] = q[:, r]
    return Q[a_sel, J]


def tile_write(regs):
    """C layout regs[mt][lane, r] -> tile[h][row] (LDS image)"""
    MT = len(regs)
    t = np.zeros((16 * MT, 16), np.float32)
    for mt in range(MT):
        for r in range(4):
            t[16 * mt + 4 * G + r, J] = regs[mt][:, r]
    return t


def tile_read(tile, mt):
    """lane (g,i) reads tile[16mt+i][4g..4g+3] -> 4 operand registers (k-step ks <-> row 4g+ks)"""
    return [tile[16 * mt + J, 4 * G + ks] for ks in range(4)]


def loss_grad_task(S, cpack, tpack, bpack, obss, actions, rewards, dones, filled, b0, t0, t1, gamma, double_q, acc):
    """one wave task: agent-local arrays obss[T+1][B][D], actions[T][B], rewards[T][B],
    dones[T+1][B], filled[T][B]; episodes b0..b0+15; transitions t0 <= t < t1.
    `acc` = dict of running accumulators (dW1,dW2,dW3 in C layout, db*, loss)."""
    T3, T2 = bpack
    B = obss.shape[1]
    rowok = (b0 + J) < B
    bj = np.minimum(b0 + J, B - 1)

    def load_x(t):
        x = []
        for ks in range(S.KS1):
            d = 4 * ks + G
            ok = (d < S.D) & rowok
            v = np.zeros(64, np.float32)
            v[ok] = obss[t, bj[ok], d[ok]]
            x.append(v)
        return x

    tq_next = np.zeros(64, np.float32)
    for t in range(t1, t0 - 1, -1):
        x = load_x(t)
        h1, h2, q = forward(S, cpack, x)
        if t < t1:
            # TD error of transition t, then backward of row-block t
            a_sel = actions[t, bj].astype(np.int64)
            y = rewards[t, bj] + gamma * tq_next * (1.0 - dones[t + 1, bj])
            fl = np.where(rowok, filled[t, bj], 0.0).astype(np.float32)
            delta = (gather_rows(q, a_sel) - y).astype(np.float32)
            acc["loss"] += float((fl * delta * delta)[G == 0].sum())
            dqs = (2.0 * fl * delta).astype(np.float32)
            dQ = np.stack([np.where(4 * G + r == a_sel, dqs, 0.0) for r in range(4)], 1).astype(np.float32)
            # dW3[a][h2] += dQ^T x H2 ; db3
            tq_tile = tile_write([dQ])
            ta = tile_write(h2)
            aop = tile_read(tq_tile, 0)
            for nt in range(S.MT):
                bop = tile_read(ta, nt)
                for ks in range(4):
                    acc["dW3"][nt] = mfma(aop[ks], bop[ks], acc["dW3"][nt])
            acc["db3"] += dQ
            # dH2^T = W3^T dQ^T, masked by relu
            dH2 = []
            for mt in range(S.MT):
                a_ = np.zeros((64, 4), np.float32)
                for r in range(4):
                    a_ = mfma(T3[mt, :, r], dQ[:, r], a_)
                dH2.append(np.where(h2[mt] > 0, a_, 0).astype(np.float32))
                acc["db2"][mt] += dH2[mt]
            tg = tile_write(dH2)
            ta = tile_write(h1)
            for mt in range(S.MT):
                aop = tile_read(tg, mt)
                for nt in range(S.MT):
                    bop = tile_read(ta, nt)
                    for ks in range(4):
                        acc["dW2"][mt][nt] = mfma(aop[ks], bop[ks], acc["dW2"][mt][nt])
            dH1 = []
            for mt1 in range(S.MT):
                a_ = np.zeros((64, 4), np.float32)
                for mt2 in range(S.MT):
                    for r in range(4):
                        a_ = mfma(T2[mt1, mt2, :, r], dH2[mt2][:, r], a_)
                dH1.append(np.where(h1[mt1] > 0, a_, 0).astype(np.float32))
                acc["db1"][mt1] += dH1[mt1]
            tg = tile_write(dH1)
            for nt in range(S.DP // 16):
                bop = []
                for ks in range(4):
                    row = b0 + 4 * G + ks
                    d = 16 * nt + J
                    ok = (row < B) & (d < S.D)
                    v = np.zeros(64, np.float32)
                    v[ok] = obss[t, row[ok], d[ok]]
                    bop.append(v)
                for mt in range(S.MT):
                    aop = tile_read(tg, mt)
                    for ks in range(4):
                        acc["dW1"][mt][nt] = mfma(aop[ks], bop[ks], acc["dW1"][mt][nt])
        if t > t0:
            # bootstrap value for transition t-1
            _, _, tq = forward(S, tpack, x)
            a_p = argmax_rows(S, q) if double_q else argmax_rows(S, tq)
            tq_next = gather_rows(tq, a_p)


def new_acc(S):
    z = lambda: np.zeros((64, 4), np.float32)
    return dict(
        loss=0.0,
        dW1=[[z() for _ in range(S.DP // 16)] for _ in range(S.MT)],
        dW2=[[z() for _ in range(S.MT)] for _ in range(S.MT)],
        dW3=[z() for _ in range(S.MT)],
        db1=[z() for _ in range(S.MT)],
        db2=[z() for _ in range(S.MT)],
        db3=z(),
    )


def write_out(S, acc):
    """accumulators -> canonical gradient block (unscaled: sum over rows)"""
    g = np.zeros(S.NPARAM, np.float32)
    for mt in range(S.MT):
        for nt in range(S.DP // 16):
            for r in range(4):
                d = 16 * nt + J
                ok = d < S.D
                g[S.oW1 + (16 * mt + 4 * G[ok] + r) * S.D + d[ok]] = acc["dW1"][mt][nt][ok, r]
        for nt in range(S.MT):
            for r in range(4):
                g[S.oW2 + (16 * mt + 4 * G + r) * S.H + 16 * nt + J] = acc["dW2"][mt][nt][:, r]
    for nt in range(S.MT):
        for r in range(4):
            a = 4 * G + r
            ok = a < S.A
            g[S.oW3 + a[ok] * S.H + 16 * nt + J[ok]] = acc["dW3"][nt][ok, r]
    # bias rows: reduce over the 16 lanes j of each g
    for mt in range(S.MT):
        for r in range(4):
            for gg in range(4):
                g[S.ob1 + 16 * mt + 4 * gg + r] = acc["db1"][mt][G == gg, r].sum()
                g[S.ob2 + 16 * mt + 4 * gg + r] = acc["db2"][mt][G == gg, r].sum()
    for r in range(4):
        for gg in range(4):
            a = 4 * gg + r
            if a < S.A:
                g[S.ob3 + a] = acc["db3"][G == gg, r].sum()
    return g


def idqn_loss_grad(S, params, tparams, obss, actions, rewards, dones, filled, gamma, double_q, n_chunks=1):
    """params/tparams [P][NPARAM]; batch in the reference layout. Returns (loss, grad[P][NPARAM])."""
    P, T1, B, _ = obss.shape
    T = T1 - 1
    grads = np.zeros((P, S.NPARAM), np.float32)
    loss = 0.0
    for p in range(P):
        cpack, tpack, bpack = fwd_pack(S, params[p]), fwd_pack(S, tparams[p]), bwd_pack(S, params[p])
        acc = new_acc(S)
        bounds = [round(c * T / n_chunks) for c in range(n_chunks + 1)]
        for b0 in range(0, B, 16):
            for c in range(n_chunks):
                if bounds[c + 1] > bounds[c]:
                    loss_grad_task(S, cpack, tpack, bpack, obss[p], actions[p], rewards[p], dones, filled, b0,
                                   bounds[c], bounds[c + 1], gamma, double_q, acc)
        grads[p] = write_out(S, acc)
        loss += acc["loss"]
    nf = float(filled.sum())
    return loss / nf, grads / nf
