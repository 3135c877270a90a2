"""Lane-exact numpy emulation of the rollout kernel's dense-layer dataflow
(ethz_safe_learning_amd/csrc/cem_device.h) on top of the REAL host packing code
(cem_pack_weights_host), used by the CPU test-suite to validate the weight-stream
layout and the accumulator-as-operand trick without a GPU.

v_mfma_f32_16x16x4_f32 lane maps (cdna_hip_programming.md section 3):
  A[i][k]: lane = 16*k + i      B[k][j]: lane = 16*k + j
  D[m][j]: lane = 16*(m//4) + j, register m%4
"""
import numpy as np


def perm_hidden(w, phi):
    """csrc/cem_device.h cem_perm_hidden: the wave's own blocks 2w, 2w+1 first."""
    return 2 * w + phi if phi < 2 else (phi - 2 if phi - 2 < 2 * w else phi)


def perm_l0(w, nfw, phi):
    """csrc/cem_device.h cem_perm_l0: the wave's own input blocks w, w+4, .. first."""
    if phi < nfw:
        return w + 4 * phi
    return [F for F in range(4 * nfw) if (F & 3) != w][phi - nfw]


def mfma_16x16x4(a, b, acc):
    """a[64], b[64] one float per lane; acc[64,4] -> acc + A.B in the D lane map."""
    A = a.reshape(4, 16).T.astype(np.float64)        # A[i][k]
    B = b.reshape(4, 16).astype(np.float64)          # B[k][j]
    D = A @ B                                        # D[m][j]
    out = acc.astype(np.float64).copy()
    for lane in range(64):
        q, j = lane >> 4, lane & 15
        for r in range(4):
            out[lane, r] += D[4 * q + r, j]
    return out


class TileEmulator:
    """One workgroup (4 waves) processing one tile of 16*RC rows of one member."""

    def __init__(self, packed_member, dims):
        # dims: dict(O, A, L, KB_in, KB_obs, NFW, wave_groups[4], wave_off_f4[4])
        self.d = dims
        self.streams = []
        for w in range(4):
            off = dims['wave_off_f4'][w] * 4
            n = dims['wave_groups'][w]
            self.streams.append(packed_member[off:off + n * 512].reshape(n, 2, 64, 4))
        self.pos = [0, 0, 0, 0]

    def pop(self, w):
        g = self.streams[w][self.pos[w]]
        self.pos[w] = (self.pos[w] + 1) % self.d['wave_groups'][w]
        return g

    def stage(self, X, kf, rc, l0=False):
        """X[c][F][lane][r] (the LDS exchange image) -> per wave (acc0, acc1)[c][lane][4].  Each wave visits the
        k-blocks in its own order (own blocks first), which is also the order of its weight stream."""
        outs = []
        for w in range(4):
            acc0 = np.zeros((rc, 64, 4))
            acc1 = np.zeros((rc, 64, 4))
            for P in range(kf):
                F = perm_l0(w, kf // 4, P) if l0 else perm_hidden(w, P)
                # the second accumulator of a hidden stage visits the wave's own two input blocks swapped (cem_mfma_stage)
                Fb = F if l0 else perm_hidden(w, P ^ 1 if P < 2 else P)
                g = self.pop(w)
                for r in range(4):
                    for c in range(rc):
                        acc0[c] = mfma_16x16x4(g[0][:, r], X[c, F, :, r], acc0[c])
                        acc1[c] = mfma_16x16x4(g[1][:, r], X[c, Fb, :, r], acc1[c])
            outs.append((acc0, acc1))
        return outs

    def forward(self, x_rows, biases_h, b_mu, b_var):
        """x_rows [16*rc, O+A] already scaled -> (mu_pre[rows,O], var_pre[rows,O]) following the kernel:
        hidden stages publish relu(acc + b) to X[c][2w+g]; head stage of block Fo gives mu/var of features
        16Fo + 4q + r on lane (q, j)."""
        d = self.d
        rows = x_rows.shape[0]
        rc = rows // 16
        Din = d['O'] + d['A']
        # input image: X[c][F][lane=(q,j)][r] = x[16c + j][16F + 4q + r]
        X = np.zeros((rc, 8, 64, 4))
        for c in range(rc):
            for F in range(d['KF0']):
                for lane in range(64):
                    q, j = lane >> 4, lane & 15
                    for r in range(4):
                        f = 16 * F + 4 * q + r
                        X[c, F, lane, r] = x_rows[16 * c + j, f] if f < Din else 0.0
        hidden = []
        for l in range(d['L']):
            outs = self.stage(X, d['KF0'] if l == 0 else 8, rc, l0=(l == 0))
            Xn = np.zeros((rc, 8, 64, 4))
            for w in range(4):
                for g in range(2):
                    G = 2 * w + g
                    for lane in range(64):
                        q = lane >> 4
                        bias = biases_h[l][16 * G + 4 * q:16 * G + 4 * q + 4]
                        for c in range(rc):
                            Xn[c, G, lane] = np.maximum(outs[w][g][c][lane] + bias, 0.0)
            X = Xn
            # decode X back to [rows, 128] for the caller
            h = np.zeros((rows, 128))
            for c in range(rc):
                for G in range(8):
                    for lane in range(64):
                        q, j = lane >> 4, lane & 15
                        h[16 * c + j, 16 * G + 4 * q:16 * G + 4 * q + 4] = X[c, G, lane]
            hidden.append(h)
        mu = np.zeros((rows, d['O']))
        var = np.zeros((rows, d['O']))
        # heads: per wave, owned blocks in order i = 0..NFW-1 (one stage each)
        per_wave = [[] for _ in range(4)]
        for i in range(d['NFW']):
            for w in range(4):
                Fo = w + 4 * i
                if Fo < d['KB_obs']:
                    acc_m = np.zeros((rc, 64, 4))
                    acc_v = np.zeros((rc, 64, 4))
                    for P in range(8):
                        F = perm_hidden(w, P)
                        Fb = perm_hidden(w, P ^ 1 if P < 2 else P)
                        g = self.pop(w)
                        for r in range(4):
                            for c in range(rc):
                                acc_m[c] = mfma_16x16x4(g[0][:, r], X[c, F, :, r], acc_m[c])
                                acc_v[c] = mfma_16x16x4(g[1][:, r], X[c, Fb, :, r], acc_v[c])
                    per_wave[w].append((Fo, acc_m, acc_v))
        for w in range(4):
            for Fo, acc_m, acc_v in per_wave[w]:
                for c in range(rc):
                    for lane in range(64):
                        q, j = lane >> 4, lane & 15
                        for r in range(4):
                            f = 16 * Fo + 4 * q + r
                            if f < d['O']:
                                mu[16 * c + j, f] = acc_m[c][lane, r] + b_mu[f]
                                var[16 * c + j, f] = acc_v[c][lane, r] + b_var[f]
        return hidden, mu, var


def dims_of(obs_dim, act_dim, n_layers):
    Din = obs_dim + act_dim
    KB_in, KB_obs = (Din + 15) // 16, (obs_dim + 15) // 16
    NFW = (KB_in + 3) // 4
    KF0 = 4 * NFW                      # layer-0 groups per wave, zero padded (ring phase)
    groups, offs, off = [], [], 0
    for w in range(4):
        g = KF0 + 8 * (n_layers - 1) + 8 * sum(1 for i in range(NFW) if w + 4 * i < KB_obs)
        groups.append(g)
        offs.append(off)
        off += g * 128
    return dict(O=obs_dim, A=act_dim, L=n_layers, KB_in=KB_in, KB_obs=KB_obs, NFW=NFW, KF0=KF0, wave_groups=groups,
                wave_off_f4=offs, member_stride_f4=off + 256)
