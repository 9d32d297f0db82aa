"""Test double for the sharded VBx protocol (fluidaudio_amd.sharding.vbx_refine_sharded): the slice-record arithmetic of csrc/vbx.hip in
numpy, so that the host orchestration (frame ranges, one all-gather per iteration, identical stopping decisions on every rank) runs in
the world_size-2 gloo test without a GPU.  Test infrastructure only — nothing in the product imports it.
Formulas: VBxClustering.swift:190-282 (set-up), :312-432 (speaker statistics), :441-572 (E-step), :586-647 (pi, ELBO)."""
import numpy as np

SLICES = 64


class NumpyVbxShard:
    def __init__(self, rho_local, labels_local, T_total, S, phi, rank, world, Fa=0.07, Fb=0.8):
        import torch
        self.torch = torch
        per = -(-T_total // SLICES)
        zn = SLICES // world
        self.z = range(rank * zn, (rank + 1) * zn)
        self.t0g = min(rank * zn * per, T_total)
        self.per, self.Tg = per, T_total
        X = np.ascontiguousarray(rho_local, np.float64)
        self.T, self.D = X.shape
        self.S, self.Fa, self.Fb = S, Fa, Fb
        self.phi = np.maximum(np.asarray(phi, np.float64), 1e-12)
        self.rho = X * np.sqrt(self.phi)
        self.G = -0.5 * ((X * X).sum(1) + self.D * np.log(2 * np.pi))
        lab = np.clip(np.asarray(labels_local, np.int64), 0, S - 1)
        g = np.zeros((self.T, S))
        g[np.arange(self.T), lab] = 7.0
        g = np.exp(g - 7.0)
        g /= g.sum(1, keepdims=True)
        self.gamma = g / g.sum(1, keepdims=True)
        self.pi = np.full(S, 1.0 / S)
        self.ll = np.zeros(self.T)
        self.stride = S * (self.D + 1) + 1

    def _records(self):
        out = np.zeros((len(self.z), self.stride))
        for i, z in enumerate(self.z):
            t0, t1 = z * self.per - self.t0g, min((z + 1) * self.per, self.Tg) - self.t0g
            t0, t1 = max(t0, 0), max(t1, 0)
            if t1 > t0:
                g, r = self.gamma[t0:t1], self.rho[t0:t1]
                rec = np.concatenate([g.T @ r, g.sum(0)[:, None]], axis=1)
                out[i, :-1] = rec.ravel()
                out[i, -1] = self.ll[t0:t1].sum()
        return self.torch.from_numpy(out.ravel().copy())

    def begin(self):
        return self._records()

    def _stats(self, full):
        f = full.numpy().reshape(SLICES, self.stride)
        rec = np.zeros((self.S, self.D + 1))
        for z in range(SLICES):                                  # slice order, like the device
            rec = rec + f[z, :-1].reshape(self.S, self.D + 1)
        return rec, f[:, -1]

    def iterate(self, full):
        rec, _ = self._stats(full)
        ns = rec[:, self.D]
        weight = (self.Fa / self.Fb) * ns
        self.invL = 1.0 / np.maximum(1.0 + weight[:, None] * self.phi[None, :], 1e-12)
        self.alpha = (rec[:, :self.D] * self.invL) * (self.Fa / self.Fb)
        phiT = ((self.alpha * self.alpha + self.invL) * self.phi[None, :]).sum(1)
        logpi = np.log(np.maximum(self.pi, 1e-8))
        lp = ((self.rho @ self.alpha.T + phiT[None, :] * -0.5) + self.G[:, None]) * self.Fa + logpi[None, :]
        mx = lp.max(1, keepdims=True) if self.T else lp
        e = np.exp(lp - mx)
        s = e.sum(1, keepdims=True)
        self.gamma = e / s
        self.ll = (mx + np.log(s)).ravel()
        return self._records()

    def finish(self, full):
        rec, llz = self._stats(full)
        ll = 0.0
        for v in llz:
            ll += v
        pi = rec[:, self.D]
        ps = pi.sum()
        self.pi = pi / ps if ps > 0 and np.isfinite(ps) else np.full(self.S, 1.0 / self.S)
        return float(ll + self.Fb * 0.5 * (np.log(self.invL) - self.invL - self.alpha * self.alpha + 1.0).sum())

    def result(self):
        return self.gamma, self.pi, np.argmax(self.gamma, axis=1).astype(np.int32)
