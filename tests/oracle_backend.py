"""A backend with the interface of odx.backend.HipBackend, computing with the numpy oracle on
CPU tensors.  TEST INFRASTRUCTURE: it lets the CPU suite (and the gloo world_size-2 tests) drive
the product's host logic — solver.py, dist.py, the drop-in modules — without a GPU.  It is never
importable from the product package."""
import numpy as np
import scipy.linalg as sla
import torch

from oracle import falkon_ref as fr


class Features:
    def __init__(self, X):
        self.X = X
        self.n, self.D = X.shape
        self.ld = self.D
        self.sq = (X.double() ** 2).sum(1)


class _Precond:
    pass


class _Knm:
    pass


class OracleBackend:
    name = "numpy-oracle"

    def __init__(self, dtype_k=np.float32):
        self.dtype_k = dtype_k
        self.device = torch.device("cpu")

    def vec(self, x):
        return torch.as_tensor(x, dtype=torch.float64).contiguous()

    def zeros(self, n, dtype=torch.float64):
        return torch.zeros(n, dtype=dtype)

    def synchronize(self):
        pass

    def features(self, X):
        X = torch.as_tensor(X).to(torch.float32).contiguous()
        return Features(X)

    def rows(self, F, idx):
        idx = torch.as_tensor(idx, dtype=torch.int64).reshape(-1)
        return Features(F.X[idx].contiguous())

    def precond(self, Zf, sigma, lam, eps):
        ref = fr.Preconditioner(Zf.X.numpy().astype(np.float64), sigma, lam, eps, np.float64)
        P = _Precond()
        Ti, Ai = np.linalg.inv(ref.T), np.linalg.inv(ref.A)
        P.LTit, P.LTi, P.LAit, P.LAi = (torch.from_numpy(np.ascontiguousarray(a)) for a in (Ti, Ti.T, Ai, Ai.T))
        P.M, P.ld = Zf.n, Zf.n
        P.info = torch.zeros(1, dtype=torch.int32)
        return P

    def check_precond(self, P):
        pass

    def knm(self, F, Zf, sigma, out=None):
        K = _Knm()
        K.K = torch.from_numpy(fr.gaussian_kernel(F.X.numpy(), Zf.X.numpy(), sigma, self.dtype_k))
        K.n, K.M, K.ld = F.n, Zf.n, Zf.n
        return K

    def knm_rhs(self, F, Zf, sigma, w, out=None, rhs_out=None):
        K = self.knm(F, Zf, sigma, out=out)
        return K, self.ktk(K, w=w, out=rhs_out)

    def ktk(self, K, v=None, w=None, out=None):
        Kd = K.K.double()
        t = Kd @ v if v is not None else torch.zeros(K.n, dtype=torch.float64)
        if w is not None:
            t = t + w
        r = Kd.t() @ t
        if out is not None:
            out.copy_(r)
            return out
        return r

    fold = True     # offer the two-vector pass (solver.py folds the periodic full residual into it)

    def can_ktk2(self, K):
        return self.fold

    def ktk2(self, K, v1, v2, out1=None, out2=None):
        return self.ktk(K, v=v1, out=out1), self.ktk(K, v=v2, out=out2)

    def cg_residual(self, B, AX, AP, state, R):
        if state[2] != 0:
            return
        R.copy_(B - (AX + state[3] * AP))

    def trmv(self, P, name, x, alpha=1.0, beta=0.0, z=None, out=None):
        r = alpha * (getattr(P, name) @ x)
        if beta != 0.0:
            r = r + beta * z
        if out is not None:
            out.copy_(r)
            return out
        return r

    def cg_init(self, B, X, R, Pv, state):
        X.zero_()
        R.copy_(B)
        Pv.copy_(B)
        s = float((B * B).sum())
        state[0], state[1], state[2], state[3] = s, s, 0.0, 0.0

    def cg_step(self, X, R, Pv, AP, state, cg_eps, full_grad):
        if state[2] != 0:
            return
        a = state[0] / ((Pv * AP).sum() + cg_eps)
        X.add_(a * Pv)
        if not full_grad:
            R.sub_(a * AP)
        state[3] = a

    def cg_finish(self, R, Pv, state, cg_eps, tol):
        if state[2] != 0:
            return
        s = (R * R).sum()
        if float(torch.sqrt(torch.abs(s))) < tol:
            state[1], state[2] = s, 1.0
            return
        b = s / (state[0] + cg_eps)
        Pv.mul_(b).add_(R)
        state[1], state[0] = s, s

    def axpby(self, a, x, b, y):
        y.mul_(b).add_(a * x)

    def mmv(self, F, Zf, sigma, V, ranges=None, out=None, max_range=None):
        V = torch.as_tensor(V, dtype=torch.float64)
        if V.dim() == 1:
            V = V[:, None]
        res = torch.from_numpy(fr.kernel_mmv(F.X.numpy().astype(np.float64), Zf.X.numpy().astype(np.float64),
                                             V.numpy(), sigma, np.float64)).float()
        if out is not None:
            out.copy_(res)
            return out
        return res

    def gemm_nt(self, Fa, Fb):
        return (Fa.X.double() @ Fb.X.double().t()).float()

    def roi_align(self, feat, rois, spatial_scale, output_size, sampling_ratio=0):
        from oracle import roi_ref
        return torch.from_numpy(roi_ref.roi_align(feat.numpy(), rois.numpy(), spatial_scale, output_size, sampling_ratio)).float()

    def roi_align_fpn(self, feats, rois, scales, output_size, sampling_ratio=2, return_levels=False):
        from oracle import roi_ref
        out = torch.from_numpy(roi_ref.roi_align_fpn([f.numpy() for f in feats], rois.numpy(), list(scales), output_size,
                                                     sampling_ratio)).float()
        if return_levels:
            return out, torch.from_numpy(roi_ref.fpn_levels(rois.numpy()[:, 1:5], list(scales))).int()
        return out

    def paste_masks(self, masks, boxes, im_h, im_w, thresh=0.5, padding=1):
        from oracle import roi_ref
        m, b = masks.numpy(), boxes.numpy()
        return torch.from_numpy(np.stack([roi_ref.paste_mask(m[i], b[i], im_h, im_w, thresh, padding) for i in range(len(m))])
                                if len(m) else np.zeros((0, im_h, im_w), dtype=bool))

    def nms(self, boxes, scores, thr):
        from oracle import roi_ref
        return torch.from_numpy(roi_ref.nms(boxes.numpy(), scores.numpy(), thr))

    # RLS
    def rls_gram(self, F, idx, Yt, G, XtY):
        X = F.X[idx].double()
        Xb = torch.cat([X, torch.ones(len(X), 1, dtype=torch.float64)], 1)
        D1 = Xb.shape[1]
        G[:, :D1] += torch.tril(Xb.t() @ Xb)
        XtY[:, :D1] += Yt[:, :len(X)] @ Xb

    def rls_solve(self, G, D, lam, XtY):
        D1 = D + 1
        A = G[:, :D1].numpy()
        A = np.tril(A) + np.tril(A, -1).T + lam * np.eye(D1)
        R = sla.cholesky(A, lower=True)
        W = np.stack([sla.solve_triangular(R.T, sla.solve_triangular(R, XtY[k, :D1].numpy(), lower=True), lower=False)
                      for k in range(4)])
        return torch.from_numpy(W), torch.zeros(1, dtype=torch.int32)

    def gemm_nt_f64(self, A, B):
        return A.double() @ B.double().t()

    def rls_predict_rows(self, F, idx, W):
        X = (F.X if idx is None else F.X[idx]).double()
        return X @ W[:, :-1].t() + W[:, -1]
