"""CPU restatement (numpy f64) of the RLS box regressors — TEST ORACLE ONLY.

Pinned: checked against golden vectors produced by the reference's own
RegionRefinerTrainer / RegionPredictor (tests/golden/make_golden.py -> rls_golden.npz).

train  : RegionRefinerTrainer.train / solve
         (src/modules/region-refiner/region_refiner_trainer/train_region_refiner.py:25-119)
apply  : RegionPredictor.predict (region_predictor/predict_regions.py:16-80)
decode : decode_boxes_detector (src/py_od_utils.py:247-274)
"""
import numpy as np
import scipy.linalg as sla


def train_class(Xi, Yi, lam):
    """One class: Xi (n, D) features, Yi (n, 4) targets -> dict(mu, T, T_inv, W (4, D+1), losses (4, n))."""
    Xi = np.asarray(Xi, dtype=np.float64)
    Yi = np.asarray(Yi, dtype=np.float64).copy()
    Xb = np.concatenate([Xi, np.ones((Xi.shape[0], 1))], axis=1)          # :57-58
    mu = Yi.mean(0)                                                        # :61
    Yi -= mu                                                               # :62
    S = Yi.T @ Yi / Yi.shape[0]                                            # :63
    D, W = np.linalg.eigh(S)                                               # :64 (S symmetric)
    T = W @ np.diag(1.0 / np.sqrt(D + 0.001)) @ W.T                        # :66
    T_inv = W @ np.diag(np.sqrt(D + 0.001)) @ W.T                          # :67
    Yw = Yi @ T                                                            # :68
    G = Xb.T @ Xb + lam * np.eye(Xb.shape[1])                              # :102
    R = sla.cholesky(G, lower=True)                                        # :103
    Wt, losses = [], []
    for k in range(4):
        z = sla.solve_triangular(R, Xb.T @ Yw[:, k], lower=True)           # :114
        w = sla.solve_triangular(R.T, z, lower=False)                      # :115
        Wt.append(w)
        losses.append(0.5 * (Xb @ w - Yw[:, k]) ** 2)                      # :116
    return {"mu": mu, "T": T, "T_inv": T_inv, "W": np.stack(Wt), "losses": np.stack(losses)}


def train(C, X, Y, num_classes, lam, is_rpn=False):
    """All classes; returns a list of dicts (None for a class without rows), class loop starts at
    1 for the detector and 0 for the RPN (:27-30)."""
    C = np.asarray(C).reshape(-1)
    out = []
    for i in range(0 if is_rpn else 1, num_classes):
        I = np.nonzero(C == i)[0]
        out.append(None if len(I) == 0 else train_class(X[I], Y[I], lam))
    return out


def decode(ex_box, Y, img_w, img_h, plus):
    src_w = ex_box[:, 2] - ex_box[:, 0] + plus
    src_h = ex_box[:, 3] - ex_box[:, 1] + plus
    cx = ex_box[:, 0] + 0.5 * src_w
    cy = ex_box[:, 1] + 0.5 * src_h
    pcx, pcy = Y[:, 0] * src_w + cx, Y[:, 1] * src_h + cy
    pw, ph = np.exp(Y[:, 2]) * src_w, np.exp(Y[:, 3]) * src_h
    return np.stack([np.maximum(pcx - 0.5 * pw, 0), np.maximum(pcy - 0.5 * ph, 0),
                     np.minimum(pcx + 0.5 * pw - 1, img_w - 1), np.minimum(pcy + 0.5 * ph - 1, img_h - 1)], axis=1)


def apply(models, boxes, feat, gt, img_size):
    """RegionPredictor.predict for one image: -> (R, C+1, 4) with the input boxes in slot 0."""
    keep = np.nonzero(gt == 0)[0]
    F = np.asarray(feat, dtype=np.float64)[keep]
    ex = np.asarray(boxes, dtype=np.float64)
    out = [ex]
    for m in models:
        Y = F @ m["W"][:, :-1].T + m["W"][:, -1]
        Y = Y @ m["T_inv"] + m["mu"]
        out.append(decode(ex, Y, img_size[0], img_size[1], np.spacing(1)))
    return np.concatenate(out, axis=1).reshape(ex.shape[0], len(models) + 1, 4)


def decode_boxes_detector(ex_box, bbox_pred, img_size):
    ex = np.asarray(ex_box, dtype=np.float64)
    P = np.asarray(bbox_pred, dtype=np.float64)
    out = np.zeros_like(P)
    for k in range(P.shape[1] // 4):
        out[:, 4 * k:4 * k + 4] = decode(ex, P[:, 4 * k:4 * k + 4], img_size[0], img_size[1], 1.0)
    return out
