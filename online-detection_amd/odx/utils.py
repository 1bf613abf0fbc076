"""Helpers the experiment drivers call between the feature extractor and the trainers (A10):
feature statistics, normalisation, device moves, the on-disk feature cache and box decoding.
Behaviour follows src/py_od_utils.py (function-by-function citations below); the global torch
RNG is consumed in the same order as there, so seeded runs sample the same rows."""
import glob
import math
import os

import numpy as np
import torch
import yaml


def _device(cpu_tensor=False):
    return 'cpu' if (cpu_tensor or not torch.cuda.is_available()) else 'cuda'


def computeFeatStatistics_torch(positives, negatives, num_samples=4000, features_dim=2048, cpu_tensor=False,
                                pos_fraction=None):
    """mean / std / mean-norm over a random sample of rows (py_od_utils.py:59-95): per class
    ceil(num_samples/C * pos_fraction) positives and, per negative batch,
    ceil(num_samples/C * (1 - pos_fraction) / max_batches) negatives, drawn with replacement."""
    device = _device(cpu_tensor)
    print('Computing features statistics')
    if pos_fraction is None:
        pos_fraction = 1 / 10
        neg_fraction = 9 / 10
    else:
        neg_fraction = 1 - pos_fraction
    num_classes = len(positives)
    take_pos = math.ceil((num_samples / num_classes) * pos_fraction)
    max_batches = max([len(nb) for nb in negatives] + [0])
    take_neg = math.ceil(((num_samples / num_classes) * neg_fraction) / max_batches)
    picked = []
    for i in range(num_classes):
        if len(positives[i]) != 0:
            picked.append(positives[i][torch.randint(len(positives[i]), (take_pos,))].to(device))
        for batch in negatives[i]:
            if len(batch) != 0:
                picked.append(batch[torch.randint(len(batch), (take_neg,))].to(device))
    if picked:
        sampled = torch.cat([p.view(-1, features_dim) for p in picked])
    else:
        sampled = torch.empty((0, features_dim), device=device)
    norms = torch.norm(sampled, dim=1)
    out_dev = _device()
    return {'mean': torch.mean(sampled, dim=0).to(out_dev), 'std': torch.std(sampled, dim=0).to(out_dev),
            'mean_norm': torch.mean(norms).to(out_dev)}


def computeFeatStatistics(positives, negatives, feature_folder, is_rpn, num_samples=4000, basedir=None):
    """The older statistics helper (py_od_utils.py:8-56): the cached `stats` file of the feature folder if it loads, else
    mean / std / mean-norm over rows drawn through the global NUMPY RNG (one `randint` per class with positives, one
    `choice` per non-empty negative batch, in that order), seeded with the first row of the last class that has
    positives, and saved to that file.  Returns the tuple (mean, std, mean_norm).  `basedir`: directory the relative
    cache path starts from (default: the drop-in `src` directory, as the reference's starts from its own)."""
    if basedir is None:
        basedir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'src')
    if not is_rpn:
        stats_path = os.path.join(basedir, '..', 'Data', 'feat_cache', feature_folder, 'stats')
    else:
        stats_path = os.path.join(basedir, '..', 'Data', 'feat_cache_RPN', feature_folder, 'rpn_stats')
    try:
        l = torch.load(stats_path)
        return torch.as_tensor(l['mean']), torch.as_tensor(l['std']), torch.as_tensor(l['mean_norm'])
    except Exception:
        pass
    print('Computing features statistics')
    num_classes = len(positives)
    take_from_pos = math.ceil((num_samples / num_classes) * (1 / 10))
    take_from_neg = math.ceil(((num_samples / num_classes) * (9 / 10)) / len(negatives[0]))
    rows, norms = [], []
    for i in range(num_classes):                       # the seed row: the LAST class with positives wins
        if len(positives[i]) != 0:
            first = positives[i][0].cpu().numpy()
            rows, norms = [first[None, :]], [np.linalg.norm(first).reshape(1, 1)]
    for i in range(num_classes):
        if len(positives[i]) != 0:
            picked = positives[i][np.random.randint(len(positives[i]), size=take_from_pos)].cpu().numpy()
            rows.append(picked), norms.append(np.linalg.norm(picked, axis=1)[:, None])
        for batch in negatives[i]:
            if len(batch) != 0:
                picked = batch[np.random.choice(len(batch), size=take_from_neg)].cpu().numpy()
                rows.append(picked), norms.append(np.linalg.norm(picked, axis=1)[:, None])
    sampled, ns = np.vstack(rows), np.vstack(norms)
    mean, std, mean_norm = torch.tensor(np.mean(sampled, axis=0)), torch.tensor(np.std(sampled, axis=0)), torch.tensor(np.mean(ns))
    torch.save({'mean': mean, 'std': std, 'mean_norm': mean_norm}, stats_path)
    return mean, std, mean_norm


def zScores(feat, mean, mean_norm, target_norm=20):
    """py_od_utils.py:98-102"""
    feat = torch.tensor(feat)
    return (feat - mean) * (target_norm / mean_norm)


def normalize_COXY(COXY, stats, cpu=False):
    """py_od_utils.py:105-111"""
    mean = stats['mean'].to('cpu') if cpu else stats['mean']
    COXY['X'] = (COXY['X'] - mean) * (20 / stats['mean_norm'].item())
    return COXY


def falkon_models_to_cuda(models):
    """py_od_utils.py:113-118 (device = the GPU when there is one)"""
    dev = _device()
    for m in models:
        if m is not None:
            m.ny_points_ = m.ny_points_.to(dev)
            m.alpha_ = m.alpha_.to(dev)
    return models


def _load_batches(features_dir, prefix, clss_id):
    n = len(glob.glob(os.path.join(features_dir, '{}_cl_{}_*'.format(prefix, clss_id))))
    return [torch.load(os.path.join(features_dir, '{}_cl_{}_batch_{}'.format(prefix, clss_id, b))) for b in range(n)]


def load_features_classifier(features_dir, is_segm=False, cpu_tensor=False, sample_ratio=1, cfg_feature_extraction=None):
    """Feature cache reader (py_od_utils.py:120-200): files positives_cl_{c}_batch_{b} /
    negatives_cl_{c}_batch_{b}; detection keeps the negative batches as a list per class,
    segmentation concatenates them; optional re-shuffling into ITERATIONS batches of BATCH_SIZE."""
    n_pos = len(glob.glob(os.path.join(features_dir, 'positives_*')))
    n_neg = len(glob.glob(os.path.join(features_dir, 'negatives_*')))
    shuffle_features, bs_shuffled, nb_shuffled = False, 2000, 2
    if cfg_feature_extraction is not None:
        with open(cfg_feature_extraction) as fid:
            params = yaml.load(fid, Loader=yaml.FullLoader)
        mb = params.get('MINIBOOTSTRAP', {})
        for key, tag in (('RPN', 'RPN'), ('DETECTOR', 'detector')):
            if key in mb and tag in features_dir:
                shuffle_features = mb[key].get('SHUFFLE_NEGATIVES', shuffle_features)
                nb_shuffled = mb[key].get('ITERATIONS', nb_shuffled)
                bs_shuffled = mb[key].get('BATCH_SIZE', bs_shuffled)

    def merged(parts):
        try:
            if cpu_tensor:
                return torch.cat(parts).to('cpu')
            t = torch.cat(parts)
            if sample_ratio < 1:
                t = t[torch.randint(len(t), (int(len(t) * sample_ratio),))]
            return t
        except Exception:
            return torch.empty((0))

    positives, negatives = [], []
    loaded_pos = loaded_neg = clss_id = 0
    while loaded_pos < n_pos or loaded_neg < n_neg:
        pos_i = _load_batches(features_dir, 'positives', clss_id)
        loaded_pos += len(pos_i)
        positives.append(merged(pos_i))
        neg_i = _load_batches(features_dir, 'negatives', clss_id)
        loaded_neg += len(neg_i)
        negatives.append(merged(neg_i) if is_segm else neg_i)
        clss_id += 1
    if not is_segm and shuffle_features:
        negatives = shuffle_negatives(negatives, batch_size=bs_shuffled, num_batches=nb_shuffled)
    return positives, negatives


def load_features_regressor(features_dir, samples_fraction=1.0):
    """reg_{x,c,y}_batch_{i} reader (py_od_utils.py:202-224)"""
    nb = len(glob.glob(os.path.join(features_dir, 'reg_x_*')))
    Xs, Cs, Ys = [], [], []
    for i in range(nb):
        C_i = torch.load(os.path.join(features_dir, 'reg_c_batch_{}'.format(i)))
        X_i = torch.load(os.path.join(features_dir, 'reg_x_batch_{}'.format(i)))
        Y_i = torch.load(os.path.join(features_dir, 'reg_y_batch_{}'.format(i)))
        if samples_fraction < 1.0:
            ind = torch.randperm(len(C_i))[:int(len(C_i) * samples_fraction)]
            X_i, C_i, Y_i = X_i[ind], C_i[ind], Y_i[ind]
        Xs.append(X_i)
        Cs.append(C_i)
        Ys.append(Y_i)
    return {'C': torch.cat(Cs), 'O': None, 'X': torch.cat(Xs), 'Y': torch.cat(Ys)}


def load_positives_from_COXY(COXY, del_COXY=False, samples_fraction=1.0):
    """py_od_utils.py:226-239: class ids present in C are assumed to be 0 (background) .. K"""
    positives = []
    for i in range(len(torch.unique(COXY['C']))):
        ids = torch.where(COXY['C'] == i + 1)[0]
        if samples_fraction < 1.0:
            ids = ids[torch.randperm(len(ids))[:int(len(ids) * samples_fraction)]]
        positives.append(COXY['X'][ids])
        if del_COXY:
            rest = torch.where(COXY['C'] != i + 1)[0]
            COXY['X'] = COXY['X'][rest]
            COXY['C'] = COXY['C'][rest]
    return positives


def minibatch_positives(positives, num_batches):
    """py_od_utils.py:241-245"""
    for i in range(len(positives)):
        positives[i] = list(torch.split(positives[i], int(len(positives[i]) / num_batches)))
    return positives


def decode_boxes_detector(boxes, bbox_pred):
    """Decode (R, 4 K) regression outputs against R proposals with the +1 width convention and
    clamp to the image (py_od_utils.py:247-274)."""
    ex = boxes.bbox
    w = ex[:, 2] - ex[:, 0] + 1
    h = ex[:, 3] - ex[:, 1] + 1
    cx = ex[:, 0] + 0.5 * w
    cy = ex[:, 1] + 0.5 * h
    pcx = bbox_pred[:, 0::4] * w[:, None] + cx[:, None]
    pcy = bbox_pred[:, 1::4] * h[:, None] + cy[:, None]
    pw = torch.exp(bbox_pred[:, 2::4]) * w[:, None]
    ph = torch.exp(bbox_pred[:, 3::4]) * h[:, None]
    out = torch.zeros_like(bbox_pred)
    out[:, 0::4] = torch.clamp(pcx - 0.5 * pw, min=0)
    out[:, 1::4] = torch.clamp(pcy - 0.5 * ph, min=0)
    out[:, 2::4] = torch.clamp(pcx + 0.5 * pw - 1, max=boxes.size[0] - 1)
    out[:, 3::4] = torch.clamp(pcy + 0.5 * ph - 1, max=boxes.size[1] - 1)
    return out


def shuffle_negatives(negatives, batch_size=None, num_batches=None):
    """Pool a class's negative batches, permute, re-split (py_od_utils.py:276-294)."""
    out = []
    for per_class in negatives:
        bs = len(per_class[0]) if batch_size is None else batch_size
        pool = torch.cat(per_class)
        nb = math.ceil(len(pool) / bs) if num_batches is None else num_batches
        perm = torch.randperm(len(pool))
        out.append([pool[perm[min(j * bs, len(perm)):min((j + 1) * bs, len(perm))]] for j in range(nb)])
    return out


def mask_iou(mask_a, mask_b):
    """IoU between two sets of boolean masks (N, H, W) x (K, H, W) -> (N, K) (py_od_utils.py:297-331)."""
    if mask_a.shape[1:] != mask_b.shape[1:]:
        raise IndexError
    a = mask_a.reshape(mask_a.shape[0], -1).astype(np.float64)
    b = mask_b.reshape(mask_b.shape[0], -1).astype(np.float64)
    inter = a @ b.T
    union = a.sum(1)[:, None] + b.sum(1)[None, :] - inter
    return (inter / union).astype(np.float32)
