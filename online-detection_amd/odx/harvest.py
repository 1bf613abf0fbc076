"""Train-time feature harvesting for the on-line detector (A11): turns the pooled RoI features
of one image at a time into the structures the on-line trainers consume —
    positives[c]        (p_c, D)   ground-truth RoI features of class c
    negatives[c][b]     (<= BATCH_SIZE, D) minibootstrap batches of background RoIs of class c
    COXY                regressor rows: X features, Y box targets, C class ids
and collects `test_boxes` entries at test time.

Behaviour follows ROIBoxHead (mrcnn_modified/modeling/roi_heads/box_head/box_head_getProposals.py):
initialize_online_detection_params :37-88, add_new_class :90-99, forward_train :107-292,
forward_test :295-334, and the final assembly of FeatureExtractorDetector.train
(feature_extractor_detector/extract_features_detector.py:255-292).  The global torch RNG is
consumed in the reference's order (one randint per class per image), so seeded runs pick the
same rows.  Details kept on purpose: proposals arrive with the ground-truth boxes prepended
(generalized_rcnn_getProposals.py:90-96) and regression targets are taken against those
prepended rows (:181); coordinates are clamped to [0, size-1]; IoU uses the +1 convention
(utils/evaluations.py:4-18); `negatives_to_pick = ceil(BATCH_SIZE * ITERATIONS / NUM_IMAGES)`
rows per class per image are spread over the class's still-open batches.

MI355X form: IoU against all ground-truth boxes is one broadcast op instead of a per-box loop; the
negatives of ALL classes of an image land in one preallocated (class, batch, row, D) device tensor with
one gather + one index_copy_ (the reference appends per class and open batch, several hundred small
copies per image, and grows its batches with O(n^2) torch.cat chains); the positives / regressor rows
of all ground-truth boxes come from one nonzero() and one copy per buffer.  The host keeps what the
reference fixes there: the order of the global RNG draws and the integer bookkeeping of open batches.
"""
import math
import os
import threading

import numpy as np
import torch

from . import options as _options


class _Staging:
    """One page-locked block per host thread, split in two halves that are filled in turn: a small host tensor is copied into the
    next free bytes and goes to the device from there with an ASYNCHRONOUS copy.  A half is entered again only after the copies
    issued from it the last time have completed (one event per stream that used it, recorded when the half was left; by then
    the other half — a megabyte of index lists, tens of images — has gone by, so the wait is never a wait)."""
    HALF = 1 << 20

    def __init__(self):
        self.buf = torch.empty(2 * self.HALF, dtype=torch.uint8).pin_memory()
        self.half, self.off = 0, 0
        self.used = [set(), set()]          # streams that copied out of each half since it was entered
        self.fence = [[], []]               # events behind the last copies out of each half

    def put(self, host, device):
        n = host.numel() * host.element_size()
        if self.off + n > self.HALF:
            self.fence[self.half] = []
            for st in self.used[self.half]:
                ev = torch.cuda.Event()
                ev.record(st)
                self.fence[self.half].append(ev)
            self.half, self.off = 1 - self.half, 0
            for ev in self.fence[self.half]:
                ev.synchronize()
            self.used[self.half] = set()
        base = self.half * self.HALF + self.off
        slot = self.buf[base:base + n].view(host.dtype).view(host.shape)
        slot.copy_(host)
        self.off += (n + 63) // 64 * 64
        out = torch.empty(host.shape, dtype=host.dtype, device=device)
        out.copy_(slot, non_blocking=True)
        self.used[self.half].add(torch.cuda.current_stream(out.device))
        return out


_staging = threading.local()


def to_device(host, device):
    """A small host tensor (index lists, draws, labels) on `device`: a plain copy.  odx.options staged_uploads=True: through a persistent
    page-locked staging block and an asynchronous copy (_Staging) — in isolation the difference is large (a `.to(device)` of
    pageable memory waits, on the host, for whatever the GPU is running: 1.5 ms per copy beside a busy stream against 15-20 us
    staged, tools/h2d_probe.py; the harvesters make ~11 such copies per image beside the next group's forward), but the harvest
    loop as a whole did not move with it (4.4-6.3 ms per image run to run either way: the wait only moves to the group's one
    host read — what the loop waits for is the harvesters' ~100 small dependent kernels per image getting their turn beside the
    forward's chip-filling products).  Opt-in until that is understood.  (Round 5 had also tried a freshly pinned block per
    list: 12 ms per image, the allocation.)"""
    device = torch.device(device)
    nbytes = host.numel() * host.element_size()
    if (device.type != "cuda" or host.is_cuda or nbytes == 0 or nbytes > _Staging.HALF // 4 or not host.is_contiguous()
            or not _options.current().staged_uploads):
        return host.to(device)
    st = getattr(_staging, "block", None)
    if st is None:
        st = _staging.block = _Staging()
    return st.put(host, device)


def _label_backend(t):
    """The backend whose labelling kernels serve tensors on t's device (HipBackend on the GPU: odx_rpn_label_f32, odx_det_label_f32,
    odx_box_targets_f32), or None — the tensor statements below are then what runs (the CPU, the tests' oracle backend)."""
    if not t.is_cuda:
        return None
    from . import backend as _backend
    be = _backend.get_backend()
    return be if hasattr(be, "det_label") and hasattr(be, "rpn_label") else None


def _box_targets(ex, tg):
    """((gx - sx) / sw, (gy - sy) / sh, log(gw / sw), log(gh / sh)) of (example, target) box rows — one launch on the GPU."""
    be = _label_backend(ex)
    if be is not None:
        return be.box_targets(ex, tg)
    sw, sh = ex[:, 2] - ex[:, 0] + 1, ex[:, 3] - ex[:, 1] + 1
    sx, sy = ex[:, 0] + 0.5 * sw, ex[:, 1] + 0.5 * sh
    gw, gh = tg[:, 2] - tg[:, 0] + 1, tg[:, 3] - tg[:, 1] + 1
    gx, gy = tg[:, 0] + 0.5 * gw, tg[:, 1] + 0.5 * gh
    return torch.stack(((gx - sx) / sw, (gy - sy) / sh, torch.log(gw / sw), torch.log(gh / sh)), dim=1)


def box_iou_plus1(gt, prop):
    """(G, R) IoU, +1 pixel convention, 0 where boxes do not touch (utils/evaluations.py:4-18)."""
    xmin = torch.max(gt[:, None, 0], prop[None, :, 0])
    ymin = torch.max(gt[:, None, 1], prop[None, :, 1])
    xmax = torch.min(gt[:, None, 2], prop[None, :, 2])
    ymax = torch.min(gt[:, None, 3], prop[None, :, 3])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    inter = w * h
    ga = ((gt[:, 2] - gt[:, 0] + 1) * (gt[:, 3] - gt[:, 1] + 1))[:, None]
    pa = ((prop[:, 2] - prop[:, 0] + 1) * (prop[:, 3] - prop[:, 1] + 1))[None, :]
    ov = inter / (ga + pa - inter)
    return torch.where((w > 0) & (h > 0), ov, torch.zeros_like(ov))


class _Growing:
    """Append-only (rows, D) buffer with amortised growth.  With `mark_every` it also remembers where the
    reference would have closed a batch (a batch is closed by the first append that brings it to >= mark_every
    rows), so that the on-disk feature cache can be cut into the reference's files (odx/storage.py)."""

    def __init__(self, D, device, dtype=torch.float32, cap=256, mark_every=None):
        self.buf = torch.empty((cap, D) if D else (cap,), dtype=dtype, device=device)
        self.n = 0
        self.mark_every, self.marks = mark_every, []

    def append(self, rows, seg_lens=None):
        """seg_lens: the row counts of the appends the reference would have made for these rows (one per ground-truth box
        / anchor type): the rows land with ONE device copy, the batch-closing rule is still applied append by append."""
        k = rows.shape[0]
        if self.n + k > self.buf.shape[0]:
            cap = max(2 * self.buf.shape[0], self.n + k)
            nb = torch.empty((cap,) + tuple(self.buf.shape[1:]), dtype=self.buf.dtype, device=self.buf.device)
            nb[: self.n] = self.buf[: self.n]
            self.buf = nb
        if k:
            self.buf[self.n:self.n + k] = rows if rows.device == self.buf.device else rows.to(self.buf.device)
        at = self.n
        self.n += k
        if self.mark_every:
            for seg in ([k] if seg_lens is None else seg_lens):
                at += seg
                if at - (self.marks[-1] if self.marks else 0) >= self.mark_every:
                    self.marks.append(at)

    def view(self):
        return self.buf[: self.n]

    def batches(self, marks=None):
        """The rows cut at the recorded batch ends (the last, open batch included even when empty — the
        reference always has one more batch open)."""
        edges = [0] + list(self.marks if marks is None else marks) + [self.n]
        return [self.buf[a:b] for a, b in zip(edges[:-1], edges[1:])]


class _Slot:
    """One (class, batch) window of the negatives store of a _SlotGrid, with the interface of a _Growing buffer
    (n, view, append).  Its fill count lives in the grid's host matrix `fill` (the vectorised planner works on whole rows of
    it); `n` reads and writes that entry."""

    def __init__(self, grid, cls, b):
        self.grid, self.cls, self.b = grid, cls, b
        self.mark_every, self.marks = None, []

    @property
    def n(self):
        return int(self.grid.fill[self.cls, self.b])

    @n.setter
    def n(self, v):
        self.grid.fill[self.cls, self.b] = v

    def view(self):
        return self.grid.store[self.cls, self.b, : self.n]

    def append(self, rows):
        k = rows.shape[0]
        self.grid.store[self.cls, self.b, self.n:self.n + k] = rows.to(self.grid.store.device)
        self.n += k


class _SlotGrid:
    """Negatives of the fill mode: a batch never grows past batch_size rows, so every (class, batch) has a fixed home in
    ONE (classes, batches, batch_size, D) tensor.  `plan` / `plan_rows` record, on the host, where the rows sampled for an
    image go (the reference's bookkeeping, integer arithmetic only); `commit` moves all of them with one gather + one
    index_copy_ — the reference (and round 1 here) appends them piece by piece, one small device copy per class and
    open batch: several hundred per image at 30 classes x 10 batches."""

    def __init__(self, D, iterations, batch_size, device):
        self.D, self.iterations, self.batch_size, self.device = D, iterations, batch_size, device
        self.store = torch.empty((0, iterations, batch_size, D), dtype=torch.float32, device=device)
        self.fill = np.zeros((0, iterations), dtype=np.int64)          # rows in every (class, batch) window
        self.classes = 0
        self.dst, self.src, self.runs = [], [], []

    def _grow(self, cap):
        new = torch.empty((cap,) + tuple(self.store.shape[1:]), dtype=torch.float32, device=self.device)
        new[: self.classes] = self.store[: self.classes]
        self.store = new
        fill = np.zeros((cap, self.iterations), dtype=np.int64)
        fill[: self.fill.shape[0]] = self.fill
        self.fill = fill

    def add_class(self):
        if self.classes == self.store.shape[0]:                       # grow (construction reserves the exact count)
            self._grow(max(1, 2 * self.store.shape[0]))
        c = self.classes
        self.classes += 1
        return [_Slot(self, c, b) for b in range(self.iterations)]

    def reserve(self, classes):
        if classes > self.store.shape[0]:
            self._grow(classes)

    def plan(self, slot, src_start, k):
        if k > 0:
            base = (slot.cls * self.iterations + slot.b) * self.batch_size + slot.n
            self.dst.append((base, k))
            self.src.append(src_start)
            slot.n += k

    def plan_rows(self, cls, take, src):
        """The plans of whole classes at once: take (len(cls), batches) rows into every window of the classes `cls` (host
        arrays), read from src (same shape) on in the image's sampled rows; class-major, batch-minor — the order of the
        box-by-box `plan` calls this replaces (fill_plan)."""
        ci, bi = np.nonzero(take)
        if ci.size:
            c = np.asarray(cls, dtype=np.int64)[ci]
            ks = take[ci, bi]
            base = (c * self.iterations + bi) * self.batch_size + self.fill[c, bi]
            self.runs.append((base, src[ci, bi], ks))
            self.fill[c, bi] += ks

    def commit(self, rows):
        """rows: the sampled rows of the image, all classes back to back (the `src_start` frame of plan)."""
        if self.dst:
            r = np.asarray(self.dst, dtype=np.int64).reshape(-1, 2)
            self.runs.append((r[:, 0], np.asarray(self.src, dtype=np.int64), r[:, 1]))
            self.dst, self.src = [], []
        if not self.runs:
            return
        # (several hundred (start, count) runs per image: expanded with a handful of array operations, not one arange per run)
        base = np.concatenate([r[0] for r in self.runs])
        start = np.concatenate([r[1] for r in self.runs])
        ks = np.concatenate([r[2] for r in self.runs])
        self.runs = []
        ramp = np.arange(int(ks.sum()), dtype=np.int64) - np.repeat(np.cumsum(ks) - ks, ks)
        dst = np.repeat(base, ks) + ramp
        src = np.repeat(start, ks) + ramp
        idx = to_device(torch.from_numpy(np.stack((dst, src))), rows.device)      # one host -> device copy
        # the rows take the store's dtype / device here (f64 or f16 features, features of another device): index_copy_
        # itself accepts neither, and by now `plan` has already counted the rows
        picked = rows.index_select(0, idx[1]).to(device=self.store.device, dtype=self.store.dtype)
        self.store.view(-1, self.D).index_copy_(0, idx[0].to(self.store.device), picked)


def fill_plan(fill, cb, per_batch, batch_size, k, n_rows, phantom):
    """Where the reference's batch walk puts the rows sampled for one image, for ALL classes at once (host integer arrays).

    The walk, per class (box_head_getProposals.py:226-290, rpn_getProposals.py:283-331): from the class's current batch on,
    a full batch (fill >= batch_size) is skipped and counted (`current_batch += 1`), an open one receives
    min(per_batch, its room, k - taken[, n - taken]) rows, and the walk ends when k rows are taken.  `phantom` (the detector's
    form): a class with n < k rows still advances `taken` by the full amounts — its rows simply run out (neg[taken:end]
    is empty past n) —; without it (the on-line RPN's form) the amounts are limited by n and the walk, never reaching k,
    visits every batch.
    fill (nc, B): rows in every batch of the nc classes; cb (nc,): their current batch; n_rows (nc,): rows sampled per class.
    Returns take (nc, B): rows that really land in each batch; lo (nc, B): their offset inside the class's sampled rows;
    skipped (nc,): full batches the walk passed."""
    nc, B = fill.shape
    bidx = np.arange(B, dtype=np.int64)[None, :]
    active = bidx >= cb[:, None]
    room = batch_size - fill
    full = active & (room <= 0)
    want = np.where(active & (room > 0), np.minimum(per_batch, room), 0)
    limit = np.full(nc, k, dtype=np.int64) if phantom else np.minimum(k, n_rows)
    cum = np.cumsum(want, axis=1)
    before = cum - want
    amount = np.clip(limit[:, None] - before, 0, want)                 # what the walk counts as taken in each batch
    reaches = (np.minimum(cum, limit[:, None]) >= k) & (want > 0)      # the batch at which `taken == k` ends the walk
    stop = np.where(reaches.any(axis=1), reaches.argmax(axis=1), B - 1)
    skipped = (full & (bidx <= stop[:, None])).sum(axis=1)
    lo = np.minimum(before, n_rows[:, None])
    hi = np.minimum(before + amount, n_rows[:, None])
    return hi - lo, lo, skipped


def clamp_boxes_(b, img_size):
    b[:, 0].clamp_(0, img_size[0] - 1)
    b[:, 2].clamp_(0, img_size[0] - 1)
    b[:, 1].clamp_(0, img_size[1] - 1)
    b[:, 3].clamp_(0, img_size[1] - 1)
    return b


class DetectorHarvester:
    def __init__(self, feat_dim, num_classes, iterations, batch_size, num_images, neg_iou_thresh=0.3,
                 reg_min_overlap=0.6, compute_gt_positives=True, shuffle_negatives=False, device=None):
        self.D = feat_dim
        self.iterations = iterations
        self.batch_size = batch_size
        self.num_images = num_images
        self.neg_iou_thresh = neg_iou_thresh
        self.reg_min_overlap = reg_min_overlap
        self.compute_gt_positives = compute_gt_positives
        self.shuffle_negatives = shuffle_negatives
        self.device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        self.num_classes = 0
        self._pos, self._neg, self.current_batch = [], [], []
        self.still_to_complete = []
        self._grid = None if shuffle_negatives else _SlotGrid(feat_dim, iterations, batch_size, self.device)
        if self._grid is not None:
            self._grid.reserve(num_classes)
        for _ in range(num_classes):
            self.add_new_class()
        self.negatives_to_pick = None
        self._X = _Growing(feat_dim, self.device, mark_every=batch_size)
        self._Y = _Growing(4, self.device)
        self._C = _Growing(1, self.device)
        self.O = None
        self.test_boxes = []

    def add_new_class(self):
        """Incremental use (box_head_getProposals.py:90-99): one more class with empty batches."""
        self.still_to_complete.append(self.num_classes)
        self.num_classes += 1
        self._pos.append(_Growing(self.D, self.device, mark_every=self.batch_size))
        if self.shuffle_negatives:
            self._neg.append([_Growing(self.D, self.device, cap=self.batch_size)])
        else:
            self._neg.append(self._grid.add_class())
        self.current_batch.append(0)

    # ------------------------------------------------------------------ train time
    def add_image(self, x, proposals, gt_bbox, gt_labels_list, img_size):
        """x (R, D) pooled features of `proposals` (R, 4) whose first len(gt) rows are the ground
        truth boxes; gt_bbox (G, 4); gt_labels_list: class ids 1..C of the G boxes.
        = commit(prepare(...)) with the image's own host read; a caller that harvests several images at a time (the feature
        extractor's group loop) prepares them all, reads every harvester's counts in ONE copy and commits image by image."""
        ctx = self.prepare(x, proposals, gt_bbox, gt_labels_list, img_size)
        self.commit(ctx, ctx["block"].tolist() if ctx["block"] is not None else [])

    def prepare(self, x, proposals, gt_bbox, gt_labels_list, img_size):
        """Everything of add_image that runs on the device BEFORE the host has to decide anything, and nothing that reads or
        changes this object's state: overlaps, associations, the regression pairs' flags, the negatives' candidate flags of
        every class present in the image.  ctx["block"]: the counts the host needs (device, int64) or None."""
        if self.negatives_to_pick is None:
            self.negatives_to_pick = math.ceil((self.batch_size * self.iterations) / self.num_images)
        x = x.reshape(x.size(0), -1)
        be = _label_backend(x)
        if be is not None and 0 < gt_bbox.shape[0] <= be.LABEL_MAX_GT and all(1 <= l <= self.num_classes for l in gt_labels_list):
            # the same labels from ONE launch (odx_det_label_f32: clamping, overlaps, per-class maxima, first-maximum association,
            # the pairs' and the candidates' flags and their counts) instead of ~45 tensor operations; one small upload
            G, R = int(gt_bbox.shape[0]), int(proposals.shape[0])
            in_image = sorted({l - 1 for l in gt_labels_list})
            up = to_device(torch.tensor([l - 1 for l in gt_labels_list] + in_image, dtype=torch.int32), x.device)
            prop_d, overlap, sel, cmask, counters = be.det_label(gt_bbox, up[:G], proposals, self.num_classes, img_size, self.reg_min_overlap,
                                                                 self.neg_iou_thresh, up[G:])
            return {"x": x, "overlap": overlap, "labels": list(gt_labels_list), "R": R, "G": G, "block": counters, "prop_d": prop_d,
                    "cls": up[:G].to(torch.int64), "sel": sel, "in_image": in_image, "cmask": cmask}
        prop = clamp_boxes_(proposals.clone().float(), img_size)
        gt = clamp_boxes_(gt_bbox.clone().float(), img_size)
        R, G = prop.shape[0], gt.shape[0]
        overlap = torch.zeros((R, self.num_classes), dtype=torch.float32, device=x.device)
        assoc = torch.full((R,), -1, dtype=torch.int64, device=x.device)
        if G:
            iou = box_iou_plus1(gt.to(x.device), prop.to(x.device))                     # (G, R)
            for j in range(G):                                                           # per-class maxima
                c = gt_labels_list[j] - 1
                overlap[:, c] = torch.max(overlap[:, c], iou[j])
            # gt with the largest IoU per proposal; strict '>' in the reference keeps the FIRST maximum
            best, arg = iou.max(dim=0)
            first = (iou == best[None, :]).float().argmax(dim=0)
            assoc = torch.where(best > 0, first, assoc)
        ctx = {"x": x, "overlap": overlap, "labels": list(gt_labels_list), "R": R, "G": G, "block": None}
        if G:
            # All ground-truth boxes at once (the reference walks them one by one, box_head_getProposals.py:151-226):
            # sel[j, r] = proposal r regresses onto box j; the pairs are listed box-major, row-minor — the order of the
            # reference's appends — by a stable sort of the flags (commit), without nonzero()'s synchronisation.
            cls = to_device(torch.tensor([l - 1 for l in gt_labels_list], dtype=torch.int64), x.device)
            sel = (overlap[:, cls].t() > self.reg_min_overlap) & (assoc[None, :] == torch.arange(G, device=x.device)[:, None])
            # ONE host read per image for this harvester: the rows per ground-truth box and the negatives' candidate counts of
            # the classes in the image (which of them still collect negatives is this object's state: commit picks the columns)
            in_image = sorted({l - 1 for l in gt_labels_list if 0 <= l - 1 < self.num_classes})
            cmask = (overlap[:, in_image] < self.neg_iou_thresh) if in_image else None                   # (R, len(in_image))
            ctx.update(prop_d=prop.to(x.device), cls=cls, sel=sel, in_image=in_image, cmask=cmask,
                       block=torch.cat((sel.sum(1), cmask.sum(0))) if in_image else sel.sum(1))
        return ctx

    def commit(self, ctx, both):
        """The stateful rest of add_image: `both` = ctx["block"] on the host."""
        x, overlap, gt_labels_list, R, G = ctx["x"], ctx["overlap"], ctx["labels"], ctx["R"], ctx["G"]
        if G:
            prop_d, cls, sel = ctx["prop_d"], ctx["cls"], ctx["sel"]
            if self.compute_gt_positives:
                for c in sorted(set(gt_labels_list)):                      # rows of a class in ground-truth order
                    rows_c = [i for i, l in enumerate(gt_labels_list) if l == c]
                    self._pos[c - 1].append(x[rows_c].view(-1, self.D), seg_lens=[1] * len(rows_c))
            classes_neg = list(range(self.num_classes)) if self.shuffle_negatives else list(self.still_to_complete)
            present = [i for i in classes_neg if i + 1 in gt_labels_list]
            seg, call = both[:G], both[G:]
            cols = [ctx["in_image"].index(i) for i in present]
            if cols == list(range(len(ctx["in_image"]))):
                cmask, ccounts = ctx["cmask"], call
            else:
                cmask = ctx["cmask"][:, cols] if cols else None
                ccounts = [call[c] for c in cols]
            flat = torch.argsort((~sel).reshape(-1).to(torch.int8), stable=True)[:sum(seg)]
            j_idx, r_idx = flat // R, flat % R
            ex, tgt = prop_d[r_idx], prop_d[j_idx]                          # tgt: the prepended ground-truth rows
            self._Y.append(_box_targets(ex, tgt), seg_lens=seg)
            self._C.append((cls[j_idx] + 1).to(torch.float32).view(-1, 1), seg_lens=seg)
            self._X.append(x[r_idx].view(-1, self.D), seg_lens=seg)
        else:
            present, cmask, ccounts = [], None, []
        if not self.shuffle_negatives:
            self._fill_batches(x, overlap, gt_labels_list, (present, cmask, ccounts))
        else:
            classes = list(range(self.num_classes))
            feats_all, lens = self._sample_all(x, overlap, classes, gt_labels_list, (present, cmask, ccounts))
            at = 0
            for i, n_i in zip(classes, lens):
                last = self._neg[i][-1]
                last.append(feats_all[at:at + n_i])
                at += n_i
                if last.n >= self.batch_size:
                    self._neg[i].append(_Growing(self.D, self.device, cap=self.batch_size))

    def _sample_all(self, x, overlap, classes, gt_labels_list, known=None):
        """The negatives of one image for every class of `classes`, in that order, with one gather
        (box_head_getProposals.py:213-222 per class): returns the sampled rows of
        all classes back to back and the number of rows of each.  The draws come from the global RNG class by class as
        in the reference (one randint per class: over all R rows for a class that is not in the image, over its
        candidates — rows overlapping its boxes by less than NEG_IOU_THRESH — for a class that is, none when it has no
        candidate).  known = (present classes, their candidate flags (R, n), their counts) when add_image has already brought
        the counts to the host with its own read; the candidates of a class are listed by a stable sort of its flags (ascending
        row index: what nonzero() gives, without a synchronisation per class), and every draw goes to the device in ONE copy."""
        k = self.negatives_to_pick
        present = [i for i in classes if i + 1 in gt_labels_list]
        if known is not None and known[0] == present:
            masks, clist = known[1], known[2]
        else:
            masks = overlap[:, present] < self.neg_iou_thresh if present else None               # (R, len(present))
            clist = masks.sum(0).tolist() if present else []
        counts = dict(zip(present, (int(c) for c in clist)))
        col = {i: j for j, i in enumerate(present)}
        cand_order = torch.argsort((~masks).to(torch.int8), dim=0, stable=True) if present else None    # column j: class present[j]'s candidates first
        draws, plan, lens = [], [], []
        for i in classes:
            if i in counts:
                if counts[i] > 0:
                    plan.append(("dev", i, len(draws)))
                    draws.append(torch.randint(counts[i], (k,)))
                    lens.append(k)
                else:
                    lens.append(0)
            else:
                plan.append(("host", i, len(draws)))
                draws.append(torch.randint(x.size(0), (k,)))
                lens.append(k)
        if not plan:
            return torch.empty((0, self.D), dtype=x.dtype, device=x.device), lens
        # one copy for all classes' draws (+ per draw the column of its class's candidate list, -1 for a class that is not in the
        # image), and the rows of ALL classes from a handful of launches: sel[:, p] = the candidate order of draw p's class,
        # gathered at the draws — what `cand_order[:, col[i]][up[d]]` class by class gave (two launches per class before)
        colv = torch.tensor([col[i] if kind == "dev" else -1 for kind, i, _ in plan], dtype=torch.int64)
        up = to_device(torch.cat((torch.stack(draws), colv[:, None]), dim=1), x.device)
        dr, cv = up[:, :k], up[:, k]
        if cand_order is None or not any(kind == "dev" for kind, _, _ in plan):
            idx = dr.reshape(-1)
        else:
            sel = cand_order.index_select(1, cv.clamp_min(0))                                  # (R, draws)
            idx = torch.where(cv[:, None] >= 0, torch.gather(sel, 0, dr.t()).t(), dr).reshape(-1)
        return x[idx].view(-1, self.D), lens

    def _fill_batches(self, x, overlap, gt_labels_list, known=None):
        classes = list(self.still_to_complete)
        feats_all, lens = self._sample_all(x, overlap, classes, gt_labels_list, known)
        if classes:
            # where the reference would append every class's rows (box_head_getProposals.py:226-290), as integer bookkeeping on
            # the host — all classes by one fill_plan (a Python walk over every (class, open batch) before: several hundred
            # steps per image); the rows themselves move once, for all classes, in _SlotGrid.commit
            cls = np.asarray(classes, dtype=np.int64)
            n_rows = np.asarray(lens, dtype=np.int64)
            cb = np.asarray([self.current_batch[i] for i in classes], dtype=np.int64)
            per_batch = math.ceil(self.negatives_to_pick / self.iterations)
            take, lo, skipped = fill_plan(self._grid.fill[cls], cb, per_batch, self.batch_size, self.negatives_to_pick, n_rows, phantom=True)
            at = np.cumsum(n_rows) - n_rows
            self._grid.plan_rows(cls, take, at[:, None] + lo)
            for i, sk in zip(classes, skipped.tolist()):
                if sk:
                    self.current_batch[i] += sk
                    if self.current_batch[i] >= self.iterations:
                        self.still_to_complete.remove(i)
        self._grid.commit(feats_all)

    # ------------------------------------------------------------------ test time
    def add_test_image(self, x, proposals, gt_label_count, img_size):
        """forward_test (:295-334): store boxes / features / gt flags for stand-alone scoring."""
        x = x.reshape(x.size(0), -1)
        prop = clamp_boxes_(proposals.clone().float(), img_size)
        gt = np.zeros((prop.shape[0], 1), dtype=bool)
        gt[:gt_label_count] = True
        self.test_boxes.append({'boxes': prop.cpu().numpy(), 'feat': x.cpu().numpy(), 'gt': gt, 'img_size': np.array(img_size)})

    # ------------------------------------------------------------------ results
    def finalize(self, use_only_gt_positives=True):
        """negatives, positives, COXY as returned by FeatureExtractorDetector.train
        (extract_features_detector.py:255-292)."""
        COXY = {'C': self._C.view().clone(), 'O': self.O, 'X': self._X.view().clone(), 'Y': self._Y.view().clone()}
        positives = [p.view().clone() for p in self._pos] if (use_only_gt_positives and self.compute_gt_positives) else None
        negatives = []
        for i in range(self.num_classes):
            if self.shuffle_negatives:
                total = torch.cat([g.view() for g in self._neg[i]])
                perm = torch.randperm(len(total))
                bs = self.batch_size
                negatives.append([total[perm[min(j * bs, len(perm)):min((j + 1) * bs, len(perm))]] for j in range(self.iterations)])
            else:
                negatives.append([g.view().clone() for g in self._neg[i]])
        return negatives, positives, COXY


class RPNHarvester:
    """Train-time harvesting for the on-line RPN (A12): one FALKON problem per anchor type.
    Behaviour follows RPNModule.forward (mrcnn_modified/modeling/rpn/rpn_getProposals.py:180-463):
      * anchors (location-major, type-minor) that cross the image border are dropped for good
        (:224-225); anchor k sits at feature cell ((k // A) // W, (k // A) % W) and belongs to
        classifier k % A (:211-215);
      * per image, IoU (+1 convention) of every kept anchor with every ground-truth box; each anchor
        is associated with its best ground truth (:262-271);
      * negatives of type i: anchors of that type with IoU < NEG_IOU_THRESH, a with-replacement
        sample of `negatives_to_pick` when there are more, spread over the still-open batches
        (:274-331) — or appended to one growing list when SHUFFLE_NEGATIVES (:333-362);
      * positives: anchors with IoU > POS_IOU_THRESH, plus, for a ground truth none of whose
        coordinates appears among the positives' associated boxes (the reference's tensor `in`
        test, :369), the anchors associated with it that reach its maximum IoU (:366-381);
      * features are the (D,) columns of the RPN activation at the anchors' cells; positives also
        yield regressor rows X / Y (box targets against the associated ground truth) / C (anchor type).
    The reference gathers each row through an index_select square + diagonal pick (:316-321); here it
    is one advanced-indexing gather.
    """

    def __init__(self, feat_dim, num_classes, iterations, batch_size, num_images, neg_iou_thresh=0.3,
                 pos_iou_thresh=0.7, shuffle_negatives=False, device=None):
        self.D, self.A = feat_dim, num_classes
        self.iterations, self.batch_size, self.num_images = iterations, batch_size, num_images
        self.neg_iou_thresh, self.pos_iou_thresh = neg_iou_thresh, pos_iou_thresh
        self.shuffle_negatives = shuffle_negatives
        self.device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        self.negatives_to_pick = None
        self.anchors = None
        self._pos = [_Growing(feat_dim, self.device, mark_every=batch_size) for _ in range(num_classes)]
        if shuffle_negatives:
            self._grid = None
            self._neg = [[_Growing(feat_dim, self.device, cap=batch_size)] for _ in range(num_classes)]
        else:
            self._grid = _SlotGrid(feat_dim, iterations, batch_size, self.device)
            self._grid.reserve(num_classes)
            self._neg = [self._grid.add_class() for _ in range(num_classes)]
        self.current_batch = [0] * num_classes
        self._X, self._Y, self._C = _Growing(feat_dim, self.device, mark_every=batch_size), _Growing(4, self.device), _Growing(1, self.device)
        self.O = None

    def _setup(self, anchors_all, img_size, W):
        n = anchors_all.shape[0]
        k = torch.arange(n, device=anchors_all.device)
        loc = k // self.A
        vis = (anchors_all[:, 0] >= 0) & (anchors_all[:, 1] >= 0) & (anchors_all[:, 2] < img_size[0]) & (anchors_all[:, 3] < img_size[1])
        self.anchors = anchors_all[vis]
        self.rows, self.cols = (loc // W)[vis], (loc % W)[vis]
        self.cls = (k % self.A)[vis]
        # Anchor types without a visible anchor are dropped the way the reference does it — by
        # removing from the list it is iterating (:227-229), which skips the element after every
        # removal: of a run of empty types only every other one is dropped.  Kept on purpose.
        self.still_to_complete = list(range(self.A))
        self.invisible = []
        for i in self.still_to_complete:
            if not bool((self.cls == i).any()):
                self.still_to_complete.remove(i)
                self.invisible.append(i)
        self.anchors_ids = list(self.still_to_complete)

    def _gather(self, t, sel):
        return t[:, self.rows[sel], self.cols[sel]].t().reshape(-1, self.D)

    def add_image(self, t, anchors_all, img_size, gt_bbox):
        """t (D, H, W) RPN activation of the image; anchors_all (H*W*A, 4) from grid_anchors;
        img_size (width, height); gt_bbox (G >= 1, 4) in the same pixel frame.

        ONE host read per image: everything the host decides — how many candidates each anchor type has, which ground-truth
        boxes get their best anchor added as a positive (the reference's `elem in tensor` walk, rpn_getProposals.py:383-400),
        how many positives each type ends up with — is decided from one small block of counts brought over together; the
        reference (and this class until round 4) read them piece by piece, a synchronisation for the candidates, one for the
        positives and two per ground-truth box, each of which waits for the harvest's kernels to get their turn beside the
        next images' forward."""
        ctx = self.prepare(t, anchors_all, img_size, gt_bbox)
        self.commit(ctx, ctx["block"].tolist())

    def prepare(self, t, anchors_all, img_size, gt_bbox):
        """The device work of add_image in front of its host read, independent of this object's batch state; ctx["block"]: what
        the host needs (device, f64).  A caller harvesting several images at a time reads the blocks of all of them — and of the
        other harvesters — in one copy and commits image by image (OnlineFeatureExtractor's group loop)."""
        if self.negatives_to_pick is None:
            self.negatives_to_pick = math.ceil((self.batch_size * self.iterations) / self.num_images)
        if self.anchors is None:
            self._setup(anchors_all.to(t.device), img_size, t.shape[2])
        dev = t.device
        A = self.A
        gt = gt_bbox.to(dev).float()
        G = gt.shape[0]
        be = _label_backend(t)
        if be is not None and 0 < G <= be.LABEL_MAX_GT:
            # the labels below from two launches (odx_rpn_label_f32) instead of ~45 tensor operations
            _, assoc, neg_mask, over, extra, cnt = be.rpn_label(gt, self.anchors, self.cls, A, self.neg_iou_thresh, self.pos_iou_thresh)
            block = torch.cat((cnt.to(torch.float64), gt.reshape(-1).to(torch.float64)))
            return {"t": t, "G": G, "block": block, "neg_mask": neg_mask, "over": over, "extra": extra, "assoc": assoc}
        iou_all = box_iou_plus1(gt, self.anchors)                                   # (G, n_vis)
        if G > 1:
            ious, idx = torch.max(iou_all, dim=0)
            assoc = gt[idx]
        else:
            ious = iou_all.reshape(-1)
            assoc = gt[0].expand(self.anchors.shape[0], 4)
        neg_mask = ious < self.neg_iou_thresh
        over = ious > self.pos_iou_thresh
        onehot = self.cls[:, None] == torch.arange(A, device=dev)[None, :]           # (n_vis, A)
        # per ground-truth box j (the reference walks them in order):
        #   mine_j   anchors associated with a box equal to box j in all four coordinates
        #   extra_j  those of them at their best IoU — what is appended when box j has no positive yet
        #   hit_j    how many anchors over the threshold are associated with a box sharing ANY coordinate with box j, position by
        #            position (the reference's `g in positive_gts` on tensors: any equal element)
        mine = (assoc[None, :, :] == gt[:, None, :]).all(dim=2)                       # (G, n_vis)
        best = torch.where(mine, ious[None, :], torch.full_like(ious, -1.0)[None, :]).max(dim=1)[0]
        extra = mine & (ious[None, :] == best[:, None])
        share = (assoc[None, :, :] == gt[:, None, :]).any(dim=2)                      # (G, n_vis)
        f64 = torch.float64
        block = torch.cat(((neg_mask[:, None] & onehot).sum(0).to(f64),                # A      candidates per type
                           (share & over[None, :]).sum(1).to(f64),                    # G      hit_j
                           mine.any(dim=1).to(f64),                                   # G      box j has anchors at all
                           (over[:, None] & onehot).sum(0).to(f64),                   # A      positives over the threshold per type
                           (extra[:, :, None] & onehot[None, :, :]).sum(1).reshape(-1).to(f64),     # G A    extras of box j per type
                           gt.reshape(-1).to(f64)))                                   # 4 G    the boxes (for the coordinate test)
        return {"t": t, "G": G, "block": block, "neg_mask": neg_mask, "over": over, "extra": extra, "assoc": assoc}

    def commit(self, ctx, block):
        """The stateful rest of add_image: `block` = ctx["block"] on the host."""
        t, G, neg_mask, over, extra, assoc = ctx["t"], ctx["G"], ctx["neg_mask"], ctx["over"], ctx["extra"], ctx["assoc"]
        dev, A = t.device, self.A
        types = list(self.still_to_complete if not self.shuffle_negatives else range(A))
        counts = [int(v) for v in block[:A]]
        hit = block[A:A + G]
        has_mine = block[A + G:A + 2 * G]
        over_type = [int(v) for v in block[A + 2 * G:2 * A + 2 * G]]
        extra_type = [[int(v) for v in block[2 * A + 2 * G + j * A:2 * A + 2 * G + (j + 1) * A]] for j in range(G)]
        gth = [block[2 * A + 2 * G + G * A + 4 * j:2 * A + 2 * G + G * A + 4 * j + 4] for j in range(G)]
        # ---- negatives: the with-replacement draws type by type from the global RNG, the bookkeeping of the open batches
        order = torch.argsort(torch.where(neg_mask, self.cls, torch.full_like(self.cls, A)), stable=True)
        starts, acc = [], 0
        for c in counts:
            starts.append(acc)
            acc += c
        picks, lens = [], []
        for i in types:
            c = counts[i]
            p = torch.randint(c, (self.negatives_to_pick,)) if c > self.negatives_to_pick else torch.arange(c)
            picks.append(p + starts[i])
            lens.append(p.numel())
        # ---- positives: which boxes get their best anchors added (on the host, from the counts), in the reference's order
        rank_of = [0] * G                      # 0: not selected; k >= 1: its extras come k-th behind the over-threshold anchors
        chosen = []
        for j in range(G):
            if hit[j] > 0 or any(any(a == b for a, b in zip(gth[j], gth[j2])) for j2 in chosen):
                continue
            if has_mine[j] > 0:
                chosen.append(j)
                rank_of[j] = len(chosen)
        per_type = [over_type[a] + sum(extra_type[j][a] for j in chosen) for a in range(A)]
        n_pos = sum(per_type)
        n_neg = sum(lens)
        # one host -> device copy for everything the device needs back: the sampled candidates' positions, the boxes' ranks
        up = to_device(torch.cat(picks + [torch.tensor(rank_of, dtype=torch.int64)]), dev) if (picks or G) else None
        pick_idx, rank_dev = up[:n_neg], up[n_neg:]
        feats_all = self._gather(t, order[pick_idx]) if n_neg else torch.empty((0, self.D), dtype=t.dtype, device=dev)
        if self.shuffle_negatives:
            at = 0
            for i, n_i in zip(types, lens):
                last = self._neg[i][-1]
                last.append(feats_all[at:at + n_i])
                at += n_i
                if last.n >= self.batch_size:
                    self._neg[i].append(_Growing(self.D, self.device, cap=self.batch_size))
        else:
            if types:
                # the batch walk of every anchor type at once (fill_plan; rpn_getProposals.py:283-331: the amounts limited by the
                # rows a type really has); the rows move once for all types (commit below)
                tarr, n_rows = np.asarray(types, dtype=np.int64), np.asarray(lens, dtype=np.int64)
                cb = np.asarray([self.current_batch[i] for i in types], dtype=np.int64)
                per_batch = math.ceil(self.negatives_to_pick / self.iterations)
                take, lo, skipped = fill_plan(self._grid.fill[tarr], cb, per_batch, self.batch_size, self.negatives_to_pick, n_rows, phantom=False)
                at = np.cumsum(n_rows) - n_rows
                self._grid.plan_rows(tarr, take, at[:, None] + lo)
                for i, sk in zip(types, skipped.tolist()):
                    if sk:
                        self.current_batch[i] += sk
                        if self.current_batch[i] >= self.iterations:
                            self.still_to_complete.remove(i)
            self._grid.commit(feats_all)
        # The reference walks the anchor types that have positives in ascending order and, per type, appends its rows
        # (rpn_getProposals.py:383-449): over-threshold anchors first (ascending anchor index), then the added best anchors box
        # by box.  Here: ONE stable sort by (type, group) gives that order for all types at once, one gather fetches every
        # positive's features, and each buffer receives one copy.
        if G > 1:
            sel_extra = extra & (rank_dev > 0)[:, None]
            grp = (sel_extra.to(torch.int64) * rank_dev[:, None]).max(dim=0)[0]       # the rank of the box whose extra this anchor is
        else:
            grp = extra[0].to(torch.int64) * rank_dev[0]
        is_pos = over | (grp > 0)
        grp = torch.where(over, torch.zeros_like(grp), grp)
        key = torch.where(is_pos, self.cls * (G + 1) + grp, torch.full_like(self.cls, A * (G + 1)))
        sel = torch.argsort(key, stable=True)[:n_pos]
        pcls_sorted = self.cls[sel]
        feat = self._gather(t, sel)
        ex, tg = self.anchors[sel], assoc[sel]
        seg = [k for k in per_type if k]
        at = 0
        for i, k in enumerate(per_type):
            if k:
                self._pos[i].append(feat[at:at + k])
                at += k
        self._Y.append(_box_targets(ex, tg), seg_lens=seg)
        self._C.append(pcls_sorted.to(torch.float32).view(-1, 1), seg_lens=seg)
        self._X.append(feat, seg_lens=seg)

    def finalize(self):
        """negatives, positives, COXY as returned by FeatureExtractorRPN.train
        (feature_extractor_RPN/extract_features_RPN.py:218)."""
        COXY = {'C': self._C.view().clone(), 'O': self.O, 'X': self._X.view().clone(), 'Y': self._Y.view().clone()}
        positives = [p.view().clone() for p in self._pos]
        negatives = []
        for i in range(self.A):
            if self.shuffle_negatives:
                total = torch.cat([g.view() for g in self._neg[i]])
                perm = torch.randperm(len(total))
                bs = self.batch_size
                negatives.append([total[perm[min(j * bs, len(perm)):min((j + 1) * bs, len(perm))]] for j in range(self.iterations)])
            else:
                negatives.append([g.view().clone() for g in self._neg[i]])
        return negatives, positives, COXY


def project_masks_on_boxes(masks, boxes, M, boxes_host=None):
    """Crop each object's full-image binary mask to its box and resize to M x M
    (mask_head_getProposals.py:15-46 -> SegmentationMask.crop / resize of maskrcnn_benchmark's binary-mask
    representation: integer-rounded crop window, bilinear resize without corner alignment, truncated
    back to {0, 1}).  masks (G, H, W) bool/uint8, boxes (G, 4) xyxy -> (G, M, M) float32.
    All objects at once on the masks' device (the reference does it object by object on the CPU, :35): the crop
    windows come to the host in one read, the bilinear resize of every crop is evaluated as four gathers from the full
    masks with the source coordinates / weights of torch's upsample_bilinear2d (align_corners=False: src = scale (i + 0.5)
    - 0.5 clamped at 0, second tap clamped to the crop's last pixel).
    PARITY UNPINNED (the mask containers are maskrcnn_benchmark's, not vendored)."""
    G = masks.shape[0]
    if G == 0:
        return torch.empty(0, dtype=torch.float32, device=masks.device)
    H, W = masks.shape[1], masks.shape[2]
    win = []
    for bx in (boxes.tolist() if boxes_host is None else boxes_host):    # one host read for all objects (none when the caller has them)
        x1, y1, x2, y2 = [int(round(float(v))) for v in bx]
        x1, y1 = min(max(x1, 0), W - 1), min(max(y1, 0), H - 1)
        x2, y2 = min(max(x2, 0), W - 1), min(max(y2, 0), H - 1)
        win.append((x1, y1, max(x2, x1 + 1), max(y2, y1 + 1)))
    dev = masks.device
    wn = np.asarray(win, dtype=np.int64)                        # (G, 4)
    i = np.arange(M, dtype=np.float32)[None, :]

    def taps(first, size):
        """Source taps of one axis on the host, in f32 exactly as upsample_bilinear2d forms them (the scale is a host-side
        f32 quotient there too); returned as absolute pixel coordinates in the full mask."""
        size = size[:, None]
        scale = size.astype(np.float32) / np.float32(M)
        src = np.maximum(scale * (i + np.float32(0.5)) - np.float32(0.5), np.float32(0.0))
        i0 = np.minimum(np.floor(src).astype(np.int64), size - 1)
        i1 = np.minimum(i0 + 1, size - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return first[:, None] + i0, first[:, None] + i1, np.float32(1.0) - l1, l1

    xa, xb, wxa, wxb = taps(wn[:, 0], wn[:, 2] - wn[:, 0])       # (G, M) each
    ya, yb, wya, wyb = taps(wn[:, 1], wn[:, 3] - wn[:, 1])
    idx = to_device(torch.from_numpy(np.stack((xa, xb, ya, yb))), dev)
    wts = to_device(torch.from_numpy(np.stack((wxa, wxb, wya, wyb))), dev)
    xa, xb, ya, yb = idx[0], idx[1], idx[2], idx[3]
    wxa, wxb, wya, wyb = wts[0], wts[1], wts[2], wts[3]
    g = torch.arange(G, device=dev)[:, None, None]
    mf = masks

    def tap(yy, xx):
        return mf[g, yy[:, :, None], xx[:, None, :]].to(torch.float32)

    top = wxa[:, None, :] * tap(ya, xa) + wxb[:, None, :] * tap(ya, xb)
    bot = wxa[:, None, :] * tap(yb, xa) + wxb[:, None, :] * tap(yb, xb)
    r = wya[:, :, None] * top + wyb[:, :, None] * bot
    return (r + 1e-6).to(torch.uint8).float()                   # truncation, made robust to 0.99999994-style fuzz


class MaskHarvester:
    """Train-time harvesting for the on-line segmentation head (A13): every pixel of the S x S mask
    activation of a ground-truth RoI is a D-dimensional row; pixels whose projected ground-truth mask
    is >= 0.5 are positives of the object's class, the others negatives of that same class, each
    sub-sampled to SAMPLING_FACTOR with a fresh randperm (ROIMaskHead.forward,
    mrcnn_modified/modeling/roi_heads/mask_head/mask_head_getProposals.py:83-143)."""

    def __init__(self, feat_dim, num_classes, batch_size=20000, sampling_factor=0.3, device=None):
        self.D, self.batch_size, self.sampling_factor = feat_dim, batch_size, sampling_factor
        self.device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        self.num_classes = 0
        self._pos, self._neg = [], []
        for _ in range(num_classes):
            self.add_new_class()

    def add_new_class(self):
        self.num_classes += 1
        self._pos.append(_Growing(self.D, self.device, cap=1024, mark_every=self.batch_size))
        self._neg.append(_Growing(self.D, self.device, cap=1024, mark_every=self.batch_size))

    def add_image(self, mask_features, masks_gt, gt_labels_list):
        """mask_features (G, D, S, S) = relu(conv5_mask(head features of the ground-truth RoIs));
        masks_gt (G, S, S) from project_masks_on_boxes; gt_labels_list class ids 1..C.
        All objects of the image at once: per object the pixels are ordered positives-first by a stable sort (ascending
        pixel index inside each group, as torch.where lists them), the sizes come to the host in one read, the random
        sub-sampling draws are made object by object, positives then negatives, from the global RNG as in the reference,
        and every class buffer receives one copy."""
        ctx = self.prepare(mask_features, masks_gt, gt_labels_list)
        if ctx is not None:
            self.commit(ctx, ctx["block"].tolist())                               # the one host read

    def prepare(self, mask_features, masks_gt, gt_labels_list):
        """The device work in front of add_image's host read (stateless); None for an image without objects."""
        G = len(mask_features)
        if G == 0:
            return None
        D, S2 = mask_features.size(1), mask_features.size(2) * mask_features.size(3)
        dev = mask_features.device
        rows = mask_features.permute(0, 2, 3, 1).reshape(G * S2, D)
        is_pos = masks_gt.reshape(G, S2).to(dev) >= 0.5
        order = torch.argsort((~is_pos).to(torch.int8), dim=1, stable=True)        # positives first, each group ascending
        return {"G": G, "S2": S2, "rows": rows, "order": order, "labels": list(gt_labels_list), "block": is_pos.sum(1)}

    def commit(self, ctx, npos):
        """The draws and the copies of add_image: npos = ctx["block"] on the host."""
        G, S2, rows, order, gt_labels_list = ctx["G"], ctx["S2"], ctx["rows"], ctx["order"], ctx["labels"]
        dev = rows.device
        npos = [int(v) for v in npos]
        picks = {}                                                                # (class, kind) -> [(object, tensor of local picks)]
        for i in range(G):
            c = gt_labels_list[i] - 1
            for kind, start, cnt in ((0, 0, npos[i]), (1, npos[i], S2 - npos[i])):
                if self.sampling_factor < 1.0:
                    p = torch.randperm(cnt)[:int(self.sampling_factor * cnt)]
                else:
                    p = torch.arange(cnt)
                picks.setdefault((c, kind), []).append((i, p + start))
        # ONE upload and ONE gather for all (class, kind) groups of the image (a cat + full + stack + upload + two gathers per
        # group before): the picks are laid out group after group, every class buffer then receives its slice
        groups = list(picks.items())
        obj = np.concatenate([np.full(len(p), i, dtype=np.int64) for _, lst in groups for i, p in lst])
        loc = torch.cat([p for _, lst in groups for _, p in lst])
        idx = to_device(torch.stack((torch.from_numpy(obj), loc)), dev)
        sel = order[idx[0], idx[1]] + idx[0] * S2
        picked = rows.index_select(0, sel)
        at = 0
        for (c, kind), lst in groups:
            k = sum(len(p) for _, p in lst)
            (self._pos if kind == 0 else self._neg)[c].append(picked[at:at + k], seg_lens=[len(p) for _, p in lst])
            at += k

    def finalize(self):
        """negatives, positives as tensors per class (extract_features_detector.py:273-278)."""
        return [g.view().clone() for g in self._neg], [g.view().clone() for g in self._pos]
