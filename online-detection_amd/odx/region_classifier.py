"""On-line region classifier: the Minibootstrap hard-negative mining loop around a classifier
plug-in, z-score normalisation and stand-alone scoring.  Behaviour follows
src/modules/region-classifier/OnlineRegionClassifier.py:19-227 (host tensors) and
OnlineRegionClassifier_incore.py:16-224 (device tensors); the two differ in where tensors
live, in that the in-core variant skips the easy-negative pruning after the last batch
(_incore.py:130) and in the `return_caches` / `normalized` options (_incore.py:79-82,158-183).

State machine per class i with positives P and negative batches N_0..N_{B-1}
(thresholds from the YAML: HARD_THRESH -0.7, EASY_THRESH -0.9):
    cache = (P, N_0);  model = train(cache)
    for j >= 1:  cache.neg += N_j[ predict(model, N_j) > HARD ];  model = train(cache)
    after each train (in-core: except the last): cache.neg = cache.neg[ predict(model, cache.neg) >= EASY ]
A class without positives or without negatives gets model None.
"""
import os
import time

import numpy as np
import torch
import yaml

from . import backend as _backend
from .boxlist import get_boxlist_class


def _device():
    return 'cuda' if torch.cuda.is_available() else 'cpu'


def _randint_advances_one_draw_per_value(g, shape):
    """Whether `torch.randint(bound, (size,))` leaves a CPU generator where `torch.randint(2, (size,))` leaves it, for the
    (positive bound, positive draws, negative bound, negative draws) a fit of the stock index rule makes: the property
    `_reference_stream_positions` predicts every class's position in the global stream with (true for bounds below 2^24
    on the torch builds seen; undocumented, hence asked of the running build).  Works on clones; `g` is not advanced."""
    pos_bound, pos_draw, neg_bound, room = shape
    a, b = torch.Generator(), torch.Generator()
    a.set_state(g.get_state())
    b.set_state(g.get_state())
    if pos_draw:
        torch.randint(int(pos_bound), (int(pos_draw),), generator=a)
    torch.randint(int(neg_bound), (int(room),), generator=a)
    torch.randint(2, (int(pos_draw) + int(room),), generator=b)
    return bool(torch.equal(a.get_state(), b.get_state()))


_STREAMS = {}


def _work_streams(k, tag="class", reserve=2):
    """k side streams for the classes of a round, chosen once per (device, calling stream) among streams that share a
    hardware queue neither with the calling stream nor with each other (odx/streams.py) — the first `reserve` of those are
    left to fit_batch's half chains (none in the streams mode, whose fits run whole on their class's stream).  With the
    runtime's 4 queues that leaves ONE for the classes' builds and predictions in the batched mode (their kernels are wide
    enough to fill the chip one after the other), more with GPU_MAX_HW_QUEUES raised."""
    from . import streams as _streams
    own = _streams.distinct(reserve + k)
    if len(own) > reserve:
        return _streams.spread(own[reserve:], k)
    key = (torch.cuda.current_device(), tag)          # (no queue to spare: plain streams, as before)
    have = _STREAMS.setdefault(key, [])
    while len(have) < k:
        have.append(torch.cuda.Stream())
    return have[:k]


class OnlineRegionClassifierBase:
    incore = True

    def __init__(self, classifier, positives, negatives, stats=None, cfg_path=None, is_rpn=False,
                 is_segmentation=False):
        if cfg_path is not None:
            with open(cfg_path) as fid:
                self.cfg = yaml.load(fid, Loader=yaml.FullLoader)
            if is_rpn:
                self.cfg = self.cfg['RPN']
            section = self.cfg['ONLINE_SEGMENTATION' if is_segmentation else 'ONLINE_REGION_CLASSIFIER']
            self.classifier_options = section['CLASSIFIER']
            self.lam = section['CLASSIFIER']['lambda']
            self.sigma = section['CLASSIFIER']['sigma']
            self.hard_tresh = section['MINIBOOTSTRAP']['HARD_THRESH']
            self.easy_tresh = section['MINIBOOTSTRAP']['EASY_THRESH']
            self.mean = 0
            self.std = 0
            self.mean_norm = 0
            self.is_rpn = is_rpn
        else:
            print('Config file path not given. cfg variable set to None.')
            self.cfg = None
        self.classifier = classifier
        self.negatives = negatives
        self.positives = positives
        self.num_classes = len(self.cfg['CHOSEN_CLASSES'])
        if is_rpn:
            self.num_classes += 1
        if stats:
            self.stats = stats
            self.mean = stats['mean']
            self.std = stats['std']
            self.mean_norm = stats['mean_norm']
        self.normalized = False
        self.is_segmentation = is_segmentation
        self.return_caches = False
        self.class_streams = 0          # > 0: classes trained concurrently on that many streams (opts['class_streams'])
        self.class_batch = 0            # > 0: the classes of a round fitted by one batched call, on that many streams (opts['class_batch'])
        # the experiment drivers call trainRegionClassifier() without options: odx.options class_batch / class_streams /
        # class_shard / reference_order (environment: ODX_CLASS_BATCH, ...) select the same opt-ins without touching a driver
        # (opts still win)
        from . import options as _options
        o = _options.load()
        self.class_batch = int(o.class_batch)
        self.class_streams = int(o.class_streams)
        # 'auto': the reference's order of draws with the classes advancing together when that can be predicted (GPU,
        # stock index rule); 'sequential': always the class-by-class loop (opts['reference_order'], ODX_REFERENCE_ORDER)
        self.reference_order = o.reference_order
        self.class_rng = False          # one RNG stream per class for the Nystroem draws (opts['class_rng']; implied by the three modes above/below)
        self.class_shard = bool(o.class_shard)   # classes round-robin over the ranks of torch.distributed, models gathered at the end (opts['class_shard'])

    def loadRegionClassifier(self) -> None:
        pass

    def processOptions(self, opts):
        for key, attr in (('num_classes', 'num_classes'), ('imset_train', 'train_imset'),
                          ('classifier_options', 'classifier_options'), ('is_rpn', 'is_rpn'), ('lam', 'lam'),
                          ('sigma', 'sigma')):
            if key in opts:
                setattr(self, attr, opts[key])
        if 'class_rng' in opts:
            self.class_rng = bool(opts['class_rng'])
        if 'reference_order' in opts:
            self.reference_order = str(opts['reference_order'])
        if 'class_shard' in opts:
            self.class_shard = bool(opts['class_shard'])
        if self.incore:
            if 'return_caches' in opts:
                self.return_caches = opts['return_caches']
            if 'normalized' in opts:
                self.normalized = opts['normalized']
            if 'class_streams' in opts:
                self.class_streams = int(opts['class_streams'])
            if 'class_batch' in opts:
                self.class_batch = int(opts['class_batch'])

    def updateModel(self, cache):
        X_neg, X_pos = cache['neg'], cache['pos']
        X = torch.cat((X_pos, X_neg), 0)
        dev = X.device if self.incore else 'cpu'
        y = torch.cat((torch.ones(len(X_pos), device=dev), -torch.ones(len(X_neg), device=dev)), 0)
        if self.sigma is not None and self.lam is not None:
            print('Updating model with lambda: {} and sigma: {}'.format(self.lam, self.sigma))
            return self.classifier.train(X, y, sigma=self.sigma, lam=self.lam)
        print('Updating model with default lambda and sigma')
        return self.classifier.train(X, y)

    def _host(self, t):
        return t if self.incore else t.cpu()

    # ---- classes over ranks ("reference-regime minibootstrap": independent units, no data-path collective) --
    def _ranks(self):
        """(rank, world) of the class sharding: (0, 1) unless opts['class_shard'] and an initialised process group."""
        import torch.distributed as dist
        if self.class_shard and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _owned(self, i):
        rank, world = self._ranks()
        return i % world == rank

    def _class_seed(self):
        """One draw from the global RNG (class c's own stream is seeded seed + c); rank 0's draw on every rank."""
        seed = torch.randint(2 ** 62, (1,))
        if self._ranks()[1] > 1:
            import torch.distributed as dist
            comm = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
            seed = seed.to(comm)
            dist.broadcast(seed, src=0)
        return int(seed.item())

    def _gather_models(self, model):
        """Every rank ends with every class's model: class i's (ny_points_, alpha_) broadcast by its owner i % world.
        Two small collectives per class after the training — none during it."""
        rank, world = self._ranks()
        if world == 1:
            return model
        import torch.distributed as dist
        comm = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        C = len(model)
        shapes = torch.zeros((C, 2), dtype=torch.int64)
        for i, m in enumerate(model):
            if m is not None and i % world == rank:
                shapes[i, 0], shapes[i, 1] = m.ny_points_.shape
        shapes = shapes.to(comm)
        dist.all_reduce(shapes)
        shapes = shapes.tolist()
        for i in range(C):
            M, D = shapes[i]
            if M == 0:
                continue
            owner = i % world
            if owner == rank:
                where = model[i].ny_points_.device
                ny = model[i].ny_points_.to(comm, torch.float32).contiguous()
                alpha = model[i].alpha_.to(comm, torch.float64).contiguous()
            else:
                where = _device() if self.incore else 'cpu'
                ny = torch.empty((M, D), dtype=torch.float32, device=comm)
                alpha = torch.empty((M, 1), dtype=torch.float64, device=comm)
            dist.broadcast(ny, src=owner)
            dist.broadcast(alpha, src=owner)
            if owner != rank:
                if not hasattr(self.classifier, 'model_from_tensors'):
                    raise RuntimeError("opts['class_shard'] needs a classifier with model_from_tensors (odx FALKONWrapper)")
                model[i] = self.classifier.model_from_tensors(ny.to(where), alpha.to(where), sigma=self.sigma, lam=self.lam)
        return model

    def trainWithMinibootstrap(self, negatives, positives, output_dir=None):
        caches, model = [], []
        t_start = time.time()
        # per-class RNG streams (opts['class_rng'], implied by opts['class_shard']): class c draws its Nystroem centres
        # from its own generator seeded from ONE draw of the global stream, so a class's model does not depend on which
        # other classes this process trains; default: the reference's class-major use of the global stream
        rng = None
        if self.class_rng or self._ranks()[1] > 1:
            seed0 = self._class_seed()
            rng = {i: torch.Generator().manual_seed(seed0 + i).get_state() for i in range(self.num_classes - 1)}
        for i in range(self.num_classes - 1):
            if len(positives[i]) == 0 or len(negatives[i]) == 0 or not self._owned(i):
                model.append(None)
                caches.append({})
                continue
            print('---------------------- Training Class number {} ----------------------'.format(i))
            nb = len(negatives[i])
            for j in range(nb):
                t_iter = time.time()
                last = j == nb - 1
                if j == 0:
                    caches.append({'pos': self._host(positives[i]), 'neg': self._host(negatives[i][0])})
                    model.append(None)
                else:
                    t_hard = time.time()
                    batch = self._host(negatives[i][j])
                    scores = self.classifier.predict(model[i], batch)
                    hard_idx = torch.where(scores > self.hard_tresh)[0]
                    caches[i]['neg'] = torch.cat((caches[i]['neg'], batch[hard_idx]), 0)
                    print('Hard negatives selected in {} seconds'.format(time.time() - t_hard))
                    print('Chosen {} hard negatives from the {}th batch'.format(len(hard_idx), j))
                print('Traning with {} positives and {} negatives'.format(len(caches[i]['pos']), len(caches[i]['neg'])))
                t_update = time.time()
                if rng is None:
                    model[i] = self.updateModel(caches[i])
                else:
                    with torch.random.fork_rng(devices=[]):
                        torch.set_rng_state(rng[i])
                        model[i] = self.updateModel(caches[i])
                        rng[i] = torch.get_rng_state()
                print('Model updated in {} seconds'.format(time.time() - t_update))
                t_easy = time.time()
                if len(caches[i]['neg']) != 0 and not (self.incore and last):
                    scores = self.classifier.predict(model[i], caches[i]['neg'])
                    keep_idx = torch.where(scores >= self.easy_tresh)[0]
                    removed = len(caches[i]['neg']) - len(keep_idx)
                    caches[i]['neg'] = caches[i]['neg'][keep_idx]
                    print('Easy negatives selected in {} seconds'.format(time.time() - t_easy))
                    print('Removed {} easy negatives. {} Remaining'.format(removed, len(caches[i]['neg'])))
                    print('Iteration {}th done in {} seconds'.format(j, time.time() - t_iter))
                if last and not self.return_caches:
                    caches[i] = None  # free the class's cache
                    if torch.cuda.is_available():
                        torch.cuda.empty_cache()
        training_time = time.time() - t_start
        print('Online Classifier trained in {} seconds'.format(training_time))
        if output_dir:
            if self.is_rpn:
                head = "RPN's Online Classifier training time"
            elif self.is_segmentation:
                head = "Online Segmentation training time"
            else:
                head = "Detector's Online Classifier training time"
            with open(os.path.join(output_dir, "result.txt"), "a") as fid:
                fid.write("{}: {}min:{}s \n".format(head, int(training_time / 60), round(training_time % 60)))
        if self.return_caches:
            self.caches = caches
        return model

    def trainWithMinibootstrapStreams(self, negatives, positives, output_dir=None):
        """The same per-class state machine with the classes advancing together, one negative batch at a time, their
        fits and predictions issued on `class_streams` HIP streams (opt-in: opts['class_streams'] = k, in-core only).
        A fit of the reference regime (M ~ 2000, n ~ 1e4) is a chain of ~ 400 small dependent kernels that leaves
        most of the chip idle; classes are independent, so their chains overlap.  The host synchronises per class only
        where the sizes of the next step depend on scores (hard / easy negative selection).

        What differs from trainWithMinibootstrap: the order in which the global torch RNG would be consumed.  The
        reference draws Nystroem centres class by class (all batches of class 0, then class 1, ...), and whether a fit
        draws at all depends on the outcome of the previous pruning, so that order cannot be kept while classes overlap.
        This mode takes ONE draw from the global stream and gives class c its own mt19937 stream seeded from it: every
        class sees the same rule (FALKONWrapper.compute_indices_selection) with independent draws; results do not depend
        on the number of streams."""
        from . import solver
        C = self.num_classes - 1
        k = max(1, int(self.class_streams))
        main = torch.cuda.current_stream()
        streams = _work_streams(k, reserve=0)
        for s in streams:
            s.wait_stream(main)
        seed0 = self._class_seed()
        active = [i for i in range(C) if len(positives[i]) != 0 and len(negatives[i]) != 0 and self._owned(i)]
        rng = {i: torch.Generator().manual_seed(seed0 + i).get_state() for i in active}
        caches, model = [{} for _ in range(C)], [None] * C
        t_start = time.time()
        nb = max([len(negatives[i]) for i in active] or [0])
        for j in range(nb):
            todo = [i for i in active if j < len(negatives[i])]
            scores = {}
            if j > 0:
                for i in todo:
                    with torch.cuda.stream(streams[i % k]):
                        scores[i] = self.classifier.predict(model[i], negatives[i][j])
            with solver.deferred_pivot_checks():
                for i in todo:
                    with torch.cuda.stream(streams[i % k]):
                        if j == 0:
                            caches[i] = {'pos': positives[i], 'neg': negatives[i][0]}
                        else:
                            hard_idx = torch.where(scores[i] > self.hard_tresh)[0]
                            caches[i]['neg'] = torch.cat((caches[i]['neg'], negatives[i][j][hard_idx]), 0)
                            print('Class {}: chosen {} hard negatives from the {}th batch'.format(i, len(hard_idx), j))
                        print('Class {}: traning with {} positives and {} negatives'.format(
                            i, len(caches[i]['pos']), len(caches[i]['neg'])))
                        with torch.random.fork_rng(devices=[]):
                            torch.set_rng_state(rng[i])
                            model[i] = self.updateModel(caches[i])
                            rng[i] = torch.get_rng_state()
            prune = [i for i in todo if len(caches[i]['neg']) != 0 and j != len(negatives[i]) - 1]
            for i in prune:
                with torch.cuda.stream(streams[i % k]):
                    scores[i] = self.classifier.predict(model[i], caches[i]['neg'])
            for i in prune:
                with torch.cuda.stream(streams[i % k]):
                    keep_idx = torch.where(scores[i] >= self.easy_tresh)[0]
                    removed = len(caches[i]['neg']) - len(keep_idx)
                    caches[i]['neg'] = caches[i]['neg'][keep_idx]
                    print('Class {}: removed {} easy negatives. {} Remaining'.format(i, removed, len(caches[i]['neg'])))
        for s in streams:
            main.wait_stream(s)
        for i in active:                      # made on a side stream, used from here on by the caller's stream
            for t in (model[i].alpha_, model[i].ny_points_) + (tuple(caches[i].values()) if self.return_caches else ()):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(main)
            if not self.return_caches:
                caches[i] = None
        training_time = time.time() - t_start
        print('Online Classifier trained in {} seconds'.format(training_time))
        if output_dir:
            head = ("RPN's Online Classifier training time" if self.is_rpn else
                    "Online Segmentation training time" if self.is_segmentation else "Detector's Online Classifier training time")
            with open(os.path.join(output_dir, "result.txt"), "a") as fid:
                fid.write("{}: {}min:{}s \n".format(head, int(training_time / 60), round(training_time % 60)))
        if self.return_caches:
            self.caches = caches
        return model

    def _fits_together(self, negatives, positives):
        """Whether the K_nM blocks and caches of all classes of a round fit beside each other (the class-by-class loop
        holds one class at a time): an upper bound of their bytes against a quarter of the free device memory."""
        try:
            M = int(self.classifier.nyst_centers)
            rows = sum(len(positives[i]) + sum(len(b) for b in negatives[i]) for i in range(self.num_classes - 1)
                       if len(positives[i]) != 0 and len(negatives[i]) != 0)
            D = next((int(positives[i].shape[1]) for i in range(self.num_classes - 1) if len(positives[i]) != 0), 0)
            need = rows * (M * 4 + D * 12) + (self.num_classes - 1) * 4 * M * M * 8
            return need < torch.cuda.mem_get_info()[0] // 4
        except Exception:
            return False

    def _reference_stream_positions(self, negatives, positives):
        """Where in the GLOBAL torch RNG stream every class's Nystroem draws start when the classes are trained one after
        the other (the reference's order), predicted without training anything.  The stock index rule draws, per fit,
        `half` values when a class has more than `half` positives and `room` values when its cache holds more than `room`
        negatives (wrappers.py:_indices_tensor; torch.randint advances the generator by one 32-bit draw per value for
        every bound below 2^24).  Positives never change, so the first count is known; the second is PREDICTED as `room`
        for every fit (a cache is 2000+ rows against room <= M) and checked fit by fit by the caller.  Returns
        (per-class generator states, per-class room, state after all classes) or None when the rule is not the stock one
        or a bound is out of range."""
        from .wrappers import FALKONWrapperBase
        clf = self.classifier
        stock = (isinstance(clf, FALKONWrapperBase) and type(clf).compute_indices_selection is FALKONWrapperBase.compute_indices_selection
                 and type(clf)._indices_tensor is FALKONWrapperBase._indices_tensor
                 and "compute_indices_selection" not in clf.__dict__ and "_indices_tensor" not in clf.__dict__)
        if not stock or self._ranks()[1] > 1:
            return None
        M = int(clf.nyst_centers)
        half = int(M / 2)
        g = torch.Generator()
        g.set_state(torch.get_rng_state())
        states, rooms = {}, {}
        checked = set()
        for i in range(self.num_classes - 1):
            if len(positives[i]) == 0 or len(negatives[i]) == 0:
                continue
            n_pos = len(positives[i])
            if n_pos + sum(len(b) for b in negatives[i]) >= 2 ** 24:
                return None
            pos_draw = half if n_pos > half else 0
            rooms[i] = M - min(n_pos, half)
            states[i] = g.get_state()
            count = len(negatives[i]) * (pos_draw + max(rooms[i], 0))
            if rooms[i] <= 0:
                return None
            # run-time self-check of the consumption model (once per distinct draw shape): this torch build must advance
            # the generator by exactly `size` draws of randint(2, ..) for the bounds and sizes this class will use —
            # otherwise every later class would silently start at a wrong position; the plain loop is the answer then
            shape = (n_pos if pos_draw else 0, pos_draw, max(len(b) for b in negatives[i]) + 1, rooms[i])
            if shape not in checked:
                if not _randint_advances_one_draw_per_value(g, shape):
                    return None
                checked.add(shape)
            if count:
                torch.randint(2, (count,), generator=g)
        return states, rooms, g.get_state()

    GROUP_MIN_CLASSES = 4          # classes per group from which a round is handed to the classifier in two groups

    def trainWithMinibootstrapBatched(self, negatives, positives, output_dir=None, reference_rng=None):
        """The per-class state machine with the classes advancing together one negative batch at a time (as in
        trainWithMinibootstrapStreams) and ALL fits of a round made by one `classifier.train_batch` call: their
        preconditioners — per class a dependent chain of ~400 small factorisation kernels, the bulk of a fit at the
        reference's sizes (M ~ 2000, n ~ 3-10 k) — come out of ONE batched launch chain (odx_falkon_precond_batched_f64),
        the K_nM builds and CG loops of the classes run side by side on `class_batch` streams.  Opt-in
        (opts['class_batch'] = k, in-core only).  Each class sees exactly the arithmetic of its own sequential fit (the
        batched factors are bit-identical to the single-class ones): with the same Nystroem indices the models equal the
        sequential mode's bit for bit.  As in the streams mode the indices themselves come from one RNG stream per class
        (seeded from one draw of the global stream), because the reference's class-major order of global draws cannot be
        kept when classes advance together."""
        from . import solver
        C = self.num_classes - 1
        k = max(1, int(self.class_batch)) if self.class_batch > 0 else 4
        main = torch.cuda.current_stream()
        streams = _work_streams(k)
        active = [i for i in range(C) if len(positives[i]) != 0 and len(negatives[i]) != 0 and self._owned(i)]
        if reference_rng is None:
            seed0 = self._class_seed()
            rng = {i: torch.Generator().manual_seed(seed0 + i).get_state() for i in active}
            rooms = None
        else:
            # every class draws from the global stream's own segment for it (see _reference_stream_positions): the
            # reference's class-major draws exactly, as long as every fit draws what was predicted — checked below
            rng, rooms, final_state = reference_rng
            rng = dict(rng)
        caches, model = [{} for _ in range(C)], [None] * C
        t_start = time.time()
        nb = max([len(negatives[i]) for i in active] or [0])

        next_hard = {}          # class -> hard-negative rows of its NEXT batch under its current model (scored with the pruning predicts)
        # The classes advance in TWO groups through a round when the classifier can take them group by group
        # (FALKONWrapper.train_batch_begin): the host's part of a round — read the selections (a synchronisation), gather every
        # class's rows, draw its centres: ~5 ms for 30 classes — is done for the first group while the second group's
        # predictions of the previous round still run, and for the second group while the first group's factorisation chain
        # already does; as ONE group those 5 ms passed with the GPU idle, a tenth of a round.  The group of a class is fixed
        # (it owns the events its selections wait for); per class nothing changes.
        two = hasattr(self.classifier, 'train_batch_begin') and len(active) >= 2 * self.GROUP_MIN_CLASSES
        ng = 2 if two else 1            # (three groups: 0.455-0.46 s, four: 0.445 s, two: 0.44 s on the same box)
        cut = (len(active) + ng - 1) // ng
        parts = [active[g * cut:(g + 1) * cut] for g in range(ng)]
        pending = {}            # group -> (prune, ahead, scores, events) of its last queued predictions, selections not read yet

        def settle(g):
            """The host's part of group g's previous round: every selection of the phase — the rows a class keeps (score >= easy
            threshold) and the hard negatives of its next batch (score > hard threshold) — from ONE nonzero() over the
            concatenated scores, i.e. one host synchronisation per group and round instead of one per torch.where; each
            class's indices are its slice of the result, in the same (ascending) order torch.where gives."""
            if g not in pending:
                return
            prune, ahead, both, events = pending.pop(g)
            for ev in events:
                main.wait_event(ev)
            keep, hard = self._select_rows([both[i][0] for i in prune], [both[i][1] for i in ahead])
            for i, hard_idx in zip(ahead, hard):
                next_hard[i] = hard_idx
            for i, keep_idx in zip(prune, keep):
                removed = len(caches[i]['neg']) - len(keep_idx)
                caches[i]['neg'] = caches[i]['neg'][keep_idx]
                print('Class {}: removed {} easy negatives. {} Remaining'.format(i, removed, len(caches[i]['neg'])))

        for j in range(nb):
            todo = [i for i in active if j < len(negatives[i])]
            order = []              # the round's classes in the order they are handed to the classifier

            def with_class_rng(pos, fn):
                i = order[pos]
                with torch.random.fork_rng(devices=[]):
                    torch.set_rng_state(rng[i])
                    out = fn()
                    rng[i] = torch.get_rng_state()
                return out

            def training_sets(members):
                if j == 0:
                    for i in members:
                        caches[i] = {'pos': positives[i], 'neg': negatives[i][0]}
                else:
                    for i in members:
                        hard_idx = next_hard.pop(i)
                        caches[i]['neg'] = torch.cat((caches[i]['neg'], negatives[i][j][hard_idx]), 0)
                        print('Class {}: chosen {} hard negatives from the {}th batch'.format(i, len(hard_idx), j))
                if rooms is not None and any(len(caches[i]['neg']) <= rooms[i] for i in members):
                    return None      # a fit that draws fewer values than predicted: the later classes' positions in the stream are off
                Xs, ys = [], []
                for i in members:
                    X_pos, X_neg = caches[i]['pos'], caches[i]['neg']
                    print('Class {}: traning with {} positives and {} negatives'.format(i, len(X_pos), len(X_neg)))
                    Xs.append(torch.cat((X_pos, X_neg), 0))
                    # labels on the host: the Nystroem index rule reads them (two nonzero() per class — host synchronisations
                    # when the labels live on the GPU); the fit uploads its own f64 copy
                    ys.append(torch.cat((torch.ones(len(X_pos)), -torch.ones(len(X_neg))), 0))
                return Xs, ys

            # every Cholesky status of the round is read once (a host synchronisation), when the block closes: after the round's
            # fits AND the predictions that use them are queued — read right behind the fits, the host would start queueing the
            # 60 predictions of a round only when the GPU has gone idle
            with solver.deferred_pivot_checks():
                groups = [[i for i in part if i in todo] for part in parts]
                if two:
                    rnd = self.classifier.train_batch_begin(sigma=self.sigma, lam=self.lam, streams=streams, expect_total=len(todo))
                    for g, members in enumerate(groups):
                        settle(g)
                        sets = training_sets(members)
                        if sets is None:
                            # (advisor, round 4) an earlier group's chain may already be queued on the side streams and the other
                            # group's predictions of the previous round may still be pending: join all of it before the
                            # caller's fallback reuses the rows and caches on the main stream
                            rnd.abort()
                            for s in streams:
                                main.wait_stream(s)
                            for gg in list(pending):
                                for ev in pending.pop(gg)[3]:
                                    main.wait_event(ev)
                            return None
                        order += members
                        rnd.add(sets[0], sets[1], index_rng=with_class_rng)
                    fitted = rnd.finish()
                else:
                    settle(0)
                    sets = training_sets(todo)
                    if sets is None:
                        return None
                    order += todo
                    fitted = self.classifier.train_batch(sets[0], sets[1], sigma=self.sigma, lam=self.lam, index_rng=with_class_rng,
                                                         streams=streams)
                for i, m in zip(order, fitted):
                    model[i] = m
                # the two predicts a class's fresh model is used for — pruning its cache now, mining its next batch at the
                # start of the next round — in ONE phase on the streams (the next batch is known; the reference scores it first
                # thing in the next iteration with this same model, OnlineRegionClassifier_incore.py:112-116), group by group:
                # a group's selections wait for ITS predictions only (an event per stream behind them)
                for s in streams:
                    s.wait_stream(main)
                for g, members in enumerate(groups):
                    prune = [i for i in members if len(caches[i]['neg']) != 0 and j != len(negatives[i]) - 1]
                    ahead = [i for i in members if j + 1 < len(negatives[i])]
                    both = {}
                    for i in sorted(set(prune) | set(ahead)):
                        with torch.cuda.stream(streams[i % k]):
                            both[i] = (self.classifier.predict(model[i], caches[i]['neg']) if i in prune else None,
                                       self.classifier.predict(model[i], negatives[i][j + 1]) if i in ahead else None)
                    events = []
                    for s in {id(s): s for s in streams}.values():
                        ev = torch.cuda.Event()
                        ev.record(s)
                        events.append(ev)
                    pending[g] = (prune, ahead, both, events)
        for g in range(len(parts)):
            settle(g)
        for s in streams:
            main.wait_stream(s)
        for i in active:
            if not self.return_caches:
                caches[i] = None
        if reference_rng is not None:
            torch.set_rng_state(final_state)          # the global stream where the class-by-class loop leaves it
        training_time = time.time() - t_start
        print('Online Classifier trained in {} seconds'.format(training_time))
        if output_dir:
            head = ("RPN's Online Classifier training time" if self.is_rpn else
                    "Online Segmentation training time" if self.is_segmentation else "Detector's Online Classifier training time")
            with open(os.path.join(output_dir, "result.txt"), "a") as fid:
                fid.write("{}: {}min:{}s \n".format(head, int(training_time / 60), round(training_time % 60)))
        if self.return_caches:
            self.caches = caches
        return model

    def _select_rows(self, easy_scores, hard_scores):
        """Row indices with score >= easy threshold for each vector of `easy_scores` and with score > hard threshold for
        each of `hard_scores` — what torch.where(...)[0] returns vector by vector, from one nonzero() and one host read."""
        vecs = [v.reshape(-1) for v in easy_scores] + [v.reshape(-1) for v in hard_scores]
        if not vecs:
            return [], []
        lens = [int(v.numel()) for v in vecs]
        flat = torch.cat(vecs)
        ends = np.cumsum(lens)
        split = int(ends[len(easy_scores) - 1]) if easy_scores else 0
        mask = torch.empty(flat.shape, dtype=torch.bool, device=flat.device)
        torch.ge(flat[:split], self.easy_tresh, out=mask[:split])
        torch.gt(flat[split:], self.hard_tresh, out=mask[split:])
        nz_host = mask.nonzero().reshape(-1).cpu().numpy()
        cuts = np.searchsorted(nz_host, ends, side='left')
        # positions inside each vector, made on the host from the one array that was read back and uploaded once (a
        # subtraction per vector on the GPU was 60 launches per round)
        counts = np.diff(np.concatenate(([0], cuts)))
        local = torch.from_numpy(nz_host - np.repeat(ends - np.asarray(lens), counts)).to(flat.device)
        out, a = [], 0
        for b in cuts:
            out.append(local[a:int(b)])
            a = int(b)
        return out[:len(easy_scores)], out[len(easy_scores):]

    def trainRegionClassifier(self, opts=None, output_dir=None):
        if opts is not None:
            self.processOptions(opts)
        print('Training Online Region Classifier')
        negatives, positives = self.negatives, self.positives
        if not self.incore:
            ref = negatives[0][0].device
            self.mean, self.std, self.mean_norm = self.mean.to(ref), self.std.to(ref), self.mean_norm.to(ref)
        if not self.normalized:
            for i in range(self.num_classes - 1):
                if len(positives[i]):
                    positives[i] = self.zScores(positives[i])
                for j in range(len(negatives[i])):
                    if len(negatives[i][j]):
                        negatives[i][j] = self.zScores(negatives[i][j])
            self.normalized = True
        if self.incore and self.class_batch > 0 and torch.cuda.is_available() and hasattr(self.classifier, 'train_batch'):
            model = self.trainWithMinibootstrapBatched(negatives, positives, output_dir=output_dir)
        elif self.incore and self.class_streams > 0 and torch.cuda.is_available():
            model = self.trainWithMinibootstrapStreams(negatives, positives, output_dir=output_dir)
        else:
            model = None
            if (self.incore and self.reference_order == 'auto' and not self.class_rng and torch.cuda.is_available()
                    and hasattr(self.classifier, 'train_batch') and self._fits_together(negatives, positives)):
                # the reference's order of Nystroem draws, the classes advancing together: every class reads its own
                # segment of the global RNG stream, predicted in advance; a fit that would draw differently (a cache that
                # shrank below the room for negative centres) makes the attempt return None and the plain loop runs
                where = self._reference_stream_positions(negatives, positives)
                if where is not None:
                    model = self.trainWithMinibootstrapBatched(negatives, positives, output_dir=output_dir, reference_rng=where)
            self.last_order = 'reference order, classes together' if model is not None else 'reference order, class by class'
            if model is None:
                model = self.trainWithMinibootstrap(negatives, positives, output_dir=output_dir)
        model = self._gather_models(model)
        if self.incore and self.return_caches:
            return model, self.caches
        return model

    def testRegionClassifier(self, model, test_boxes):
        print('Online Region Classifier testing')
        BoxList = get_boxlist_class()
        dev = _device()
        predictions = []
        total = 0.0
        try:
            for c in range(self.num_classes - 1):
                model[c].ny_points_ = model[c].ny_points_.to(dev)
                model[c].alpha_ = model[c].alpha_.to(dev)
        except Exception:
            pass
        if not self.incore:
            self.mean, self.std, self.mean_norm = (torch.as_tensor(self.mean).to(dev), torch.as_tensor(self.std).to(dev),
                                                   torch.as_tensor(self.mean_norm).to(dev))
        fused = None
        if (dev == 'cuda' and self.num_classes > 2 and len(model) >= self.num_classes - 1
                and all(m is not None and getattr(m, 'alpha_', None) is not None and hasattr(getattr(m, 'kernel', None), 'sigma')
                        for m in model[:self.num_classes - 1])
                and len({float(m.kernel.sigma) for m in model[:self.num_classes - 1]}) == 1):
            from .heads import _OnlineHead
            fused = _OnlineHead(list(model[:self.num_classes - 1]), None, None)
        for entry in test_boxes:
            if entry is None:
                continue
            keep = np.nonzero(entry['gt'] == 0)
            boxes = entry['boxes'][keep, :][0]
            X_test = torch.tensor(entry['feat'][keep, :][0], device=dev)
            t0 = time.time()
            if self.mean_norm != 0:
                X_test = self.zScores(X_test)
            scores = -torch.ones((len(boxes), self.num_classes))
            if fused is not None:
                # all classes with ONE fused scoring launch over the concatenated centres (what the in-network detector
                # head does, odx.heads._OnlineHead._scores) instead of num_classes - 1 predicts per image
                scores[:, 1:] = fused._scores(_backend.get_backend().features(X_test), len(boxes)).to(scores.device)
            else:
                for c in range(self.num_classes - 1):
                    scores[:, c + 1] = torch.squeeze(self.classifier.predict(model[c], X_test))
            total += time.time() - t0
            b = BoxList(torch.from_numpy(boxes), (entry['img_size'][0], entry['img_size'][1]), mode="xyxy")
            b.add_field("scores", scores.to('cpu'))
            predictions.append(b)
        print('Average image testing time: {} seconds.'.format(total / len(test_boxes)))
        return predictions

    def predict(self, dataset) -> None:
        pass

    def zScores(self, feat, target_norm=20):
        feat = feat - self.mean
        return feat * (target_norm / self.mean_norm.item())
