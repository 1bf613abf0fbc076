"""Where the drop-in FeatureExtractor / AccuracyEvaluator get their image stream and network when the caller passes none.

The reference's drivers construct `FeatureExtractor(cfg)` / `AccuracyEvaluator(cfg)` and call them without any
`cfg_options['samples']` (experiments/run_experiment_online_rpn_ood_oos.py:72,78-83,291-307): there the images come from
maskrcnn_benchmark dataset classes named in the YAML and the weights from a checkpoint path — both outside this repository.
So that such a driver runs UNCHANGED, the two things are looked up through the environment:

    ODX_SAMPLES = "module:callable"    callable(split, cfg_path) -> iterable of (image, gt_boxes, gt_labels[, masks[, difficult]])
                                       split is "train" or "test"
    ODX_MODEL   = "module:callable"    callable(cfg_path) -> an odx.extract.OnlineDetectionModel / odx.fpn.OnlineDetectionModelFPN

An explicit cfg_options['samples'] / ['model'] always wins; with neither the facades raise as before.
"""
import importlib
import os


def _resolve(var):
    spec = os.environ.get(var, "")
    if not spec:
        return None
    mod, _, name = spec.partition(":")
    if not mod or not name:
        raise ValueError("%s must look like 'module:callable', got %r" % (var, spec))
    fn = getattr(importlib.import_module(mod), name, None)
    if not callable(fn):
        raise ValueError("%s: %s has no callable %r" % (var, mod, name))
    return fn


def fill(cfg_options, split, cfg_path):
    """cfg_options with 'samples' / 'model' filled in from ODX_SAMPLES / ODX_MODEL where the caller gave none (a copy: the
    drivers' shared default dict is never written to)."""
    opts = dict(cfg_options or {})
    if "samples" not in opts:
        fn = _resolve("ODX_SAMPLES")
        if fn is not None:
            opts["samples"] = fn(split, cfg_path)
    if opts.get("model") is None:
        fn = _resolve("ODX_MODEL")
        if fn is not None:
            opts["model"] = fn(cfg_path)
    return opts
