#!/usr/bin/env python3
"""Makes tests/golden/driver_run.npz + driver_run.json (container-only: needs /root/reference).

1. A tiny seeded network harvests RPN / detector / segmentation rows of six synthetic images on the CPU (tests' oracle
   backend) and spills them to the reference's on-disk feature-cache layout (odx/storage.py).  The cache files, packed into
   one array file, are the fixture's INPUT.
2. The reference's own driver, experiments/run_experiment_online_rpn_ood_oos.py, is executed UNCHANGED — its source text is
   compiled and run as __main__ with `__file__` pointing into online-detection_amd/experiments/, so that its
   sys.path.append lines (:6-12) find this repository's drop-in modules under online-detection_amd/src — with
       --load_RPN_detector_segmentation_features --save_RPN_detector_segmentation_models --CPU
   on those caches.  It constructs FeatureExtractor / AccuracyEvaluator without any cfg_options; images and network reach the
   drop-ins through ODX_SAMPLES / ODX_MODEL (odx/providers.py).
3. What the run left in its output directory (model files' structure / shapes / norms, result.txt lines without wall-clock
   times, the mAP lines) is summarised into driver_run.json: the fixture's EXPECTED OUTPUT.
Nothing of the reference is stored: arrays and a summary of outputs only."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import odx  # noqa: E402
from tests import driver_fixture as df  # noqa: E402
from tests.oracle_backend import OracleBackend  # noqa: E402


def run_reference_driver(out_dir, cfg_path):
    """exec the reference driver's source, unmodified, as __main__."""
    fake = os.path.join(ROOT, "online-detection_amd", "experiments", os.path.basename(df.REF_DRIVER))
    src = open(df.REF_DRIVER).read()
    argv, env = list(sys.argv), dict(os.environ)
    path = list(sys.path)
    sys.argv = [fake, "--output_dir", out_dir, "--load_RPN_detector_segmentation_features", "--save_RPN_detector_segmentation_models",
                "--CPU", "--config_file_feature_extraction", cfg_path, "--config_file_online_rpn_detection_segmentation", cfg_path]
    os.environ["ODX_SAMPLES"] = "tests.driver_fixture:samples"
    os.environ["ODX_MODEL"] = "tests.driver_fixture:model"
    try:
        torch.manual_seed(df.SEED)
        exec(compile(src, fake, "exec"), {"__name__": "__main__", "__file__": fake})
    finally:
        sys.argv = argv
        sys.path[:] = path
        os.environ.clear()
        os.environ.update(env)


def main():
    odx.set_backend(OracleBackend(np.float64))
    tmp = tempfile.mkdtemp()
    cfg_path = os.path.join(tmp, "cfg.yaml")
    yaml.safe_dump(df.config(), open(cfg_path, "w"))
    from tests import dropin
    fe = dropin.load("feature_extractor").FeatureExtractor(cfg_path)
    torch.manual_seed(7)
    assert fe.extractFeaturesRPNDetector(True, output_dir=tmp, save_features=True, extract_features_segmentation=True,
                                         cfg_options={"samples": df.samples("train"), "model": df.model()}) is None
    arrays = df.pack_cache(tmp)
    os.remove(os.path.join(tmp, "result.txt"))
    run_reference_driver(tmp, cfg_path)
    summary = df.summarize(tmp)
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, "driver_run.npz"), **arrays)
    json.dump(summary, open(os.path.join(here, "driver_run.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(summary["result_lines"], indent=1))
    print("%d cache files, %.0f KB packed" % (len(arrays), os.path.getsize(os.path.join(here, "driver_run.npz")) / 1e3))


if __name__ == "__main__":
    main()
