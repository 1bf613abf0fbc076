"""Worker of test_class_sharded_minibootstrap_on_gpu (tests/test_gpu_modules.py): one rank of a `torch.distributed.run`
launch; trains its classes (i % world == rank) in the class-batch mode on the HIP backend, gathers all models, saves them."""
import io
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def problem(path):
    import yaml
    D, C, ITER, M = 64, 5, 3, 96
    cfg = {"NUM_CLASSES": C + 1, "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                                              "CLASSIFIER": {"lambda": 0.001, "sigma": 8, "M": M, "kernel_type": "gauss"}},
           "CHOSEN_CLASSES": {i: "c%d" % i for i in range(C + 1)}}
    with open(path, "w") as fid:
        yaml.safe_dump(cfg, fid)
    g = torch.Generator().manual_seed(61)
    mus = torch.randn(C, D, generator=g) * 1.2
    pos, neg = [], []
    for c in range(C):
        npos = [150, 0, 90, 200, 30][c]
        pos.append((mus[c] + 0.6 * torch.randn(npos, D, generator=g)).cuda() if npos else torch.empty((0, D)).cuda())
        neg.append([(mus[(c + 1 + j % 2) % C] * (0.4 + 0.15 * j) + 0.9 * torch.randn(200, D, generator=g)).cuda() for j in range(ITER)])
    stats = {"mean": torch.zeros(D).cuda(), "std": torch.ones(D).cuda(), "mean_norm": torch.tensor(8.0).cuda()}
    return pos, neg, stats


def train(path, opts):
    from tests import dropin
    pos, neg, stats = problem(path)
    orc_mod = dropin.load("OnlineRegionClassifier_incore")
    wrap_mod = dropin.load("FALKONWrapper_with_centers_selection_incore")
    torch.manual_seed(5)
    with redirect_stdout(io.StringIO()):
        models = orc_mod.OnlineRegionClassifier(wrap_mod.FALKONWrapper(cfg_path=path), pos, neg, stats, cfg_path=path).trainRegionClassifier(opts=opts)
    return [None if m is None else (m.ny_points_.cpu(), m.alpha_.cpu()) for m in models]


if __name__ == "__main__":
    out_dir = sys.argv[1]
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")          # both ranks share the box's one GPU: RCCL needs one device per rank
    try:
        torch.save(train(os.path.join(out_dir, "cfg_%d.yaml" % rank), {"class_batch": 2, "class_shard": True}),
                   os.path.join(out_dir, "models_%d.pt" % rank))
    finally:
        dist.destroy_process_group()
