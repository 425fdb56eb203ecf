"""`python train.py --config-name train_binary_class_clf [group=option] [key=value]` — same entry point and config
surface as the reference's train.py:9-90 (Hydra defaults lists under configs/), wired to the MI355X hot path.

The reference's datasets need proprietary mammography data; with `dataset.config.synthetic: true` (the default here)
the loaders are seeded synthetic batches with the collate_fn's keys and shapes.  Multi-GPU: launch with
`python -m torch.distributed.run --nproc-per-node N train.py ...` (one process per GPU, RCCL).
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from mmgclip.config import compose                                              # noqa: E402
from mmgclip.dataset.synthetic import SyntheticLoader                           # noqa: E402
from mmgclip.experiments.experiments_controller import create_experiment       # noqa: E402
from mmgclip.networks.mmgclip_model import PIXEL_ENCODERS, _get                # noqa: E402
from mmgclip.utils.global_utils import seeding                                 # noqa: E402
from mmgclip.utils.logger import logger                                        # noqa: E402
from mmgclip import distributed                                                # noqa: E402


def build_loaders(cfg, rank=0):
    if not _get(cfg, "dataset.config.synthetic", False):
        raise NotImplementedError("only `dataset.config.synthetic: true` is available: the reference's Radboud data "
                                  "pipeline (mmgclip/dataset/dataset.py) is outside this hot path (SURVEY.md §2 #9)")
    n = int(_get(cfg, "dataset.config.synthetic_samples", 4964))
    n_train = int(n * cfg.dataset.split.train_split_ratio)
    n_val = int((n - n_train) * cfg.dataset.split.test_split_ratio)
    pixels = cfg.networks.image_encoder.name in PIXEL_ENCODERS
    kw = dict(S=cfg.tokenizer.config.sequence_length, with_impression=cfg.loss.config.loss_name == "MMGCLIPLoss")
    if pixels:
        kw.update(image_size=_get(cfg, "networks.image_encoder.image_size", 224), in_chans=_get(cfg, "networks.image_encoder.in_chans", 1))
    else:
        kw.update(feature_dim=cfg.networks.image_encoder.image_features_dimension)
    bt, bv = cfg.dataloader.train.batch_size, cfg.dataloader.valid.batch_size
    train = SyntheticLoader(max(1, n_train // bt), bt, seed=cfg.base.seed + 1000 * rank, **kw)
    valid = SyntheticLoader(max(1, n_val // bv), bv, seed=cfg.base.seed + 500000 + 1000 * rank, **kw)
    # the held-out half of the non-training samples (the reference's test split, dataset.py: test_split_ratio); only rank 0 scores it
    btst = cfg.dataloader.test.batch_size
    test = SyntheticLoader(max(1, (n - n_train - n_val) // btst), btst, seed=cfg.base.seed + 900000, **{**kw, "with_impression": False})
    return train, valid, test


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-name", default="train_binary_class_clf")
    ap.add_argument("--config-path", default=os.path.join(HERE, "configs"))
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)
    cfg = compose(args.config_path, args.config_name, args.overrides)
    seeding(cfg.base.seed)
    comm = distributed.init_from_env()
    train_loader, val_loader, test_loader = build_loaders(cfg, comm.rank if comm else 0)
    logger.info(f"train batches: {len(train_loader)}, valid batches: {len(val_loader)}, test batches: {len(test_loader)}")
    experiment = create_experiment(cfg.experiments.config.experiment_name)(
        config=cfg, train_dataloader=train_loader, valid_dataloader=val_loader, test_dataloader=test_loader, tokenizer=None, comm=comm)
    experiment.run()
    return experiment


if __name__ == "__main__":
    main()
