"""`ClassifierExperiment` — the caller of the hot path, mirroring the life cycle of
mmgclip/experiments/ClassifierExperiment.py:34-344: build model / loss / AdamW / scheduler / early stopper, then
`run()`: for each epoch `train()` (zero_grad(set_to_none) -> model(batch) -> criterion(**outputs) -> backward ->
optimizer.step(); scheduler stepped ONCE PER EPOCH, so epoch 1 runs at lr 0: SURVEY.md §0) and `validate()`.

Scope: the train loop and `validate()` (loss + zero-shot prompt AUROC: 1 malignancy prompt, 4 mass
shape prompts, 8 BI-RADS prompts; prompt embeddings cached once per call instead of re-encoded per batch) are reproduced;
`test()` (the offline Evaluator with bootstrap CIs, confusion matrices and plots) is out of scope.
Additive options: `distributed.global_loss`, `optimizer.config.fused` (FusedAdamW over the parameter arenas).
"""
import os
import time

import numpy as np
import torch

os.environ["TOKENIZERS_PARALLELISM"] = "false"

from ..callbacks.early_stopping import EarlyStopper                      # noqa: E402
from ..dataset.synthetic import BENIGN_MALIGNANT, MASS_SHAPES            # noqa: E402
from ..loss.loss_controller import create_loss                           # noqa: E402
from ..networks.mmgclip_model import MMGCLIP as model, _get              # noqa: E402
from ..params import begin_step                                          # noqa: E402
from ..scheduler.warmup_cosine import LinearWarmupCosineAnnealingLR      # noqa: E402
from ..utils.global_utils import create_directory_if_not_exists         # noqa: E402
from ..utils.logger import logger                                        # noqa: E402
from ..utils.train_utils import epoch_time                               # noqa: E402


def _label_value(label, enum):
    """Reference datasets hand labels over as ints or as enum member names (mmgclip/prompts/enums.py)."""
    return enum[label] if isinstance(label, str) else int(label)


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


def _summary_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:                 # tensorboard is not installed in the build image: scalars are only logged
        return _NullWriter()


class ClassifierExperiment:
    def __init__(self, config=None, train_dataloader=None, valid_dataloader=None, test_dataloader=None, tokenizer=None,
                 comm=None):
        self._time_start = self._time_end = None
        self.train_dataloader, self.valid_dataloader, self.test_dataloader = train_dataloader, valid_dataloader, test_dataloader
        self.tokenizer = tokenizer
        self.config = config
        self.current_epoch = 0
        self.comm = comm
        logger.info(f"Experiment Parameters: name={self.__class__.__name__}")
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        assert self.config is not None, 'Error in initializing the model. Missing training config object.'

        self.model = model(config=config).to(self.device)
        self.model.count_parameters(self.model)

        loss_cls = create_loss(self.config.loss.config.loss_name)
        use_global = comm is not None and _get(config, "distributed.global_loss", True)
        try:
            self.criterion = loss_cls(comm=comm if use_global else None).to(self.device)
        except TypeError:
            self.criterion = loss_cls().to(self.device)
        logger.info(f"Using {self.criterion.__class__.__name__} loss.")

        lr, wd = self.config.optimizer.config.learning_rate, self.config.optimizer.config.weight_decay
        if _get(config, "optimizer.config.fused", False):
            # the towers' parameter arenas do not exist yet (they are built at the first forward): FusedAdamW finds them from
            # the parameters at every step.  EVERY parameter is listed, frozen ones included, exactly as the reference hands
            # `model.parameters()` to torch.optim.AdamW (ClassifierExperiment.py:74): the param-group layout of a checkpoint's
            # optimizer_state_dict is then the same in both directions (frozen parameters never get a gradient and are skipped)
            from ..optim import FusedAdamW
            self.optimizer = FusedAdamW(self.model.parameters(), lr=lr, weight_decay=wd)
        else:
            self.optimizer = torch.optim.AdamW(self.model.parameters(), lr=lr, weight_decay=wd)

        if self.config.scheduler.name == "cosine":
            self.scheduler = LinearWarmupCosineAnnealingLR(self.optimizer, total_steps=self.config.scheduler.config.epochs,
                                                           warmup_steps=self.config.scheduler.config.warmup_epochs)
        elif self.config.scheduler.name == "ReduceLROnPlateau":
            self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, 'min',
                                                                        patience=self.config.scheduler.config.patience)
        logger.info(f"Using {self.scheduler.__class__.__name__}")

        self.ckp_path = create_directory_if_not_exists(self.config.checkpoints.checkpoints_export_dir)
        self.ckp_path = os.path.join(self.ckp_path, self.config.checkpoints.checkpoints_file_name)
        # data-parallel run: every rank takes the same early-stopping decisions (the validation loss is averaged over the
        # ranks in run()), rank 0 alone writes the checkpoint file and the TensorBoard scalars
        self._lead = comm is None or comm.rank == 0
        self.early_stopper = EarlyStopper(patience=self.config.base.patience, delta=0, trace_func=logger.warning, save=self._lead)
        self.writer = _summary_writer(self.config.base.tensorboard_export_dir) if self._lead else _NullWriter()
        self._sync = None

    # ---- data parallelism (additive: the reference is single-process) --------------------------------------------
    def _towers(self):
        return [t for t in (getattr(self.model, "image_encoder", None), self.model.text_encoder)
                if t is not None and hasattr(t, "post_backward_hook")]

    def _grad_sync(self):
        """GradSync over the tower arenas + every other trainable parameter; built once the arenas exist, i.e. after the first
        forward.  With the global-batch loss the gradients of the ranks are SUMMED (the loss carries 1/(2N) of the global mean);
        with `distributed.global_loss: false` every rank has its own local-batch loss and the gradients are AVERAGED."""
        if self._sync is None:
            from ..distributed import GradSync
            arenas = [t.arena for t in self._towers() if t.arena is not None and t.arena.any_trainable()]
            owned = {id(p) for a in arenas for p in a.params}
            extra = [p for p in self.model.parameters() if p.requires_grad and id(p) not in owned]
            use_global = _get(self.config, "distributed.global_loss", True)
            self._sync = GradSync(self.comm, arenas, extra, scale=1.0 if use_global else 1.0 / self.comm.world_size)
            for t in self._towers():
                t.post_backward_hook = self._sync.reduce_arena_async
        return self._sync

    def train(self):
        """One epoch; returns the mean of the per-step losses (ClassifierExperiment.py:93-132)."""
        self.model.train()
        loss_list = []
        parallel = self.comm is not None and self.comm.active
        for index, batch in enumerate(self.train_dataloader):
            self.optimizer.zero_grad(set_to_none=True)
            if parallel:
                for t in self._towers():
                    begin_step(t)
            outputs = self.model(batch)
            loss, labels = self.criterion(**outputs)
            sync = self._grad_sync() if parallel else None      # (installs the towers' post-backward hooks on first use)
            loss.backward()
            if hasattr(self.model, "join_streams"):
                self.model.join_streams()                       # (two-stream mode: the text tower's backward ran on a side stream)
            if sync is not None:
                sync.finish()                                   # every gradient is the all-rank sum before the optimizer reads it
            self.optimizer.step()
            loss_list.append(loss.item())
        self.scheduler.step()
        epoch_loss = np.mean(loss_list)
        self.writer.add_scalar('loss/train', epoch_loss, self.current_epoch + 1)
        return epoch_loss

    def _prompt_token_sets(self):
        """{metric: tokens of its k prompts}: the tokenizer when the caller supplied one, else hashed stand-in ids."""
        from ..dataset.synthetic import synthetic_prompt_tokens, validation_prompts
        S = self.config.tokenizer.config.sequence_length
        sets = {}
        for name, strings in validation_prompts(self.config.experiments.config.metrics).items():
            if self.tokenizer is not None:
                sets[name] = self.tokenizer(strings, padding="max_length", truncation=True, return_tensors="pt", max_length=S)
            else:
                sets[name] = synthetic_prompt_tokens(strings, S, self.model.text_encoder.config.vocab_size)
        return sets

    @staticmethod
    def _auc(y_true, y_score):
        from sklearn import metrics
        y_true = np.asarray(y_true)
        if y_true.min() == y_true.max():
            return float("nan")              # one class only in this split: AUROC undefined
        fpr, tpr, _ = metrics.roc_curve(y_true, y_score)
        return float(metrics.auc(fpr, tpr))

    def validate(self):
        """Validation loss + zero-shot prompt AUROC (ClassifierExperiment.py:134-289): malignancy (1 prompt), mass shape
        (4 prompts, one-vs-rest mean), BI-RADS (8 prompts, one-vs-rest mean).  Returns the reference's 5-tuple
        (loss, auc_malig, auc_shapes, auc_birads, auc_mean), -1 for metrics that are not configured.

        The reference re-encodes the same prompts through BERT for every batch (:192-229); here their embeddings are
        computed ONCE per call and every batch only runs the [n,D] x [D,k] logit kernel."""
        from .. import head
        self.model.eval()
        metric_names = self.config.experiments.config.metrics
        loss_list, targets, preds = [], {}, {}
        with torch.no_grad():
            prompt_emb = {}
            for name, tokens in self._prompt_token_sets().items():
                tf = self.model.encode_text({"text_tokens": tokens})
                te = self.model.text_projection_layer(tf) if self.model.text_projection_layer is not None else tf
                prompt_emb[name] = head.L2Normalize.apply(te).contiguous()
                targets[name], preds[name] = [], []
            for batch in self.valid_dataloader:
                outputs = self.model(batch)                      # :166 (no `validation` flag: MMGCLIPLoss keeps its second text pass)
                loss, _ = self.criterion(**outputs)              # :169 the configured criterion, t2t term included
                loss_list.append(loss.item())
                labels = batch["prompt_labels"]
                scale = outputs["logit_scale"].reshape(1).float().contiguous()
                for name, te in prompt_emb.items():
                    _, _, sims = head.rows_forward(outputs["image_embeddings"].contiguous(), te, scale, 0, True)   # [n,k]
                    preds[name].append(sims.cpu().numpy())
                    if name == "malig":         # int (exam-report dataset) or the enum member's name (image-label dataset): :179-185
                        targets[name].extend(_label_value(l["BenignMalignantDatasetLabels"], BENIGN_MALIGNANT) for l in labels)
                    elif name == "shapes":      # :199-202
                        targets[name].extend(_label_value(l["MassShapeLabels"], MASS_SHAPES) for l in labels)
                    else:
                        targets[name].extend(-1 if l["BIRADS"] == "unknown" else int(l["BIRADS"]) for l in labels)
        epoch_loss = float(np.mean(loss_list)) if loss_list else float("nan")
        self.writer.add_scalar('loss/val', epoch_loss, self.current_epoch + 1)
        aucs = {}
        for name in preds:
            p = np.concatenate(preds[name], 0)
            t = np.asarray(targets[name])
            if name == "malig":
                aucs[name] = self._auc(t, p[:, 0])
            else:
                off = 1 if name == "birads" else 0        # BI-RADS prompt idx 0 is "unknown" (label -1)
                aucs[name] = float(np.nanmean([self._auc(t == (i - off), p[:, i]) for i in range(p.shape[1])]))
            self.writer.add_scalar(f'auc/val/{name}', aucs[name], self.current_epoch + 1)
        mean_auc = float(np.mean(list(aucs.values()))) if len(aucs) > 1 else -1
        if len(aucs) > 1:
            self.writer.add_scalar('auc/val/average', mean_auc, self.current_epoch + 1)
        return (epoch_loss, aucs.get("malig", -1), aucs.get("shapes", -1), aucs.get("birads", -1),
                mean_auc if len(metric_names) > 1 else -1)

    def test(self):
        """Test cycle on the test loader (ClassifierExperiment.py:291-301): `Evaluator(...).evaluate_experiment()`."""
        from ..evaluator import Evaluator
        logger.info("Running testing evaluator script.")
        return Evaluator(config=self.config, test_dataloader=self.test_dataloader, tokenizer=self.tokenizer,
                         model=self.model).evaluate_experiment()

    def run(self):
        self._time_start = time.time()
        for self.current_epoch in range(self.config.scheduler.config.epochs):
            t0 = time.time()
            train_loss = self.train()
            val = self.validate() if self.valid_dataloader is not None else (train_loss, -1, -1, -1, -1)
            val_loss = val[0]
            if self.comm is not None and self.comm.active:       # same early-stopping decision on every rank
                t = torch.tensor([float(val_loss)], dtype=torch.float64, device=self.device if self.device.type == "cuda" and
                                 torch.distributed.get_backend(self.comm.group) == "nccl" else "cpu")
                val_loss = float(self.comm.all_reduce_sum(t)[0]) / self.comm.world_size
            mins, secs = epoch_time(t0, time.time())
            self.writer.add_scalar('lr', self.optimizer.param_groups[0]['lr'], self.current_epoch + 1)
            self.early_stopper(val_loss, self.current_epoch, self.model, self.optimizer, self.ckp_path)
            logger.info(f"Epoch: {self.current_epoch + 1:02} | Time: {mins}m {secs}s | train loss {train_loss:.4f} | "
                        f"val loss {val_loss:.4f} | val AUC malig/shapes/birads/mean {val[1]:.3f}/{val[2]:.3f}/{val[3]:.3f}/{val[4]:.3f}")
            if self.early_stopper.early_stop:
                logger.info("Early stopping")
                break
        # the end-of-run test pass (ClassifierExperiment.py:337-339); rank 0 alone writes the result files
        if len(_get(self.config, "dataset.eval.enum_classes", [])) > 0 and self.test_dataloader is not None and self._lead:
            self.test_results = self.test()
        self._time_end = time.time()
        logger.info(f"Run complete. Total time: {time.strftime('%H:%M:%S', time.gmtime(self._time_end - self._time_start))}")
