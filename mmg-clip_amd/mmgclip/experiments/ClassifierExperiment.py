"""`ClassifierExperiment` — the caller of the hot path, mirroring the life cycle of
mmgclip/experiments/ClassifierExperiment.py:34-344: build model / loss / AdamW / scheduler / early stopper, then
`run()`: for each epoch `train()` (zero_grad(set_to_none) -> model(batch) -> criterion(**outputs) -> backward ->
optimizer.step(); scheduler stepped ONCE PER EPOCH, so epoch 1 runs at lr 0: SURVEY.md §0) and `validate()`.

Scope: the train loop and the validation LOSS are reproduced; the prompt-AUROC metrics of
`validate()` / `test()` (sklearn + re-encoding 1/4/8 prompts per batch) are listed as "next" and not built here.
Additive options: `distributed.global_loss`, `optimizer.config.fused` (FusedAdamW over the parameter arenas).
"""
import os
import time

import numpy as np
import torch

os.environ["TOKENIZERS_PARALLELISM"] = "false"

from ..callbacks.early_stopping import EarlyStopper                      # noqa: E402
from ..loss.loss_controller import create_loss                           # noqa: E402
from ..networks.mmgclip_model import MMGCLIP as model, _get              # noqa: E402
from ..scheduler.warmup_cosine import LinearWarmupCosineAnnealingLR      # noqa: E402
from ..utils.global_utils import create_directory_if_not_exists         # noqa: E402
from ..utils.logger import logger                                        # noqa: E402
from ..utils.train_utils import epoch_time                               # noqa: E402


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
            from ..optim import FusedAdamW
            arenas = [getattr(m, "arena", None) for m in (getattr(self.model, "image_encoder", None), self.model.text_encoder)]
            self.optimizer = FusedAdamW([p for p in self.model.parameters() if p.requires_grad], lr=lr, weight_decay=wd,
                                        arenas=arenas)
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
        self.early_stopper = EarlyStopper(patience=self.config.base.patience, delta=0, trace_func=logger.warning)
        self.writer = _summary_writer(self.config.base.tensorboard_export_dir)

    def train(self):
        """One epoch; returns the mean of the per-step losses (ClassifierExperiment.py:93-132)."""
        self.model.train()
        loss_list = []
        for index, batch in enumerate(self.train_dataloader):
            self.optimizer.zero_grad(set_to_none=True)
            outputs = self.model(batch)
            loss, labels = self.criterion(**outputs)
            loss.backward()
            self.optimizer.step()
            loss_list.append(loss.item())
        self.scheduler.step()
        epoch_loss = np.mean(loss_list)
        self.writer.add_scalar('loss/train', epoch_loss, self.current_epoch + 1)
        return epoch_loss

    def validate(self):
        """Validation loss (the prompt-AUROC part of ClassifierExperiment.py:134-289 is out of scope, see module doc)."""
        self.model.eval()
        loss_list = []
        with torch.no_grad():
            for batch in self.valid_dataloader:
                outputs = self.model(batch, validation=True)
                outputs.pop("text_embeddings2", None)
                loss, _ = self.criterion(**outputs) if "text_embeddings2" not in self.criterion.forward.__code__.co_varnames \
                    else create_loss("CLIPLoss")()(**outputs)
                loss_list.append(loss.item())
        val_loss = float(np.mean(loss_list)) if loss_list else float("nan")
        self.writer.add_scalar('loss/val', val_loss, self.current_epoch + 1)
        return val_loss

    def run(self):
        self._time_start = time.time()
        for self.current_epoch in range(self.config.scheduler.config.epochs):
            t0 = time.time()
            train_loss = self.train()
            val_loss = self.validate() if self.valid_dataloader is not None else train_loss
            mins, secs = epoch_time(t0, time.time())
            self.writer.add_scalar('lr', self.optimizer.param_groups[0]['lr'], self.current_epoch + 1)
            self.early_stopper(val_loss, self.current_epoch, self.model, self.optimizer, self.ckp_path)
            logger.info(f"Epoch: {self.current_epoch + 1:02} | Time: {mins}m {secs}s | train loss {train_loss:.4f} | "
                        f"val loss {val_loss:.4f}")
            if self.early_stopper.early_stop:
                logger.info("Early stopping")
                break
        self._time_end = time.time()
        logger.info(f"Run complete. Total time: {time.strftime('%H:%M:%S', time.gmtime(self._time_end - self._time_start))}")
