import torch


class EarlyStopper:
    """Patience counter on the validation loss + checkpoint writer; state machine and checkpoint dict layout of
    mmgclip/callbacks/early_stopping.py:5-66 ({'epoch','model_state_dict','optimizer_state_dict','val_loss',
    'best_score','counter'}, extra `<epoch>_model.pth` every 100 epochs)."""

    def __init__(self, patience=5, verbose=False, delta=0, trace_func=print, save=True):
        self.patience, self.verbose, self.delta, self.trace_func = patience, verbose, delta, trace_func
        self.save = save          # additive: False on the non-zero ranks of a data-parallel run (one writer per file)
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = float('inf')

    def __call__(self, validation_loss, epoch, model, optimizer, path):
        score = -validation_loss
        improved = self.best_score is None or not (score < self.best_score + self.delta)
        if improved:
            first = self.best_score is None
            self.best_score = score
            self.save_checkpoint(validation_loss, model, optimizer, epoch, path)
            if not first:
                self.counter = 0
        else:
            self.counter += 1
            self.trace_func(f'EarlyStopping counter: {self.counter} out of {self.patience}')
            if self.counter >= self.patience:
                self.early_stop = True

    def save_checkpoint(self, valid_loss, model, optimizer, epoch, path):
        self.trace_func(f"Valid loss improved from {self.val_loss_min:.6f} to {valid_loss:.6f}. Saving model ...")
        if not self.save:
            self.val_loss_min = valid_loss
            return
        checkpoint = {'epoch': epoch, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(),
                      'val_loss': valid_loss, 'best_score': self.best_score, 'counter': self.counter}
        torch.save(checkpoint, path)
        if (epoch != 0) and (epoch % 100 == 0):
            torch.save(checkpoint, path.replace('model.pth', f'{epoch}_model.pth'))
        self.val_loss_min = valid_loss
