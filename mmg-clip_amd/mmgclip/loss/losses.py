"""Loss classes with the reference's names, constructor and call signatures (mmgclip/loss/losses.py:6-216), computing
on the fp32 MI355X head kernels.  All return `(loss: 0-d tensor, labels: int64 [n])`.

`criterion(**outputs)` works exactly as in mmgclip/experiments/ClassifierExperiment.py:112.  When the model was run
with `materialize_logits=False` the CLIP loss takes the fused path (no [n,n] matrices in HBM)."""
import torch
import torch.nn as nn

from .. import head


class CLIPLoss(nn.Module):
    """(CE(logits_per_image, arange) + CE(logits_per_text, arange)) / 2 — losses.py:28-44 (labels live on the logits'
    device instead of the reference's hard-coded `.cuda()`)."""

    def __init__(self, comm=None):
        super().__init__()
        self.comm = comm          # mmgclip.distributed.Comm for the global-batch loss (None = local batch)

    def forward(self, logits_per_image=None, logits_per_text=None, **kwargs):
        if logits_per_image is None or (self.comm is not None and self.comm.active):
            ie, te, s = kwargs["image_embeddings"], kwargs["text_embeddings"], kwargs["logit_scale"]
            loss = head.fused_clip_loss(ie, te, s, self.comm)
            n = ie.shape[0]
            off = 0 if self.comm is None else self.comm.rank * n
            return loss, torch.arange(off, off + n, device=ie.device)
        n, _ = logits_per_image.shape
        labels = torch.arange(n, device=logits_per_image.device)
        loss_i = head.cross_entropy(logits_per_image)
        loss_t = head.cross_entropy(logits_per_text)
        return (loss_i + loss_t) / 2, labels


class MMGCLIPLoss(nn.Module):
    """CLIP loss on recomputed logits + t2t_weight * symmetric CE between the two text views — losses.py:57-96."""

    def __init__(self, t2t_weight=0.5, comm=None):
        super().__init__()
        self.t2t_weight = t2t_weight
        self.comm = comm

    def forward(self, image_embeddings, text_embeddings, text_embeddings2, logit_scale, **kwargs):
        loss_clip = head.fused_clip_loss(image_embeddings, text_embeddings, logit_scale, self.comm)
        loss_t2t = head.fused_clip_loss(text_embeddings2, text_embeddings, logit_scale, self.comm)
        n = image_embeddings.shape[0]
        return loss_clip + loss_t2t * self.t2t_weight, torch.arange(n, device=image_embeddings.device)


class AveragedMedicalCLIPLoss(nn.Module):
    """Text-similarity label clustering + column-averaged logits — losses.py:98-216.

    Everything runs on the device: similarities (fp32 MFMA head kernel), the greedy threshold clustering
    (`mmg_greedy_threshold_labels`: the reference's Python double loop with one device read per element, losses.py:148-162,
    as one kernel; the only host traffic is the 4-byte cluster count), the per-cluster column means and both cross-entropies.
    `_assign_labels` / `_average_logits` keep the reference's signatures (a list of labels in and out) for callers that use
    the helpers directly.
    """

    def __init__(self, similarity_threshold=0.65):
        super().__init__()
        self.similarity_threshold = similarity_threshold

    def _mesaure_embeddings_similarity(self, embeddings):
        e = head.L2Normalize.apply(embeddings.detach())
        one = torch.ones(1, device=e.device)
        _, _, sim = head.rows_forward(e.contiguous(), e.contiguous(), one, 0, True)
        return sim

    def _assign_labels(self, cosine_sim_matrix, threshold=0.65):
        labels, _, _ = head.greedy_threshold_labels(cosine_sim_matrix, threshold)
        return labels.tolist()

    def _average_logits(self, logits, list_labels):
        labels = torch.as_tensor(list_labels, device=logits.device, dtype=torch.int64)
        k = int(labels.max().item()) + 1
        counts = torch.bincount(labels, minlength=logits.shape[1]).to(torch.int32)
        return head.ClusterMeanCols.apply(logits, labels, counts, k)

    def forward(self, image_embeddings, text_embeddings, logit_scale, logits_per_image, logits_per_text, **kwargs):
        sim = self._mesaure_embeddings_similarity(text_embeddings)
        labels, counts, k = head.greedy_threshold_labels(sim, self.similarity_threshold)
        averaged = head.ClusterMeanCols.apply(logits_per_image, labels, counts, k)
        loss_i = head.cross_entropy(averaged, labels)
        loss_t = head.cross_entropy(logits_per_text, labels)      # as the reference: [n,n] logits vs cluster ids
        return (loss_i + loss_t) / 2, labels
