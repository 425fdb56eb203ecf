"""`MMGCLIP(config)` — drop-in for mmgclip/networks/mmgclip_model.py:12-166 on the MI355X kernels.

Same constructor, attributes (`config, device, text_encoder, image_projection_layer, text_projection_layer,
logit_scale`, `image_encoder` when configured), methods (`count_parameters`, `encode_images`, `encode_text`,
`forward(batch, **kwargs)`) and output-dict keys.  Differences are additive and config-gated:
  * image encoders that take pixels (`ConvNextTinyEncoder`, `ConvNextBaseEncoder`); `ConvNextTiny` keeps the reference
    behaviour (pre-extracted features pass through, no module built);
  * `networks.learnable_logit_scale` (default false = the reference's behaviour on a GPU, where `.to(device)` turns
    the Parameter into a constant tensor: SURVEY.md §0);
  * `forward(..., materialize_logits=False)` skips the two [n,n] matrices for the fused loss path.
"""
import os

import numpy as np

import torch
import torch.nn as nn

from .. import head
from ..utils.logger import logger
from .bert import EosPool
from .network_controller import getNetworkClass
from .projection_controller import get_projection_head

PIXEL_ENCODERS = ("ConvNextTinyEncoder", "ConvNextBaseEncoder", "ViTB16Encoder")


def _get(cfg, path, default=None):
    cur = cfg
    for k in path.split("."):
        try:
            cur = cur[k] if isinstance(cur, dict) else getattr(cur, k)
        except (KeyError, AttributeError):
            return default
    return cur


class MMGCLIP(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        assert config is not None, 'Error in initializing the model. Missing training config object.'
        self.config = config
        if not torch.cuda.is_available():
            logger.warning("WARNING: No CUDA device is found. The MI355X kernels have no CPU fallback.")
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

        enc_name = self.config.networks.image_encoder.name
        if enc_name == "ResNet50Encoder":
            self.image_encoder = getNetworkClass(enc_name)(
                pretrained=True, image_features_dimension=self.config.networks.image_encoder.image_features_dimension).to(self.device)
        elif enc_name in PIXEL_ENCODERS:
            ie = self.config.networks.image_encoder
            self.image_encoder = getNetworkClass(enc_name)(
                pretrained=_get(ie, "pretrained_path"), image_features_dimension=ie.image_features_dimension,
                in_chans=_get(ie, "in_chans", 1), scale16=_get(ie, "scale16", True), micro_batch=_get(ie, "micro_batch", 64),
                freeze=_get(ie, "freeze", False),
                checkpoint=_get(ie, "checkpoint", False),
                **({"image_size": _get(ie, "image_size", 224)} if enc_name == "ViTB16Encoder" else
                   {"fp8": _get(ie, "fp8", False)})).to(self.device)
            logger.info(f"Using {self.image_encoder.__class__.__name__}")

        te = self.config.networks.text_encoder
        self.text_encoder = getNetworkClass(te.name)(
            pretrained=self.config.tokenizer.config.tokenizer_name, freeze=_get(te, "freeze", True),
            random_init=_get(te, "random_init", False), dropout=bool(_get(te, "dropout", True))).to(self.device)

        if self.config.projection.config.projection_name != "ZeroProjection":
            head_cls = get_projection_head(self.config.projection.config.projection_name)
            self.image_projection_layer = head_cls(
                embedding_dim=self.config.networks.image_encoder.image_features_dimension,
                projection_dim=self.config.projection.config.output_projection_dimension,
                dropout=self.config.networks.dropout.config.dropout).to(self.device)
            self.text_projection_layer = head_cls(
                embedding_dim=self.text_encoder.model_output_dimension,
                projection_dim=self.config.projection.config.output_projection_dimension,
                dropout=self.config.networks.dropout.config.dropout).to(self.device)
            logger.info(f"Embeddings are projected to {self.config.projection.config.output_projection_dimension} "
                        f"features using {self.config.projection.config.projection_name}.")
        else:
            self.image_projection_layer = None
            self.text_projection_layer = None

        init = torch.ones([]) * np.log(1 / self.config.networks.logit_temperature)
        if _get(self.config, "networks.learnable_logit_scale", False) or self.device.type == "cpu":
            self.logit_scale = nn.Parameter(init.float().to(self.device))
        else:
            self.register_buffer("logit_scale", init.float().to(self.device), persistent=False)

    def count_parameters(self, model):
        """Logs a table of trainable parameters and returns their total count (mmgclip_model.py:54-74)."""
        rows, total = [], 0
        for name, parameter in model.named_parameters():
            if not parameter.requires_grad:
                continue
            rows.append((name, parameter.numel()))
            total += parameter.numel()
        width = max([len(r[0]) for r in rows] + [7])
        table = "\n".join([f"| {'Modules'.ljust(width)} | Parameters |"] + [f"| {n.ljust(width)} | {c:>10} |" for n, c in rows])
        logger.info(f"\n{table}")
        logger.info(f"Total Trainable Params: {total}")
        return total

    def encode_images(self, batch):
        """[n,1,F,1,1] pre-extracted features -> [n,F] (mmgclip_model.py:76-93); pixel encoders take `batch['image']`
        (or a 4-D `image_features`) and run the ConvNeXt tower."""
        name = self.config.networks.image_encoder.name
        if name in PIXEL_ENCODERS:
            pix = batch["image"] if "image" in batch else batch["image_features"]
            return self.image_encoder(pix.to(self.device))
        flattened_embeddings = torch.flatten(batch['image_features'].to(self.device), 1)
        if name == "ResNet50Encoder":
            return self.image_encoder(flattened_embeddings)
        return flattened_embeddings

    def encode_text(self, batch, text_pooling='eos'):
        """BERT last hidden state pooled at the [SEP] position = attention_mask.sum(-1) - 1 (mmgclip_model.py:95-115)."""
        src = batch['text_tokens']
        host_mask = src['attention_mask'] if ('attention_mask' in src and not src['attention_mask'].is_cuda) else None
        tokens = src.to(self.device) if hasattr(src, "to") else {k: v.to(self.device) for k, v in src.items()}
        if host_mask is not None and tokens['attention_mask'] is not host_mask:
            # prompt lengths read from the batch while it is still on the host: the unpadded text tower needs them there
            m = tokens['attention_mask']
            m._mmg_seq_lens = (m._version, type(self.text_encoder).sequence_lengths(host_mask))
        if isinstance(batch, dict):
            batch['text_tokens'] = tokens                      # the reference's BatchEncoding.to() is in-place
        hidden = self.text_encoder.hidden_states(tokens)       # bf16 [n*S, H]
        if text_pooling == 'eos':
            B, S = tokens['input_ids'].shape
            return EosPool.apply(hidden, tokens['attention_mask'], B, S)
        raise NotImplementedError(f"{text_pooling} method is not implemented...")

    # ---- the two towers on two HIP streams (default with a pixel image encoder; MMG_TEXT_STREAM=0 / networks.text_stream: false = off) ----
    # The text tower's kernels are small (~11 k tokens: its GEMMs fill half the GPU) next to the image tower's.  The text tower - every
    # pass of it, so that its backwards stay serialised on one stream - is enqueued on a side stream before the image tower goes onto the
    # current one, and autograd replays each backward on the stream of its forward, so both directions overlap (same-run A/B at C2:
    # 332.4 -> 327.5 ms/step).  Cross-stream tensors are registered with the allocator (record_stream); join_streams() makes the
    # current stream wait for the side stream (call it after backward, before optimizer.step()).
    text_stream_enabled = True      # set False on an instance to run both towers on the caller's stream (bench.py's profiled steps)

    def _text_stream(self):
        if not self.text_stream_enabled:
            return None
        on = os.environ.get("MMG_TEXT_STREAM")
        on = (on == "1") if on is not None else bool(_get(self.config.networks, "text_stream", True))
        if not on or not torch.cuda.is_available() or self.config.networks.image_encoder.name not in PIXEL_ENCODERS:
            return None
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream()
        return self._side_stream

    def join_streams(self):
        side = getattr(self, "_side_stream", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    def forward(self, batch, **kwargs):
        side = self._text_stream()
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)                                   # parameters updated by the optimizer, tokens copied by the caller
            # the image tower is enqueued first: its large kernels keep the GPU busy while the host is still launching the text
            # tower's several hundred small ones (MMG_TEXT_FIRST=1: the other order, for A/B runs)
            text_first = os.environ.get("MMG_TEXT_FIRST", "0") == "1"
            if not text_first:
                image_features = self.encode_images(batch)
            with torch.cuda.stream(side):
                text_features = self.encode_text(batch, text_pooling='eos')
            if text_first:
                image_features = self.encode_images(batch)
            main.wait_stream(side)
            text_features.record_stream(main)
        else:
            image_features = self.encode_images(batch)
            text_features = self.encode_text(batch, text_pooling='eos')

        image_embeddings = self.image_projection_layer(image_features) if self.image_projection_layer is not None else image_features
        text_embeddings = self.text_projection_layer(text_features) if self.text_projection_layer is not None else text_features

        image_embeddings = head.L2Normalize.apply(image_embeddings)        # :128
        text_embeddings = head.L2Normalize.apply(text_embeddings)          # :129
        logit_scale = self.logit_scale.exp()                               # :132

        output = {"image_embeddings": image_embeddings, "text_embeddings": text_embeddings, "logit_scale": logit_scale}
        if kwargs.get("materialize_logits", True):
            li, lt = head.ScaledLogits.apply(image_embeddings, text_embeddings, logit_scale)   # :135-136
            output["logits_per_image"], output["logits_per_text"] = li, lt

        if self.config.loss.config.loss_name == "MMGCLIPLoss" and not kwargs.get('validation', False) == True:  # noqa: E712
            batch['text_tokens'] = batch['image_impression_tokens']        # :160 (mutates the batch, like the reference)
            if side is not None:                                           # the same tower: same stream as its first pass
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    text_features2 = self.encode_text(batch, text_pooling='eos')
                torch.cuda.current_stream().wait_stream(side)
                text_features2.record_stream(torch.cuda.current_stream())
            else:
                text_features2 = self.encode_text(batch, text_pooling='eos')
            text_embeddings2 = self.text_projection_layer(text_features2)
            output['text_embeddings2'] = head.L2Normalize.apply(text_embeddings2)
        return output


class PromptClassifier(nn.Module):
    """Zero-shot classifier wrapper around an MMGCLIP model — drop-in for mmgclip/networks/mmgclip_model.py:168-257.

    `forward(image_features, class_list, visualize, image_id, ground_truth)` tokenises the class prompts
    (`padding="max_length"`, `max_length = config.tokenizer.config.sequence_length`), runs the model in eval mode without
    gradients and returns `classes_similarities = softmax(logits_per_image)` [n, k], `similarities_argmax` (of the first
    image, like the reference) and `class_list`.  `tokenizer` (additive argument): a callable with the HF tokenizer call
    signature; by default `AutoTokenizer.from_pretrained(config.tokenizer.config.tokenizer_name)` as in the reference
    (`:185`), and - only where that vocabulary cannot be loaded (offline box) - the hashed stand-in ids of
    `dataset/synthetic.py`, with a warning.  The bar plot (`:212-255`) is drawn when matplotlib is importable.
    """

    def __init__(self, model=None, tokenizer=None):
        super().__init__()
        self.model = model
        self.device = model.device
        self.tokenizer = tokenizer
        if self.tokenizer is None:
            name = self.model.config.tokenizer.config.tokenizer_name
            try:
                from transformers import AutoTokenizer
                self.tokenizer = AutoTokenizer.from_pretrained(pretrained_model_name_or_path=name)
            except Exception as e:      # no vocabulary on an offline machine
                logger.warning(f"tokenizer `{name}` unavailable ({type(e).__name__}); using hashed stand-in token ids")

    def _tokens(self, class_list):
        S = self.model.config.tokenizer.config.sequence_length
        if self.tokenizer is not None:
            return self.tokenizer(class_list, padding="max_length", truncation=True, return_tensors="pt", max_length=S)
        from ..dataset.synthetic import synthetic_prompt_tokens
        vocab = self.model.text_encoder.model.embeddings.word_embeddings.weight.shape[0]
        return synthetic_prompt_tokens(list(class_list), S, vocab_size=vocab)

    def forward(self, image_features, class_list, visualize=True, image_id=None, ground_truth=None):
        inputs = {"image_features": image_features, "text_tokens": self._tokens(class_list)}
        self.model.eval()
        with torch.no_grad():
            classes_similarities = self.model(inputs)['logits_per_image']        # images are rows, prompts are columns
            classes_similarities = classes_similarities.softmax(dim=-1)
        outputs = {"classes_similarities": classes_similarities,
                   "similarities_argmax": torch.argmax(classes_similarities, dim=-1)[0].item(),
                   "class_list": class_list}
        if visualize:
            assert image_id is not None, "For visualizing results, image_id value is required."
            try:
                import matplotlib.pyplot as plt
            except ImportError:
                logger.warning("matplotlib is not installed: skipping the probability plot")
                return outputs
            probs = classes_similarities.detach().cpu().numpy().reshape(-1)[:len(class_list)]
            y = np.arange(len(class_list))
            plt.figure(figsize=(7, 6))
            plt.barh(y, probs)
            plt.gca().invert_yaxis()
            plt.yticks(y, class_list)
            plt.xlabel("probability")
            plt.title(f"{image_id}" + (f"  TP: {ground_truth}" if ground_truth else ""))
            plt.show()
        return outputs
