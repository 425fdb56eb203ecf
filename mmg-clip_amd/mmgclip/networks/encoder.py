"""Encoder classes with the reference's names and call signatures (mmgclip/networks/encoder.py:15-156), computing on the
HIP towers (convnext.py / bert.py).

  ConvNextTiny      offline loader API of the reference (`from_pretrained(model_path)`, `forward(x) -> [B,768,1,1]`);
                    accepts a torchvision-layout state dict (.pth/.pt with `features.*` keys) instead of TorchScript.
  ResNet50Encoder   present for API completeness; the torchvision ResNet path is out of this hot path's scope
                     and raises on construction.
  BertEncoder       HF-layout BERT; `forward(x: mapping) -> last_hidden_state [B,S,H]`; frozen by default as in the
                    reference (encoder.py:141-142).
New in-graph image encoders (SURVEY.md §7 decision 1): ConvNextTinyEncoder, ConvNextBaseEncoder — take pixels.
"""
import json
import os

import torch
import torch.nn as nn

from ..utils.logger import logger
from .bert import BertConfigLite, BertTower
from .convnext import ConvNextTower
from .resnet import ResNetTower
from .vit import ViTTower

PROJECTION_HEAD_DIM = 512


def _load_state_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


class ConvNextTiny(nn.Module):
    """`model = None` until `from_pretrained(path)`; `forward(x[B,C,H,W]) -> [B,768,1,1]` (features -> avgpool)."""

    def __init__(self):
        super().__init__()
        self.model = None
        self._tower = None

    def from_pretrained(self, model_path=None):
        assert os.path.isfile(model_path), "Model `.pt` file doesn't exist."
        sd = _load_state_file(model_path)
        sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
        in_chans = sd["features.0.0.weight"].shape[1]
        self._tower = ConvNextTower("tiny", in_chans=in_chans, scale16=False)
        self._tower.model.load_state_dict({k: v for k, v in sd.items() if k.startswith("features.")}, strict=True)
        for p in self._tower.parameters():
            p.requires_grad = False
        self.model = self._tower.model
        return self.model

    def forward(self, x):
        if self.model is None:
            raise ImportError("Model was not loaded correctly. Call `from_pretrained` and pass the model file path first.")
        return self._tower(x).reshape(x.shape[0], -1, 1, 1)


class ResNet50Encoder(ResNetTower):
    """`ResNet50Encoder(pretrained=True, image_features_dimension=768)`; attrs `model`, `model_output_dimension` = 2048; everything
    frozen except `layer4` (mmgclip/networks/encoder.py:57-89).  `pretrained` may be a path to a torchvision resnet50 state dict
    (.pth / .safetensors); the hub download of the reference (`models.resnet50(pretrained=True)`) is impossible offline, so
    `pretrained=True` without MMGCLIP_RESNET50_WEIGHTS set means torchvision's random initialisation (logged)."""

    def __init__(self, pretrained=True, image_features_dimension=768):
        super().__init__()
        logger.info("Initializing 'resnet50' as the image encoder.")
        path = pretrained if isinstance(pretrained, str) else os.environ.get("MMGCLIP_RESNET50_WEIGHTS")
        if path:
            assert os.path.isfile(path), f"ResNet-50 weights not found: {path}"
            sd = _load_state_file(path)
            sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
            self.model.load_state_dict(sd, strict=True)
        elif pretrained:
            logger.info("no local ResNet-50 weights (MMGCLIP_RESNET50_WEIGHTS): random initialisation")


class _ConvNextEncoder(ConvNextTower):
    VARIANT = "tiny"

    def __init__(self, pretrained=None, image_features_dimension=None, in_chans=1, scale16=True, micro_batch=64, freeze=False,
                 checkpoint=False, fp8=False):
        super().__init__(self.VARIANT, in_chans=in_chans, scale16=scale16, micro_batch=micro_batch, checkpoint=checkpoint,
                         fp8=fp8)
        if isinstance(pretrained, str) and os.path.isfile(pretrained):
            sd = _load_state_file(pretrained)
            sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
            self.model.load_state_dict({k: v for k, v in sd.items() if k.startswith("features.")}, strict=True)
        if image_features_dimension is not None and image_features_dimension != self.model_output_dimension:
            raise ValueError(f"{type(self).__name__} produces {self.model_output_dimension} features, config asks for "
                             f"{image_features_dimension}")
        if freeze:
            for p in self.parameters():
                p.requires_grad = False
        logger.info(f"Initializing '{type(self).__name__}' as the image encoder.")


class ConvNextTinyEncoder(_ConvNextEncoder):
    VARIANT = "tiny"


class ConvNextBaseEncoder(_ConvNextEncoder):
    VARIANT = "base"


class ViTB16Encoder(ViTTower):
    """ViT-B/16 on raw pixels (BASELINE config C4; torchvision `vit_b_16` state-dict layout, class-token pooling)."""

    def __init__(self, pretrained=None, image_features_dimension=None, in_chans=1, scale16=True, micro_batch=256, freeze=False,
                 image_size=224, layers=12, checkpoint=False):
        super().__init__(image_size=image_size, in_chans=in_chans, layers=layers, scale16=scale16, micro_batch=micro_batch,
                         checkpoint=checkpoint)
        if isinstance(pretrained, str) and os.path.isfile(pretrained):
            sd = _load_state_file(pretrained)
            self.model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("heads.")}, strict=True)
        if image_features_dimension is not None and image_features_dimension != self.model_output_dimension:
            raise ValueError(f"ViTB16Encoder produces {self.model_output_dimension} features, config asks for {image_features_dimension}")
        if freeze:
            for p in self.parameters():
                p.requires_grad = False
        logger.info("Initializing 'ViTB16Encoder' as the image encoder.")


class BertEncoder(BertTower):
    """`BertEncoder(pretrained=<local dir | hub name>)`; attrs `model`, `model_output_dimension` (encoder.py:131-144).

    `pretrained` may be a local directory holding `config.json` and `model.safetensors` / `pytorch_model.bin`
    (HF layout).  `dropout` (additive knob, networks.text_encoder.dropout, default true): HF's training-mode dropout, as the
    reference runs it (model.train(), ClassifierExperiment.py:97); eval() switches it off.  A hub name cannot be fetched offline: it falls back to a seeded random initialisation of the
    Bio_ClinicalBERT architecture only when `random_init=True` (or MMGCLIP_RANDOM_INIT=1), otherwise raises.
    """

    def __init__(self, pretrained=None, freeze=True, random_init=False, config=None, micro_batch=4096, dropout=True):
        cfg, sd = config, None
        if isinstance(pretrained, str) and os.path.isdir(pretrained):
            with open(os.path.join(pretrained, "config.json")) as fh:
                cfg = BertConfigLite(**json.load(fh))
            for fn in ("model.safetensors", "pytorch_model.bin"):
                if os.path.isfile(os.path.join(pretrained, fn)):
                    sd = _load_state_file(os.path.join(pretrained, fn))
                    break
            if sd is None:
                raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {pretrained}")
        elif not (random_init or os.environ.get("MMGCLIP_RANDOM_INIT") == "1"):
            raise OSError(f"cannot fetch '{pretrained}' (no network): pass a local Hugging Face directory, or set "
                          f"networks.text_encoder.random_init / MMGCLIP_RANDOM_INIT=1 for seeded random weights")
        super().__init__(cfg or BertConfigLite(), micro_batch=micro_batch, dropout=dropout)
        logger.info(f"Initializing pretrained `{pretrained}` as the text encoder and tokenizer.")
        if sd is not None:
            sd = {k[len("bert."):] if k.startswith("bert.") else k: v for k, v in sd.items()}
            missing, unexpected = self.model.load_state_dict(sd, strict=False)
            missing = [k for k in missing if not k.startswith("pooler.")]
            if missing:
                raise KeyError(f"checkpoint lacks {missing[:5]} ...")
        if freeze:                                     # encoder.py:141-142
            for param in self.model.parameters():
                param.requires_grad = False

    def hidden_states(self, x, packed=None):
        """fp32 [B*S, H] last hidden state (device layout, autograd-connected).  By default only the rows of valid tokens are
        computed (right-padded prompts; padding rows come back as zeros) - all that EOS pooling reads."""
        return BertTower.forward(self, x["input_ids"], x.get("attention_mask"), x.get("token_type_ids"), packed=packed)

    def forward(self, x):
        """x: mapping with input_ids / attention_mask / token_type_ids -> last_hidden_state fp32 [B,S,H] (encoder.py:156),
        padding positions included (HF computes them too)."""
        B, S = x["input_ids"].shape
        return self.hidden_states(x, packed=False).float().view(B, S, self.config.hidden_size)
