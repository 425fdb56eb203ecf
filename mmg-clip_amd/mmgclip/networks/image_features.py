"""Offline image-feature extraction with the reference's API (mmgclip/networks/image_features.py:11-122): one
`<image>.pth` file holding a `[1, 768, 1, 1]` fp32 tensor per input image, produced by `features -> avgpool` of the
ConvNeXt-T archive on the full-resolution, single-channel image after `x*65535`, `(x - 32767.5)/32767.5` (:95-101).
This is the on-disk format the reference-faithful training path reads (mmgclip/dataset/dataset.py:336).

Differences: the encoder is the HIP ConvNeXt tower loaded from a torchvision-layout state dict (not TorchScript); images of
any size >= 32x32 are accepted (strided layers floor, e.g. 1906x818 -> 59x25 -> pooled), and several images of equal size
can go through one launch.  Failures are appended to `failed.txt` like the reference does (:119-122).
"""
import os

import numpy as np
import torch

from ..utils.global_utils import create_directory_if_not_exists
from ..utils.logger import logger
from .encoder import ConvNextTiny


def load_image(path):
    """PIL image -> fp32 [1, H, W] in [0, 1] (what torchvision.transforms.ToTensor yields for 8- and 16-bit PNGs)."""
    from PIL import Image
    img = Image.open(path)
    arr = np.array(img)
    if arr.ndim == 3:
        arr = arr[..., 0]
    if arr.dtype == np.uint8:
        x = arr.astype(np.float32) / 255.0
    elif arr.dtype in (np.uint16, np.int32, np.int16):
        x = arr.astype(np.float32) / 65535.0
    else:
        x = arr.astype(np.float32)
    return torch.from_numpy(x)[None]


class ImageFeatureExtractor:
    def __init__(self, config=None, dataset=None):
        assert config is not None, 'Error in initializing the feature extractor. Missing training config object.'
        self.config = config
        self.dataset = self._validate_dataset(dataset)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.image_encoder = ConvNextTiny()
        self.image_encoder.from_pretrained(self.config.networks.image_encoder.convnext_tiny_clf_path)
        self.image_encoder.to(self.device).eval()
        self.export_dir = create_directory_if_not_exists(self.config.base.features_export_dir)

    def _validate_dataset(self, dataset):
        import pandas as pd
        if isinstance(dataset, pd.DataFrame):
            if 'image_path' not in dataset.columns:
                raise ValueError("Error in the `dataset` dataframe passed. The dataframe doesn't contain the column `image_path`.")
        elif isinstance(dataset, str):
            raise NotImplementedError('Handling a string directory dataset is not yet implemented.')
        else:
            raise ValueError("Missing value for `dataset`. Please pass a valid Path or a dataset dataframe.")
        return dataset

    def _export_name(self, image_path):
        rel = image_path.split('2D_100micron/')[-1] if '2D_100micron/' in image_path else os.path.basename(image_path)
        return os.path.join(self.export_dir, os.path.splitext(rel)[0] + '.pth')

    def extract(self):
        logger.info(f"Extracting and exporting features into {self.export_dir} directory.")
        with torch.no_grad():
            for _, row in self.dataset.iterrows():
                try:
                    image = load_image(row['image_path']).unsqueeze(0).to(self.device)        # [1,1,H,W] in [0,1]
                    image = (65535.0 * image - 32767.5) / 32767.5                              # :95-99
                    features = self.image_encoder(image)                                      # [1,768,1,1]
                    out = self._export_name(row['image_path'])
                    create_directory_if_not_exists(os.path.dirname(out))
                    torch.save(features.detach().cpu(), out)
                except Exception as e:                                                        # noqa: BLE001 (as the reference)
                    with open(os.path.join(self.export_dir, 'failed.txt'), "a") as fh:
                        fh.write(row['image_path'] + '\n' + str(e) + '\n\n')


class StudyFeatureExtractor:
    """Per-exam features (reference image_features.py:124-263): the first `n_images_per_study` files of each `study_path`
    go through the tower, then `concatenate_features_method` in {maxpool, avgpool, stack, concat} joins the [768] vectors;
    one `<8-digit patient id>.pth` per study.  Views of one study that share a size run as one tower launch."""

    def __init__(self, config=None, dataset=None):
        assert config is not None, 'Error in initializing the feature extractor. Missing training config object.'
        self.config = config
        self.dataset = self._validate_dataset(dataset)
        self.export_dir = os.path.join(self.config.base.features_export_dir)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.image_encoder = ConvNextTiny()
        self.image_encoder.from_pretrained(self.config.networks.image_encoder.convnext_tiny_clf_path)
        self.image_encoder.to(self.device).eval()

    def _validate_dataset(self, dataset):
        import pandas as pd
        if isinstance(dataset, pd.DataFrame):
            if 'study_path' not in dataset.columns:
                raise ValueError("Error in the `dataset` dataframe passed. The dataframe doesn't contain the following column `study_path`.")
        elif isinstance(dataset, str):
            raise NotImplementedError('Handling a string directory dataset is not yet implemented.')
        else:
            raise ValueError("Missing value for `dataset`. Please pass a valid Path or a dataset dataframe.")
        return dataset

    def _get_patient_id(self, path):
        import re
        match = re.search(r'\d{8}', path)
        if match:
            return match.group()

    def _encode_views(self, paths):
        views = [(65535.0 * load_image(p) - 32767.5) / 32767.5 for p in paths]
        feats = [None] * len(views)
        by_size = {}
        for i, v in enumerate(views):
            by_size.setdefault(tuple(v.shape), []).append(i)
        for idx in by_size.values():
            out = self.image_encoder(torch.stack([views[i] for i in idx]).to(self.device))   # [k,768,1,1]
            for j, i in enumerate(idx):
                feats[i] = out[j].reshape(-1)
        return feats

    def extract(self):
        dcfg = self.config.dataset.config
        method, n_views = dcfg.concatenate_features_method, dcfg.n_images_per_study
        logger.info(f"Extracting and exporting features into {self.export_dir} directory.")
        logger.info(f"Concatenating {n_views} images using {method} method.")
        with torch.no_grad():
            for _, row in self.dataset.iterrows():
                study_path = row['study_path']
                try:
                    names = os.listdir(study_path)[:n_views]
                    feats = self._encode_views([os.path.join(study_path, n) for n in names])
                    if method == "maxpool":
                        joint = torch.stack(feats, dim=0).max(dim=0)[0]
                    elif method == "concat":
                        joint = torch.cat(feats, dim=0)
                    elif method == "stack":
                        joint = torch.stack(feats, dim=0)
                    elif method == "avgpool":
                        joint = torch.stack(feats, dim=0).mean(dim=0)
                    else:
                        raise ValueError("Not implemented feature vector concatenation method")
                    rel = study_path.split('2D_100micron/')[-1] if '2D_100micron/' in study_path else os.path.basename(study_path.rstrip('/'))
                    out_dir = os.path.join(self.export_dir, rel)
                    create_directory_if_not_exists(out_dir)
                    torch.save(joint.detach().cpu(), os.path.join(out_dir, '{}.pth'.format(self._get_patient_id(path=study_path))))
                except Exception as e:                                                        # noqa: BLE001 (as the reference)
                    with open(os.path.join(self.export_dir, 'failed.txt'), "a") as fh:
                        fh.write(study_path + '\n' + str(e) + '\n\n')


image_feature_extractor = ImageFeatureExtractor
study_feature_extractor = StudyFeatureExtractor
