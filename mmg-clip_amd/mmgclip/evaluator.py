"""`Evaluator` — the scoring half of mmgclip/evaluator.py: `encode_text`, `encode_image`, `calculate_ci`
and `zeroshot_label_prompt` (evaluator.py:67-94, 321-478) on the MI355X path.

What runs where: the k class prompts go through the text tower ONCE ([k,512] normalised embeddings); every batch of image
embeddings is scored by the `[n,512] x [512,k]` logit kernel of the contrastive head (`mmg_clip_rows_fwd`, the same kernel the
training loss uses) with `exp(logit_scale)` folded in; softmax, AUROC / accuracy / F1 and the 1000-sample bootstrap of the
binary tasks stay on the host (scipy / sklearn), as in the reference.  Returns the reference's `results` dict:
`{prompt: {'auc', 'accuracy'}, ..., ['auc_ci_mean', 'auc_ci_lower', 'auc_ci_higher' for two-prompt tasks], 'accuracy',
'f1score'}`.  `evaluate_experiment` (evaluator.py:564-654) walks the test loader once (image embeddings from the device, prompt
labels from the batches), dispatches over `config.dataset.eval.enum_classes` x `config.dataset.eval.method` and writes `results.txt`
under `config.base.results_export_dir`.  Not reproduced: the ROC / histogram PNG files (evaluator.py:386-460, plotting), the PrettyTable
object of `zeroshot_eval` (a list of row dicts with the same columns instead) and `clf_conf_matrix` ("confustion_matrix": it scores the
image classifier head of the ConvNeXt TorchScript archive, which the repo does not ship) - configured there, it is logged and skipped.
"""
import numpy as np
import torch

from . import head
from .utils.logger import logger

# prompt templates per label key (evaluator.py:333-345); None = built from the class names
LABEL_PROMPTS = {
    "BenignMalignantDatasetLabels": "Finding suggesting {}.",
    "MassShapeLabels": "Mass shape is {}.",
    "MassMarginLabels": "Mass margin is {}.",
    "HasMassLabels": ["No mass was observed.", "Findings revealed a mass."],
    "HasArchDistortion": ["Normal architecture is visible.", "Displayed architectural distortion."],
    "HasCalcification": ["No calcifications are present.", "Finding suggesting calcifications."],
}
# label enums of the reference (mmgclip/prompts/enums.py:13-43): name -> value; `evaluate_experiment` looks them up by class name
ENUMS = {
    "HasArchDistortion": {"noarchitecturaldistortion": 0, "displayedarchitecturaldistortion": 1},
    "BenignMalignantDatasetLabels": {"benign": 0, "malignant": 1},
    "HasMassLabels": {"nomass": 0, "mass": 1},
    "HasCalcification": {"negative": 0, "hascalcification": 1},
    "MassShapeLabels": {"unknown": 0, "oval": 1, "round": 2, "irregular": 3},
    "MassMarginLabels": {"unknown": 0, "circumscribed": 1, "obscured": 2, "spiculated": 3, "illdefined": 4},
}
# enum member name -> the wording the model was trained on (mmgclip/utils/data_utils.py:921-960)
_CLASS_WORDING = {"illdefined": "ill defined", "nomass": "no mass", "noncalcified": "non-calcified", "hascalcification": "has calcification",
                  "noarchitecturaldistortion": "no architectural distortion",
                  "displayedarchitecturaldistortion": "displayed architectural distortion"}


def process_class_list(class_list):
    if not isinstance(class_list, list):
        raise ValueError("`class_list` has to be a list of classes.")
    return [_CLASS_WORDING.get(c, c) for c in class_list]


def label_prompts(key, classes_dict):
    """The sentences `zeroshot_label_prompt` scores for label `key` (evaluator.py:329-345)."""
    tpl = LABEL_PROMPTS[key]
    if isinstance(tpl, list):
        return list(tpl)
    return [tpl.format(name) for name in process_class_list(list(classes_dict.keys()))]


class Evaluator:
    def __init__(self, config, test_dataloader=None, tokenizer=None, model=None):
        """`model`: a trained MMGCLIP instance; without one the checkpoint at config.checkpoints.* is loaded (evaluator.py:43-57)."""
        import os
        from .networks.mmgclip_model import MMGCLIP
        self.config = config
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.test_dataloader = test_dataloader
        self.tokenizer = tokenizer
        if model is not None:
            self.model = model
        else:
            ckp_path = os.path.join(config.checkpoints.checkpoints_export_dir, config.checkpoints.checkpoints_file_name)
            assert os.path.isfile(ckp_path), f'Checkpoint file path "{ckp_path}" does not exist.'
            ckp = torch.load(ckp_path, weights_only=False)
            self.model = MMGCLIP(config=config).to(self.device)
            self.model.load_state_dict(ckp["model_state_dict"])
        self.model.eval()
        self._has_proj = config.projection.config.projection_name != "ZeroProjection"

    # ---- embeddings (evaluator.py:67-87); `as_numpy=False` keeps them on the device for the scoring kernel ------------------
    def _tokens(self, strings):
        S = self.config.tokenizer.config.sequence_length
        if self.tokenizer is not None:
            return self.tokenizer(strings, padding="longest", truncation=True, return_tensors="pt", max_length=S)
        from .dataset.synthetic import synthetic_prompt_tokens
        vocab = self.model.text_encoder.model.embeddings.word_embeddings.weight.shape[0]
        return synthetic_prompt_tokens(list(strings), S, vocab_size=vocab)

    def encode_text(self, text_tokens, as_numpy=True):
        if isinstance(text_tokens, (str, list)):
            text_tokens = {"text_tokens": self._tokens([text_tokens] if isinstance(text_tokens, str) else text_tokens)}
        with torch.no_grad():
            emb = self.model.encode_text(text_tokens, text_pooling='eos')
            emb = self.model.text_projection_layer(emb) if self._has_proj else emb
            emb = head.L2Normalize.apply(emb)
        return emb.detach().cpu().numpy() if as_numpy else emb

    def encode_image(self, batch, as_numpy=True):
        with torch.no_grad():
            emb = self.model.encode_images(batch)
            emb = self.model.image_projection_layer(emb) if self._has_proj else emb
            emb = head.L2Normalize.apply(emb)
        return emb.detach().cpu().numpy() if as_numpy else emb

    def calculate_ci(self, scores):
        sorted_scores = np.sort(scores)
        return np.mean(scores), sorted_scores[int(0.025 * len(sorted_scores))], sorted_scores[int(0.975 * len(sorted_scores))]

    # ---- scoring ------------------------------------------------------------------------------------------------------------
    def prompt_similarities(self, image_embeddings, prompts, use_logits=True):
        """[n,k] `exp(logit_scale) * I @ T.t()` (use_logits) or the plain cosine similarity, before the softmax."""
        ie = torch.as_tensor(image_embeddings, dtype=torch.float32, device=self.device).contiguous()
        te = self.encode_text(prompts, as_numpy=False).contiguous()
        scale = self.model.logit_scale.exp().reshape(1).float() if use_logits else torch.ones(1, device=self.device)
        _, _, sims = head.rows_forward(ie, te, scale.contiguous(), 0, True)
        return sims.cpu().numpy()

    def zeroshot_label_prompt(self, image_embeddings, label_names, classes_dict, key, use_logits=True, n_iterations=1000):
        from scipy.special import softmax
        from sklearn import metrics
        logger.info(f"Evaluating zero-shot prompt configuration for {key}.")
        label_names = [process_class_list([label[key]]) for label in label_names]
        prompts = label_prompts(key, classes_dict)
        similarities = softmax(self.prompt_similarities(image_embeddings, prompts, use_logits), axis=1)
        y_true = np.array([classes_dict[label[0].replace(' ', '').replace('-', '')] for label in label_names])
        y_pred = np.argmax(similarities, axis=-1)
        results = {}
        for idx, value in enumerate(prompts):
            results[value] = {'auc': metrics.roc_auc_score(y_true == idx, similarities[:, idx]),
                              'accuracy': np.mean((y_pred == idx) == (y_true == idx))}
        if len(prompts) == 2:                                   # bootstrap CI of the positive class's AUROC (:417-468)
            pos = similarities[:, 1]
            scores = []
            for _ in range(n_iterations):
                indices = np.random.choice(len(pos), len(pos), replace=True)
                if len(np.unique(y_true[indices])) == 2:
                    scores.append(metrics.roc_auc_score(y_true[indices] == 1, pos[indices]))
            results['auc_ci_mean'], results['auc_ci_lower'], results['auc_ci_higher'] = self.calculate_ci(scores)
        results['accuracy'] = metrics.accuracy_score(y_true, y_pred)
        results['f1score'] = metrics.f1_score(y_true, y_pred, average='binary' if len(prompts) <= 2 else 'micro')
        return results

    def zeroshot_eval(self, image_embeddings, label_names, classes_dict, key, use_logits=True):
        """Class-wise "No <class>" / "<class>" prompt pairs (evaluator.py:258-319): rows of {Class, AUROC, Accuracy, F1}."""
        from scipy.special import softmax
        from sklearn import metrics
        logger.info(f"Evaluating zero-shot prompt configuration for {key}.")
        label_names = [process_class_list([label[key]]) for label in label_names]
        rows = []
        for class_name in process_class_list(list(classes_dict.keys())):
            sims = softmax(self.prompt_similarities(image_embeddings, [f'No {class_name}', f'{class_name}'], use_logits), axis=1)
            y_true = np.array([1 if class_name in label else 0 for label in label_names])
            fpr, tpr, _ = metrics.roc_curve(y_true, sims[:, 1])
            y_pred = np.argmax(sims, axis=1)
            rows.append({"Class": class_name, "AUROC": metrics.auc(fpr, tpr), "Accuracy": metrics.accuracy_score(y_true, y_pred),
                         "F1": metrics.f1_score(y_true, y_pred)})
        return rows

    @staticmethod
    def _named_labels(prompt_labels, key, classes_dict):
        """The reference's datasets hand label NAMES over (`label[key]` is an enum member's name, evaluator.py:326); synthetic
        loaders hand the enum VALUE: both are accepted."""
        by_value = {v: k for k, v in classes_dict.items()}
        return [{**label, key: by_value[label[key]] if not isinstance(label[key], str) else label[key]} for label in prompt_labels]

    def evaluate_experiment(self):
        """The end-of-run test pass (evaluator.py:564-654): returns the list of results it also writes to results.txt."""
        import os
        from .utils.global_utils import create_directory_if_not_exists
        self.model.eval()
        prompt_labels, image_embeddings = [], []
        with torch.no_grad():
            for batch in self.test_dataloader:
                prompt_labels.extend(batch['prompt_labels'])
                image_embeddings.append(self.encode_image(batch, as_numpy=False))
        image_embeddings = torch.cat(image_embeddings, 0)
        methods = list(self.config.dataset.eval.method)
        out_dir = create_directory_if_not_exists(self.config.base.results_export_dir)
        experiments_results = []
        for enum_class_name in self.config.dataset.eval.enum_classes:
            if enum_class_name not in ENUMS:
                raise ValueError(f"Invalid enum class: {enum_class_name}")
            classes_dict = dict(ENUMS[enum_class_name])
            if any(enum_class_name not in label for label in prompt_labels):
                logger.warning(f"test batches carry no `{enum_class_name}` labels: skipped")
                continue
            labels = self._named_labels(prompt_labels, enum_class_name, classes_dict)
            if "zeroshot" in methods:
                create_directory_if_not_exists(os.path.join(out_dir, 'zeroshot'))
                results = self.zeroshot_eval(image_embeddings=image_embeddings, label_names=labels, classes_dict=classes_dict, key=enum_class_name)
                logger.info(f"Results From zero-shot evaluation...\n{results}\n")
                experiments_results.append(results)
            if "zeroshot_label_prompt" in methods:
                create_directory_if_not_exists(os.path.join(out_dir, 'zeroshot_label_prompt'))
                results = self.zeroshot_label_prompt(image_embeddings=image_embeddings, label_names=labels, classes_dict=classes_dict,
                                                     key=enum_class_name)
                logger.info(f"Results From zero-shot label prompt evaluation...\n{results}\n")
                experiments_results.append(results)
            if "confustion_matrix" in methods:       # (sic: the reference's spelling of the config value)
                logger.warning("`confustion_matrix` scores the image classifier of the ConvNeXt TorchScript archive, which is not part "
                               "of this build: skipped")
        with open(os.path.join(out_dir, 'results.txt'), 'w') as file:
            for result in experiments_results:
                file.write(str(result) + '\n\n')
        return experiments_results
