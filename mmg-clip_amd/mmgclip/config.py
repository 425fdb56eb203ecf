"""Built-in composer for the reference's Hydra config surface (hydra-core / omegaconf / attrdict are not installable
offline: SURVEY.md §5).  Implements exactly the subset the reference's configs use:

  * `defaults:` lists with `_self_`, list-valued group selections (`- optimizer:\\n    - adamw`), nested groups
    (`dataset/percentage`, `networks/dropout`), an option written with a `.yaml` suffix; no `@package` directives, so
    the group path is the config key (configs/train_binary_class_clf.yaml:1-25);
  * `key=value` / `group=option` / `+key=value` command-line overrides;
  * interpolations `${a.b.c}`, `${now:%Y-%m-%d}`, `${hydra:run.dir}`.
Returns a `Config` (dict with attribute access, like the AttrDict the reference wraps around OmegaConf: train.py:11-14).
"""
import copy
import datetime
import os
import re

import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return Config({k: Config.wrap(v) for k, v in obj.items()})
        if isinstance(obj, list):
            return [Config.wrap(v) for v in obj]
        return obj


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg, dotted, value):
    cur = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        cur = cur.setdefault(p, {})
    cur[parts[-1]] = value


_SCI = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _fix_scalars(node):
    """PyYAML (YAML 1.1) reads `5e-5` as a string; OmegaConf reads it as a float — follow OmegaConf."""
    if isinstance(node, dict):
        return {k: _fix_scalars(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_fix_scalars(v) for v in node]
    if isinstance(node, str) and _SCI.match(node):
        return float(node)
    return node


def _load_yaml(path):
    with open(path) as fh:
        return _fix_scalars(yaml.safe_load(fh) or {})


def _parse_value(text):
    return _fix_scalars(yaml.safe_load(text))


def compose(config_dir, config_name, overrides=()):
    root = _load_yaml(os.path.join(config_dir, config_name if config_name.endswith(".yaml") else config_name + ".yaml"))
    defaults = root.pop("defaults", [])
    group_choice, key_over = {}, []
    for ov in overrides:
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        if os.path.isdir(os.path.join(config_dir, k)):
            group_choice[k] = v
        else:
            key_over.append((k, _parse_value(v)))
    cfg, self_done = {}, False
    for entry in defaults:
        if entry == "_self_":
            _merge(cfg, root)
            self_done = True
            continue
        (group, opts), = entry.items()
        opts = opts if isinstance(opts, list) else [opts]
        if group in group_choice:
            opts = [group_choice[group]]
        for opt in opts:
            name = opt[:-5] if str(opt).endswith(".yaml") else str(opt)
            sub = _load_yaml(os.path.join(config_dir, group, name + ".yaml"))
            node = {}
            _set_path(node, group.replace("/", "."), sub)
            _merge(cfg, node)
    if not self_done:
        _merge(cfg, root)
    for k, v in key_over:
        _set_path(cfg, k, v)
    now = datetime.datetime.now()
    run_dir = cfg.get("hydra", {}).get("run", {}).get("dir", "outputs/${now:%Y-%m-%d}/${now:%H-%M-%S}")

    def resolve_str(s, depth=0):
        def repl(m):
            expr = m.group(1)
            if expr.startswith("now:"):
                return now.strftime(expr[4:])
            if expr == "hydra:run.dir":
                return resolve_str(run_dir, depth + 1)
            cur = cfg
            for part in expr.split("."):
                cur = cur[part]
            return str(resolve(cur, depth + 1))
        assert depth < 16, "interpolation cycle"
        return re.sub(r"\$\{([^}]+)\}", repl, s)

    def resolve(node, depth=0):
        if isinstance(node, dict):
            return {k: resolve(v, depth) for k, v in node.items()}
        if isinstance(node, list):
            return [resolve(v, depth) for v in node]
        if isinstance(node, str) and "${" in node:
            out = resolve_str(node, depth)
            return out
        return node

    cfg = resolve(cfg)
    cfg.pop("hydra", None)
    return Config.wrap(cfg)
