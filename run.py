#!/usr/bin/env python
"""``python run.py lightning_module=eben lightning_module.generator.p=2 ++trainer.max_steps=20``

Counterpart of vibravox ``run.py:25-80`` for the EBEN path.  With hydra + lightning installed the
reference's own ``run.py`` drives these modules through the re-targeted ``configs/``;
neither is available in this image, so this entry point carries a small resolver for the subset of
Hydra the EBEN configs use -- ``defaults`` lists with ``group@package: option``, ``_target_`` /
``_partial_`` instantiation, ``${key}`` interpolation and ``a.b=c`` / ``+a.b=c`` / ``++a.b=c`` overrides --
and a plain training loop calling ``EBENLightningModule.training_step``.
"""
from __future__ import annotations

import importlib
import os
import re
import sys
from functools import partial
from typing import Any, Dict, List

import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(ROOT, "configs")


def _load(group: str, option: str) -> Dict[str, Any]:
    with open(os.path.join(CONFIG_DIR, group, option + ".yaml") if group else os.path.join(CONFIG_DIR, option + ".yaml")) as f:
        return yaml.safe_load(f) or {}


def _compose(group: str, option: str, group_choice: Dict[str, str]) -> Dict[str, Any]:
    cfg = _load(group, option)
    defaults = cfg.pop("defaults", [])
    out: Dict[str, Any] = {}
    for entry in defaults:
        (key, opt), = entry.items()
        sub_group, _, package = key.partition("@")
        package = package or sub_group
        path = f"{group}/{sub_group}" if group else sub_group
        opt = group_choice.get(f"{group}.{package}" if group else package, opt)
        out[package] = _compose(path, opt, group_choice)
    out.update(cfg)
    return out


def _set(cfg: Dict[str, Any], dotted: str, value: Any, create: bool) -> None:
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if k not in node:
            if not create:
                raise KeyError(f"Could not override '{dotted}': no key '{k}' (use +{dotted}=... to add it)")
            node[k] = {}
        node = node[k]
    if keys[-1] not in node and not create:
        raise KeyError(f"Could not override '{dotted}': key not in config (use +{dotted}=... to add it)")
    node[keys[-1]] = value


def _resolve(node: Any, root: Dict[str, Any]) -> Any:
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        if re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+", node):  # YAML 1.1 reads 3e-4 as a string
            return float(node)

        def sub(m):
            cur: Any = root
            for k in m.group(1).split("."):
                cur = cur[k]
            return str(_resolve(cur, root))
        whole = re.fullmatch(r"\$\{([\w\.]+)\}", node)
        if whole:
            cur: Any = root
            for k in whole.group(1).split("."):
                cur = cur[k]
            return _resolve(cur, root)
        return re.sub(r"\$\{([\w\.]+)\}", sub, node)
    return node


def compose(overrides: List[str]) -> Dict[str, Any]:
    group_choice, value_overrides = {}, []
    for ov in overrides:
        key, _, val = ov.partition("=")
        bare = key.lstrip("+")
        is_group = os.path.isdir(os.path.join(CONFIG_DIR, *bare.split("."))) or bare in ("lightning_module", "lightning_datamodule", "trainer")
        if is_group and os.path.exists(os.path.join(CONFIG_DIR, *bare.split("."), val + ".yaml")):
            group_choice[bare] = val
        else:
            value_overrides.append((key, yaml.safe_load(val)))
    cfg = _compose("", "run", group_choice)
    for key, val in value_overrides:
        create = key.startswith("+")
        _set(cfg, key.lstrip("+"), val, create)
    return _resolve(cfg, cfg)


def instantiate(node: Any) -> Any:
    """hydra.utils.instantiate for ``_target_`` / ``_partial_`` nodes (recursive)."""
    if isinstance(node, dict) and "_target_" in node:
        kwargs = {k: instantiate(v) for k, v in node.items() if not k.startswith("_")}
        mod, _, name = node["_target_"].rpartition(".")
        fn = getattr(importlib.import_module(mod), name)
        for k, v in kwargs.items():
            if isinstance(v, list) and k in ("betas", "fft_sizes", "hop_sizes", "win_lengths"):
                kwargs[k] = tuple(v)
        return partial(fn, **kwargs) if node.get("_partial_") else fn(**kwargs)
    if isinstance(node, dict):
        return {k: instantiate(v) for k, v in node.items()}
    return node


def main(argv: List[str]) -> int:
    from vibravox_amd._env import configure_hw_queues
    configure_hw_queues()   # data-parallel ranks: before the first HIP call
    import torch

    cfg = compose(argv)
    torch.manual_seed(42)  # run.py:74 seed_everything(42)
    assert torch.cuda.is_available(), "the EBEN HIP path needs an MI355X (there is no CPU fallback)"
    device = torch.device("cuda", 0)
    datamodule = instantiate(cfg["lightning_datamodule"])
    module = instantiate(cfg["lightning_module"]).to(device)
    trainer = cfg.get("trainer", {})
    if trainer.get("precision") is not None:   # vibravox configs/trainer/ddp.yaml:23-25: the trainer's precision selects the arithmetic plan
        module.set_precision(trainer["precision"])
    loader = datamodule.train_dataloader()
    import gc
    for step in range(int(trainer.get("max_steps", 10))):
        if step == 3:   # set-up and warm-up objects are long-lived: keep the cyclic GC from re-walking them (~100 ms stalls)
            gc.collect()
            gc.freeze()
        module.training_step(next(loader), step)
        if step % int(trainer.get("log_every_n_steps", 1)) == 0:
            logs = " ".join(f"{k.split('/')[-2][0]}/{k.split('/')[-1]}={float(v):.4f}" for k, v in module.logged.items())
            print(f"step {step}: {logs}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
