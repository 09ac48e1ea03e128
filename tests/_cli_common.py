"""Synthetic on-disk fixtures for the entry-point tests: a local `--source_model` folder (tiny CLIP text tower + a minimal
CLIP BPE tokenizer), a YAML with the reference's config schema at a tiny geometry, ccnet-layout checkpoints, PNG pairs and
the TSV that `create_datalist.py` would have written."""
import json
import os

import numpy as np
import torch

from _common import TINY, TINY_CE, TINY_VAE

SCHED_YAML = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0205, schedule="scaled_linear_beta",
                  prediction_type="v_prediction")
TEXT_GEOM = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, hidden_act="gelu",
                 max_position_embeddings=77)


def write_tokenizer(d):
    os.makedirs(d, exist_ok=True)
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789.,-")
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    merges = ["m r", "mr i</w>", "b r", "br e", "bre a", "brea s", "breas t</w>"]
    for m in merges:
        a, b = m.split()
        vocab[a + b] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n" + "\n".join(merges) + "\n")
    json.dump({"model_max_length": 77, "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>",
               "unk_token": "<|endoftext|>", "pad_token": "<|endoftext|>", "tokenizer_class": "CLIPTokenizer"},
              open(os.path.join(d, "tokenizer_config.json"), "w"))
    return len(vocab)


def write_config(path, acq=True):
    import yaml
    unet = {k: v for k, v in TINY.items()}
    cfg = {"controlnet": {"params": {**unet, **TINY_CE}},
           "ldm": {"params": {**unet, "out_channels": 4}, "scheduler": dict(SCHED_YAML)},
           "stage1": {"params": dict(TINY_VAE)}}
    with open(path, "w") as fh:
        yaml.safe_dump(cfg, fh)
    return cfg


def write_pairs(root, n=3, size=128, seed=5):
    """n (source, target) PNG pairs named like the dataset's slices: `<patient>_0000_<slice>.png` -> `_000<phase>_`."""
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    rng = np.random.RandomState(seed)
    rows = []
    for i in range(n):
        phase = 1 + (i % 2)
        src = os.path.join(root, f"pat{i:02d}_0000_s{i + 10}.png")
        tgt = os.path.join(root, f"pat{i:02d}_000{phase}_s{i + 10}.png")
        yy, xx = np.mgrid[0:size, 0:size]
        base = 127 + 100 * np.sin(xx / (7.0 + i)) * np.cos(yy / (9.0 + i))
        for p, off in ((src, 0), (tgt, 20)):
            a = np.clip(base + off + rng.randint(-10, 10, (size, size)), 0, 255).astype(np.uint8)
            Image.fromarray(a).save(p)
        rows.append(dict(source=src, target=tgt, report=f"mri breast {i}", report_raw=f"MRI breast {i}",
                         patient_id=f"pat{i:02d}", acquisition_time=float(90 + 130 * i), segmentation_mask="none"))
    return rows


def write_tsv(path, rows):
    import pandas as pd
    pd.DataFrame(rows).to_csv(path, sep="\t", index=False)


def namespace(**kw):
    from ccnet_amd.run_controlnet_inference import parse_args
    a = parse_args([])
    for k, v in kw.items():
        setattr(a, k, v)
    return a
