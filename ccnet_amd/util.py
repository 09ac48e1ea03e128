"""Host-side helpers of the inference entry point, MI355X build: the functions `run_controlnet_inference.py` imports from
the reference's `util.py`, with the same names, arguments and error behaviour, built on this package's networks.

  * `get_datalist`, `get_test_dataloader`      -- util.py:52-100, 103-149 (TSV -> dicts; PNG -> bicubic resize -> [-1, 1];
                                                  CLIP tokenizer on the report text; batches in file order, no shuffling)
  * `get_models_for_controlnet`                -- util.py:1209-1328 (config mutations :1221-1223, :1251-1253, :1261-1263,
                                                  :1288-1292; `.pt` checkpoint layouts `ckpt["state_dict"]`, `ckpt["diffusion"]`
                                                  or the bare state dict; diffusers UNet tried first, then the monai one)
  * `load_controlnet`                          -- util.py:1330-1379 (`ControlNetModel.from_unet` | `ControlNet(**params)`)
  * `get_scale_factor`, `Stage1Wrapper`, `setup_logger` -- util.py:1381-1394, 1473-1492, 801-818
  * `sample_from_controlnet`, `sample_from_ldm`, `save_controlnet_inference_sample` are re-exported from
    `ccnet_amd.sampling` / `ccnet_amd.io`.

What differs from the reference, on purpose: nothing is fetched (hub ids raise OSError: `--source_model` must be a local
folder; mlflow URIs raise NotImplementedError), checkpoints are read with `weights_only=True`, the YAML is read with
PyYAML's safe loader into plain dicts (OmegaConf is not needed for `config["a"]["b"]` / `.get`), and the loader is a plain
generator (MONAI's CacheDataset / worker processes are host plumbing outside the hot path) whose image transform runs on
the GPU (`ccnet_amd.io.load_images`).
"""
from __future__ import annotations

import logging
import sys
from pathlib import Path
from typing import Dict, Iterator, List, Optional

import torch
import torch.nn as nn

from .diffusers_dialect import AutoencoderKL as DiffusersAutoencoderKL
from .diffusers_dialect import ControlNetModel, UNet2DConditionModel
from .io import load_images, save_controlnet_inference_sample  # noqa: F401  (re-export)
from .nets import AutoencoderKL, ControlNet, DiffusionModelUNet, adjust_class_embedding
from .sampling import sample_from_controlnet, sample_from_ldm  # noqa: F401  (re-export)
from .schedulers import DDPMScheduler

FILENAME = "filename"
FILENAME_TARGET = "filename_target"
REPORT = "report"
REPORT_RAW = "report_raw"
SOURCE = "source"
TARGET = "target"
PATIENT_ID = "patient_id"
ACQ_TIME = "acquisition_time"
SEGMENTATION_MASK = "segmentation_mask"
INDEX = "index"  # (not a reference key) row number in the TSV: addresses the sample's noise

DEFAULT_REPORT = "dynamic contrast-enhanced magnetic resonance image of a breast."


def load_config(path) -> dict:
    """`OmegaConf.load(args.config_file)` (run_controlnet_inference.py:96) as plain nested dicts."""
    import yaml
    with open(path) as fh:
        return yaml.safe_load(fh)


def setup_logger(filename="logs.log") -> None:
    logging.basicConfig(filename=str(filename), filemode="a",
                        format="%(asctime)s,%(msecs)d %(name)s %(levelname)s %(message)s", datefmt="%Y-%m-%d %H:%M:%S",
                        level=logging.DEBUG)
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    h = logging.StreamHandler(sys.stdout)
    h.setLevel(logging.INFO)
    h.setFormatter(logging.Formatter("%(asctime)s | %(levelname)s | %(message)s", "%Y-%m-%d %H:%M:%S"))
    logger.addHandler(h)


def get_datalist(ids_path: str, default_report_text: str = DEFAULT_REPORT, use_default_report_text: bool = False,
                 upper_limit: Optional[int] = None) -> List[dict]:
    import pandas as pd
    df = pd.read_csv(ids_path, sep="\t")
    if upper_limit is not None:
        df = df[212:212 + int(upper_limit)]  # the reference's window (util.py:62), kept as is
    rows = []
    for _, row in df.iterrows():
        if use_default_report_text or REPORT not in row or REPORT_RAW not in row:
            report = report_raw = default_report_text
        else:
            report, report_raw = f"{row[REPORT]}", f"{row[REPORT_RAW]}"
        rows.append({FILENAME: f"{row[SOURCE]}", FILENAME_TARGET: f"{row[TARGET]}", SOURCE: f"{row[SOURCE]}",
                     TARGET: f"{row[TARGET]}", REPORT: report, REPORT_RAW: report_raw, PATIENT_ID: f"{row[PATIENT_ID]}",
                     ACQ_TIME: float(row[ACQ_TIME]), SEGMENTATION_MASK: f"{row[SEGMENTATION_MASK]}"})
    logging.info(f"Found {len(rows)} image pairs.")
    return rows


class _TestLoader:
    """Iterable of collated batches with the reference loader's keys and shapes: lists of strings for the name / text
    fields, SOURCE / TARGET fp32 [B, 1, H, W] in [-1, 1] (on `device`), REPORT int64 [B, 1, 77], ACQ_TIME float64 [B]."""

    def __init__(self, rows, batch_size, tokenizer, img_height, img_width, device, indices=None):
        self.indices = list(range(len(rows))) if indices is None else list(indices)  # position of each row in the TSV
        self.rows, self.bs, self.tok = [rows[i] for i in self.indices], max(1, int(batch_size)), tokenizer
        self.h, self.w, self.device = img_height, img_width, device

    def __len__(self):
        return (len(self.rows) + self.bs - 1) // self.bs

    def __iter__(self) -> Iterator[Dict]:
        for i in range(0, len(self.rows), self.bs):
            chunk = self.rows[i:i + self.bs]
            b = {k: [r[k] for r in chunk] for k in (FILENAME, FILENAME_TARGET, REPORT_RAW, PATIENT_ID, SEGMENTATION_MASK)}
            b[SOURCE] = load_images([r[SOURCE] for r in chunk], self.h, self.w, device=self.device)
            b[TARGET] = load_images([r[TARGET] for r in chunk], self.h, self.w, device=self.device)
            ids = [self.tok(r[REPORT], padding="max_length", max_length=self.tok.model_max_length, truncation=True,
                            return_tensors="pt").input_ids for r in chunk]  # custom_transforms.py:16-24: (1, 77) each
            b[REPORT] = torch.stack(ids, dim=0)
            b[ACQ_TIME] = torch.tensor([r[ACQ_TIME] for r in chunk], dtype=torch.float64)
            b[INDEX] = self.indices[i:i + self.bs]
            yield b


def get_test_dataloader(batch_size: int, test_ids: str, num_workers: int = 8, img_height: int = 512, img_width: int = 512,
                        upper_limit: Optional[int] = None, use_default_report_text: bool = False, tokenizer=None,
                        device="cuda", rank: int = 0, world_size: int = 1) -> _TestLoader:
    """util.py:103-149.  `tokenizer`: the CLIPTokenizer (the reference's transform fetches it from the hub; here the caller
    passes the one it loaded from the local `--source_model` folder).  `num_workers` is accepted and unused.  `rank` /
    `world_size`: this process's contiguous shard of the rows (the reference runs one process)."""
    if tokenizer is None:
        raise ValueError("get_test_dataloader needs the CLIP tokenizer loaded from the local --source_model folder")
    rows = get_datalist(ids_path=test_ids, upper_limit=upper_limit, use_default_report_text=use_default_report_text)
    from .distributed import shard_indices
    mine = shard_indices(len(rows), world_size, rank)  # one process per GPU: rank r takes a contiguous block of rows
    return _TestLoader(rows, batch_size, tokenizer, img_height, img_width, device, indices=mine)


class Stage1Wrapper(nn.Module):
    """util.py:1473-1492: `.model` is what the sampling loop decodes with; forward() encodes to a sampled latent."""

    def __init__(self, model: nn.Module, stack_input_to_rgb: bool = False) -> None:
        super().__init__()
        self.model = model
        self.stack_input_to_rgb = stack_input_to_rgb

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[1] == 1 and self.stack_input_to_rgb:
            x = torch.cat((x, x, x), dim=1)
            return self.model.encode(x).latent_dist.sample()
        z_mu, z_sigma = self.model.encode(x)
        return self.model.sampling(z_mu, z_sigma)


def _load_pt(path, model, key):
    """`ckpt[key]`, else the file as a bare state dict (util.py:1269-1276, 1302-1319, 1361-1375); weights_only=True."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(ckpt, dict) and key in ckpt and isinstance(ckpt[key], dict):
        try:
            model.load_state_dict(ckpt[key])
            return
        except Exception as e1:  # the reference falls through to the bare-state-dict attempt on any failure
            logging.info(f"Could not load ckpt[{key!r}] from {path}: {e1}. Trying the file as a bare state dict.")
    model.load_state_dict(ckpt)


def _no_mlflow(uri, what):
    raise NotImplementedError(f"{what}={uri!r}: only '.pt' / '.pth' checkpoint files are loadable here "
                              "(mlflow model URIs need the tracking server, util.py:1278,1325,1377)")


def get_models_for_controlnet(args, config, scheduler=None, stage1=None, diffusion=None, device=None):
    cp, lp = config["controlnet"]["params"], config["ldm"]["params"]
    if args.use_pretrained == 1:
        stage1 = Stage1Wrapper(DiffusersAutoencoderKL.from_pretrained(args.source_model, subfolder="vae"),
                               stack_input_to_rgb=True)
        cp["in_channels"] = 4
        lp["in_channels"] = 4
        lp["out_channels"] = 4
    elif args.use_pretrained == 2:
        stage1 = Stage1Wrapper(DiffusersAutoencoderKL.from_pretrained(args.source_model, subfolder="vae"),
                               stack_input_to_rgb=True)
        diffusion = UNet2DConditionModel.from_pretrained(args.source_model, subfolder="unet")
        # (the reference swaps the hub's PNDM scheduler for the config's DDPM one here and then -- util.py:1281,1328 --
        # falls off the end of the function returning None; this build returns the models it loaded)
        cp["in_channels"] = 4
        cp["num_channels"] = [320, 640, 1280]
        cp["num_head_channels"] = [0, 640, 1280]
    if stage1 is None:
        if args.is_stage1_fine_tuned:
            cp["in_channels"] = 4
            lp["in_channels"] = 4
            lp["out_channels"] = 4
        if ".pt" not in str(args.stage1_uri):
            _no_mlflow(args.stage1_uri, "stage1_uri")
        s1 = AutoencoderKL(**config["stage1"].get("params", dict()))
        _load_pt(args.stage1_uri, s1, "state_dict")
        stage1 = Stage1Wrapper(model=s1)
    if diffusion is None:
        if args.is_ldm_fine_tuned:
            cp["in_channels"] = 4
            cp["num_channels"] = [320, 640, 1280]
            cp["num_head_channels"] = [0, 640, 1280]
            cp["norm_eps"] = 1e-5
        if ".pt" not in str(args.ddpm_uri):
            _no_mlflow(args.ddpm_uri, "ddpm_uri")
        # either dialect may be in the file: the diffusers UNet of --source_model first, then the monai-style one
        builders = []
        if Path(str(args.source_model), "unet").is_dir():
            builders.append(lambda: UNet2DConditionModel.from_pretrained(args.source_model, subfolder="unet"))
        builders.append(lambda: DiffusionModelUNet(**lp))
        errors = []
        for build in builders:
            try:
                m = build()
                if args.cond_on_acq_times:
                    adjust_class_embedding(model=m)
                _load_pt(args.ddpm_uri, m, "diffusion")
            except Exception as e:
                errors.append(e)
                continue
            diffusion = m
            break
        if diffusion is None:
            raise ValueError(f"Could not load the diffusion model from {args.ddpm_uri}. ({errors})")
        scheduler = DDPMScheduler(**config["ldm"].get("scheduler", dict())) if scheduler is not None else scheduler
    return config, stage1, diffusion, scheduler


def load_controlnet(config, args, diffusion, device):
    if args.init_from_unet:
        prev = None
        if diffusion is not None and getattr(diffusion, "config", None) is not None \
                and diffusion.config.num_class_embeds is not None:
            prev = diffusion.config.num_class_embeds
            diffusion.config.num_class_embeds = None  # from_unet must not copy a class embedding (util.py:1336-1338)
        controlnet = ControlNetModel.from_unet(unet=diffusion, conditioning_channels=1,
                                               controlnet_conditioning_channel_order="rgb")
        if args.cond_on_acq_times:
            diffusion.config.num_class_embeds = prev
            adjust_class_embedding(model=controlnet)
    else:
        controlnet = ControlNet(**config["controlnet"].get("params", dict()))
        if args.cond_on_acq_times:
            adjust_class_embedding(model=controlnet)
    if ".pt" not in str(args.controlnet_uri):
        _no_mlflow(args.controlnet_uri, "controlnet_uri")
    _load_pt(args.controlnet_uri, controlnet, "controlnet")
    return controlnet


def get_scale_factor(stage1) -> int:
    cfg = getattr(stage1, "config", None)
    if cfg is not None and getattr(cfg, "block_out_channels", None) is not None:
        n = len(cfg.block_out_channels)
    elif getattr(stage1, "num_channels", None) is not None:
        n = len(stage1.num_channels)
    else:
        n = 4
    return 2 ** (n - 1)
