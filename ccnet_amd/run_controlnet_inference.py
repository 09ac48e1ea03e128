"""ccnet's inference entry point on the MI355X build: same flags, YAML schema, checkpoint layout, file naming and output
folders as the reference's `src/python/training/run_controlnet_inference.py` (flags :27-67, scheduler :98-114, models
:116-138, empty prompt :140-151, loader :153-161, per-sample loop :165-258 with the phase / skip-if-exists bookkeeping
:169-194, save :258).

    python -m ccnet_amd.run_controlnet_inference --config_file configs/controlnet/controlnet_v1.yaml \
        --source_model /local/stable-diffusion-2-1-base --ddpm_uri ldm.pth --controlnet_uri controlnet.pth \
        --stage1_uri vae.pth --test_ids test.tsv --output_dir out --scheduler_type ddim --num_inference_steps 200 \
        --guidance_scale 1.0 --controlnet_conditioning_scale 1.1 --scale_factor 0.1 --cond_on_acq_times --batch_size 16

The arithmetic of the loop body (:211-256) runs in `ccnet_amd.sampling.Sampler` (HIP graph of ControlNet + UNet + CFG +
scheduler step, then AEKL decode).  Differences from the reference, all on the host side: `--source_model` must be a LOCAL
folder (no network); `--batch_size` > 1 is honoured (the reference's loop only works at 1): images of a
batch share one graph replay per timestep.  The initial noise of a sample is addressed by (seed, row number in the TSV)
instead of being the next draw of the global CPU generator (:212) -- that stream also feeds the third-party model
constructors and the loader's iterator in the reference, so it pins nothing across implementations -- which makes a run's
INPUTS (noise, prompt, conditioning image) independent of batch size, of which files already existed (resume) and of how rows
are split over processes.  The arithmetic applied to a row is the same function in every packing, evaluated in bf16 with
tile plans that depend on how many rows share the batch (kernel choice, split-K factor, streaming thresholds), so the same
row sampled in batches of different size agrees to bf16 rounding (PSNR > 40 dB between batch 1 and batch 16 at the real
geometry, tests/test_g1_parity_gpu.py), and bit for bit only at equal batch size, in any order within the batch.  Under
`torch.distributed.run` (WORLD_SIZE / RANK in the environment) every rank generates a contiguous block of the rows on its
own GPU and writes its own files; there is nothing to exchange.
"""
from __future__ import annotations

import argparse
import logging
import os
import random
from pathlib import Path

import numpy as np
import torch

from . import util as U
from .sampling import Sampler, encode_prompts
from .schedulers import DDIMScheduler, DDPMScheduler
from .synthetic import make_noise
from .text_encoder import CLIPTextModel


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=2, help="Random seed to use.")
    p.add_argument("--output_dir", help="Path to save the generated images")
    p.add_argument("--experiment", help="Experiment name to keep track.")
    p.add_argument("--config_file", help="YAML with ldm, controlnet and stage1 params.")
    p.add_argument("--upper_limit", default=None, help="Maximum number of images to be generated.")
    p.add_argument("--source_model", type=str, default="stabilityai/stable-diffusion-2-1-base",
                   help="LOCAL folder of the source model (vae / unet / text_encoder / tokenizer sub-folders)")
    p.add_argument("--stage1_uri", help="stage1 .pt/.pth checkpoint")
    p.add_argument("--ddpm_uri", help="diffusion .pt/.pth checkpoint")
    p.add_argument("--controlnet_uri", default="controlnet_best_model.pth", help="controlnet .pt/.pth checkpoint")
    p.add_argument("--only_phase_1", action="store_true", help="Generate only first phase.")
    p.add_argument("--only_phase_2", action="store_true", help="Generate only second phase.")
    p.add_argument("--init_from_unet", action="store_true", help="ControlNetModel.from_unet instead of the monai-style ControlNet.")
    p.add_argument("--use_pretrained", type=int, default=0, help="0 = trained checkpoints, 1 = pretrained stage 1, 2 = stage 1 and diffusion")
    p.add_argument("--is_stage1_fine_tuned", action="store_true")
    p.add_argument("--is_ldm_fine_tuned", action="store_true")
    p.add_argument("--cond_on_acq_times", action="store_true", help="Pass MRI acquisition times as class_labels.")
    p.add_argument("--use_default_report_text", action="store_true")
    p.add_argument("--num_inference_steps", type=int)
    p.add_argument("--x_size", type=int, default=64, help="Latent space x size.")
    p.add_argument("--y_size", type=int, default=64, help="Latent space y size.")
    p.add_argument("--guidance_scale", type=float, default=None)
    p.add_argument("--controlnet_conditioning_scale", type=float, default=1.0)
    p.add_argument("--scale_factor", type=float, default=0.01)
    p.add_argument("--scheduler_type", type=str, default=None, help="'ddim' or 'ddpm'")
    p.add_argument("--test_ids", help="TSV with the test pairs.")
    p.add_argument("--img_height", type=int, default=512)
    p.add_argument("--img_width", type=int, default=512)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--num_workers", type=int)
    return p.parse_args(argv)


def set_determinism(seed: int) -> None:
    """monai.utils.set_determinism(seed) as far as this path consumes randomness: torch / numpy / random generators."""
    torch.manual_seed(seed)
    np.random.seed(seed % (2 ** 32))
    random.seed(seed)


def final_file_name(filename: str, filename_target: str) -> str:
    """run_controlnet_inference.py:170-176,189: the synthetic slice is named after the SOURCE file with the phase digits of
    the TARGET file and a `_synthetic.png` suffix."""
    name = filename
    for digit in range(6):
        if f"_000{digit}_" in filename_target:
            name = name.replace("_0000_", f"_000{digit}_")
            break
    return name


def is_skipped(name: str, only_phase_1: bool, only_phase_2: bool) -> bool:
    """:179-187 (including its asymmetric second test, kept as written)."""
    skipped = True
    if only_phase_1 and "0001" in name:
        skipped = False
    if only_phase_2 and "0002" not in name:
        skipped = False
    if not only_phase_1 and not only_phase_2:
        skipped = False
    return skipped


def build_scheduler(args, config):
    sc = config["ldm"]["scheduler"]
    if args.scheduler_type == "ddim":
        scheduler = DDIMScheduler(num_train_timesteps=sc["num_train_timesteps"], beta_start=sc["beta_start"],
                                  beta_end=sc["beta_end"], schedule=sc["schedule"], prediction_type=sc["prediction_type"],
                                  clip_sample=False)
        scheduler.set_timesteps(num_inference_steps=args.num_inference_steps)
        return scheduler
    if args.scheduler_type == "ddpm":
        return DDPMScheduler(**config["ldm"].get("scheduler", dict()))
    raise NotImplementedError(f"scheduler {args.scheduler_type} is not implemented. Please use either 'ddpm' or 'ddim'.")


@torch.no_grad()
def main(args):
    from transformers import CLIPTokenizer  # host-side string work
    set_determinism(seed=args.seed)
    output_dir = Path(args.output_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    U.setup_logger(output_dir / f"run_inference_controlnet_{args.experiment}.log")
    logging.info(f"Arguments: {str(args)}")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else torch.cuda.current_device())
    torch.cuda.set_device(device)

    config = U.load_config(args.config_file)
    scheduler = build_scheduler(args, config)
    config, stage1, diffusion, _ = U.get_models_for_controlnet(config=config, args=args, scheduler=scheduler, device=device)
    controlnet = U.load_controlnet(config, args, diffusion, device)
    text_encoder = CLIPTextModel.from_pretrained(args.source_model, subfolder="text_encoder")
    tokenizer = CLIPTokenizer.from_pretrained(str(Path(args.source_model) / "tokenizer"))
    stage1, diffusion, controlnet, text_encoder = (m.to(device).eval() for m in (stage1, diffusion, controlnet, text_encoder))

    ids = tokenizer([""], padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                    return_tensors="pt").input_ids
    empty_prompt_embeds = encode_prompts(text_encoder, ids.to(device))

    loader = U.get_test_dataloader(batch_size=args.batch_size, test_ids=args.test_ids, num_workers=args.num_workers,
                                   upper_limit=None if args.upper_limit is None else int(args.upper_limit),
                                   img_height=args.img_height, img_width=args.img_width,
                                   use_default_report_text=args.use_default_report_text, tokenizer=tokenizer, device=device,
                                   rank=rank, world_size=world)
    args.scale_factor = U.get_scale_factor(stage1=stage1) if args.scale_factor is None else args.scale_factor
    sampler = Sampler(controlnet, diffusion, scheduler, stage1.model, guidance_scale=args.guidance_scale,
                      conditioning_scale=args.controlnet_conditioning_scale, scale_factor=args.scale_factor)
    latent_ch = config["controlnet"]["params"]["in_channels"]
    written = []
    for x in loader:
        keep, names = [], []
        for j, (fn, fnt) in enumerate(zip(x[U.FILENAME], x[U.FILENAME_TARGET])):
            name = final_file_name(fn, fnt)
            if is_skipped(name, args.only_phase_1, args.only_phase_2):
                continue
            name = Path(name).name.replace(".png", "_synthetic.png")
            if (output_dir / "PIL" / name).exists() or (output_dir / name).exists():
                logging.debug(f"Image {name} already exists in the output dir {output_dir}. Skipping.")
                continue
            keep.append(j)
            names.append(name)
        if not keep:
            continue
        sel = torch.tensor(keep, device=device)
        images, cond = x[U.TARGET].to(device)[sel], x[U.SOURCE].to(device)[sel]
        reports = x[U.REPORT].to(device)[sel]
        acq = x[U.ACQ_TIME].to(device)[sel].float() if args.cond_on_acq_times else None
        prompt_embeds = encode_prompts(text_encoder, reports.squeeze(1))
        noise = make_noise([x[U.INDEX][j] for j in keep], args.seed, latent_ch, args.x_size, args.y_size).to(device)
        res = sampler.sample(noise, cond, prompt_embeds, empty_prompt_embeds, acq)
        written += U.save_controlnet_inference_sample(run_dir=output_dir, filename=names, sample=res["image"], cond=cond,
                                                      images=images)
    logging.info(f"Wrote {len(written)} files below {output_dir}.")
    return written


if __name__ == "__main__":
    main(parse_args())
