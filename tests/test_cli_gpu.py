"""The entry point end to end on the GPU: `ccnet_amd.run_controlnet_inference.main(args)` driven by a synthetic TSV / PNG
directory, a local --source_model folder and ccnet-layout checkpoints; its PNG outputs are checked against the oracle's
restatement of the reference loop (run_controlnet_inference.py:140-258) on the same files, seeds and weights."""
import json
import os

import numpy as np
import pytest
import torch

from _cli_common import TEXT_GEOM, namespace, write_config, write_pairs, write_tokenizer, write_tsv
from _common import TINY, TINY_CE, TINY_VAE

pytestmark = pytest.mark.gpu


def _ref_u8(a, f32_scale=False):
    a = a if f32_scale else a.astype(np.float64)
    scaled = np.array((a - (-1.0)) / 2.0, dtype=float)
    return np.rint(np.clip(0.0 + scaled * 1.0, 0.0, 1.0) * 255).astype(np.uint8)


def test_entry_point_against_oracle_loop(cuda, tmp_path):
    import ccnet_amd.nets as pn
    import oracle.nets as on
    import oracle.schedulers as osch
    from PIL import Image
    from safetensors.torch import save_file
    from ccnet_amd import run_controlnet_inference as cli
    from ccnet_amd.synthetic import init_weights
    from ccnet_amd.text_encoder import CLIPTextModel
    from oracle.sampler import sample as osample, to_uint8 as o_u8
    from oracle.text_encoder import clip_text_forward
    from transformers import CLIPTokenizer

    # ---- files on disk
    src = tmp_path / "sd"
    vocab = write_tokenizer(str(src / "tokenizer"))
    geom = dict(TEXT_GEOM, vocab_size=vocab, eos_token_id=vocab - 1)
    torch.manual_seed(3)
    te = CLIPTextModel(**geom).eval()
    with torch.no_grad():
        for n, p in te.named_parameters():
            if "embedding" in n:
                p.copy_(0.5 * torch.randn_like(p))
    os.makedirs(src / "text_encoder")
    json.dump(dict(geom, model_type="clip_text_model"), open(src / "text_encoder" / "config.json", "w"))
    te_sd = {k: v.clone().contiguous() for k, v in te.state_dict().items()}
    save_file(te_sd, str(src / "text_encoder" / "model.safetensors"))
    config = write_config(tmp_path / "cfg.yaml")
    un, cn, vae = pn.DiffusionModelUNet(**config["ldm"]["params"]), pn.ControlNet(**config["controlnet"]["params"]), \
        pn.AutoencoderKL(**config["stage1"]["params"])
    for m, s in ((un, 71), (cn, 72)):
        pn.adjust_class_embedding(m)
        init_weights(m, s)
    init_weights(vae, 73)
    torch.save({"epoch": 1, "diffusion": un.state_dict()}, tmp_path / "ldm.pth")
    torch.save({"epoch": 1, "controlnet": cn.state_dict(), "best_loss": 1.0}, tmp_path / "controlnet.pth")
    torch.save({"epoch": 1, "state_dict": vae.state_dict()}, tmp_path / "vae.pth")
    rows = write_pairs(str(tmp_path / "png"), n=3, size=128)
    write_tsv(tmp_path / "test.tsv", rows)
    out = tmp_path / "out"
    args = namespace(seed=42, output_dir=str(out), experiment="t", config_file=str(tmp_path / "cfg.yaml"),
                     source_model=str(src), stage1_uri=str(tmp_path / "vae.pth"), ddpm_uri=str(tmp_path / "ldm.pth"),
                     controlnet_uri=str(tmp_path / "controlnet.pth"), cond_on_acq_times=True, num_inference_steps=6,
                     x_size=16, y_size=16, guidance_scale=2.0, controlnet_conditioning_scale=1.1, scale_factor=0.5,
                     scheduler_type="ddim", test_ids=str(tmp_path / "test.tsv"), img_height=128, img_width=128, batch_size=2)

    # ---- the entry point (batches of 2 + 1)
    written = cli.main(args)
    names = [f"pat{i:02d}_000{1 + i % 2}_s{i + 10}_synthetic.png" for i in range(3)]
    assert sorted(p.name for p in written if p.parent.name == "PIL") == names
    assert sorted(p.name for p in written if p.parent.name == "PIL_comparison") == names

    # ---- the oracle's loop over the same files (batch 1, as the reference runs it)
    tok = CLIPTokenizer.from_pretrained(str(src / "tokenizer"))
    ou, oc, ov = on.DiffusionModelUNet(**config["ldm"]["params"]), on.ControlNet(**config["controlnet"]["params"]), \
        on.AutoencoderKL(**config["stage1"]["params"])
    on.adjust_class_embedding(ou)
    on.adjust_class_embedding(oc)
    ou.load_state_dict(un.state_dict()); oc.load_state_dict(cn.state_dict()); ov.load_state_dict(vae.state_dict())
    for m in (ou, oc, ov):
        m.eval()

    def embed(texts):
        ids = tok(texts, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
        return clip_text_forward(te_sd, ids, geom["num_attention_heads"], geom["hidden_act"], 1e-5, geom["eos_token_id"])[0]

    empty = embed([""])
    from ccnet_amd.synthetic import make_noise
    for i, r in enumerate(rows):
        a_src = np.asarray(Image.open(r["source"])).astype(np.float32)
        a_tgt = np.asarray(Image.open(r["target"])).astype(np.float32)
        # MONAI's PNG reader hands over the transposed array; the 128 -> 128 bicubic resize is the identity
        cond = torch.from_numpy(a_src.T / 127.5 - 1.0)[None, None]
        real = torch.from_numpy(a_tgt.T / 127.5 - 1.0)[None, None]
        noise = make_noise([i], 42, 4, 16, 16)  # addressed by (seed, row): independent of batching and of resumed runs
        sch = osch.DDIMScheduler(clip_sample=False, **config["ldm"]["scheduler"])
        sch.set_timesteps(6)
        ref = osample(oc, ou, sch, noise, cond, embed([r["report"]]), empty, guidance_scale=2.0, conditioning_scale=1.1,
                      acquisition_time=torch.tensor([r["acquisition_time"]]), stage1=ov, scale_factor=0.5)
        want = o_u8(ref["image"])[0]
        got = np.asarray(Image.open(out / "PIL" / names[i]))
        d = np.abs(got.astype(int) - want.astype(int))
        print(f"{names[i]}: uint8 mean |diff| {d.mean():.3f} max {d.max()}")
        assert got.shape == (128, 128) and d.mean() < 1.0 and d.max() <= 12  # bf16 sampler vs fp32 oracle, 8-bit scale
        trip = np.asarray(Image.open(out / "PIL_comparison" / names[i]))
        assert trip.shape == (128, 384) and np.array_equal(trip[:, 128:256], got)
        assert np.array_equal(trip[:, :128], _ref_u8(cond.numpy(), True)[0, 0])      # integer edges: exact
        assert np.array_equal(trip[:, 256:], _ref_u8(real.numpy(), True)[0, 0])

    # ---- idempotent resume (:189-192): nothing is regenerated; a deleted file is, and only that one
    stamp = {n: os.path.getmtime(out / "PIL" / n) for n in names}
    assert cli.main(args) == []
    assert all(os.path.getmtime(out / "PIL" / n) == stamp[n] for n in names)
    before = np.asarray(Image.open(out / "PIL" / names[1])).copy()
    os.remove(out / "PIL" / names[1])
    assert cli.main(namespace(**{**vars(args), "only_phase_1": True})) == []  # sample 1 is a phase-2 slice: filtered out
    again = cli.main(args)
    assert [p.name for p in again if p.parent.name == "PIL"] == [names[1]]
    # the regenerated slice (alone in its batch now) is the slice of the first run: noise is addressed by row, and a row's
    # arithmetic does not depend on the batch it travels in beyond bf16 rounding of other tile shapes
    d = np.abs(np.asarray(Image.open(out / "PIL" / names[1])).astype(int) - before.astype(int))
    assert d.mean() < 1.0 and d.max() <= 12
