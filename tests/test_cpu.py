"""CPU-side suite (no GPU): the oracle against known-answer constants and the committed golden vectors, the host
logic of the product (schedulers, state-dict surface, sharding), and the C-ABI library's exported symbols."""
import json
import math
import os
import re

import numpy as np
import pytest
import torch

from _common import SCHED, TINY, TINY_CE, TINY_VAE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "tiny_golden.npz")


# ------------------------------------------------------------------ oracle: closed-form / known-answer (Appendix C)
def test_schedule_constants_appendix_c():
    import oracle.schedulers as osch
    s = osch.DDIMScheduler(clip_sample=False, **SCHED)
    assert abs(s.betas[0].item() - 0.0015) < 1e-9 and abs(s.betas[1].item() - 0.001508110) < 1e-9
    assert abs(s.betas[999].item() - 0.0205) < 1e-8
    known = {0: 0.998499990, 1: 0.996994138, 20: 0.967288375, 500: 0.107676789, 960: 0.000212549,
             980: 0.000142899, 995: 0.000105263, 999: 0.000096911}
    for t, v in known.items():
        assert abs(s.alphas_cumprod[t].item() - v) < 5e-9, (t, s.alphas_cumprod[t].item())
    assert abs(s.alphas_cumprod[999].item() ** 0.5 - 0.00984433) < 1e-7
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    s.set_timesteps(200)
    assert s.timesteps.tolist() == list(range(995, -1, -5))


def test_ddim_last_step_returns_x0_and_vpred_algebra():
    import oracle.schedulers as osch
    s = osch.DDIMScheduler(clip_sample=False, **SCHED)
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    prev, x0 = s.step(v, 0, x)
    assert torch.allclose(prev, x0, atol=1e-6)          # final_alpha_cumprod = 1  =>  x_prev = x0
    a = s.alphas_cumprod[500]
    prev, x0 = s.step(v, 500, x)
    assert torch.allclose(x0, a.sqrt() * x - (1 - a).sqrt() * v, atol=1e-6)
    eps = a.sqrt() * v + (1 - a).sqrt() * x
    ap = s.alphas_cumprod[480]
    assert torch.allclose(prev, ap.sqrt() * x0 + (1 - ap).sqrt() * eps, atol=1e-6)
    # x0/eps are an orthonormal rotation of (x, v): x = sqrt(a) x0 + sqrt(1-a) eps
    assert torch.allclose(a.sqrt() * x0 + (1 - a).sqrt() * eps, x, atol=1e-5)


def test_ddpm_clip_and_noise_injection():
    import oracle.schedulers as osch
    s = osch.DDPMScheduler(**SCHED)
    assert s.clip_sample is True and s.timesteps.tolist()[:3] == [999, 998, 997]
    g = torch.Generator().manual_seed(1)
    x, v, n = (torch.randn(1, 4, 8, 8, generator=g) * 3 for _ in range(3))
    _, x0 = s.step(v, 700, x, noise=n)
    assert x0.abs().max() <= 1.0
    p0, _ = s.step(v, 0, x, noise=n)
    p0b, _ = s.step(v, 0, x, noise=n * 5)
    assert torch.equal(p0, p0b)                           # no noise at t = 0


def test_cfg_identities():
    from oracle.sampler import cfg_combine
    g = torch.Generator().manual_seed(2)
    out = torch.randn(4, 4, 8, 8, generator=g)
    u, t = out.chunk(2)
    assert torch.equal(cfg_combine(out, None), t)
    assert torch.equal(cfg_combine(out, 0), u)
    assert torch.allclose(cfg_combine(out, 1.0), t, atol=1e-6)
    assert torch.allclose(cfg_combine(out, 7.0), u + 7.0 * (t - u))


def test_timestep_embedding_cos_first():
    from oracle.nets import get_timestep_embedding
    e = get_timestep_embedding(torch.tensor([0, 10]), 8)
    assert torch.allclose(e[0], torch.tensor([1.0, 1, 1, 1, 0, 0, 0, 0]))
    assert abs(e[1, 0].item() - math.cos(10.0)) < 1e-6 and abs(e[1, 4].item() - math.sin(10.0)) < 1e-6


def test_param_counts_match_published_sd21():
    """parameter counts of the restated geometry = published SD-2.1 UNet / ControlNet sizes."""
    import oracle.nets as on
    from _common import G1, G1_CE
    with torch.device("meta"):
        u = on.DiffusionModelUNet(out_channels=4, **G1)
        c = on.ControlNet(**G1, **G1_CE)
    nu = sum(p.numel() for p in u.parameters())
    nc = sum(p.numel() for p in c.parameters())
    assert abs(nu / 1e6 - 865.9) < 1.0, nu     # SD-2.1 UNet ~ 866 M (without the ccnet class MLP)
    assert abs(nc / 1e6 - 364.7) < 1.5, nc


# ------------------------------------------------------------------ oracle vs committed golden vectors
def _oracles():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    return make_golden


@pytest.fixture(scope="module")
def gold():
    return {k: torch.from_numpy(v) for k, v in np.load(GOLD).items()}


def test_oracle_reproduces_golden(gold):
    mg = _oracles()
    import oracle.schedulers as osch
    from oracle.sampler import sample as osample, to_uint8
    cn, un, vae = mg.build_oracles()
    assert abs(mg.checksum(cn.state_dict()) - gold["ck_cn"].item()) < 1e-3 * gold["ck_cn"].item()
    assert abs(mg.checksum(un.state_dict()) - gold["ck_un"].item()) < 1e-3 * gold["ck_un"].item()
    x, ctx, cond, acq = gold["x"], gold["ctx"], gold["cond"], gold["acq"]
    t = torch.tensor([500])
    with torch.no_grad():
        down, mid = cn(x, t, cond, 0.9, ctx, acq.reshape(-1, 1))
        out = un(x, t, ctx, acq.reshape(-1, 1), down, mid)
    assert torch.allclose(down[0], gold["cn_down0"], atol=1e-4)
    assert torch.allclose(mid, gold["cn_mid"], atol=1e-4)
    assert torch.allclose(out, gold["unet_out"], atol=1e-4)
    sch = osch.DDIMScheduler(clip_sample=False, **SCHED)
    sch.set_timesteps(8)
    res = osample(cn, un, sch, x, cond, ctx, gold["empty"], guidance_scale=3.0, conditioning_scale=1.0,
                  acquisition_time=acq, stage1=vae, scale_factor=0.5, record_latents=True)
    assert torch.allclose(torch.stack(res["latents"]), gold["ddim_latents"], atol=2e-4)
    assert torch.allclose(res["image"], gold["ddim_image"], atol=1e-3)
    assert np.abs(to_uint8(res["image"]).astype(int) - gold["ddim_u8"].numpy().astype(int)).max() <= 1


def test_oracle_invariants_zero_controlnet(gold):
    """Upstream zero-initialised zero-convs => ControlNet adds nothing; conditioning_scale scales linearly."""
    mg = _oracles()
    cn, un, _ = mg.build_oracles()
    x, ctx, cond, acq = gold["x"], gold["ctx"], gold["cond"], gold["acq"].reshape(-1, 1)
    t = torch.tensor([500])
    with torch.no_grad():
        d1, m1 = cn(x, t, cond, 1.0, ctx, acq)
        d2, m2 = cn(x, t, cond, 0.5, ctx, acq)
        assert torch.allclose(m2, 0.5 * m1, atol=1e-6) and torch.allclose(d2[3], 0.5 * d1[3], atol=1e-6)
        d0, m0 = cn(x, t, cond, 0.0, ctx, acq)
        assert torch.allclose(un(x, t, ctx, acq, d0, m0), gold["unet_plain"], atol=1e-5)


# ------------------------------------------------------------------ product host logic
def test_product_scheduler_host_constants_match_oracle():
    import ccnet_amd.schedulers as ps
    import oracle.schedulers as osch
    p, o = ps.DDIMScheduler(clip_sample=False, **SCHED), osch.DDIMScheduler(clip_sample=False, **SCHED)
    assert torch.equal(p.alphas_cumprod, o.alphas_cumprod) and torch.equal(p.betas, o.betas)
    for n in (50, 200, 1000, 7):
        p.set_timesteps(n)
        o.set_timesteps(n)
        assert torch.equal(p.timesteps, o.timesteps)
    p.set_timesteps(50)
    c = p._coef(980)
    a_t, a_p = o.alphas_cumprod[980], o.alphas_cumprod[960]
    assert c[:4] == [float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5)]
    assert p._coef(0)[2:4] == [1.0, 0.0] and c[5] == 0.0 and c[6] == 2.0
    d = ps.DDPMScheduler(**SCHED)
    assert d.clip_sample and d._coef(0)[4] == 0.0 and d._coef(10)[7] == 1.0
    with pytest.raises(ValueError):
        p.set_timesteps(2000)
    with pytest.raises(NotImplementedError):
        ps.DDIMScheduler(schedule="cosine")


def test_state_dict_surface_and_loading():
    import ccnet_amd.nets as pn
    import oracle.nets as on
    pu, ou = pn.DiffusionModelUNet(out_channels=4, **TINY), on.DiffusionModelUNet(out_channels=4, **TINY)
    pc, oc = pn.ControlNet(**TINY, **TINY_CE), on.ControlNet(**TINY, **TINY_CE)
    pv, ov = pn.AutoencoderKL(**TINY_VAE), on.AutoencoderKL(**TINY_VAE)
    for p, o in ((pu, ou), (pc, oc), (pv, ov)):
        ps_, os_ = p.state_dict(), o.state_dict()
        assert list(ps_.keys()) == list(os_.keys())
        assert all(ps_[k].shape == os_[k].shape for k in ps_)
        p.load_state_dict(os_)
    # upstream zero-initialises these
    assert pu.out[2].conv.weight.abs().sum() == 0 and pc.controlnet_mid_block.weight.abs().sum() == 0
    # wrapped zero-conv spelling `controlnet_down_blocks.k.conv.weight` is accepted too (Appendix A.8)
    sd = {}
    for k, v in oc.state_dict().items():
        if k.startswith(("controlnet_down_blocks.", "controlnet_mid_block.")):
            k = k.replace(".weight", ".conv.weight").replace(".bias", ".conv.bias")
        sd[k] = v
    pc.load_state_dict(sd)
    # reference dispatch is by class-name substring (util.py:943, 1051)
    assert "DiffusionModelUNet" in type(pu).__name__ and "ControlNetModel" not in type(pc).__name__
    pn.adjust_class_embedding(pu)
    on.adjust_class_embedding(ou)
    assert pu.num_class_embeds == 99999 and list(pu.state_dict().keys()) == list(ou.state_dict().keys())
    with pytest.raises(ValueError):
        pn.adjust_class_embedding(torch.nn.Linear(1, 1))


def test_product_has_no_cpu_fallback():
    import ccnet_amd.nets as pn
    import ccnet_amd.schedulers as ps
    from ccnet_amd.lib import CcnetError
    pu = pn.DiffusionModelUNet(out_channels=4, **TINY)
    with pytest.raises(CcnetError):
        pu(torch.zeros(1, 4, 16, 16), torch.tensor([1]), torch.zeros(1, 77, 64))
    s = ps.DDIMScheduler(clip_sample=False, **SCHED)
    s.set_timesteps(10)
    with pytest.raises(CcnetError):
        s.step(torch.zeros(1, 4, 8, 8), 900, torch.zeros(1, 4, 8, 8))
    with pytest.raises(CcnetError):
        pn.DiffusionModelUNet(spatial_dims=2, in_channels=4, out_channels=4, num_channels=[48, 96],
                              attention_levels=[False, False], norm_num_groups=16)


def test_reference_yaml_schema_loads():
    """configs/controlnet/controlnet_v1.yaml schema (values restated in tests/golden/controlnet_v1.yaml) builds the
    three product networks after the reference's own config mutation for the SD VAE (util.py:1221-1223)."""
    import yaml
    import ccnet_amd.nets as pn
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "controlnet_v1.yaml")))
    cfg["controlnet"]["params"]["in_channels"] = 4
    cfg["ldm"]["params"]["in_channels"] = 4
    cfg["ldm"]["params"]["out_channels"] = 4
    with torch.device("meta"):
        c = pn.ControlNet(**cfg["controlnet"]["params"])
        u = pn.DiffusionModelUNet(**cfg["ldm"]["params"])
        v = pn.AutoencoderKL(**cfg["stage1"]["params"])
    assert c.block_out_channels == [256, 512, 768] and u.out_channels == 4 and v.latent_channels == 3
    import ccnet_amd.schedulers as ps
    s = ps.DDPMScheduler(**cfg["ldm"]["scheduler"])
    assert s.prediction_type == "v_prediction" and s.num_train_timesteps == 1000


def test_synthetic_inputs_are_index_addressed():
    from ccnet_amd.synthetic import make_batch
    a = make_batch([3, 4, 5], seed=42, latent=(4, 16, 16), image=(64, 64), ctx_dim=64, acq=True)
    b = make_batch([5], seed=42, latent=(4, 16, 16), image=(64, 64), ctx_dim=64, acq=True)
    for k in ("noise", "cond", "prompt_embeds", "acquisition_time"):
        assert torch.equal(a[k][2], b[k][0]), k
    assert a["cond"].min() >= -1 and a["cond"].max() <= 1 and a["cond"].shape == (3, 1, 64, 64)
    assert 60 <= a["acquisition_time"].min() and a["acquisition_time"].max() <= 1000
    assert not torch.equal(a["noise"][0], a["noise"][1])


# ------------------------------------------------------------------ C ABI
def test_cabi_exports_every_declared_symbol():
    from ccnet_amd import lib
    hdr = open(os.path.join(ROOT, "include", "ccnet_hip.h")).read()
    declared = set(re.findall(r"\b(ccnet_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    l = lib.load()
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in include/ccnet_hip.h but not exported"
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    assert l.ccnet_arch().decode() == "gfx950" and l.ccnet_version() >= 1
    assert l.ccnet_strerror(-1).decode().startswith("invalid")
    import ctypes
    assert ctypes.sizeof(lib.ConvParams) == 9 * 8 + 23 * 4 + 4 + 3 * 8 + 16 + 16 + 8 + 8  # struct layout of ccnet_conv_params
    from ccnet_amd import ops
    assert ctypes.sizeof(ops.FfParams) == 7 * 8 + 5 * 4 + 4           # ccnet_ff_params: 7 pointers, 5 int32, float
    assert ctypes.sizeof(ops.XattnParams) == 8 * 8 + 8 * 4 + 4 + 4    # ccnet_xattn_params: 8 pointers, 8 int32, float (+ tail padding)
    assert l.ccnet_ff_packed_bytes(320, 1280) == 80 * 30720 and l.ccnet_ff_packed_bytes(640, 2560) == 0
    assert l.ccnet_xattn_packed_bytes(320, 5) == 20 * 20480 and l.ccnet_xattn_kv_image_bytes(320, 5) == 102400
    assert l.ccnet_xattn_packed_bytes(640, 10) == 0


def test_xattn_pack_and_kv_image_layouts():
    """Host-side layouts of the fused cross-attention kernel (include/ccnet_hip.h: ccnet_xattn_params) against their definitions,
    element by element on a few probes."""
    import math
    from ccnet_amd import ops
    g = torch.Generator().manual_seed(5)
    wq, wo = torch.randn(320, 320, generator=g), torch.randn(320, 320, generator=g)
    bq = torch.randn(320, generator=g)
    wp, bqp = ops.xattn_pack(wq, bq, wo, 5)
    qs = math.log2(math.e) / 8.0
    pi = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]
    flat = wp.reshape(-1)
    wqs = (wq * qs).to(torch.bfloat16)
    for blk, r, unit, e in [(0, 0, 0, 0), (3, 17, 11, 5), (9, 31, 39, 7), (4, 6, 8, 2)]:
        row = 32 * blk + r
        src_row = (row // 16) * 16 + pi[row % 16]
        src_unit = unit ^ ((r >> 1) & 7)                       # position `unit` of row r holds unit `unit ^ swz(r)`
        assert flat[blk * 10240 + r * 320 + unit * 8 + e] == wqs[src_row, src_unit * 8 + e]
        assert bqp[row] == (bq * qs)[src_row]
    wob = wo.to(torch.bfloat16)
    for h, s_, hh, n, t in [(0, 0, 0, 0, 0), (2, 3, 1, 200, 6), (4, 1, 0, 319, 3)]:
        dim = 64 * h + 16 * s_ + (4 * hh + t if t < 4 else 4 + 4 * hh + t)
        assert flat[10 * 10240 + ((h * 4 + s_) * 2 + hh) * 2560 + n * 8 + t] == wob[n, dim]
    kv = torch.randn(2 * 77, 640, generator=g).to(torch.bfloat16)
    img = ops.xattn_kv_image(kv, 2, 77, 320)
    assert img.shape == (2, 51200)
    for i, key, unit, e in [(0, 0, 0, 0), (1, 76, 39, 7), (1, 33, 12, 4)]:
        assert img[i, key * 320 + unit * 8 + e] == kv[i * 77 + key, (unit ^ ((key >> 1) & 7)) * 8 + e]
    assert (img[:, 77 * 320:80 * 320] == 0).all()              # padded keys
    for i, s_, hh, c, t in [(0, 0, 0, 0, 0), (1, 4, 1, 300, 2), (1, 2, 0, 17, 7)]:
        key = 16 * s_ + (4 * hh + t if t < 4 else 4 + 4 * hh + t)
        exp = kv[i * 77 + key, 320 + c] if key < 77 else 0
        assert img[i, 25600 + ((s_ * 2 + hh) * 320 + c) * 8 + t] == exp
    assert ops.xattn_kv_image(torch.zeros(2 * 96, 640, dtype=torch.bfloat16), 2, 96, 320) is None


# ------------------------------------------------------------------ sharding (world_size 2, gloo)
def _worker(rank, world, port, n_total, bs, q):
    import torch.distributed as dist
    from ccnet_amd.distributed import sample_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccnet_amd.synthetic import make_noise

    def run_batch(idx):  # stand-in for the GPU sampler: a deterministic function of the seed-indexed noise
        n = make_noise(idx, 42, 4, 8, 8)
        return {"latent": n * 2 + 1, "image_u8": (n[:, 0].abs() * 10).to(torch.uint8)}

    out, mine = sample_sharded(run_batch, n_total, bs, world, rank, dist)
    q.put((rank, list(mine), out["latent"].clone(), out["image_u8"].clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,bs", [(10, 3), (7, 4), (1, 4)])  # (1, 4): rank 1 owns nothing and still gathers
def test_sharded_sampling_gloo_world2(n_total, bs):
    import torch.multiprocessing as mp
    from ccnet_amd.distributed import sample_sharded, shard_indices
    from ccnet_amd.synthetic import make_noise
    assert list(shard_indices(10, 4, 3)) == [9] and list(shard_indices(8, 8, 7)) == [7]
    assert [len(shard_indices(9, 8, r)) for r in range(8)] == [2, 2, 2, 2, 1, 0, 0, 0]  # trailing ranks can be empty
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, bs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, _ = sample_sharded(lambda idx: {"latent": make_noise(idx, 42, 4, 8, 8) * 2 + 1,
                                         "image_u8": (make_noise(idx, 42, 4, 8, 8)[:, 0].abs() * 10).to(torch.uint8)},
                            n_total, bs, 1, 0)
    covered = sorted(i for _, mine, _, _ in got for i in mine)
    assert covered == list(range(n_total))
    for _, _, lat, img in got:   # every rank holds the full, index-ordered result, identical to world_size 1
        assert torch.equal(lat, ref["latent"]) and torch.equal(img, ref["image_u8"])


# ------------------------------------------------------------------ diffusers dialect (host logic)
SD21_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True,
                 norm_eps=1e-5)


def test_diffusers_dialect_sd21_keys_and_parameter_count():
    """The SD-2.1-base UNet in diffusers naming: 686 tensors, 865,910,724 parameters (the published figure for
    `stabilityai/stable-diffusion-2-1-base/unet`) and the well-known key spellings."""
    from ccnet_amd import diffusers_dialect as dd
    with torch.device("meta"):
        u = dd.UNet2DConditionModel(**SD21_UNET)
    sd = u.state_dict()
    assert len(sd) == 686
    assert sum(v.numel() for v in sd.values()) == 865_910_724
    for k, shape in {
        "conv_in.weight": (320, 4, 3, 3), "time_embedding.linear_1.weight": (1280, 320),
        "time_embedding.linear_2.bias": (1280,), "down_blocks.0.resnets.0.conv1.weight": (320, 320, 3, 3),
        "down_blocks.1.resnets.0.conv_shortcut.weight": (640, 320, 1, 1),
        "down_blocks.0.attentions.0.proj_in.weight": (320, 320),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight": (320, 1024),
        "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight": (2560, 320),
        "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.weight": (320, 1280),
        "down_blocks.0.downsamplers.0.conv.weight": (320, 320, 3, 3),
        "mid_block.attentions.0.proj_out.weight": (1280, 1280), "mid_block.resnets.1.time_emb_proj.weight": (1280, 1280),
        "up_blocks.0.resnets.2.conv_shortcut.weight": (1280, 2560, 1, 1),
        "up_blocks.0.upsamplers.0.conv.weight": (1280, 1280, 3, 3),
        "up_blocks.3.attentions.2.transformer_blocks.0.norm3.bias": (320,),
        "conv_norm_out.weight": (320,), "conv_out.weight": (4, 320, 3, 3),
    }.items():
        assert tuple(sd[k].shape) == shape, k
    assert "down_blocks.3.attentions.0.norm.weight" not in sd and "up_blocks.3.upsamplers.0.conv.weight" not in sd
    assert u.config.block_out_channels == (320, 640, 1280, 1280) and u.config["cross_attention_dim"] == 1024
    # head dim 64 at every level, Transformer2DModel GroupNorm eps fixed at 1e-6
    assert u._net.num_head_channels_ == (64, 64, 64, 64) and u._net.transformer_norm_eps == 1e-6
    assert "DiffusionModelUNet" not in type(u).__name__ and not hasattr(u, "block_out_channels")


def test_diffusers_dialect_round_trip_and_class_embedding():
    from ccnet_amd import diffusers_dialect as dd
    import ccnet_amd.nets as pn
    tiny = dict(block_out_channels=(64, 128), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), layers_per_block=1, cross_attention_dim=64,
                attention_head_dim=(1, 2))
    for lin in (False, True):
        u = dd.UNet2DConditionModel(use_linear_projection=lin, **tiny)
        for p in u.parameters():
            p.data.normal_()
        sd = u.state_dict()
        assert sd["down_blocks.0.attentions.0.proj_in.weight"].dim() == (2 if lin else 4)
        for k in sd:  # the two translations are inverse of each other
            assert dd.monai_key_to_diffusers(dd.diffusers_key_to_monai(k)) == k
        u2 = dd.UNet2DConditionModel(use_linear_projection=lin, **tiny)
        u2.load_state_dict(sd)
        a, b = u._net.state_dict(), u2._net.state_dict()
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
        with pytest.raises(RuntimeError):
            u2.load_state_dict({k: v for k, v in list(sd.items())[1:]})  # strict, as nn.Module
    # acquisition-time conditioning on a diffusers-dialect model (util.py:1113-1130): config + MLP, in the state dict
    pn.adjust_class_embedding(u)
    assert u.config.num_class_embeds == 99999 and isinstance(u.class_embedding, torch.nn.Sequential)
    assert "class_embedding.2.weight" in u.state_dict()
    # ControlNetModel.from_unet: same geometry, encoder weights copied, zero convs zero, own conditioning embedding
    u.config.num_class_embeds = None  # util.py:1337-1340
    cn = dd.ControlNetModel.from_unet(u, conditioning_channels=1)
    pn.adjust_class_embedding(cn)
    csd = cn.state_dict()
    assert torch.equal(csd["down_blocks.0.resnets.0.conv1.weight"], sd["down_blocks.0.resnets.0.conv1.weight"])
    assert torch.equal(csd["mid_block.attentions.0.proj_in.weight"], sd["mid_block.attentions.0.proj_in.weight"])
    assert csd["controlnet_down_blocks.0.weight"].abs().sum() == 0 and csd["controlnet_mid_block.bias"].abs().sum() == 0
    assert tuple(csd["controlnet_cond_embedding.conv_in.weight"].shape) == (16, 1, 3, 3)
    assert tuple(csd["controlnet_cond_embedding.blocks.5.weight"].shape) == (256, 96, 3, 3)
    assert "ControlNetModel" in type(cn).__name__ and cn.config.num_class_embeds == 99999
    with pytest.raises(NotImplementedError):
        dd.UNet2DConditionModel(class_embed_type="timestep", **tiny)
    with pytest.raises(OSError):
        dd.UNet2DConditionModel.from_pretrained("stabilityai/stable-diffusion-2-1-base", subfolder="unet")


def test_diffusers_dialect_from_pretrained_local_folder(tmp_path):
    import json
    from safetensors.torch import save_file
    from ccnet_amd import diffusers_dialect as dd
    tiny = dict(block_out_channels=[64, 128], down_block_types=["CrossAttnDownBlock2D", "DownBlock2D"],
                up_block_types=["UpBlock2D", "CrossAttnUpBlock2D"], layers_per_block=1, cross_attention_dim=64,
                attention_head_dim=[1, 2], use_linear_projection=True)
    u = dd.UNet2DConditionModel(**tiny)
    for p in u.parameters():
        p.data.normal_()
    d = tmp_path / "unet"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(tiny, _class_name="UNet2DConditionModel", _diffusers_version="0.32.0")))
    save_file({k: v.contiguous() for k, v in u.state_dict().items()}, str(d / "diffusion_pytorch_model.safetensors"))
    v = dd.UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    a, b = u._net.state_dict(), v._net.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a) and not v.training
    # ccnet checkpoint files: {"diffusion": state_dict} or the bare state dict (util.py:1305-1323)
    torch.save({"diffusion": u.state_dict(), "epoch": 3}, tmp_path / "ldm.pt")
    w = dd.load_ccnet_checkpoint(str(tmp_path / "ldm.pt"), dd.UNet2DConditionModel(**tiny), "diffusion")
    torch.save(u.state_dict(), tmp_path / "bare.pt")
    x = dd.load_ccnet_checkpoint(str(tmp_path / "bare.pt"), dd.UNet2DConditionModel(**tiny), "diffusion")
    for m in (w, x):
        assert all(torch.equal(a[k], m._net.state_dict()[k]) for k in a)


def test_diffusers_dialect_sd_vae_keys_and_round_trip():
    """SD-VAE (`stabilityai/stable-diffusion-2-1-base/vae`): 248 tensors, 83,653,863 parameters (published figure)."""
    from ccnet_amd import diffusers_dialect as dd
    sdvae = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                 up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, sample_size=768, scaling_factor=0.18215)
    with torch.device("meta"):
        v = dd.AutoencoderKL(**sdvae)
    sd = v.state_dict()
    assert len(sd) == 248 and sum(t.numel() for t in sd.values()) == 83_653_863
    for k, shape in {"encoder.conv_in.weight": (128, 3, 3, 3), "encoder.mid_block.attentions.0.to_q.weight": (512, 512),
                     "encoder.mid_block.attentions.0.group_norm.weight": (512,), "encoder.conv_out.weight": (8, 512, 3, 3),
                     "quant_conv.weight": (8, 8, 1, 1), "post_quant_conv.bias": (4,),
                     "decoder.up_blocks.0.resnets.2.conv1.weight": (512, 512, 3, 3),
                     "decoder.up_blocks.2.resnets.0.conv_shortcut.weight": (256, 512, 1, 1),
                     "decoder.up_blocks.3.resnets.2.norm2.bias": (128,), "decoder.conv_out.weight": (3, 128, 3, 3)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd and v.config.scaling_factor == 0.18215
    tiny = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 2,
                up_block_types=["UpDecoderBlock2D"] * 2, block_out_channels=[64, 128], layers_per_block=1)
    a = dd.AutoencoderKL(**tiny)
    for p in a.parameters():
        p.data.normal_()
    sd = a.state_dict()
    b = dd.AutoencoderKL(**tiny)
    b.load_state_dict(sd)
    # pre-0.18 spelling of the attention block (query/key/value/proj_attn) loads too
    old = {k.replace(".to_q.", ".query.").replace(".to_k.", ".key.").replace(".to_v.", ".value.")
            .replace(".to_out.0.", ".proj_attn."): v for k, v in sd.items()}
    c = dd.AutoencoderKL(**tiny)
    c.load_state_dict(old)
    x, y, z = a._net.state_dict(), b._net.state_dict(), c._net.state_dict()
    assert all(torch.equal(x[k], y[k]) and torch.equal(x[k], z[k]) for k in x)
    with pytest.raises(RuntimeError):
        b.load_state_dict({**sd, "decoder.bogus.weight": torch.zeros(1)})


def test_autoencoder_attention_state_dict_matches_oracle():
    import ccnet_amd.nets as pn
    import oracle.nets as on
    cfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=[64, 128], latent_channels=4,
               num_res_blocks=1, attention_levels=[False, True], with_encoder_nonlocal_attn=True,
               with_decoder_nonlocal_attn=True)
    for extra in ({}, dict(_decoder_num_res_blocks=2, _final_act=True, _encoder_out_channels=8)):
        p, o = pn.AutoencoderKL(**cfg, **extra), on.AutoencoderKL(**cfg, **extra)
        a, b = p.state_dict(), o.state_dict()
        assert list(a) == list(b) and all(a[k].shape == b[k].shape for k in a)
        assert any(".to_q." in k for k in a)


# ------------------------------------------------------------------ §8(f)4: text tower (oracle pin + host surface)
def test_text_oracle_is_pinned_to_transformers():
    """The oracle's CLIP text tower against transformers' own CLIPTextModel (the module the reference imports,
    run_controlnet_inference.py:22,126) on seeded random weights: both activations, both EOS pooling rules."""
    tr = pytest.importorskip("transformers")
    from oracle.text_encoder import clip_text_forward
    for act, eos in (("gelu", 999), ("quick_gelu", 2)):
        torch.manual_seed(5)
        cfg = tr.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                num_attention_heads=2, hidden_act=act, eos_token_id=eos, bos_token_id=998, pad_token_id=1)
        hf = tr.CLIPTextModel(cfg).eval()
        ids = torch.randint(3, 990, (2, 77))
        ids[:, 30] = 999
        with torch.no_grad():
            o = hf(ids)
        hid, pool = clip_text_forward(hf.state_dict(), ids, 2, act, cfg.layer_norm_eps, eos)
        assert (hid - o.last_hidden_state).abs().max().item() < 1e-4
        assert (pool - o.pooler_output).abs().max().item() < 1e-4


def test_text_tower_surface_and_known_parameter_count(tmp_path):
    from safetensors.torch import save_file
    from ccnet_amd.text_encoder import CLIPTextModel
    from ccnet_amd.lib import CcnetError
    # SD-2.1-base text_encoder/config.json geometry: 340,387,840 parameters (the published size of that tower)
    with torch.device("meta"):
        big = CLIPTextModel(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                            hidden_act="gelu", projection_dim=512)
    assert sum(p.numel() for p in big.parameters()) == 340_387_840
    assert len(big.state_dict()) == 2 + 23 * 16 + 2
    small = dict(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                 hidden_act="gelu")
    m = CLIPTextModel(**small)
    sd = m.state_dict()
    # both key generations load: transformers 4.x prefixes `text_model.` and carries a position_ids buffer
    bare = {k[len("text_model."):]: v for k, v in sd.items()}
    m2 = CLIPTextModel(**small)
    m2.load_state_dict(bare)
    old = dict(sd)
    old["text_model.embeddings.position_ids"] = torch.arange(77).unsqueeze(0)
    m2.load_state_dict(old)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # local folder round trip; hub ids raise
    d = tmp_path / "sd21" / "text_encoder"
    d.mkdir(parents=True)
    (d / "config.json").write_text(json.dumps(dict(small, architectures=["CLIPTextModel"], model_type="clip_text_model")))
    save_file({k: v.contiguous() for k, v in bare.items()}, str(d / "model.safetensors"))
    m3 = CLIPTextModel.from_pretrained(str(tmp_path / "sd21"), subfolder="text_encoder")
    assert m3.config.hidden_size == 128 and not m3.training
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    with pytest.raises(OSError):
        CLIPTextModel.from_pretrained("stabilityai/stable-diffusion-2-1-base", subfolder="text_encoder")
    with pytest.raises(CcnetError):  # no CPU path
        m3(torch.zeros(1, 77, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        CLIPTextModel(hidden_size=128, num_attention_heads=4)
