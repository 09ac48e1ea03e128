"""End-to-end parity of the HIP path against the fp32 oracle (oracle/, parity unpinned -- see its header) on the
same seeded weights and inputs.  Tolerances (stated per BASELINE.json north_star):
  * single network forward, bf16 compute vs fp32: relative L2 error <= 2e-2, and max-abs <= 6e-2 * max|ref|
  * sampling loop: per-step latent max-abs error <= 0.1 on O(1) latents (reported for every step), final decoded
    image PSNR >= 40 dB (data range 2.0 for [-1,1] images)
"""
import pytest
import torch

from _common import G1, G1_CE, SCHED, TINY, TINY_CE, TINY_VAE, build_pair, psnr, rel_l2

pytestmark = pytest.mark.gpu

REL_L2 = 2e-2
MAX_REL = 6e-2


def _inputs(B, dev, seed=0, hw=16, img=128, D=64, tokens=77):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, hw, hw, generator=g)
    ctx = 0.5 * torch.randn(B, tokens, D, generator=g)
    cond = torch.rand(B, 1, img, img, generator=g) * 2 - 1
    return x, ctx, cond


def _check(out, ref, what):
    r = rel_l2(out, ref)
    m = (out.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{what}: rel_l2={r:.3e} max_rel={m:.3e} ref_max={ref.abs().max().item():.3f}")
    assert r <= REL_L2 and m <= MAX_REL, f"{what}: rel_l2={r:.3e} max_rel={m:.3e}"


def test_unet_forward_tiny(cuda):
    p, o = build_pair("unet", TINY, 1, cuda)
    x, ctx, _ = _inputs(2, cuda)
    t = torch.tensor([500])
    out = p(x.to(cuda), t.to(cuda), ctx.to(cuda))
    with torch.no_grad():
        ref = o(x, t, ctx)
    assert out.shape == ref.shape and out.dtype == torch.float32
    _check(out, ref, "unet tiny")


def test_unet_controlnet_single_head_geometry(cuda):
    """Repo-YAML style geometry (G2): attention only at the deeper levels, one head of C channels per level."""
    from _common import TINY_G2
    pc, oc = build_pair("controlnet", {**TINY_G2, **TINY_CE}, 21, cuda)
    pu, ou = build_pair("unet", TINY_G2, 22, cuda)
    x, ctx, cond = _inputs(2, cuda, seed=23)
    t = torch.tensor([640])
    down, mid = pc(x.to(cuda), t.to(cuda), cond.to(cuda), 1.0, ctx.to(cuda))
    out = pu(x.to(cuda), t.to(cuda), ctx.to(cuda), None, down, mid)
    with torch.no_grad():
        rd, rm = oc(x, t, cond, 1.0, ctx)
        ref = ou(x, t, ctx, None, rd, rm)
    _check(mid.float(), rm, "G2-style controlnet mid")
    _check(out, ref, "G2-style unet + residuals")


def test_controlnet_and_unet_with_residuals_tiny(cuda):
    pc, oc = build_pair("controlnet", {**TINY, **TINY_CE}, 2, cuda, acq=True)
    pu, ou = build_pair("unet", TINY, 3, cuda, acq=True)
    x, ctx, cond = _inputs(2, cuda, seed=4)
    t = torch.tensor([980])
    cls = torch.tensor([[120.0], [800.0]])
    down, mid = pc(x.to(cuda), t.to(cuda), cond.to(cuda), 0.9, ctx.to(cuda), cls.to(cuda))
    with torch.no_grad():
        rdown, rmid = oc(x, t, cond, 0.9, ctx, cls)
    assert len(down) == len(rdown)
    for k, (d, r) in enumerate(zip(down, rdown)):
        assert tuple(d.shape) == tuple(r.shape)
        _check(d, r, f"controlnet residual {k}")
    _check(mid, rmid, "controlnet mid")
    out = pu(x.to(cuda), t.to(cuda), ctx.to(cuda), cls.to(cuda), down, mid)
    with torch.no_grad():
        ref = ou(x, t, ctx, cls, rdown, rmid)
    _check(out, ref, "unet + residuals")
    # residuals given as ordinary fp32 NCHW tensors (the reference's own calling convention) take the copy path
    out2 = pu(x.to(cuda), t.to(cuda), ctx.to(cuda), cls.to(cuda), [r.to(cuda) for r in rdown], rmid.to(cuda))
    _check(out2, ref, "unet + fp32 NCHW residuals")


def test_structural_invariants(cuda):
    """Zero ControlNet zero-convs (upstream init) or conditioning_scale = 0  =>  UNet output == plain UNet output."""
    pc, _ = build_pair("controlnet", {**TINY, **TINY_CE}, 5, cuda)
    pu, _ = build_pair("unet", TINY, 6, cuda)
    x, ctx, cond = _inputs(2, cuda, seed=7)
    xg, cg, dg, t = x.to(cuda), ctx.to(cuda), cond.to(cuda), torch.tensor([300]).to(cuda)
    plain = pu(xg, t, cg)
    down, mid = pc(xg, t, dg, 0.0, cg)
    assert all(d.float().abs().max().item() == 0.0 for d in down) and mid.float().abs().max().item() == 0.0
    assert torch.equal(pu(xg, t, cg, None, down, mid), plain)
    sd = pc.state_dict()
    for k in sd:
        if k.startswith(("controlnet_down_blocks", "controlnet_mid_block")):
            sd[k] = torch.zeros_like(sd[k])
    pc.load_state_dict(sd)
    down, mid = pc(xg, t, dg, 1.0, cg)
    assert torch.equal(pu(xg, t, cg, None, down, mid), plain)


def test_vae_decode_encode_tiny(cuda):
    p, o = build_pair("vae", TINY_VAE, 8, cuda)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(2, 4, 16, 16, generator=g)
    img = p.decode(z.to(cuda))
    with torch.no_grad():
        ref = o.decode(z)
    _check(img, ref, "vae decode")
    x = torch.rand(2, 1, 128, 128, generator=g) * 2 - 1
    mu, sigma = p.encode(x.to(cuda))
    with torch.no_grad():
        rmu, rsig = o.encode(x)
    _check(mu, rmu, "vae encode mu")
    _check(sigma, rsig, "vae encode sigma")


@pytest.mark.parametrize("kind", ["ddim", "ddpm"])
def test_sampler_parity_tiny(cuda, kind):
    """The whole loop (ControlNet + UNet + CFG + scheduler, then AEKL decode) against the oracle's restatement of
    run_controlnet_inference.py:211-256, explicit noise, acquisition-time conditioning on."""
    import ccnet_amd.schedulers as ps
    import oracle.schedulers as osch
    from ccnet_amd.sampling import Sampler
    from oracle.sampler import sample as osample
    pc, oc = build_pair("controlnet", {**TINY, **TINY_CE}, 11, cuda, acq=True)
    pu, ou = build_pair("unet", TINY, 12, cuda, acq=True)
    pv, ov = build_pair("vae", TINY_VAE, 13, cuda)
    B, steps = 2, 12
    x, ctx, cond = _inputs(B, cuda, seed=14)
    g = torch.Generator().manual_seed(15)
    empty = 0.5 * torch.randn(1, 77, 64, generator=g)
    acq = torch.tensor([100.0, 650.0])
    if kind == "ddim":
        psch = ps.DDIMScheduler(clip_sample=False, **SCHED)
        osc = osch.DDIMScheduler(clip_sample=False, **SCHED)
        dn = None
    else:
        psch, osc = ps.DDPMScheduler(**SCHED), osch.DDPMScheduler(**SCHED)
        dn = [torch.randn(B, 4, 16, 16, generator=g) for _ in range(steps)]
    psch.set_timesteps(steps)
    osc.set_timesteps(steps)
    s = Sampler(pc, pu, psch, pv, guidance_scale=3.0, conditioning_scale=1.0, scale_factor=0.5)
    res = s.sample(x.to(cuda), cond.to(cuda), ctx.to(cuda), empty.to(cuda), acq.to(cuda),
                   ddpm_noise=None if dn is None else [n.to(cuda) for n in dn], record_latents=True)
    ref = osample(oc, ou, osc, x, cond, ctx, empty, guidance_scale=3.0, conditioning_scale=1.0,
                  acquisition_time=acq, stage1=ov, scale_factor=0.5, ddpm_noise=dn, record_latents=True)
    errs = [(a.cpu() - b).abs().max().item() for a, b in zip(res["latents"], ref["latents"])]
    print(f"{kind} per-step latent max-abs:", " ".join(f"{e:.3e}" for e in errs))
    assert max(errs) <= 0.1, f"per-step latent max-abs {max(errs):.3e}"
    ps_db = psnr(res["image"], ref["image"])
    print(f"{kind} final image PSNR {ps_db:.1f} dB")
    assert ps_db >= 40.0


@pytest.mark.parametrize("gs", [None, 0.0])
def test_sampler_guidance_none_and_zero_run_one_branch(cuda, gs, monkeypatch):
    """guidance_scale None keeps the text half of the [uncond, text] batch and 0 the unconditional half
    (run_controlnet_inference.py:245-252; src/bash/INF_CTRL_CCNET_P1.sh:21 runs 0.0).  The Sampler evaluates only the half that
    is kept (B rows instead of 2B): against the oracle's 2B loop, and against its own 2B form (CCNET_CFG_PRUNE=0) -- the same
    value per image; the two forms may pick different tile plans (row count), so equality is to fp32-summation-order / bf16
    rounding level, far below the oracle tolerance."""
    import ccnet_amd.schedulers as ps
    import oracle.schedulers as osch
    from ccnet_amd.sampling import Sampler
    from oracle.sampler import sample as osample
    pc, oc = build_pair("controlnet", {**TINY, **TINY_CE}, 11, cuda, acq=True)
    pu, ou = build_pair("unet", TINY, 12, cuda, acq=True)
    B, steps = 3, 6
    x, ctx, cond = _inputs(B, cuda, seed=24)
    g = torch.Generator().manual_seed(25)
    empty = 0.5 * torch.randn(1, 77, 64, generator=g)
    acq = torch.tensor([100.0, 650.0, 333.0])
    psch, osc = ps.DDIMScheduler(clip_sample=False, **SCHED), osch.DDIMScheduler(clip_sample=False, **SCHED)
    psch.set_timesteps(steps)
    osc.set_timesteps(steps)
    args = (x.to(cuda), cond.to(cuda), ctx.to(cuda), empty.to(cuda), acq.to(cuda))
    s = Sampler(pc, pu, psch, None, guidance_scale=gs, conditioning_scale=1.0, scale_factor=0.5)
    assert s._one_branch()
    st = s.prepare(*args[1:])
    assert st["Bc"] == B  # one branch only
    one = s.sample(*args)["latent"]
    monkeypatch.setenv("CCNET_CFG_PRUNE", "0")
    s2 = Sampler(pc, pu, psch, None, guidance_scale=gs, conditioning_scale=1.0, scale_factor=0.5)
    assert not s2._one_branch() and s2.prepare(*args[1:])["Bc"] == 2 * B
    two = s2.sample(*args)["latent"]
    ref = osample(oc, ou, osc, x, cond, ctx, empty, guidance_scale=gs, conditioning_scale=1.0, acquisition_time=acq)["latent"]
    e1, e2 = (one.cpu() - ref).abs().max().item(), (two.cpu() - ref).abs().max().item()
    d = (one - two).abs().max().item()
    print(f"guidance {gs}: one-branch vs oracle {e1:.3e}, two-branch vs oracle {e2:.3e}, one vs two {d:.3e}")
    assert e1 <= 0.1 and e2 <= 0.1 and d <= 2e-2


def test_cfg_halves_share_their_head(cuda, monkeypatch):
    """Under CFG the unconditional and the text half are the same rows until the first cross-attention (same latent,
    conditioning image, timestep, acquisition time: run_controlnet_inference.py:219-221), so the Sampler evaluates conv_in,
    the first resnet and the first transformer block up to attn2's query once (CCNET_CFG_SHARE, nets._run_attn dup_to).
    Against the oracle, and against the form that evaluates both halves from the start (the same values up to the order of
    fp32 sums: tile plans depend on the row count)."""
    import ccnet_amd.nets as pn
    import ccnet_amd.schedulers as ps
    import oracle.schedulers as osch
    from ccnet_amd.sampling import Sampler
    from oracle.sampler import sample as osample
    pc, oc = build_pair("controlnet", {**TINY, **TINY_CE}, 11, cuda, acq=True)
    pu, ou = build_pair("unet", TINY, 12, cuda, acq=True)
    B, steps = 3, 6
    x, ctx, cond = _inputs(B, cuda, seed=34)
    g = torch.Generator().manual_seed(35)
    empty = 0.5 * torch.randn(1, 77, 64, generator=g)
    acq = torch.tensor([100.0, 650.0, 333.0])
    psch, osc = ps.DDIMScheduler(clip_sample=False, **SCHED), osch.DDIMScheduler(clip_sample=False, **SCHED)
    psch.set_timesteps(steps)
    osc.set_timesteps(steps)
    args = (x.to(cuda), cond.to(cuda), ctx.to(cuda), empty.to(cuda), acq.to(cuda))
    assert pn.CFG_SHARE and pu._share_ok(True, B, 2 * B) and not pu._share_ok(True, B, B)
    res = {}
    dups = []
    real_dup = pn._dup_act
    monkeypatch.setattr(pn, "_dup_act", lambda a: (dups.append(a.N), real_dup(a))[1])
    for overlap in (True, False):
        res[("share", overlap)] = Sampler(pc, pu, psch, None, guidance_scale=3.0, scale_factor=0.5, overlap=overlap).sample(*args)["latent"]
    assert dups and all(n == B for n in dups)  # the shared head ran on B images
    monkeypatch.setattr(pn, "CFG_SHARE", False)
    n_dup = len(dups)
    res["plain"] = Sampler(pc, pu, psch, None, guidance_scale=3.0, scale_factor=0.5).sample(*args)["latent"]
    assert len(dups) == n_dup
    ref = osample(oc, ou, osc, x, cond, ctx, empty, guidance_scale=3.0, conditioning_scale=1.0, acquisition_time=acq)["latent"]
    errs = {k: (v.cpu() - ref).abs().max().item() for k, v in res.items()}
    d = (res[("share", True)] - res["plain"]).abs().max().item()
    print("vs oracle:", errs, "shared vs plain:", d)
    assert all(e <= 0.1 for e in errs.values()) and d <= 2e-2


def test_fused_blocks_in_the_replayed_graph_follow_the_prompts(cuda, monkeypatch):
    """The fused feed-forward and cross-attention launches (ccnet_ff_geglu_fused_bf16, ccnet_xattn_fused_bf16) take part in the
    sampler's captured step at >= 32768 rows of 320 channels; the cross-attention kernel reads the prompt's K / V from LDS images
    that the graph holds as static buffers.  One level of the G1 width at 64x64, 4 images with CFG (8 x 4096 rows): a second batch
    with other prompts through the SAME sampler (graph replay, refreshed buffers) equals a fresh sampler's result bit for bit --
    stale images would reproduce the first batch's prompts (run_controlnet_inference.py:211-256 runs batch after batch)."""
    import ccnet_amd.ops as pops
    import ccnet_amd.schedulers as ps
    from ccnet_amd.sampling import Sampler
    wide = dict(spatial_dims=2, in_channels=4, num_res_blocks=1, num_channels=[320], attention_levels=[True],
                with_conditioning=True, cross_attention_dim=64, num_head_channels=64, norm_eps=1e-5)
    pc, _ = build_pair("controlnet", {**wide, "conditioning_embedding_in_channels": 1,
                                      "conditioning_embedding_num_channels": [16, 32, 96, 256]}, 61, cuda, acq=True)
    pu, _ = build_pair("unet", wide, 62, cuda, acq=True)
    calls = {"xattn": 0, "ff": 0}
    real_x, real_f = pops.xattn_fused, pops.ff_fused
    monkeypatch.setattr(pops, "xattn_fused", lambda *a, **k: (calls.__setitem__("xattn", calls["xattn"] + 1), real_x(*a, **k))[1])
    monkeypatch.setattr(pops, "ff_fused", lambda *a, **k: (calls.__setitem__("ff", calls["ff"] + 1), real_f(*a, **k))[1])
    B = 4
    g = torch.Generator().manual_seed(63)
    empty = 0.5 * torch.randn(1, 77, 64, generator=g)
    acq = torch.tensor([100.0, 650.0, 333.0, 900.0])

    def batch(seed):
        x, ctx, cond = _inputs(B, cuda, seed=seed, hw=64, img=512)
        return (x.to(cuda), cond.to(cuda), ctx.to(cuda), empty.to(cuda), acq.to(cuda))

    def sampler():
        sch = ps.DDIMScheduler(clip_sample=False, **SCHED)
        sch.set_timesteps(3)
        return Sampler(pc, pu, sch, None, guidance_scale=3.0, scale_factor=0.5)

    s = sampler()
    r1 = s.sample(*batch(71))["latent"]
    assert calls["xattn"] > 0 and calls["ff"] > 0, calls          # the fused launches are part of the step
    r2 = s.sample(*batch(72))["latent"]                            # same geometry: graph replay
    r2_fresh = sampler().sample(*batch(72))["latent"]
    assert torch.isfinite(r2).all() and not torch.equal(r1, r2)
    assert torch.equal(r2, r2_fresh), (r2 - r2_fresh).abs().max().item()
    # same noise, conditioning image and acquisition times as the second batch, the FIRST batch's prompts: must differ from r2
    b2, b1 = batch(72), batch(71)
    r_mixed = sampler().sample(b2[0], b2[1], b1[2], b2[3], b2[4])["latent"]
    assert not torch.equal(r_mixed, r2)


def test_g1_single_cfg_step(cuda):
    """SD-2.1-base geometry, batch 1 with CFG (effective batch 2), one DDIM step at t=980 against the oracle."""
    import ccnet_amd.schedulers as ps
    import oracle.schedulers as osch
    from ccnet_amd.sampling import Sampler
    from ccnet_amd.synthetic import make_batch
    from oracle.sampler import sample as osample
    pc, oc = build_pair("controlnet", {**G1, **G1_CE}, 21, cuda, acq=True)
    pu, ou = build_pair("unet", G1, 22, cuda, acq=True)
    b = make_batch([0], seed=42, acq=True)
    psch, osc = ps.DDIMScheduler(clip_sample=False, **SCHED), osch.DDIMScheduler(clip_sample=False, **SCHED)
    psch.set_timesteps(50)
    osc.set_timesteps(50)
    s = Sampler(pc, pu, psch, None, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1)
    res = s.sample(b["noise"].to(cuda), b["cond"].to(cuda), b["prompt_embeds"].to(cuda),
                   b["empty_prompt_embeds"].to(cuda), b["acquisition_time"].to(cuda), max_steps=1)
    ref = osample(oc, ou, osc, b["noise"], b["cond"], b["prompt_embeds"], b["empty_prompt_embeds"],
                  guidance_scale=7.0, conditioning_scale=1.0, acquisition_time=b["acquisition_time"], max_steps=1)
    err = (res["latent"].cpu() - ref["latent"]).abs().max().item()
    print(f"G1 one-step latent max-abs {err:.3e}, rel_l2 {rel_l2(res['latent'], ref['latent']):.3e}")
    assert err <= 0.1


def test_against_committed_golden(cuda):
    """HIP path vs tests/golden/tiny_golden.npz (made by tests/golden/make_golden.py from the oracle): network
    forwards, 8-step DDIM and DDPM sampling traces, decoded image and the uint8 mapping."""
    import os
    import numpy as np
    import ccnet_amd.nets as pn
    import ccnet_amd.schedulers as ps
    from ccnet_amd.sampling import Sampler, to_uint8
    from ccnet_amd.synthetic import init_weights
    gold = {k: torch.from_numpy(v) for k, v in
            np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_golden.npz")).items()}
    cn, un, vae = pn.ControlNet(**TINY, **TINY_CE), pn.DiffusionModelUNet(out_channels=4, **TINY), \
        pn.AutoencoderKL(**TINY_VAE)
    pn.adjust_class_embedding(cn)
    pn.adjust_class_embedding(un)
    init_weights(cn, 11)
    init_weights(un, 12)
    init_weights(vae, 13)
    cn, un, vae = cn.to(cuda).eval(), un.to(cuda).eval(), vae.to(cuda).eval()
    g = {k: v.to(cuda) for k, v in gold.items() if k in ("x", "ctx", "empty", "cond", "acq")}
    t = torch.tensor([500], device=cuda)
    down, mid = cn(g["x"], t, g["cond"], 0.9, g["ctx"], g["acq"].reshape(-1, 1))
    _check(down[0], gold["cn_down0"], "golden controlnet down[0]")
    _check(down[-1], gold["cn_down_last"], "golden controlnet down[-1]")
    _check(mid, gold["cn_mid"], "golden controlnet mid")
    _check(un(g["x"], t, g["ctx"], g["acq"].reshape(-1, 1), down, mid), gold["unet_out"], "golden unet out")
    _check(un(g["x"], t, g["ctx"], g["acq"].reshape(-1, 1)), gold["unet_plain"], "golden unet plain")
    sch = ps.DDIMScheduler(clip_sample=False, **SCHED)
    sch.set_timesteps(8)
    assert sch.timesteps.tolist() == gold["ddim_ts"].tolist()
    s = Sampler(cn, un, sch, vae, guidance_scale=3.0, conditioning_scale=1.0, scale_factor=0.5)
    res = s.sample(g["x"], g["cond"], g["ctx"], g["empty"], g["acq"], record_latents=True)
    errs = [(a.cpu() - b).abs().max().item() for a, b in zip(res["latents"], gold["ddim_latents"])]
    print("golden ddim per-step latent max-abs:", " ".join(f"{e:.3e}" for e in errs))
    assert max(errs) <= 0.1
    assert psnr(res["image"], gold["ddim_image"]) >= 40.0
    du8 = (to_uint8(res["image"]).cpu().int() - gold["ddim_u8"].int()).abs()
    assert du8.float().mean().item() < 1.0 and du8.max().item() <= 12   # 8-bit image within ~1 grey level on average
    dsch = ps.DDPMScheduler(**SCHED)
    dsch.set_timesteps(8)
    s2 = Sampler(cn, un, dsch, None, guidance_scale=3.0, conditioning_scale=1.0, scale_factor=0.5)
    dres = s2.sample(g["x"], g["cond"], g["ctx"], g["empty"], g["acq"],
                     ddpm_noise=[n.to(cuda) for n in gold["ddpm_noise"]], record_latents=True)
    derr = [(a.cpu() - b).abs().max().item() for a, b in zip(dres["latents"], gold["ddpm_latents"])]
    assert max(derr) <= 0.1, derr


def test_diffusers_dialect_controlnet_and_unet(cuda):
    """The diffusers surface (UNet2DConditionModel / ControlNetModel.from_unet, diffusers kwargs through
    sample_from_controlnet / sample_from_ldm, diffusers-named state dicts with Linear proj_in/proj_out) against the
    oracle fed the translated weights; Transformer2DModel's GroupNorm eps is 1e-6 whatever norm_eps says."""
    from ccnet_amd import diffusers_dialect as dd
    from ccnet_amd.nets import adjust_class_embedding
    from ccnet_amd.sampling import sample_from_controlnet, sample_from_ldm
    from ccnet_amd.synthetic import init_weights
    import oracle.nets as on
    dcfg = dict(block_out_channels=(64, 128), down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), layers_per_block=1,
                cross_attention_dim=64, attention_head_dim=(1, 2), use_linear_projection=True, norm_eps=1e-5)
    un = dd.UNet2DConditionModel(**dcfg)
    adjust_class_embedding(un)                      # util.py:1302 (cond_on_acq_times)
    init_weights(un._net, 31)
    un.config.num_class_embeds = None               # util.py:1337-1340
    cn = dd.ControlNetModel.from_unet(un, conditioning_channels=1)
    un.config.num_class_embeds = 99999
    adjust_class_embedding(cn)
    init_weights(cn._net, 32)
    un, cn = un.to(cuda).eval(), cn.to(cuda).eval()
    # oracle twins: monai-generative modules, weights through the key translation, transformer GroupNorm eps 1e-6
    ou = on.DiffusionModelUNet(out_channels=4, **TINY)
    oc = on.ControlNet(**TINY, **TINY_CE)
    for o, d in ((ou, un), (oc, cn)):
        on.adjust_class_embedding(o)
        o.load_state_dict(dd.state_dict_to_monai({k: v.cpu() for k, v in d.state_dict().items()}))
        n = 0
        for m in o.modules():
            if type(m).__name__ == "SpatialTransformer":
                m.norm.eps = 1e-6
                n += 1
        assert n > 0
        o.eval()
    x, ctx, cond = _inputs(2, cuda, seed=33)
    t = torch.tensor([320])
    acq = torch.tensor([[100.0, 700.0]])            # (1, bs): reshaped by sample_from_* as in the reference
    down, mid = sample_from_controlnet(cn, x.to(cuda), cond.to(cuda), t.to(cuda), ctx.to(cuda), class_labels=acq.to(cuda),
                                       conditioning_scale=0.8)
    out = sample_from_ldm(un, x.to(cuda), t.to(cuda), ctx.to(cuda), down, mid, class_labels=acq.to(cuda))
    plain = sample_from_ldm(un, x.to(cuda), t.to(cuda), ctx.to(cuda), class_labels=acq.to(cuda))
    with torch.no_grad():
        a = acq.reshape(2, 1)
        rd, rm = oc(x, t, cond, 0.8, ctx, a)
        ref = ou(x, t, ctx, a, rd, rm)
        ref_plain = ou(x, t, ctx, a)
    assert isinstance(down, tuple) and len(down) == len(rd)
    _check(mid.float(), rm, "diffusers-dialect controlnet mid")
    _check(out, ref, "diffusers-dialect unet + residuals")
    _check(plain, ref_plain, "diffusers-dialect unet")
    r = un(sample=x.to(cuda), timestep=320, encoder_hidden_states=ctx.to(cuda), class_labels=acq.reshape(2, 1).to(cuda))
    assert torch.equal(r.sample, plain)             # python-number timestep, return_dict=True


def test_vae_attention_blocks_and_sd_vae_dialect(cuda):
    """Autoencoder attention blocks (monai-generative AttentionBlock: one head over all channels) and the SD-VAE
    surface of diffusers (`encode(x).latent_dist`, `decode(z).sample`, 3-resnet decoder levels, SiLU after the final
    GroupNorm, `quant_conv` on 2 x latent moments) against the oracle on translated weights."""
    from ccnet_amd import diffusers_dialect as dd
    from ccnet_amd.synthetic import init_weights
    import oracle.nets as on
    # (1) monai-generative kwargs with attention at the last level and both non-local blocks
    cfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=[64, 128], latent_channels=4,
               num_res_blocks=1, attention_levels=[False, True], with_encoder_nonlocal_attn=True,
               with_decoder_nonlocal_attn=True)
    p, o = build_pair("vae", cfg, 41, cuda)
    g = torch.Generator().manual_seed(42)
    z = torch.randn(2, 4, 16, 16, generator=g)
    x = torch.rand(2, 1, 32, 32, generator=g) * 2 - 1
    with torch.no_grad():
        ref = o.decode(z)
        rmu, rsig = o.encode(x)
    out = p.decode(z.to(cuda))
    mu, sig = p.encode(x.to(cuda))
    _check(out, ref, "AEKL decode with attention")
    _check(mu, rmu, "AEKL encode mu with attention")
    _check(sig, rsig, "AEKL encode sigma with attention")
    # (2) diffusers SD-VAE dialect
    tiny = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 2,
                up_block_types=["UpDecoderBlock2D"] * 2, block_out_channels=[64, 128], layers_per_block=1,
                latent_channels=4)
    v = dd.AutoencoderKL(**tiny)
    init_weights(v._net, 43)
    ov = on.AutoencoderKL(spatial_dims=2, in_channels=3, out_channels=3, num_channels=[64, 128], latent_channels=4,
                          num_res_blocks=1, attention_levels=[False, False], with_encoder_nonlocal_attn=True,
                          with_decoder_nonlocal_attn=True, norm_eps=1e-6, _decoder_num_res_blocks=2, _final_act=True,
                          _encoder_out_channels=8)
    ov.load_state_dict(v._net.state_dict())
    v = v.to(cuda).eval()
    x3 = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    with torch.no_grad():
        ref = ov.decode(z)
        rmu, rsig = ov.encode(x3)
    dec = v.decode(z.to(cuda))
    post = v.encode(x3.to(cuda)).latent_dist
    _check(dec.sample, ref, "SD-VAE decode")
    _check(post.mean, rmu, "SD-VAE posterior mean")
    _check(post.std, rsig, "SD-VAE posterior std")
    assert torch.equal(post.mode(), post.mean) and post.sample().shape == rmu.shape
    assert torch.allclose(post.logvar, 2 * torch.log(post.std))


def test_aekl_attention_single_image_64x64_latent(cuda):
    """One image through an AEKL whose non-local attention runs over 64 x 64 = 4096 tokens of 128 channels (monai-generative
    with_decoder_nonlocal_attn, and the shape of diffusers' SD-VAE mid block on a single 512^2 slice): batch 1 makes the
    generic attention path's GEMMs eligible for split-K, which needs the workspace at every call site."""
    cfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=[64, 128], latent_channels=4,
               num_res_blocks=1, attention_levels=[False, False], with_encoder_nonlocal_attn=False,
               with_decoder_nonlocal_attn=True)
    p, o = build_pair("vae", cfg, 45, cuda)
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(46))
    with torch.no_grad():
        ref = o.decode(z)
    out = p.decode(z.to(cuda))
    assert out.shape == (1, 1, 128, 128)
    _check(out, ref, "AEKL decode, 4096-token attention, one image")


def test_sampler_on_the_diffusers_surface(cuda):
    """run_controlnet_inference.py:211-256 with the classes the reference's SD-2.1 path builds: UNet2DConditionModel,
    ControlNetModel.from_unet, diffusers AutoencoderKL (decode returns `.sample`), DDIM, CFG, acquisition times."""
    import ccnet_amd.schedulers as ps
    import oracle.nets as on
    import oracle.schedulers as osch
    from ccnet_amd import diffusers_dialect as dd
    from ccnet_amd.nets import adjust_class_embedding
    from ccnet_amd.sampling import Sampler
    from ccnet_amd.synthetic import init_weights
    from oracle.sampler import sample as osample
    dcfg = dict(block_out_channels=(64, 128), down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), layers_per_block=1,
                cross_attention_dim=64, attention_head_dim=(1, 2), use_linear_projection=True, norm_eps=1e-5)
    un = dd.UNet2DConditionModel(**dcfg)
    cn = dd.ControlNetModel.from_unet(un, conditioning_channels=1)
    for m, seed in ((un, 51), (cn, 52)):
        adjust_class_embedding(m)
        init_weights(m._net, seed)
    vae = dd.AutoencoderKL(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                           up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[64, 64, 128, 128],
                           layers_per_block=1, latent_channels=4)
    init_weights(vae._net, 53)
    ou, oc = on.DiffusionModelUNet(out_channels=4, **TINY), on.ControlNet(**TINY, **TINY_CE)
    for o, d in ((ou, un), (oc, cn)):
        on.adjust_class_embedding(o)
        o.load_state_dict(dd.state_dict_to_monai(d.state_dict()))
        for m in o.modules():
            if type(m).__name__ == "SpatialTransformer":
                m.norm.eps = 1e-6
        o.eval()
    ov = on.AutoencoderKL(spatial_dims=2, in_channels=3, out_channels=3, num_channels=[64, 64, 128, 128],
                          latent_channels=4, num_res_blocks=1, attention_levels=[False] * 4,
                          with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True, norm_eps=1e-6,
                          _decoder_num_res_blocks=2, _final_act=True, _encoder_out_channels=8).eval()
    ov.load_state_dict(vae._net.state_dict())
    un, cn, vae = un.to(cuda).eval(), cn.to(cuda).eval(), vae.to(cuda).eval()
    B, steps = 2, 6
    x, ctx, cond = _inputs(B, cuda, seed=54)
    g = torch.Generator().manual_seed(55)
    empty = 0.5 * torch.randn(1, 77, 64, generator=g)
    acq = torch.tensor([250.0, 900.0])
    psch, osc = ps.DDIMScheduler(clip_sample=False, **SCHED), osch.DDIMScheduler(clip_sample=False, **SCHED)
    psch.set_timesteps(steps)
    osc.set_timesteps(steps)
    s = Sampler(cn, un, psch, vae, guidance_scale=3.0, conditioning_scale=1.0, scale_factor=0.5)
    res = s.sample(x.to(cuda), cond.to(cuda), ctx.to(cuda), empty.to(cuda), acq.to(cuda), record_latents=True)
    ref = osample(oc, ou, osc, x, cond, ctx, empty, guidance_scale=3.0, conditioning_scale=1.0,
                  acquisition_time=acq, stage1=ov, scale_factor=0.5, record_latents=True)
    errs = [(a.cpu() - b).abs().max().item() for a, b in zip(res["latents"], ref["latents"])]
    print("diffusers-surface per-step latent max-abs:", " ".join(f"{e:.3e}" for e in errs))
    assert max(errs) <= 0.1
    assert res["image"].shape == (B, 3, 128, 128)
    ps_db = psnr(res["image"], ref["image"])
    print(f"diffusers-surface final image PSNR {ps_db:.1f} dB")
    assert ps_db >= 40.0


# ------------------------------------------------------------------ §8(f)4: the CLIP text tower
@pytest.mark.parametrize("geom", [
    dict(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
         hidden_act="quick_gelu", eos_token_id=2),
    # SD-2.1 text_encoder geometry (ViT-H text tower: 1024 wide, 16 heads, erf GELU) at 4 of its 23 layers
    dict(vocab_size=2000, hidden_size=1024, intermediate_size=4096, num_hidden_layers=4, num_attention_heads=16,
         hidden_act="gelu", eos_token_id=1999),
])
def test_text_tower_matches_oracle(geom):
    from ccnet_amd.text_encoder import CLIPTextModel
    from oracle.text_encoder import clip_text_forward
    dev = torch.device("cuda:0")
    torch.manual_seed(123)
    m = CLIPTextModel(**geom).eval()
    with torch.no_grad():  # trained-like statistics: non-trivial norms and biases
        for n, p in m.named_parameters():
            if "layer_norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
            elif "embedding" in n:
                p.copy_(0.5 * torch.randn_like(p))
    V = geom["vocab_size"]
    ids = torch.randint(3, V - 2, (3, 77))
    ids[:, 0] = V - 2
    for b, n in enumerate((5, 40, 76)):  # EOS then padding, as the tokenizer pads prompts to 77
        ids[b, n] = V - 1
        ids[b, n + 1:] = V - 1 if geom["eos_token_id"] != 2 else 0
    ref, ref_pool = clip_text_forward(m.state_dict(), ids, geom["num_attention_heads"], geom["hidden_act"], 1e-5,
                                      geom["eos_token_id"])
    m = m.to(dev)
    out = m(ids.to(dev))
    hid, pool = out[0].cpu(), out.pooler_output.cpu()
    assert hid.shape == ref.shape and hid.dtype == torch.float32
    # bf16 residual stream over the layers: tolerance relative to the O(1) post-LayerNorm output
    err = (hid - ref).abs()
    assert err.max().item() < 0.08 and err.mean().item() < 0.01, (err.max().item(), err.mean().item())
    assert (pool - ref_pool).abs().max().item() < 0.08
    # second call (packed weights cached) is bit-identical; an in-place weight update invalidates the cache
    assert torch.equal(m(ids.to(dev))[0].cpu(), hid)
    with torch.no_grad():
        m.text_model.final_layer_norm.bias.add_(1.0)
    assert (m(ids.to(dev))[0].cpu() - hid - 1.0).abs().max().item() < 0.05
    # tuple form and the reference's reshape of [B,1,77] ids
    assert torch.equal(m(ids.unsqueeze(1).squeeze(1).to(dev), return_dict=False)[0].cpu(), m(ids.to(dev))[0].cpu())


def test_preview_sample_with_text_tower(cuda):
    """log_controlnet_samples' sampling part (util.py:840-931): default BOS + 76 x EOS prompt through the text tower, one
    image, no CFG, DDIM, AEKL decode -- against the oracle's restatement with the oracle text tower."""
    import ccnet_amd.schedulers as ps
    import oracle.schedulers as osch
    from ccnet_amd.sampling import preview_controlnet_sample
    from ccnet_amd.text_encoder import CLIPTextModel
    from oracle.sampler import preview_sample
    from oracle.text_encoder import clip_text_forward
    pc, oc = build_pair("controlnet", {**TINY, **TINY_CE}, 31, cuda, acq=True)
    pu, ou = build_pair("unet", TINY, 32, cuda, acq=True)
    pv, ov = build_pair("vae", TINY_VAE, 33, cuda)
    torch.manual_seed(34)
    geom = dict(hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=1, hidden_act="gelu")
    te = CLIPTextModel(**geom).eval()  # full 49408-token vocabulary: the default prompt uses ids 49406 / 49407
    with torch.no_grad():
        for n, p in te.named_parameters():
            if "embedding" in n:
                p.copy_(0.5 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in te.state_dict().items()}
    _, _, cond = _inputs(1, cuda, seed=35)
    noise = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(36))
    acq = torch.tensor([400.0])
    psch, osc = ps.DDIMScheduler(clip_sample=False, **SCHED), osch.DDIMScheduler(clip_sample=False, **SCHED)
    psch.set_timesteps(8)
    osc.set_timesteps(8)
    res = preview_controlnet_sample(cond, pc, pu, pv, te.to(cuda), psch, device=cuda, scale_factor=0.5,
                                    acquisition_time=acq, controlnet_conditioning_scale=0.8, noise=noise)
    ref = preview_sample(cond, oc, ou, ov, lambda ids: clip_text_forward(sd, ids, 1, "gelu", 1e-5, 49407)[0], osc, noise,
                         scale_factor=0.5, acquisition_time=acq, conditioning_scale=0.8)
    assert (res["prompt_embeds"].cpu() - ref["prompt_embeds"]).abs().max().item() < 0.08
    assert (res["latent"].cpu() - ref["latent"]).abs().max().item() <= 0.1
    db = psnr(res["image"], ref["image"])
    print(f"preview PSNR {db:.1f} dB")
    assert db >= 40.0


def test_overlapped_step_matches_sequential_step(cuda):
    """Sampler(overlap=True) runs the UNet encoder beside the ControlNet on a second stream and lets the zero-convolutions
    write skip + residual; overlap=False is the sequential form.  Same math up to one bf16 rounding per skip; the graph
    replay of the two-stream step must equal its eager run bit for bit."""
    import ccnet_amd.schedulers as ps
    from ccnet_amd.sampling import Sampler
    pc, _ = build_pair("controlnet", {**TINY, **TINY_CE}, 41, cuda, acq=True)
    pu, _ = build_pair("unet", TINY, 42, cuda, acq=True)
    pv, _ = build_pair("vae", TINY_VAE, 43, cuda)
    B = 3
    x, ctx, cond = _inputs(B, cuda, seed=44)
    empty = 0.5 * torch.randn(1, 77, 64, generator=torch.Generator().manual_seed(45))
    acq = torch.tensor([50.0, 300.0, 900.0])
    res = {}
    for name, kw in (("seq", dict(overlap=False)), ("ov_graph", dict(overlap=True)),
                     ("ov_eager", dict(overlap=True, use_graph=False))):
        sch = ps.DDIMScheduler(clip_sample=False, **SCHED)
        sch.set_timesteps(10)
        s = Sampler(pc, pu, sch, pv, guidance_scale=4.0, conditioning_scale=1.0, scale_factor=0.5, **kw)
        out = [s.sample(x.to(cuda), cond.to(cuda), ctx.to(cuda), empty.to(cuda), acq.to(cuda)) for _ in range(2)]
        assert torch.equal(out[0]["latent"], out[1]["latent"])  # replay is repeatable
        res[name] = out[1]
    assert torch.equal(res["ov_graph"]["latent"], res["ov_eager"]["latent"])
    assert torch.equal(res["ov_graph"]["image"], res["ov_eager"]["image"])
    d = (res["ov_graph"]["latent"] - res["seq"]["latent"]).abs().max().item()
    print(f"overlap vs sequential latent max-abs {d:.3e}")
    assert d <= 0.15  # two bf16 evaluation orders, each held to 0.1 against the fp32 oracle by test_sampler_parity_tiny
    assert psnr(res["ov_graph"]["image"], res["seq"]["image"].cpu()) >= 40.0


def test_layernorm_fold_option_matches_default_lowering(cuda, monkeypatch):
    """CCNET_LN_FOLD=1 (LayerNorm folded into the consuming GEMMs) against the default lowering on one UNet forward: the same
    function of the same weights, one bf16 rounding fewer per LayerNorm."""
    import ccnet_amd.nets as pn
    from ccnet_amd.synthetic import init_weights
    x, ctx, _ = _inputs(2, cuda, seed=71)
    t = torch.tensor([400]).to(cuda)
    outs = []
    for fold in (False, True):
        monkeypatch.setattr(pn, "LN_FOLD", fold)
        u = pn.DiffusionModelUNet(out_channels=4, **TINY)
        init_weights(u, 72)
        outs.append(u.to(cuda).eval()(x.to(cuda), t, ctx.to(cuda)))
    _check(outs[1], outs[0].cpu(), "LayerNorm-folded UNet vs default lowering")
    _, o = build_pair("unet", TINY, 72, cuda)
    with torch.no_grad():
        _check(outs[1], o(x, torch.tensor([400]), ctx), "LayerNorm-folded UNet vs oracle")


def test_two_phase_unet_and_embedding_table(cuda):
    """encode_phase + decode_phase is forward() cut in two (bit-identical without residuals), and a row of
    embedding_table() is what the per-step embedding MLPs produce for that timestep (fp32 summation order aside)."""
    pu, _ = build_pair("unet", TINY, 51, cuda, acq=True)
    x, ctx, _ = _inputs(2, cuda, seed=52)
    xg, cg = x.to(cuda), ctx.to(cuda)
    cls = torch.tensor([[250.0], [700.0]]).to(cuda)
    t = torch.tensor([500]).to(cuda)
    ref = pu(xg, t, cg, cls)
    st = pu.encode_phase(xg, t, cg, cls)
    assert torch.equal(pu.decode_phase(st), ref)
    table = pu.embedding_table([980, 500, 20], cls, 2)
    assert table.shape[:2] == (3, 2)
    per_step = pu._embedding(pu._pk or pu._pack(), t, cls, 2)
    assert (table[1] - per_step).abs().max().item() <= 1e-4 * per_step.abs().max().item()
    out = pu(xg, t, cg, cls, _tall=table[1])
    assert (out - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert pu.embedding_table(range(1000), cls, 2, max_bytes=1024) is None  # too big: the caller falls back to per-step


def test_cond_embedding_chunks_over_images(cuda):
    """Batches whose 512^2-resolution activations would pass the conv kernel's 2 GiB tensor limit are embedded in chunks of
    images; the result is the unchunked one."""
    pc, _ = build_pair("controlnet", {**TINY, **TINY_CE}, 61, cuda)
    _, _, cond = _inputs(5, cuda, seed=62)
    whole = pc.cond_embedding(cond.to(cuda))
    per_image = 128 * 128 * 256 * 2
    parts = pc.cond_embedding(cond.to(cuda), _max_bytes=2 * per_image)  # two images per chunk: 2 + 2 + 1
    assert (parts.N, parts.H, parts.W, parts.C) == (whole.N, whole.H, whole.W, whole.C)
    assert torch.equal(parts.t, whole.t)
