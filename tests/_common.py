"""Shared builders for the parity tests: product (HIP) networks and fp32 oracle networks with identical weights."""
import torch

TINY = dict(spatial_dims=2, in_channels=4, num_res_blocks=1, num_channels=[64, 128], attention_levels=[True, True],
            with_conditioning=True, cross_attention_dim=64, num_head_channels=64, norm_eps=1e-5)
# the repo YAML's shape of attention (configs/controlnet/controlnet_v1.yaml:8-11): no attention at level 0, ONE head per level
TINY_G2 = dict(spatial_dims=2, in_channels=4, num_res_blocks=1, num_channels=[64, 128, 192], attention_levels=[False, True, True],
               with_conditioning=True, cross_attention_dim=64, num_head_channels=[0, 128, 192], norm_eps=1e-6)
TINY_CE = dict(conditioning_embedding_in_channels=1, conditioning_embedding_num_channels=[16, 32, 96, 256])
TINY_VAE = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=[64, 128, 128, 128], latent_channels=4,
                num_res_blocks=1, attention_levels=[False] * 4, with_encoder_nonlocal_attn=False,
                with_decoder_nonlocal_attn=False)
G1 = dict(spatial_dims=2, in_channels=4, num_res_blocks=2, num_channels=[320, 640, 1280, 1280],
          attention_levels=[True, True, True, False], with_conditioning=True, cross_attention_dim=1024,
          num_head_channels=64, norm_eps=1e-5)
G1_CE = dict(conditioning_embedding_in_channels=1, conditioning_embedding_num_channels=[64, 128, 128, 256])
SCHED = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0205, schedule="scaled_linear_beta",
             prediction_type="v_prediction")


def build_pair(kind, cfg, seed, dev, acq=False):
    """Returns (product_on_gpu, oracle_on_cpu) with the same seeded weights."""
    import ccnet_amd.nets as pn
    import oracle.nets as on
    from ccnet_amd.synthetic import init_weights
    if kind == "unet":
        p, o = pn.DiffusionModelUNet(out_channels=cfg["in_channels"], **cfg), \
            on.DiffusionModelUNet(out_channels=cfg["in_channels"], **cfg)
    elif kind == "controlnet":
        p, o = pn.ControlNet(**cfg), on.ControlNet(**cfg)
    else:
        p, o = pn.AutoencoderKL(**cfg), on.AutoencoderKL(**cfg)
    if acq:
        pn.adjust_class_embedding(p)
        on.adjust_class_embedding(o)
    init_weights(p, seed)
    o.load_state_dict(p.state_dict())
    return p.to(dev).eval(), o.eval()


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def psnr(a, b, data_range=2.0):
    a, b = a.float().cpu(), b.float().cpu()
    mse = ((a - b) ** 2).mean().item()
    import math
    return 10 * math.log10(data_range ** 2 / max(mse, 1e-20))
