import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ccnet_amd.sampling import Sampler, to_uint8
from ccnet_amd.schedulers import DDPMScheduler
from ccnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
cn, un, vae = bench.build_models(dev, True)
sch = DDPMScheduler(**bench.SCHED)
sch.set_timesteps(1000)
s = Sampler(cn, un, sch, vae, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2   # configs[4] of BASELINE.json is batch 32 per GPU
b = {k: v.to(dev) for k, v in make_batch(list(range(B)), seed=42, acq=True).items()}
t0 = time.perf_counter()
r = s.sample(b['noise'], b['cond'], b['prompt_embeds'], b['empty_prompt_embeds'], b.get('acquisition_time'))
torch.cuda.synchronize()
print('DDPM 1000 steps, batch %d: %.1f s' % (B, time.perf_counter() - t0))
lat, img = r['latent'], r['image']
print('latent finite', bool(torch.isfinite(lat).all()), 'absmax %.3f' % lat.abs().max().item(), 'image finite', bool(torch.isfinite(img).all()),
      'uint8 mean %.1f' % to_uint8(img).float().mean().item())
assert torch.isfinite(lat).all() and torch.isfinite(img).all() and lat.abs().max() <= 1.0 + 1e-5  # clip_sample=True
