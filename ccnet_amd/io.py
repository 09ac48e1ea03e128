"""The data edges of the sampling path, MI355X side.

  * `load_images`  -- `LoadImaged -> EnsureChannelFirstd -> Resized(bicubic) -> ScaleIntensityRanged(0..255 -> -1..1, clip)` of
                      `get_test_dataloader` (util.py:112-131): PNG bytes are decoded on the host (PIL), everything after the
                      decode is one kernel on the GPU.  MONAI's PNG reader hands the network the TRANSPOSED (W, H) array
                      (SURVEY.md section 8(f)2); `monai_orientation=True` (default) reproduces that.
  * `save_controlnet_inference_sample` -- util.py:1396-1447: `PIL/<filename>` (synthetic slice) and
                      `PIL_comparison/<filename>` (cond | synthetic | real, side by side), `interval_mapping` -> uint8 of
                      channel 0 on the GPU (`ccnet_to_uint8`), PNG encoding on the host.  Batched: one file per image.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import ops
from .lib import CcnetError


def load_images(paths: Sequence[Union[str, Path]], img_height: int = 512, img_width: int = 512, device="cuda",
                monai_orientation: bool = True) -> torch.Tensor:
    """-> fp32 [B, 1, img_height, img_width] in [-1, 1] on `device`."""
    from PIL import Image
    out = []
    for p in paths:
        with Image.open(p) as im:
            a = np.array(im.convert("L") if im.mode not in ("L",) else im, dtype=np.uint8)  # (a writable copy)
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        out.append(ops.preprocess_u8(t[None], img_height, img_width, transpose=monai_orientation))
    return torch.cat(out, dim=0)


def _u8(x, f32_scale: bool = False) -> np.ndarray:
    x = getattr(x, "sample", x)  # diffusers DecoderOutput (util.py:1402)
    if not torch.is_tensor(x) or not x.is_cuda:
        raise CcnetError("save_controlnet_inference_sample expects GPU tensors (the uint8 mapping runs on the GPU)")
    return ops.to_uint8(x.float(), f32_scale=f32_scale).cpu().numpy()


def save_controlnet_inference_sample(run_dir, filename: Union[str, Sequence[str]], sample, cond=None, images=None,
                                     save_comparison: bool = True) -> List[Path]:
    """sample / cond / images: [B, C, H, W] in [-1, 1]; filename: one name (B = 1) or B names.  Returns the written paths."""
    from PIL import Image
    run_dir = Path(run_dir)
    s = _u8(sample)
    names = [filename] if isinstance(filename, (str, Path)) else list(filename)
    if len(names) != s.shape[0]:
        raise ValueError(f"{len(names)} file names for a batch of {s.shape[0]}")
    written = []
    (run_dir / "PIL").mkdir(parents=True, exist_ok=True)
    for i, nm in enumerate(names):
        path = run_dir / "PIL" / nm
        Image.fromarray(s[i]).save(path)
        written.append(path)
    if save_comparison and cond is not None and images is not None:
        c, r = _u8(cond, True), _u8(images, True)  # float32 arrays in the reference (util.py:1432-1435)
        (run_dir / "PIL_comparison").mkdir(parents=True, exist_ok=True)
        for i, nm in enumerate(names):
            path = run_dir / "PIL_comparison" / nm
            Image.fromarray(np.concatenate((c[i], s[i], r[i]), axis=1)).save(path)
            written.append(path)
    return written
